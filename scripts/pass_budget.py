"""GPU box: what does one more op cost inside a GENERAL k_multi pass (masked X / 2x2 / phase, table
ops with register selects), and how many fit before the pass stops being HBM bound?  Calibrates
op_cost() / pass_budget in qsv_multi.inc.  Also: the unfused reference stream with and without the
X frame and the budget."""
import sys
sys.path.insert(0, ".")
import numpy as np
from qcmrf_amd import _lib, ir, program
W = int(sys.argv[1]) if len(sys.argv) > 1 else 28
rs = np.random.RandomState(0)
def ru():
    q, _ = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2)); return q
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
eng.set_option("pass_budget", 0)
import os
RR = int(os.environ.get("QSV_R", "4"))
GR = int(os.environ.get("QSV_GR", "4"))                 # tile of the general pass (padded to it)
eng.set_option("multi_r", max(RR, GR))
eng.set_option("general_r", GR)
if os.environ.get("QSV_GLR"): eng.set_option("general_light_r", int(os.environ["QSV_GLR"]))
tg = [8, 9, 10, 11][:RR]
def run(ops, label):
    rec, data = program.encode(ops)
    for _ in range(2): eng.exec(rec, data)
    eng.sync(); eng.reset_stats(); eng.timer_begin()
    for _ in range(4): eng.exec(rec, data)
    ms = eng.timer_end() / 4
    st = eng.stats()
    nl = sum(v["launches"] for v in st["kinds"].values()) / 4
    print("%-40s ops %3d  launches/exec %.1f  %.3f ms/exec  %5.1f us/op  %4.0f GB/s" % (label, len(ops), nl, ms, ms * 1e3 / len(ops), 32 * 2.0 ** W * nl / ms / 1e6), flush=True)
    return ms
for xf in (() if os.environ.get("QSV_STREAM_ONLY") else ((0,) if os.environ.get("QSV_MICRO_ONLY") else (1, 0))):
    eng.set_option("xframe", xf)
    print("--- xframe", xf)
    for N in (4, 16, 32, 60):
        run([ir.op_x(tg[i % RR]) for i in range(N)], "plain X round-robin")
        run([ir.op_x(tg[i % RR], [2, 3], [1, 0]) for i in range(N)], "CCX (lane-bit controls)")
        run([ir.op_x(tg[i % RR], [tg[(i + 1) % RR], tg[(i + 2) % RR]], [1, 0]) for i in range(N)], "CCX (register controls)")
        run([ir.op_x(tg[i % RR], [20, 21], [1, 0]) for i in range(N)], "CCX (block-bit controls)")
        run([ir.op_x(i % 3, [tg[i % RR], 20], [1, 0]) for i in range(N)], "CCX on a LANE target (register + block control)")
        run([ir.op_u(i % 3, ru(), [20], [1]) for i in range(N)], "controlled 2x2 on a LANE target")
        run([ir.op_u(tg[i % RR], ru()) for i in range(N)], "dense 2x2 (type 2)")
        run([ir.op_u(tg[i % RR], ru(), [3], [1]) for i in range(N)], "controlled 2x2 (lane control)")
        run([ir.op_mcphase([2, tg[i % RR]], 0.3) for i in range(N)], "cp (lane + register)")
        run([ir.op_mcphase([tg[(i + 1) % RR], tg[i % RR]], 0.3) for i in range(N)], "cp (register + register)")
        run([ir.op_mux([2, tg[(i + 1) % RR]], tg[i % RR], np.array([ru() for _ in range(4)])) for i in range(N)], "mux, register select")
        run([ir.op_diag([2, tg[i % RR]], np.exp(1j * rs.randn(4))) for i in range(N)], "diag, register select")
eng.close()
if os.environ.get("QSV_MICRO_ONLY"):
    sys.exit(0)
# the reference stream, gate by gate, at this width
from qcmrf_amd import QCMRF, workloads as wl
from qcmrf_amd.backend import QsvBackend
C = wl.for_width(W)
qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
be = QsvBackend(fusion=0)
for opts in ({"xframe": 0, "pass_budget": 0}, {"xframe": 1, "pass_budget": 0}, {"xframe": 1, "pass_budget": 80}, {"xframe": 1, "pass_budget": 100},
             {"xframe": 1, "pass_budget": 120}, {"xframe": 1, "pass_budget": 150}, {"xframe": 1, "pass_budget": 200}):
    be.run(qc, shots=16, engine_options=opts)
    r = be.run(qc, shots=16, engine_options=opts, profile=True).result()
    m = r.metadata(0)
    k = m["stats"]["kinds"]
    print("fusion 0, W=%d %s: evolve %.1f ms" % (W, opts, m["time_evolve"] * 1e3),
          {n: (v["launches"], round(v["ms"] / v["launches"], 3), round(v["bytes"] / v["ms"] / 1e6 / 8000, 3)) for n, v in k.items() if v["ms"] > 0}, flush=True)
be.close()
