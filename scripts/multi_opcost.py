"""GPU box: cost per op inside a k_multi pass (general path vs simple path)."""
import sys, json
sys.path.insert(0, ".")
import numpy as np
from qcmrf_amd import _lib, ir, program
W = 28
rs = np.random.RandomState(0)
def ru():
    q, _ = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2)); return q
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
eng.set_option("multi_r", 4)
tg = [8, 9, 10, 11]
def run(ops, label):
    rec, data = program.encode(ops)
    for _ in range(2): eng.exec(rec, data)
    eng.sync(); eng.reset_stats(); eng.timer_begin()
    for _ in range(4): eng.exec(rec, data)
    ms = eng.timer_end() / 4
    st = eng.stats()
    print("%-34s ops %3d  launches/exec %.1f  %.3f ms/exec  %.1f us/op" % (label, len(ops), st["kinds"]["multi"]["launches"] / 4 if "multi" in st["kinds"] else 0, ms, ms * 1e3 / len(ops)), flush=True)
for N in (4, 16, 48):
    run([ir.op_x(tg[i % 4]) for i in range(N)], "plain X round-robin")
    run([ir.op_x(tg[i % 4], [2, 3], [1, 0]) for i in range(N)], "CCX (lane-bit controls)")
    run([ir.op_u(tg[i % 4], ru()) for i in range(N)], "dense 2x2 (type 2)")
    run([ir.op_mux([2, 3], tg[i % 4], np.array([ru() for _ in range(4)])) for i in range(N)], "mux table (simple path)")
    run([ir.op_mcphase([2, tg[i % 4]], 0.3) for i in range(N)], "cp (type 3)")
    run([ir.op_x(tg[0]) for i in range(N)], "plain X same target (1 per round)")
