"""GPU box: speed of one k_multi pass vs WHERE its target bits sit (run from the repo root)."""
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
for R in (4, 5, 6):
    eng.set_option("multi_r", R)
    for start in list(range(6, W - R + 1)):
        tg = list(range(start, start + R))
        free = [b for b in range(W) if b not in tg]
        ops = [ir.op_mux(free[:3], t, np.array([ru() for _ in range(8)])) for t in tg]
        rec, data = program.encode(ops)
        for _ in range(2): eng.exec(rec, data)
        eng.sync(); eng.timer_begin()
        for _ in range(6): eng.exec(rec, data)
        ms = eng.timer_end() / 6
        print(json.dumps({"R": R, "bits": "%d-%d" % (tg[0], tg[-1]), "ms": round(ms, 3), "GBps": round(32 * 2.0 ** W / ms / 1e6)}), flush=True)
    # scattered targets
    tg = [7, 11, 16, 21, 26, 13][:R]
    free = [b for b in range(W) if b not in tg]
    ops = [ir.op_mux(free[:3], t, np.array([ru() for _ in range(8)])) for t in tg]
    rec, data = program.encode(ops)
    for _ in range(2): eng.exec(rec, data)
    eng.sync(); eng.timer_begin()
    for _ in range(6): eng.exec(rec, data)
    ms = eng.timer_end() / 6
    print(json.dumps({"R": R, "bits": str(tg), "ms": round(ms, 3), "GBps": round(32 * 2.0 ** W / ms / 1e6)}), flush=True)
