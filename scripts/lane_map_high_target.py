"""GPU box: one 1q gate on a HIGH target as a one-op k_multi pass (tile = target + bits 6..9) under different
lane maps (engine option lane_map: 5-bit fields = address bit carried by lane bits 3, 4, 5; 1 = default rule)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcmrf_amd import _lib, ir, program
W = int(sys.argv[1]) if len(sys.argv) > 1 else 28
rs = np.random.RandomState(0)
def ru():
    q, _ = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2)); return q
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
eng.set_option("single_shortcut", 0)
eng.set_option("xframe", 0)
def t(ops, reps=6):
    rec, data = program.encode(ops)
    for _ in range(2): eng.exec(rec, data)
    eng.sync(); eng.timer_begin()
    for _ in range(reps): eng.exec(rec, data)
    return eng.timer_end() / reps
B = 32.0 * 2 ** W
maps = [("default", 1), ("plain", 0)] + [("l5->%d" % b, b << 10) for b in (10, 11, 12, 13)] + [("l4->10,l5->11", (10 << 5) | (11 << 10)), ("l3->10 l4->11 l5->12", 10 | (11 << 5) | (12 << 10))]
for q in (8, 13, 17, 20, 24, 27):
    line = "1q_t%02d " % q
    for name, code in maps:
        eng.set_option("lane_map", code)
        ms = t([ir.op_u(q, ru())])
        line += " | %s %.3f (%.3f)" % (name, ms, B / ms / 1e6 / 8000)
    print(line, flush=True)
eng.close()
