"""GPU box: do read+write k_multi passes like their register targets on the TOP bits of a big shard
(as the write-only generator does)?"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcmrf_amd import _lib, ir, program
W = int(sys.argv[1]) if len(sys.argv) > 1 else 34
rs = np.random.RandomState(0)
def rx():
    a = rs.rand() * 3
    return np.array([[np.cos(a), -1j * np.sin(a)], [-1j * np.sin(a), np.cos(a)]])
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
def run(regs, bor, stat, init, label):
    sel = [q for q in (25, 26, 19, 21) if q not in regs + bor + stat][:2]
    mux = lambda t: ir.op_mux(sel, t, np.array([rx() for _ in range(4)]))
    quiet = (1 << W) - 1
    for q in regs + bor + stat:
        quiet &= ~(1 << q)
    ops = ([ir.op_init(quiet)] if init else []) + [mux(t) for t in regs + bor + stat]
    rec, data = program.encode(ops)
    for _ in range(2): eng.exec(rec, data)
    eng.sync(); eng.reset_stats(); eng.timer_begin()
    for _ in range(4): eng.exec(rec, data)
    ms = eng.timer_end() / 4
    n = sum(v["launches"] for v in eng.stats()["kinds"].values()) / 4
    print("%-4s %-44s launches %.1f  %8.3f ms  %d GB/s" % ("init" if init else "r+w", label, n, ms, (16 if init else 32) * 2.0 ** W / ms / 1e6), flush=True)
R = lambda a, b: list(range(a, b + 1))
T = W - 1
cases = [
    (R(6, 10), [], [], "reg 6-10"),
    (R(T - 4, T), [], [], "reg top 5"),
    (R(T - 5, T - 1), [], [], "reg top-1 .. "),
    (R(T - 3, T), [], [], "reg top 4"),
    (R(T - 4, T), [6, 7, 8], [0, 1, 2], "reg top 5 + bor 6-8 + stat 0-2"),
    (R(T - 4, T), [14, 15, 16], [0, 1, 2], "reg top 5 + bor 14-16 + stat 0-2"),
    (R(6, 10), [11, 12 + 2, 15], [0, 1, 2], "reg 6-10 + bor 11,14,15 + stat 0-2"),
    (R(T - 4, T), [6, 7, 8], [], "reg top 5 + bor 6-8"),
    # round 2 (non-temporal kernels, generalised lane map): mixed tiles
    ([6] + R(T - 3, T), [], [], "reg 6 + top 4"),
    ([6, 7] + R(T - 2, T), [], [], "reg 6,7 + top 3"),
    ([12] + R(T - 3, T), [], [], "reg 12 + top 4"),
    (R(14, 18), [], [], "reg 14-18"),
    (R(20, 24), [], [], "reg 20-24"),
    (R(T - 4, T), [12, 13, 14], [0, 1, 2], "reg top 5 + bor 12-14 + stat 0-2"),
    (R(T - 4, T), [11, 12, 13], [], "reg top 5 + bor 11-13"),
    # round 3: mixed tiles with the bench pass's lane targets
    ([9, 10, 11, T - 1, T], [6, 7, 8], [0, 1, 2], "reg 9-11 + top 2, bor 6-8, stat 0-2"),
    ([6, 7, 8, T - 1, T], [9, 10, 12], [0, 1, 2], "reg 6-8 + top 2, bor 9,10,12, stat 0-2"),
    ([6, 7, T - 2, T - 1, T], [8, 9, 10], [0, 1, 2], "reg 6,7 + top 3, bor 8-10, stat 0-2"),
    (R(6, 10), [12, 13, 14], [0, 1, 2], "reg 6-10, bor 12-14, stat 0-2"),
    (R(T - 3, T) + [6], [7, 8, 9], [0, 1, 2], "reg 6 + top 4, bor 7-9, stat 0-2"),
]
if os.environ.get("QSV_CASES"):
    pick = [int(x) for x in os.environ["QSV_CASES"].split(",")]
    cases = [cases[i] for i in pick]
for init in ((False,) if os.environ.get('QSV_RW_ONLY') else (False, True)):
    for regs, bor, stat, label in cases:
        run(regs, bor, stat, init, label)
