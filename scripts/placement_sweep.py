"""GPU box: where should the register / borrowed-lane targets of a pass sit?  One k_multi pass
(5 register targets + borrowed lanes + 3 static lanes), read+write and init-fused, W qubits."""
import sys
sys.path.insert(0, ".")
import numpy as np
from qcmrf_amd import _lib, ir, program
W = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rs = np.random.RandomState(0)
def rx():
    a = rs.rand() * 3
    return np.array([[np.cos(a), -1j * np.sin(a)], [-1j * np.sin(a), np.cos(a)]])
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
def run(regs, bor, stat, init, label):
    sel = [W - 1, W - 2]
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
    st = eng.stats()["kinds"]
    n = sum(v["launches"] for v in st.values()) / 4
    print("%-4s %-44s launches %.1f  %8.3f ms  %d GB/s" % ("init" if init else "r+w", label, n, ms, (16 if init else 32) * 2.0 ** W / ms / 1e6), flush=True)
R = lambda a, b: list(range(a, b + 1))
cases = [
    (R(14, 18), R(19, 21), [0, 1, 2], "reg 14-18 bor 19-21 stat 0-2"),
    (R(6, 10), R(11, 13), [0, 1, 2], "reg 6-10 bor 11-13 stat 0-2"),
    (R(9, 13), R(6, 8), [0, 1, 2], "reg 9-13 bor 6-8 stat 0-2"),
    (R(6, 10), R(14, 16), [0, 1, 2], "reg 6-10 bor 14-16 stat 0-2"),
    (R(11, 15), R(6, 8), [0, 1, 2], "reg 11-15 bor 6-8 stat 0-2"),
    (R(11, 15), R(16, 18), [0, 1, 2], "reg 11-15 bor 16-18 stat 0-2"),
    (R(6, 10), [], [0, 1, 2], "reg 6-10 stat 0-2"),
    (R(6, 10), [], [], "reg 6-10"),
    (R(11, 15), [], [], "reg 11-15"),
    (R(14, 18), [], [], "reg 14-18"),
    (R(6, 10), [11], [], "reg 6-10 bor 11"),
    (R(6, 10), [12], [], "reg 6-10 bor 12"),
    (R(6, 10), [14], [], "reg 6-10 bor 14"),
    (R(6, 10), [20], [], "reg 6-10 bor 20"),
    (R(7, 11), [6], [], "reg 7-11 bor 6"),
    (R(11, 15), [6], [], "reg 11-15 bor 6"),
    (R(14, 18), [6], [], "reg 14-18 bor 6"),
    (R(14, 18), [19], [], "reg 14-18 bor 19"),
    (R(6, 10), [11, 12], [], "reg 6-10 bor 11-12"),
    (R(8, 12), [6, 7], [], "reg 8-12 bor 6-7"),
]
for init in (False, True):
    for regs, bor, stat, label in cases:
        run(regs, bor, stat, init, label)
