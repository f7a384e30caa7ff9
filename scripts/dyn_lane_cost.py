"""GPU box: k_multi pass with 5 register targets + k lane gates, the lane gates either on static
lane bits (3,4,5 then 0,1,2) or on borrowed lane bits (targets 11,12,13); read+write pass and
init-fused (write-only) pass.  Isolates the price of 128-byte granules and of lane gates."""
import sys
sys.path.insert(0, ".")
import numpy as np
from qcmrf_amd import _lib, ir, program
W = int(sys.argv[1]) if len(sys.argv) > 1 else 28
rs = np.random.RandomState(0)
def rx():
    a = rs.rand() * 3
    return np.array([[np.cos(a), -1j * np.sin(a)], [-1j * np.sin(a), np.cos(a)]])
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
def run(ops, label, init):
    rec, data = program.encode(ops)
    for _ in range(2): eng.exec(rec, data)
    eng.sync(); eng.reset_stats(); eng.timer_begin()
    for _ in range(5): eng.exec(rec, data)
    ms = eng.timer_end() / 5
    st = eng.stats()["kinds"]
    n = sum(v["launches"] for v in st.values()) / 5
    print("%-52s launches/exec %.1f  %.3f ms  %d GB/s" % (label, n, ms, (16 if init else 32) * 2.0 ** W / ms / 1e6), flush=True)
sel = [24, 25]
regs = [6, 7, 8, 9, 10]
mux = lambda t: ir.op_mux(sel, t, np.array([rx() for _ in range(4)]))
quiet = (1 << W) - 1
for q in regs + [11, 12, 13, 17, 19, 21, 0, 1, 2, 3, 4, 5]:
    quiet &= ~(1 << q)
for init in (False, True):
    pre = [ir.op_init(quiet)] if init else []
    kind = "init" if init else "r+w "
    for lanes, tag in (([], "none"), ([3, 4, 5], "static 3-5"), ([11, 12, 13], "borrowed 11-13"), ([17, 19, 21], "borrowed 17,19,21"),
                       ([3, 4, 5, 0, 1, 2], "static 0-5"), ([11, 12, 13, 0, 1, 2], "borrowed + static 0-2"), ([11], "borrowed 11"),
                       ([11, 12], "borrowed 11-12")):
        eng.set_option("dyn_lanes", 3)
        run(pre + [mux(t) for t in regs] + [mux(t) for t in lanes], "%s 5 reg + lanes %s" % (kind, tag), init)
