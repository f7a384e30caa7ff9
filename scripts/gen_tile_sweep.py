"""GPU box: where should the generator's register bits sit?  (write-only pass, 19 uniform factors)"""
import sys
sys.path.insert(0, ".")
import numpy as np
from qcmrf_amd import _lib, ir, program
W = int(sys.argv[1]) if len(sys.argv) > 1 else 34
eng = _lib.Engine(W)
def timeit(f, n=4):
    for _ in range(2): f()
    eng.sync(); eng.timer_begin()
    for _ in range(n): f()
    return eng.timer_end() / n
ms = timeit(lambda: eng.init_uniform((1 << W) - 1))
print("k_init                          %8.3f ms  %d GB/s" % (ms, 16 * 2.0 ** W / ms / 1e6), flush=True)
ops = [ir.op_init((1 << W) - 1)] + [ir.op_diag([1 + (k % 5), 0], np.exp(1j * np.arange(4) * (k + 1))) for k in range(8)]
rec, data = program.encode(ops)
for R in (5, 4, 6):
    eng.set_option("init_prod_r", R)
    for b0 in (0, -1, 12, 16, 20, 24, W - R - 1, W - R - 2):
        eng.set_option("init_prod_bit0", b0)
        ms = timeit(lambda: eng.exec(rec, data))
        print("R=%d  bit0=%3d                   %8.3f ms  %d GB/s" % (R, b0, ms, 16 * 2.0 ** W / ms / 1e6), flush=True)
