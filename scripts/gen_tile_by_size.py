"""GPU box: generator tile choice by shard size (8 uniform factors + 4 factors on the register bits)."""
import sys
sys.path.insert(0, ".")
import numpy as np
from qcmrf_amd import _lib, ir, program
WS = tuple(int(x) for x in sys.argv[1:]) or (26, 28, 29, 30, 31, 32, 33)
for W in WS:
    eng = _lib.Engine(W)
    def timeit(f, n=6):
        for _ in range(2): f()
        eng.sync(); eng.timer_begin()
        for _ in range(n): f()
        return eng.timer_end() / n
    ms = timeit(lambda: eng.init_uniform((1 << W) - 1))
    line = "W=%d  k_init %5.0f" % (W, 16 * 2.0 ** W / ms / 1e6)
    for nt, R, b0 in [(n_, r_, b_) for n_ in (0, 1) for (r_, b_) in ((5, 0), (4, 0), (4, -1), (5, -1), (3, -1))]:
        regs = list(range(6, 6 + R)) if b0 == 0 else list(range(W - R, W))
        ops = [ir.op_init((1 << W) - 1)] + [ir.op_diag([1 + (k % 5), 0], np.exp(1j * np.arange(4) * (k + 1))) for k in range(8)]
        ops += [ir.op_diag([regs[k], 12 + k], np.exp(1j * np.arange(4) * (k + 2))) for k in range(min(4, R))]
        rec, data = program.encode(ops)
        try:
            eng.set_option("init_prod_r", R)
        except ValueError:
            continue
        eng.set_option("init_prod_bit0", b0)
        eng.set_option("init_prod_nt", nt)
        ms = timeit(lambda: eng.exec(rec, data))
        line += "  | %sR=%d %s %5.0f" % ("nt " if nt else "", R, "low" if b0 == 0 else "top", 16 * 2.0 ** W / ms / 1e6)
    print(line, flush=True)
    eng.close()
