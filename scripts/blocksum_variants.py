"""GPU box: access pattern of the read-only |amp|^2 pass (k_blocksum; engine option blocksum_variant)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcmrf_amd import _lib
for W in (28, 32, 34):
    try:
        eng = _lib.Engine(W)
    except RuntimeError as e:
        print(W, e); continue
    eng.init_uniform((1 << W) - 1)
    eng.set_option("cache_sums", 0)
    line = "W=%d" % W
    for bv in (0, 2, 4, 6, 1):
        eng.set_option("blocksum_variant", bv)
        for _ in range(2): eng.norm()
        eng.sync(); eng.timer_begin()
        for _ in range(6): eng.norm()
        ms = eng.timer_end() / 6
        line += "  | v%d %.3f ms %.3f" % (bv, ms, 16 * 2.0 ** W / ms / 1e6 / 8000)
    print(line, flush=True)
    import numpy as np
    tab = np.random.RandomState(0).randn(2 ** 10)
    line = "W=%d expect_diag (10-qubit table)" % W
    for bv in (0, 2, 6):
        eng.set_option("blocksum_variant", bv)
        for _ in range(2): eng.expect_diag(list(range(3, 13)), tab)
        eng.sync(); eng.timer_begin()
        for _ in range(4): eng.expect_diag(list(range(3, 13)), tab)
        ms = eng.timer_end() / 4
        line += "  | v%d %.3f ms %.3f" % (bv, ms, 16 * 2.0 ** W / ms / 1e6 / 8000)
    print(line, flush=True)
    eng.close()
