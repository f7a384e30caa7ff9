"""GPU box: the index swizzle of the one-gate kernels (option swizzle) by state size -- does it cost anything where the
state fits the caches?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcmrf_amd import _lib
rs = np.random.RandomState(0)
q, _ = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2))
for W in (16, 20, 22, 24, 26, 28, 30):
    eng = _lib.Engine(W)
    eng.init_uniform((1 << W) - 1)
    line = "W=%d" % W
    for t in (2, 8, W - 1):
        for sw in (0, 1):
            eng.set_option("swizzle", sw)
            for _ in range(3): eng.apply_1q(t, q)
            eng.sync(); eng.timer_begin()
            reps = 20 if W <= 24 else 6
            for _ in range(reps): eng.apply_1q(t, q)
            ms = eng.timer_end() / reps
            line += "  | t%d swz%d %.4f ms %4.0f GB/s" % (t, sw, ms, 32 * 2.0 ** W / ms / 1e6)
    print(line, flush=True)
    eng.close()
