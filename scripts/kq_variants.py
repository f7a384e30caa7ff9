"""GPU box: dense k-qubit gates on the matrix cores (k_kq_mfma) -- grid-stride walk against a contiguous run of batches per wave."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcmrf_amd import _lib
W = 28
rs = np.random.RandomState(0)
def ru(k):
    q, _ = np.linalg.qr(rs.randn(2 ** k, 2 ** k) + 1j * rs.randn(2 ** k, 2 ** k)); return q
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
cases = [("kq5_low", [0, 1, 2, 3, 4]), ("kq5_6to10", [6, 7, 8, 9, 10]), ("kq5_mixed", [3, 9, 14, 20, 26]), ("kq5_high", [23, 24, 25, 26, 27]),
         ("kq4_mid", [8, 9, 10, 11]), ("kq3_high", [25, 26, 27]), ("kq3_low", [1, 2, 3])]
for name, qs in cases:
    u = ru(len(qs))
    line = "%-10s" % name
    for ch in (0, 1):
        eng.set_option("kq_chunked", ch)
        for _ in range(2): eng.apply_kq(qs, u)
        eng.sync(); eng.timer_begin()
        for _ in range(6): eng.apply_kq(qs, u)
        ms = eng.timer_end() / 6
        line += "  | chunked %d %.3f ms %.3f" % (ch, ms, 32 * 2.0 ** W / ms / 1e6 / 8000)
    print(line, flush=True)
eng.close()
