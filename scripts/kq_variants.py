"""GPU box: dense k-qubit gates (k_kq_mfma / k_kq_mfma3 / k_kq_tile): every kq_variant on the micro-benchmark's cases.
   python scripts/kq_variants.py        timing table
   python scripts/kq_variants.py pmc    two launches per (case, variant), for rocprofv3 --pmc"""
import os, sys
os.environ.setdefault("QSV_MEASUREMENT_KNOBS", "1")         # kq_debug: kernels with a part of their work left out (timing only)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcmrf_amd import _lib
W = 28
pmc = len(sys.argv) > 1 and sys.argv[1] == "pmc"
rs = np.random.RandomState(0)
def ru(k):
    q, _ = np.linalg.qr(rs.randn(2 ** k, 2 ** k) + 1j * rs.randn(2 ** k, 2 ** k)); return q
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
cases = [("kq5_low", [0, 1, 2, 3, 4]), ("kq5_mixed", [1, 6, 11, 17, W - 1]), ("kq5_6to10", [6, 7, 8, 9, 10]), ("kq5_high", [23, 24, 25, 26, 27]),
         ("kq4_mid", [8, 9, 10, 11]), ("kq3_high", [25, 26, 27]), ("kq3_low", [1, 2, 3])]
if pmc:
    cases = [c for c in cases if c[0] in ("kq5_low", "kq5_mixed", "kq3_high")]
for name, qs in cases:
    u = ru(len(qs))
    line = "%-10s" % name
    for variant, tile3, bpc, ch in ((3, 0, 0, 0), (6, 0, 0, 0), (6, 0, 64, 0), (7, 0, 0, 0), (7, 0, 64, 0), (8, 0, 0, 0), (8, 0, 16, 0), (8, 0, 64, 0), (3, 2, 0, 0)):
        if (tile3 and len(qs) != 3) or (pmc and (bpc or ch or variant in (0, 2))):
            continue
        eng.set_option("kq_variant", variant)
        eng.set_option("kq3_tile", tile3)
        eng.set_option("kq_blocks_per_cu", bpc)
        eng.set_option("kq_order", ch // 10 if ch >= 10 else 0)     # (10, 20, 30: kq_order 1, 2, 3)
        eng.set_option("kq_chunked", ch if 0 < ch < 10 else 0)
        eng.set_option("kq_debug", -ch if ch < 0 else 0)        # (v6 only: c-1 = no products, c-2 = no memory traffic; results are garbage)
        reps = 2 if pmc else 8
        for _ in range(0 if pmc else 2): eng.apply_kq(qs, u)
        eng.sync(); eng.timer_begin()
        for _ in range(reps): eng.apply_kq(qs, u)
        ms = eng.timer_end() / reps
        line += " | v%d%s%s%s %.3f" % (variant, "t" if tile3 else "", "b%d" % bpc if bpc else "", ("c%d" % ch) if ch else "", 32 * 2.0 ** W / ms / 1e6 / 8000)
    print(line, flush=True)
eng.close()
