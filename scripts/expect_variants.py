"""GPU box: <H>-style diagonal expectation on the resident state (qsv_expect_diag): grid-stride walk (blocksum_variant 2)
against one workgroup per 64 KiB tile + device-side fold (6, default) at 28..34 qubits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcmrf_amd import _lib
for W in [int(a) for a in sys.argv[1:]] or [28, 31, 34]:
    eng = _lib.Engine(W)
    eng.init_uniform((1 << W) - 1)
    qs = list(range(0, 14))
    tab = np.random.RandomState(1).uniform(-1, 1, size=2 ** len(qs))
    line = "W=%d" % W
    ref = None
    for bv in (2, 6):
        eng.set_option("blocksum_variant", bv)
        for _ in range(2): out = eng.expect_diag(qs, tab, 1 << (W - 1), 0)
        eng.sync(); eng.timer_begin()
        for _ in range(5): out = eng.expect_diag(qs, tab, 1 << (W - 1), 0)
        ms = eng.timer_end() / 5
        ref = ref or out
        line += "  | variant %d: %.3f ms = %.3f of 8 TB/s (%.15g, %.15g)" % (bv, ms, 16 * 2.0 ** W / ms / 1e6 / 8000, out[0], out[1])
        assert abs(out[0] - ref[0]) < 1e-12 and abs(out[1] - ref[1]) < 1e-12
    print(line, flush=True)
    eng.close()
