"""GPU box, for rocprofv3 --pmc (scripts/kq_lds_pmc.sh): the dense 5-qubit gate on targets 0..4 at 28 qubits through the
operand-layout kernel (kq_variant 3), the LDS-staged one (6), the LDS-staged one without its products (kq_debug 1), and for
comparison a plain read+write stream of the same bytes (a dense one-qubit gate on bit 20).  Three launches each."""
import os, sys
os.environ.setdefault("QSV_MEASUREMENT_KNOBS", "1")         # kq_debug: kernels with a part of their work left out (timing only)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcmrf_amd import _lib
W = 28
rs = np.random.RandomState(0)
q, _ = np.linalg.qr(rs.randn(32, 32) + 1j * rs.randn(32, 32))
h = np.array([[1, 1], [1, -1]], dtype=np.complex128) / np.sqrt(2)
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
targets = [int(t) for t in os.environ.get("KQ_TARGETS", "0,1,2,3,4").split(",")]
for variant, dbg in ((3, 0), (6, 0), (6, 1), (8, 0)):
    eng.set_option("kq_variant", variant)
    eng.set_option("kq_debug", dbg)
    eng.sync(); eng.timer_begin()
    for _ in range(3): eng.apply_kq(targets, q)
    print("variant %d debug %d: %.3f ms" % (variant, dbg, eng.timer_end() / 3), flush=True)
eng.sync(); eng.timer_begin()
for _ in range(3): eng.apply_1q(20, h)
print("1q on bit 20: %.3f ms" % (eng.timer_end() / 3), flush=True)
eng.close()
