"""qcmrf_amd -- MI355X-native fp64 statevector engine behind the np84/qcmrf entry points.

    from qcmrf_amd import QCMRF, Aer, extract_probs, fidelity, KL
    qc = QCMRF([[0, 1], [1, 2]], theta)                       # reference constructor
    counts = Aer.get_backend('qasm_simulator').run(qc, shots=10000).result().get_counts()

Host code is Python + numpy + ctypes; every amplitude is touched only by hand-written HIP
kernels in ``csrc/`` (libqsv.so, C ABI in include/qsv.h).  No PyTorch, no Triton, no CPU fallback.
"""
from .qcmrf import QCMRF, fidelity, KL, extract_probs, HAVE_QISKIT
from .backend import Aer, QsvBackend, get_backend

__all__ = ["QCMRF", "fidelity", "KL", "extract_probs", "Aer", "QsvBackend", "get_backend", "HAVE_QISKIT"]
__version__ = "0.1.0"
