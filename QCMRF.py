"""Import shim so reference-style scripts keep working unchanged:

    from QCMRF import QCMRF, extract_probs, fidelity as F, KL      (run_experiment.py:1)
"""
from qcmrf_amd.qcmrf import QCMRF, extract_probs, fidelity, KL  # noqa: F401
