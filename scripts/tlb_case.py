"""GPU box: one W-qubit grid MRF run both ways -- the default path (k_init_prod, write-only generator) and full-width gate
sweeps (fold_fresh off: init-fused pass + read/write k_multi pass) -- for rocprofv3 --pmc (scripts/pmc_tlb.sh).
   python scripts/tlb_case.py W [alloc]      alloc=arena: one hipMalloc of (nearly) all free device memory first, released
                                             before the engine allocates (does the allocation history matter?)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcmrf_amd import QCMRF, workloads as wl
from qcmrf_amd.backend import QsvBackend
W = int(sys.argv[1])
C = wl.for_width(W)
qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
be = QsvBackend()
for fold in (True, False):
    for i in range(3):
        r = be.run(qc, shots=4096, seed_simulator=5 + i, fold_fresh=fold, profile=True).result()
    st = r.metadata(0)["stats"]["kinds"]
    print("W=%d fold=%s" % (W, fold), {k: "%.3f ms %.0f GB/s" % (v["ms"] / v["launches"], v["bytes"] / v["ms"] / 1e6) for k, v in st.items() if v["launches"]}, flush=True)
be.close()
