"""GPU box: one read+write k_multi pass (one multiplexed gate per register bit) at W qubits for tiles of 3, 4 and 5 register
bits on the LOW bits (6..) and on the TOP bits of the shard: time per launch; under rocprofv3 --pmc (scripts/pmc_tlb_rows.sh) the
address-translation counters of every launch, in this order.  Question: is the 34-qubit pass's translation-miss rate a matter
of how many rows 8 GiB apart one wave touches?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcmrf_amd import _lib, ir, program
W = int(sys.argv[1]) if len(sys.argv) > 1 else 34
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rs = np.random.RandomState(0)
def rx():
    a = rs.rand() * 3
    return np.array([[np.cos(a), -1j * np.sin(a)], [-1j * np.sin(a), np.cos(a)]])
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
T = W - 1
R = lambda a, b: list(range(a, b + 1))
cases = [("low 5: bits 6-10", R(6, 10), 5), ("top 5", R(T - 4, T), 5), ("top 4", R(T - 3, T), 4), ("top 3", R(T - 2, T), 3),
         ("low 4: bits 6-9", R(6, 9), 4), ("mid 5: bits 20-24", R(20, 24), 5), ("top 2 + low 3", [6, 7, 8, T - 1, T], 5)]
for label, regs, r in cases:
    eng.set_option("multi_r", r)
    sel = [q for q in (25, 26, 19, 21) if q not in regs][:2]
    ops = [ir.op_mux(sel, t, np.array([rx() for _ in range(4)])) for t in regs]
    rec, data = program.encode(ops)
    eng.exec(rec, data)
    eng.sync(); eng.reset_stats(); eng.timer_begin()
    for _ in range(reps): eng.exec(rec, data)
    ms = eng.timer_end() / reps
    n = sum(v["launches"] for v in eng.stats()["kinds"].values()) / reps
    print("CASE %-20s R=%d launches/exec %.1f  %8.3f ms  %5.0f GB/s  %.3f of 8 TB/s" % (label, r, n, ms, 32 * 2.0 ** W / ms / 1e6, 32 * 2.0 ** W / ms / 1e6 / 8000), flush=True)
eng.close()
