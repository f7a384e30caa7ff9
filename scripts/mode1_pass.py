"""GPU box: one MODE 1 pass (general complex 2x2 tables, R = 5 register targets, read + write of a
28-qubit shard) and one MODE 0 pass (masked CX / CP / dense 2x2, R = 4), a few times each -- run
under rocprofv3 --pmc by scripts/profile_round.sh to compare their HBM traffic with the algorithmic
32 B per amplitude (a 52 B/lane spill in MODE 1; now 16 B/lane of scratch)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcmrf_amd import _lib, ir, program
W = int(sys.argv[1]) if len(sys.argv) > 1 else 28
rs = np.random.RandomState(0)
def ru():
    q, _ = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2)); return q
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
sel = [20, 21]
mode1 = [ir.op_mux(sel, t, np.array([ru() for _ in range(4)])) for t in range(6, 11)]
mode0 = [ir.op_x(8, [2, 20], [1, 0]), ir.op_u(9, ru()), ir.op_mcphase([3, 10], 0.3), ir.op_x(10, [9]), ir.op_u(11, ru(), [4], [1])]
for name, ops, r in (("mode1", mode1, 5), ("mode0", mode0, 4)):
    eng.set_option("multi_r", r)
    rec, data = program.encode(ops)
    for _ in range(2): eng.exec(rec, data)
    eng.sync(); eng.reset_stats(); eng.timer_begin()
    for _ in range(6): eng.exec(rec, data)
    ms = eng.timer_end() / 6
    st = eng.stats()["kinds"]
    print(name, "launches/exec", {k: v["launches"] / 6 for k, v in st.items()}, "%.3f ms/exec" % ms, "%.0f GB/s" % (32 * 2.0 ** W / ms / 1e6), flush=True)
eng.close()
