"""GPU box: one general (MODE 0, R = 4) pass with 60 ops of one kind on a 26-qubit state, once per
kind -- run under rocprofv3 --pmc to read the instruction mix of k_multi<4,false,0> per op kind."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcmrf_amd import _lib, ir, program
W = 26
rs = np.random.RandomState(0)
def ru():
    q, _ = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2)); return q
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
eng.set_option("multi_r", 4)
eng.set_option("xframe", 0)
tg = [8, 9, 10, 11]
kinds = {
    "ccx_lane": lambda i: ir.op_x(tg[i % 4], [2, 3], [1, 0]),
    "ccx_block": lambda i: ir.op_x(tg[i % 4], [20, 21], [1, 0]),
    "ccx_reg": lambda i: ir.op_x(tg[i % 4], [tg[(i + 1) % 4], tg[(i + 2) % 4]], [1, 0]),
    "cp_regreg": lambda i: ir.op_mcphase([tg[(i + 1) % 4], tg[i % 4]], 0.3),
    "u_dense": lambda i: ir.op_u(tg[i % 4], ru()),
    "none": None,
}
which = sys.argv[1] if len(sys.argv) > 1 else "ccx_lane"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ops = [kinds[which](i) for i in range(N)] if kinds[which] else [ir.op_x(tg[0], [2, 3], [1, 0])]
rec, data = program.encode(ops)
for _ in range(3):
    eng.exec(rec, data)
eng.sync()
print(which, N, eng.stats()["kinds"])
eng.close()
