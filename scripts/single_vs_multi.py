"""GPU box: ONE gate per program at 28 qubits -- its dedicated kernel (k_pair / k_lowt / k_diag / k_mux ...) against
the same gate as a one-op k_multi pass (engine option single_shortcut = 0), by target position."""
import sys
sys.path.insert(0, ".")
import numpy as np
from qcmrf_amd import _lib, ir, program
W = int(sys.argv[1]) if len(sys.argv) > 1 else 28
rs = np.random.RandomState(0)
def ru():
    q, _ = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2)); return q
eng = _lib.Engine(W)
eng.init_uniform((1 << W) - 1)
def t(ops, reps=6):
    rec, data = program.encode(ops)
    for _ in range(2): eng.exec(rec, data)
    eng.sync(); eng.timer_begin()
    for _ in range(reps): eng.exec(rec, data)
    return eng.timer_end() / reps
cases = [("1q_t%02d" % q, [ir.op_u(q, ru())]) for q in (0, 3, 5, 6, 8, 10, 11, 12, 13, 17, 20, 24, 27)]
cases += [("x_t27", [ir.op_x(27)]), ("cx_c3_t26", [ir.op_x(26, [3])]), ("cx_c26_t3", [ir.op_x(3, [26])]),
          ("ccx_c1c5_t12", [ir.op_x(12, [1, 5], [1, 0])]), ("cp_c12_t27", [ir.op_mcphase([12, 27], 0.3)]),
          ("diag3", [ir.op_diag([2, 9, 20], np.exp(1j * rs.randn(8)))]),
          ("mux_c3_t27", [ir.op_mux([3], 27, np.array([ru(), ru()]))]), ("mux_c3_t13", [ir.op_mux([3], 13, np.array([ru(), ru()]))])]
B = 32.0 * 2 ** W
for name, ops in cases:
    out = []
    for sc in (1, 0):
        eng.set_option("single_shortcut", sc)
        eng.set_option("xframe", 0)
        ms = t(ops)
        out.append("%s %.3f ms %.3f" % ("kernel" if sc else "k_multi", ms, B / ms / 1e6 / 8000))
    print("%-14s %s" % (name, "   ".join(out)), flush=True)
eng.close()
