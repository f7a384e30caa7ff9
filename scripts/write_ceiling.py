"""GPU box: write-only ceilings -- k_init (plain fill), the generator with one trivial factor, and
the generator with the 34-qubit circuit's 19 factors."""
import sys
sys.path.insert(0, ".")
import numpy as np
from qcmrf_amd import _lib, ir, program, QCMRF, workloads as wl
from qcmrf_amd.backend import QsvBackend
W = int(sys.argv[1]) if len(sys.argv) > 1 else 34
eng = _lib.Engine(W)
def timeit(f, n=4):
    for _ in range(2): f()
    eng.sync(); eng.timer_begin()
    for _ in range(n): f()
    return eng.timer_end() / n
ms = timeit(lambda: eng.init_uniform((1 << W) - 1))
print("k_init                        %8.3f ms  %d GB/s" % (ms, 16 * 2.0 ** W / ms / 1e6), flush=True)
for nf in (1, 5, 19):
    ops = [ir.op_init((1 << W) - 1)] + [ir.op_diag([12 + k, 1 + (k % 5)], np.exp(1j * np.arange(4) * (k + 1))) for k in range(nf)]
    rec, data = program.encode(ops)
    ms = timeit(lambda: eng.exec(rec, data))
    print("generator, %2d uniform factors %8.3f ms  %d GB/s" % (nf, ms, 16 * 2.0 ** W / ms / 1e6), flush=True)
for nf in (1, 5):
    ops = [ir.op_init((1 << W) - 1)] + [ir.op_diag([6 + k, 20 + k], np.exp(1j * np.arange(4) * (k + 1))) for k in range(nf)]
    rec, data = program.encode(ops)
    ms = timeit(lambda: eng.exec(rec, data))
    print("generator, %2d reg-bit factors %8.3f ms  %d GB/s" % (nf, ms, 16 * 2.0 ** W / ms / 1e6), flush=True)
for opt in ("fused_sums=0", "fused_sums=1"):
    k, v = opt.split("=")
    eng.set_option(k, int(v))
    ops = [ir.op_init((1 << W) - 1)] + [ir.op_diag([12, 1], np.exp(1j * np.arange(4)))]
    rec, data = program.encode(ops)
    ms = timeit(lambda: eng.exec(rec, data))
    print("generator, 1 factor, %s  %8.3f ms  %d GB/s" % (opt, ms, 16 * 2.0 ** W / ms / 1e6), flush=True)
