"""GPU box: cost of creating / destroying a small engine (what a batch of circuits of different
widths pays per width)."""
import sys, time
sys.path.insert(0, ".")
from qcmrf_amd import _lib
_lib.load(); _lib.device_count()
t0 = time.perf_counter(); e = _lib.Engine(8); print("first create %.2f ms" % ((time.perf_counter() - t0) * 1e3)); e.close()
for W in (4, 8, 12, 20):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); e = _lib.Engine(W); t1 = time.perf_counter(); e.close(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    print("W=%d create %.2f ms  close %.2f ms" % (W, min(t[0] for t in ts) * 1e3, min(t[1] for t in ts) * 1e3))
