"""Why did the lowered bench leg show the SAME generator launch at 41.7 ms when the main leg has it at 38.1 ms?
Runs the constructed circuit and its lowered form alternately on one backend, with and without per-kernel profiling,
and prints evolve time and the generator's event time for every run.  GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qcmrf_amd import QCMRF, workloads as wl
from qcmrf_amd.backend import QsvBackend
from qcmrf_amd.transpile import transpile

name, C = wl.baseline_config(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
low = transpile(qc)
be = QsvBackend()
be.run(qc, shots=4096, seed_simulator=1).result()


def go(label, circ, n, gap_ms=0.0, **kw):
    row = []
    for i in range(n):
        if gap_ms:
            time.sleep(gap_ms * 1e-3)                      # an idle device between runs, as a slow host compile leaves it
        t0 = time.perf_counter()
        r = be.run(circ, shots=4096, seed_simulator=10 + i, **kw).result()
        dt = (time.perf_counter() - t0) * 1e3
        m = r.metadata(0)
        k = m.get("stats", {}).get("kinds", {}).get("init_prod")
        row.append("%.1f/%.2f/%.2f%s" % (dt, m["time_compile"] * 1e3, m["time_evolve"] * 1e3, "/k%.2f" % (k["ms"] / k["launches"]) if k else ""))
    print("%-34s step/compile/evolve[/kernel] ms: %s" % (label, "  ".join(row)), flush=True)


for rnd in range(2):
    go("constructed", qc, 5)
    go("lowered", low, 5)
    for gap in (5, 10, 20, 100):
        go("constructed, %d ms idle before" % gap, qc, 5, gap_ms=gap)
    go("constructed profile=True", qc, 5, profile=True)
    go("lowered profile=True", low, 5, profile=True)
    go("lowered zero_tracking=0 profile", low, 3, profile=True, engine_options={"zero_tracking": 0})
    go("constructed zero_tracking=0 profile", qc, 3, profile=True, engine_options={"zero_tracking": 0})
