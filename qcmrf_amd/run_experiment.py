"""The reference's experiment driver (/root/reference/run_experiment.py:1-61) on this engine.

    python -m qcmrf_amd.run_experiment [--scale 0.5] [--shots 10000] [--reps 10] [--outdir .]

Same steps, same files: seed numpy with 1984, draw theta = -halfnorm.rvs(scale) for the 7
hard-coded graphs x REPS, dump ``models_<SCALE>.json``, build the 70 ``QCMRF`` circuits, run them
on the simulator with SHOTS shots, dump ``result_simulation_<SCALE>.json``.  Differences:
``d = sum 2^|C|`` is computed directly (the reference asks the closed-source ``kiopto_native``
for ``len(px.weights(...))``, which is the same number); ``transpile`` is applied only when Qiskit
is importable (the engine ingests the nested circuits directly); the unreachable IBM-hardware
tail (run_experiment.py:63-88) is not reproduced.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.5)
    ap.add_argument("--shots", type=int, default=10000)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--outdir", default=".")
    ap.add_argument("--seed-simulator", type=int, default=None)
    args = ap.parse_args(argv)

    np.random.seed(1984)
    from scipy.stats import halfnorm
    from . import QCMRF, Aer, HAVE_QISKIT
    from .workloads import REFERENCE_GRAPHS as GRAPHS

    THETAS = {}
    for j, C in enumerate(GRAPHS):
        d = sum(2 ** len(c) for c in C)
        for _ in range(args.reps):
            theta = -halfnorm.rvs(loc=0, scale=args.scale, size=d)
            THETAS.setdefault(j, []).append(theta.tolist())
    with open(os.path.join(args.outdir, "models_" + str(args.scale) + ".json"), "w") as f:
        f.write(json.dumps({"GRAPHS": GRAPHS, "THETAS": THETAS}, indent=4))

    CIRCS = [QCMRF(C, THETAS[j][i], with_measurements=True) for j, C in enumerate(GRAPHS) for i in range(args.reps)]
    if HAVE_QISKIT:                                   # pragma: no cover - Qiskit absent in this image
        from qiskit import transpile
        CIRCS = transpile(CIRCS, basis_gates=['cx', 'id', 'rz', 'sx', 'x'])

    simulator = Aer.get_backend('qasm_simulator')
    t0 = time.perf_counter()
    result = simulator.run(CIRCS, shots=args.shots, seed_simulator=args.seed_simulator).result()
    counts = result.get_counts()
    dt = time.perf_counter() - t0
    print("%d circuits x %d shots: %.3f s in run().result().get_counts() (%.2f ms per circuit)"
          % (len(CIRCS), args.shots, dt, dt / len(CIRCS) * 1e3), file=sys.stderr)
    with open(os.path.join(args.outdir, "result_simulation_" + str(args.scale) + ".json"), "w") as f:
        f.write(json.dumps(counts, indent=4))
    return counts


if __name__ == "__main__":
    main()
