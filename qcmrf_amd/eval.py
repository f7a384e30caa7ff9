"""The reference's result evaluation (/root/reference/eval.py, ``--mode file``) without
``kiopto_native`` / PrettyTable / IBM imports.

    python -m qcmrf_amd.eval --results result_simulation_0.5.json --scale 0.5 [--resdir .]

Same steps: regenerate the thetas of the 7 graphs x 10 reps from ``np.random.seed(1984)``
(eval.py:3,31-40), load a result file -- a list of counts dicts (norm 10 000) or
``{"quasi_dists": [...]}`` (norm 1) (eval.py:49-60) --, compare the conditional pmf of the
all-ancillas-zero outcomes with the exact Gibbs pmf (eval.py:115-128) and print per graph:
mean fidelity +- std, best fidelity, mean success rate +- std (eval.py:136-146).
``--mode gibbs|pam`` need kiopto's samplers and are not provided.
"""
from __future__ import annotations

import argparse
import errno
import json
import os

import numpy as np

from . import mrf
from .qcmrf import fidelity as F
from .workloads import REFERENCE_GRAPHS as GRAPHS

REPS = 10


def regenerate_thetas(scale):
    from scipy.stats import halfnorm
    np.random.seed(1984)
    thetas = {}
    for j, C in enumerate(GRAPHS):
        d = mrf.dimension(C)
        thetas[j] = [(-halfnorm.rvs(loc=0, scale=float(scale), size=d)).tolist() for _ in range(REPS)]
    return thetas


def evaluate(dists, thetas, norm):
    """rows of (graph, mean F, std F, best F, mean delta, std delta) in graph order"""
    rows, idx = [], 0
    for j, C in enumerate(GRAPHS):
        N = 2 ** mrf.num_vertices(C)
        L_F, L_delta = [], []
        for i in range(REPS):
            p, _ = mrf.gibbs_pmf(C, thetas[j][i])
            q = np.zeros(N)
            Z = 0
            for k, v in dists[idx].items():
                kid = int(k, 2) if isinstance(k, str) else int(k)
                if kid < N:
                    q[kid] = v
                    Z += v
            q /= Z
            L_F.append(max(min(F(p, q), 1), 0))
            L_delta.append(Z / norm)
            idx += 1
        rows.append((C, float(np.mean(L_F)), float(np.std(L_F)), float(np.max(L_F)),
                     float(np.mean(L_delta)), float(np.std(L_delta))))
    return rows


def format_table(rows):
    head = ["graph", "fidelity", "max fidelity", "success rate"]
    body = [[str(C), "%.3f ±%.3f" % (mf, sf), "%.3f" % bf, "%.3f ±%.3f" % (md, sd)]
            for C, mf, sf, bf, md, sd in rows]
    w = [max(len(r[c]) for r in [head] + body) for c in range(4)]
    line = "+" + "+".join("-" * (x + 2) for x in w) + "+"
    fmt = lambda r: "| " + " | ".join(s.center(x) for s, x in zip(r, w)) + " |"
    return "\n".join([line, fmt(head), line] + [fmt(r) for r in body] + [line])


def main(argv=None):
    ap = argparse.ArgumentParser(prog="QCMRF result evaluation.", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument("--results", type=str, default="result_simulation.json", help="result file (counts list or quasi_dists)")
    ap.add_argument("--scale", type=str, default="0.1", help="variance of the parameter prior")
    ap.add_argument("--mode", type=str, default="file", help="file (gibbs / pam need kiopto_native: unsupported)")
    ap.add_argument("--resdir", type=str, default=None, help="directory of the result file (default ./res_<scale>)")
    args = ap.parse_args(argv)
    if args.mode != "file":
        raise SystemExit("--mode %s needs kiopto_native's samplers; only --mode file is provided" % args.mode)
    thetas = regenerate_thetas(args.scale)
    fname = os.path.join(args.resdir if args.resdir is not None else "./res_" + args.scale, args.results)
    if not os.path.isfile(fname):
        raise FileNotFoundError(errno.ENOENT, os.strerror(errno.ENOENT), fname)
    data = json.load(open(fname))
    if isinstance(data, dict) and "quasi_dists" in data:
        dists, norm = data["quasi_dists"], 1
    else:
        dists, norm = data, 10_000
    rows = evaluate(dists, thetas, norm)
    print(format_table(rows))
    return rows


if __name__ == "__main__":
    main()
