#!/usr/bin/env python3
"""Regenerate tests/golden/* from the reference's committed DATA files.

Run in the build container only (/root/reference does not exist on the GPU box):

    python tests/golden/make_golden.py

What it writes (all data, no reference source text):

models_<s>.json        the reference's model files res_<s>/models*.json (GRAPHS + THETAS),
                       re-serialised compactly AFTER checking that they are reproduced bit
                       for bit by the generating law of /root/reference/run_experiment.py:3,23-33
                       (np.random.seed(1984); -halfnorm.rvs(scale=s, size=sum 2^|C|)).
aer_counts_<s>.json    res_<s>/result_simulation.json: the Qiskit-Aer count dictionaries the
                       reference's authors committed (70 circuits x 10 000 shots, unseeded).
                       These are the only Aer outputs that exist for this path.
config1.json           BASELINE config 1 (graph [[0,1],[1,2],[2,3]], THETAS["2"][0], scale 0.5):
                       oracle closed-form probability vector + amplitudes (W = 8).
eval_table.json        per graph x scale: mean fidelity / empirical success rate / analytic
                       success rate of the committed Aer counts (eval.py:115-128 arithmetic).
"""
import json
import os
import sys

import numpy as np
from scipy.stats import halfnorm

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import closed_form as cf          # noqa: E402

REF = "/root/reference"
SCALES = {"0.1": "res_0.1/models_0.1.json", "0.25": "res_0.25/models_0.25.json",
          "0.5": "res_0.5/models.json"}
REPS = 10


def regenerate(graphs, scale):
    np.random.seed(1984)
    out = {}
    for j, C in enumerate(graphs):
        d = sum(2 ** len(c) for c in C)
        out[str(j)] = [(-halfnorm.rvs(loc=0, scale=scale, size=d)).tolist() for _ in range(REPS)]
    return out


def fidelity(P, Q):
    sel = (P > 0) & (Q > 0)
    return float(np.sqrt(P[sel] * Q[sel]).sum() ** 2)


def main():
    table = {}
    for s, rel in SCALES.items():
        models = json.load(open(os.path.join(REF, rel)))
        regen = regenerate(models["GRAPHS"], float(s))
        for k in models["THETAS"]:
            assert models["THETAS"][k] == regen[k], "theta regeneration mismatch %s/%s" % (s, k)
        json.dump(models, open(os.path.join(HERE, "models_%s.json" % s), "w"), separators=(",", ":"))
        counts = json.load(open(os.path.join(REF, "res_%s/result_simulation.json" % s)))
        assert len(counts) == 70 and all(sum(c.values()) == 10000 for c in counts)
        json.dump(counts, open(os.path.join(HERE, "aer_counts_%s.json" % s), "w"),
                  separators=(",", ":"))
        idx, rows = 0, []
        for j, C in enumerate(models["GRAPHS"]):
            n = cf.model_shape(C)[0]
            F, d_emp, d_an = [], [], []
            for i in range(REPS):
                th = models["THETAS"][str(j)][i]
                p, Z = cf.gibbs_pmf(C, th)
                q = np.zeros(2 ** n)
                for key, v in counts[idx].items():
                    kid = int(key, 2)
                    if kid < 2 ** n:
                        q[kid] = v
                z = q.sum()
                F.append(min(max(fidelity(p, q / z), 0), 1))
                d_emp.append(z / 10000)
                d_an.append(Z / 2 ** n)
                idx += 1
            rows.append({"graph": C, "fidelity": float(np.mean(F)),
                         "delta_emp": float(np.mean(d_emp)), "delta_an": float(np.mean(d_an))})
        table[s] = rows
    json.dump(table, open(os.path.join(HERE, "eval_table.json"), "w"), indent=1)

    models = json.load(open(os.path.join(REF, SCALES["0.5"])))
    C, th = models["GRAPHS"][2], models["THETAS"]["2"][0]
    a = cf.amplitudes(C, th)
    json.dump({"cliques": C, "theta": th, "W": 8,
               "probabilities": cf.probabilities(C, th).tolist(),
               "amp_re": a.real.tolist(), "amp_im": a.imag.tolist()},
              open(os.path.join(HERE, "config1.json"), "w"))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
