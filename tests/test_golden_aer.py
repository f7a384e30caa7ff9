"""CPU: the oracle against the reference's own committed Qiskit-Aer outputs
(res_{0.1,0.25,0.5}/result_simulation.json -> tests/golden/aer_counts_*.json, 210 circuits x
10 000 unseeded shots) and the eval.py table derived from them.  These fixtures are the only
Aer outputs that exist for this path; they pin the distribution and every bit/ancilla/theta
convention statistically (a wrong convention gives chi^2/dof of 40..1400 instead of ~1)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import closed_form as cf


def chi2_per_dof(counts, p, shots=10000):
    W = int(np.log2(p.size))
    obs = np.zeros(p.size)
    for k, v in counts.items():
        assert len(k) == W
        obs[int(k, 2)] += v
    assert obs[p == 0].sum() == 0, "Aer produced a key outside the analytic support"
    # cells with an expectation below 5 counts are pooled into one bin (chi^2 validity)
    e = p * shots
    big = e >= 5
    o_b, e_b = list(obs[big]), list(e[big])
    if e[~big].sum() > 0:
        o_b.append(obs[~big].sum())
        e_b.append(e[~big].sum())
    o_b, e_b = np.array(o_b), np.array(e_b)
    return ((o_b - e_b) ** 2 / e_b).sum() / max(len(e_b) - 1, 1)


@pytest.mark.parametrize("scale", ["0.1", "0.25", "0.5"])
def test_closed_form_matches_all_committed_aer_runs(models, aer_counts, scale):
    idx, chis = 0, []
    for j, C in enumerate(models[scale]["GRAPHS"]):
        for rep in range(10):
            th = models[scale]["THETAS"][str(j)][rep]
            chis.append(chi2_per_dof(aer_counts[scale][idx], cf.probabilities(C, th)))
            idx += 1
    assert idx == 70
    assert 0.88 < np.mean(chis) < 1.12, np.mean(chis)
    assert max(chis) < 7.0          # one 3-dof case sits at 6.4 (p = 2.5e-4: expected once in 210 runs at the 5% level)


def _wrong_clique_lsb_first(C, th):
    """same closed form but with the clique-state index read LSB first"""
    n, m, W, dim = cf.model_shape(C)
    perm, off = [], 0
    for Cl in C:
        k = len(Cl)
        for y in range(2 ** k):
            perm.append(off + int(format(y, "0%db" % k)[::-1], 2))
        off += 2 ** k
    return cf.probabilities(C, [th[i] for i in perm])


def test_wrong_conventions_are_rejected_by_the_aer_data(models, aer_counts):
    m = models["0.5"]
    # graph 3 = [[0,1],[1,2],[2,3],[3,4]]: asymmetric enough to see every convention
    j, C = 3, m["GRAPHS"][3]
    base = 30
    good, lsb, rev = [], [], []
    n, mm, W, dim = cf.model_shape(C)
    for rep in range(10):
        th = m["THETAS"][str(j)][rep]
        cnt = aer_counts["0.5"][base + rep]
        good.append(chi2_per_dof(cnt, cf.probabilities(C, th)))
        p = _wrong_clique_lsb_first(C, th)
        obs = np.zeros(2 ** W)
        for k, v in cnt.items():
            obs[int(k, 2)] += v
        sel = p > 0
        lsb.append((((obs[sel] - p[sel] * 1e4) ** 2) / (p[sel] * 1e4)).sum() / (sel.sum() - 1))
        # variable order reversed: read the variable bits of every key backwards
        obs2 = np.zeros(2 ** W)
        for k, v in cnt.items():
            obs2[int(k[:W - n] + k[W - n:][::-1], 2)] += v
        p0 = cf.probabilities(C, th)
        sel = p0 > 0
        rev.append((((obs2[sel] - p0[sel] * 1e4) ** 2) / (p0[sel] * 1e4)).sum() / (sel.sum() - 1))
    assert np.mean(good) < 1.3
    assert np.mean(lsb) > 20 and np.mean(rev) > 20


def test_eval_table_fixture(models, aer_counts):
    """eval.py:115-128 arithmetic on the committed counts reproduces tests/golden/eval_table.json
    and the analytic success rate Z/2^n."""
    from qcmrf_amd import fidelity
    table = json.load(open(os.path.join(GOLDEN, "eval_table.json")))
    for scale in ("0.1", "0.25", "0.5"):
        idx = 0
        for j, C in enumerate(models[scale]["GRAPHS"]):
            n = cf.model_shape(C)[0]
            F, d_emp, d_an = [], [], []
            for rep in range(10):
                th = models[scale]["THETAS"][str(j)][rep]
                p, Z = cf.gibbs_pmf(C, th)
                q = np.zeros(2 ** n)
                for key, v in aer_counts[scale][idx].items():
                    if int(key, 2) < 2 ** n:
                        q[int(key, 2)] = v
                F.append(min(max(fidelity(p, q / q.sum()), 0), 1))
                d_emp.append(q.sum() / 10000)
                d_an.append(Z / 2 ** n)
                idx += 1
            row = table[scale][j]
            assert row["graph"] == C
            assert abs(np.mean(F) - row["fidelity"]) < 1e-12
            assert abs(np.mean(d_emp) - row["delta_emp"]) < 1e-12
            assert abs(np.mean(d_an) - row["delta_an"]) < 1e-12
            assert abs(np.mean(d_emp) - np.mean(d_an)) < 0.01 and np.mean(F) > 0.995


def test_theta_fixture_is_the_reference_generating_law(models):
    """run_experiment.py:3,23-33: np.random.seed(1984); -halfnorm.rvs(scale, size=sum 2^|C|)"""
    from scipy.stats import halfnorm
    for scale in ("0.1", "0.25", "0.5"):
        np.random.seed(1984)
        for j, C in enumerate(models[scale]["GRAPHS"]):
            d = sum(2 ** len(c) for c in C)
            for rep in range(10):
                th = (-halfnorm.rvs(loc=0, scale=float(scale), size=d)).tolist()
                assert th == models[scale]["THETAS"][str(j)][rep]
