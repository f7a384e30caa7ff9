"""CPU: the oracle against itself (closed form == numpy gate level == C gate level) and against
the committed golden vector of BASELINE config 1."""
import numpy as np
import pytest

from conftest import random_theta
from oracle import closed_form as cf, gate_stream as gs, sv_numpy as sv, cref


def test_closed_form_equals_gate_level_numpy(models):
    for s in ("0.1", "0.25", "0.5"):
        for j, C in enumerate(models[s]["GRAPHS"]):
            for rep in (0, 4, 9):
                th = models[s]["THETAS"][str(j)][rep]
                W = cf.model_shape(C)[2]
                state = sv.run_stream(gs.reference_stream(C, th), W)
                assert np.abs(state - cf.amplitudes(C, th)).max() < 1e-14
                assert np.abs(sv.probabilities(state) - cf.probabilities(C, th)).max() < 1e-14
                assert abs(cf.probabilities(C, th).sum() - 1) < 1e-13


def test_config1_golden(config1):
    C, th = config1["cliques"], config1["theta"]
    assert np.array_equal(cf.probabilities(C, th), np.array(config1["probabilities"]))
    a = cf.amplitudes(C, th)
    assert np.array_equal(a.real, np.array(config1["amp_re"])) and np.array_equal(a.imag, np.array(config1["amp_im"]))
    assert np.abs(sv.run_stream(gs.reference_stream(C, th), 8) - a).max() < 1e-15


def test_c_oracle_equals_numpy_and_closed_form():
    C = gs.chain_cliques(8)                         # W = 16
    th = random_theta(cf.model_shape(C)[3])
    ops = gs.reference_stream(C, th)
    r = cref.RefState(16)
    r.run_stream(ops)
    assert np.abs(r.state - sv.run_stream(ops, 16)).max() < 1e-14
    assert np.abs(r.state - cf.amplitudes(C, th)).max() < 1e-14
    assert abs(r.norm() - 1) < 1e-13
    # the fused op kinds of the C oracle against numpy
    rs = np.random.RandomState(0)
    v = rs.randn(2 ** 10) + 1j * rs.randn(2 ** 10)
    v /= np.linalg.norm(v)
    r = cref.RefState(10)
    r.state[:] = v
    ref = v.copy()
    mats = np.array([np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2))[0] for _ in range(8)])
    r.apply_mux([1, 7, 4], 8, mats); sv.apply_mux(ref, [1, 7, 4], 8, mats)
    tab = np.exp(1j * rs.randn(8))
    r.apply_diag([9, 0, 3], tab); sv.apply_diag(ref, [9, 0, 3], tab)
    r.apply_1q(2, mats[0], [5, 6], [0, 1]); sv.apply_1q(ref, 2, mats[0], [5, 6], [0, 1])
    r.apply_mcphase([0, 2, 9], 0.7, [1, 0, 1]); sv.apply_mcphase(ref, [0, 2, 9], 0.7, [1, 0, 1])
    assert np.abs(r.state - ref).max() < 1e-14
    got = r.marginal([3, 8], 1 << 5, 1 << 5)
    idx = np.arange(2 ** 10)
    sel = ((idx >> 5) & 1) == 1
    j = ((idx >> 3) & 1) | (((idx >> 8) & 1) << 1)
    assert np.abs(got - np.bincount(j[sel], weights=np.abs(ref[sel]) ** 2, minlength=4)).max() < 1e-14


def test_c_oracle_config2_chain_w20():
    C = gs.chain_cliques(10)
    th = random_theta(36)
    r = cref.RefState(20)
    r.run_stream(gs.reference_stream(C, th))
    assert np.abs(r.state - cf.amplitudes(C, th)).max() < 1e-14


def test_distribution_keys_and_gibbs(models):
    C = models["0.5"]["GRAPHS"][5]
    th = models["0.5"]["THETAS"]["5"][2]
    n, m, W, dim = cf.model_shape(C)
    d = cf.distribution(C, th)
    assert abs(sum(d.values()) - 1) < 1e-12 and all(len(k) == W for k in d)
    assert all(k[W - 1 - n] == "0" for k in d)                 # classical bit n is never written
    p, Z = cf.gibbs_pmf(C, th)
    ok = np.array([d["0" * (m + 1) + format(x, "0%db" % n)] for x in range(2 ** n)])
    assert np.abs(ok / ok.sum() - p).max() < 1e-14
    assert abs(ok.sum() - cf.success_probability(C, th)) < 1e-14


def test_grid_and_chain_shapes():
    assert cf.model_shape(gs.grid_cliques(2, 6, drop_last=1)) == (12, 15, 28, 60)
    assert cf.model_shape(gs.grid_cliques(2, 7)) == (14, 19, 34, 76)
    assert cf.model_shape(gs.chain_cliques(10)) == (10, 9, 20, 36)
    from qcmrf_amd import workloads
    assert workloads.grid(2, 6, 1) == gs.grid_cliques(2, 6, 1) and workloads.chain(10) == gs.chain_cliques(10)
    assert workloads.width(workloads.baseline_config(3)[1]) == 31
