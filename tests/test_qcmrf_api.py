"""CPU: the kept constructor / helper surface of /root/reference/QCMRF.py (no GPU involved)."""
import json

import numpy as np
import pytest

from oracle import closed_form as cf
from qcmrf_amd import QCMRF, KL, extract_probs, fidelity
import QCMRF as shim


def test_shim_module_exports_reference_names():
    assert shim.QCMRF is QCMRF and shim.fidelity is fidelity and shim.KL is KL and shim.extract_probs is extract_probs


def test_constructor_attributes(models):
    for j, C in enumerate(models["0.5"]["GRAPHS"]):
        th = models["0.5"]["THETAS"][str(j)][0]
        qc = QCMRF(C, th, with_measurements=True)
        n, m, W, dim = cf.model_shape(C)
        assert (qc.num_vertices, qc.num_nodes, qc.num_cliques, qc.dimension) == (n, n, m, dim)
        assert qc.max_clique == max(len(c) for c in C) and qc.cliques is C
        assert qc.num_qubits == W and qc.num_clbits == W and qc.name == "QCMRF"
        assert qc.theta is th
        assert np.allclose(qc.gamma, cf.gamma_of_theta(th), rtol=0, atol=0)
        assert qc.basis_gates == ['cx', 'id', 'rz', 'sx', 'x']


def test_theta_gamma_round_trip():
    C = [[0, 1], [1, 2]]
    g = [0.1, 0.2, 0.3, 0.05, 0.0, 0.33, 0.21, 0.11]
    qc = QCMRF(C, gamma=g, beta=2.0)
    assert qc.gamma is g
    want = [2 * np.log(np.cos(2 * x)) / 2.0 for x in g]
    assert qc.theta == want
    qc2 = QCMRF(C, theta=qc.theta, beta=2.0)
    assert np.allclose(qc2.gamma, g, atol=1e-15)


def test_value_errors_have_the_reference_texts():
    with pytest.raises(ValueError, match="The set of clique is not set properly. Type must be list of list of int."):
        QCMRF(((0, 1),), [0] * 4)
    with pytest.raises(ValueError, match="Type must be list of list of int"):
        QCMRF([[0.0, 1]], [0] * 4)
    with pytest.raises(ValueError, match="The parameter vector has an incorrect dimension. Expected: 8"):
        QCMRF([[0, 1], [1, 2]], [0.0] * 7)
    with pytest.raises(ValueError, match="The QCMRF parameter vector has an incorrect dimension. Expected: 4"):
        QCMRF([[0, 1]], gamma=[0.1] * 5)


def test_default_theta_draws_from_global_numpy_rng_one_scalar_at_a_time():
    np.random.seed(7)
    want = [np.random.uniform(low=-5.0, high=0) for _ in range(12)]
    np.random.seed(7)
    qc = QCMRF([[0, 1], [1, 2], [2, 3]])
    assert qc.theta == want and all(-5 <= t <= 0 for t in qc.theta)


def test_top_level_instruction_sequence_matches_reference_build():
    """QCMRF.py:204-243: n H, then per clique  h, cU_C<ii>, x, cU_C<ii>_dg, x, h, measure; final measures"""
    C = [[0, 1], [1, 2], [2, 3]]
    qc = QCMRF(C, [-0.3] * 12)
    names = [ci.operation.name for ci in qc.data]
    want = ["h"] * 4
    for ii in range(3):
        want += ["h", "cU_C%d" % ii, "x", "cU_C%d_dg" % ii, "x", "h", "measure"]
    want += ["measure"] * 4
    assert names == want
    cu = qc.data[5].operation
    assert cu.num_qubits == 6 and [c.operation.name for c in cu.definition.data] == ["and", "cp", "and"] * 4
    assert [qc.find_bit(q).index for q in qc.data[5].qubits] == [0, 1, 2, 3, 4, 5]
    assert [qc.find_bit(q).index for q in qc.data[12].qubits] == [0, 1, 2, 3, 4, 6]
    andg = cu.definition.data[0].operation
    # Qiskit's AND is a circuit holding ONE gate "and" whose definition is x.. mcx x..
    assert [c.operation.name for c in andg.definition.data] == ["and"]
    inner = andg.definition.data[0].operation
    assert [c.operation.name for c in inner.definition.data] == ["x", "x", "ccx", "x", "x"]     # y = (0,0)
    # a fresh AND per append (QCMRF.py:225,227) and per-instruction inverses (QCMRF.py:234): no object is shared
    ands = [c.operation for ci in qc.data if ci.operation.name.startswith("cU_C") for c in ci.operation.definition.data
            if c.operation.name.startswith("and")]
    assert len(ands) == 3 * 2 * 8 and len(set(map(id, ands))) == len(ands)
    assert [c.operation.name for c in qc.data[7].operation.definition.data] == ["and_dg", "cp", "and_dg"] * 4
    wb = QCMRF(C, [-0.3] * 12, with_barriers=True, with_measurements=False)
    assert [ci.operation.name for ci in wb.data].count("barrier") == 4 and "measure" not in [ci.operation.name for ci in wb.data]


def test_fidelity_kl_extract_probs():
    P = np.array([0.5, 0.25, 0.25, 0.0])
    Q = np.array([0.4, 0.4, 0.0, 0.2])
    assert abs(fidelity(P, Q) - (np.sqrt(0.2) + np.sqrt(0.1)) ** 2) < 1e-15
    assert abs(KL(P, Q) - (0.5 * np.log(0.5 / 0.4) + 0.25 * np.log(0.25 / 0.4))) < 1e-15
    assert fidelity(P, P) == pytest.approx(1.0) and KL(P, P) == 0.0
    R = {"00010": 30, "00001": 10, "00011": 20, "10001": 25, "01010": 15}
    p, delta = extract_probs(R, 2, 3)
    assert np.allclose(p, [0, 10 / 60, 30 / 60, 20 / 60]) and delta == pytest.approx(0.6)
    p, delta = extract_probs({"111": 5}, 2, 1)
    assert delta == 0 and p.sum() == 0


def test_hamiltonian_diagonal_is_minus_log_potential(models):
    C = models["0.5"]["GRAPHS"][5]
    th = models["0.5"]["THETAS"]["5"][0]
    qc = QCMRF(C, th)
    H = qc.hamiltonian_diagonal()
    p, Z = cf.gibbs_pmf(C, th)          # index with x_0 as MSB == variable v on bit n-1-v
    assert np.allclose(np.exp(-H) / Z, p, atol=1e-15)


def test_backend_surface_without_gpu():
    from qcmrf_amd import Aer, get_backend
    b = Aer.get_backend("qasm_simulator")
    assert b.name() == "qasm_simulator" and get_backend("qasm_simulator") is b
    with pytest.raises(ValueError):
        Aer.get_backend("ibm_torino")
    C = [[0, 1], [1, 2]]
    ing, pl = b.compile(QCMRF(C, [-0.2] * 8), fold_fresh=False)
    assert [o.kind for o in pl.ops] == ["init", "mux", "mux"] and sorted(pl.layout) == list(range(6))
    ing, pl = b.compile(QCMRF(C, [-0.2] * 8))
    assert [o.kind for o in pl.ops] == ["init", "diag", "diag"] and bin(pl.ops[0].mask).count("1") == 5


def test_eval_module_reproduces_the_table_from_the_committed_aer_counts(tmp_path, aer_counts):
    """qcmrf_amd.eval == eval.py --mode file arithmetic (tests/golden/eval_table.json)"""
    from conftest import GOLDEN
    from qcmrf_amd import eval as ev
    table = json.load(open(GOLDEN + "/eval_table.json"))
    for scale in ("0.1", "0.5"):
        d = tmp_path / ("res_" + scale)
        d.mkdir()
        (d / "result_simulation.json").write_text(json.dumps(aer_counts[scale]))
        rows = ev.main(["--results", "result_simulation.json", "--scale", scale, "--resdir", str(d)])
        for row, want in zip(rows, table[scale]):
            assert row[0] == want["graph"]
            assert abs(row[1] - want["fidelity"]) < 1e-12 and abs(row[4] - want["delta_emp"]) < 1e-12
    assert "fidelity" in ev.format_table(rows)
    with pytest.raises(SystemExit):
        ev.main(["--mode", "gibbs"])


def test_mrf_module_matches_oracle(models):
    from qcmrf_amd import mrf
    for j, C in enumerate(models["0.25"]["GRAPHS"]):
        th = models["0.25"]["THETAS"][str(j)][4]
        p, lnZ = mrf.gibbs_pmf(C, th)
        po, Z = cf.gibbs_pmf(C, th)
        assert np.abs(p - po).max() < 1e-15 and abs(np.exp(lnZ) - Z) < 1e-12
        assert abs(mrf.success_probability(C, th) - cf.success_probability(C, th)) < 1e-14
        assert mrf.dimension(C) == cf.model_shape(C)[3]


def test_sufficient_statistics_sum_to_hamiltonian():
    import itertools
    C = [[0, 2], [1, 2, 3]]
    th = (-np.linspace(0.1, 1.2, 12)).tolist()
    qc = QCMRF(C, th)
    H = np.zeros(16)
    i = 0
    for Cl in C:
        for y in itertools.product([0, 1], repeat=len(Cl)):
            H += qc.sufficient_statistic_diagonal(Cl, y) * (-th[i])
            i += 1
    assert np.allclose(H, qc.hamiltonian_diagonal(), atol=1e-15)
