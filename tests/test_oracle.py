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


# ---- Kronecker-product checks of every gate kind: W <= 8, dense matrices built
# ---- here from np.kron and projectors -- independent of the index arithmetic of the simulators
def _kron_embed(nq, factors):
    """tensor product with qubit 0 as the LEAST significant index bit; factors: {qubit: 2x2}"""
    U = np.eye(1, dtype=np.complex128)
    for q in range(nq):
        U = np.kron(factors.get(q, np.eye(2)), U)
    return U


def _kron_controlled(nq, t, M, ctrls=(), vals=None):
    """I + P_ctrl (x) (M - I)_t with P_ctrl the projector on the control pattern"""
    vals = [1] * len(ctrls) if vals is None else vals
    proj = {c: np.diag([1.0 - v, float(v)]) for c, v in zip(ctrls, vals)}
    return np.eye(2 ** nq) + _kron_embed(nq, {**proj, t: np.asarray(M) - np.eye(2)})


def _kron_diag(nq, qubits, table):
    U = np.zeros((2 ** nq, 2 ** nq), dtype=np.complex128)
    for j, d in enumerate(np.asarray(table)):
        U += d * _kron_embed(nq, {q: np.diag([1.0 - ((j >> b) & 1), float((j >> b) & 1)]) for b, q in enumerate(qubits)})
    return U


def _kron_mux(nq, ctrls, t, mats):
    U = np.zeros((2 ** nq, 2 ** nq), dtype=np.complex128)
    for j, M in enumerate(mats):
        U += _kron_embed(nq, {**{q: np.diag([1.0 - ((j >> b) & 1), float((j >> b) & 1)]) for b, q in enumerate(ctrls)}, t: M})
    return U


@pytest.mark.parametrize("nq", [1, 2, 3, 5, 8])
def test_every_gate_kind_against_kronecker_products(nq):
    rs = np.random.RandomState(nq)
    v = rs.randn(2 ** nq) + 1j * rs.randn(2 ** nq)
    v /= np.linalg.norm(v)

    def ru():
        return np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2))[0]

    def both(U, np_apply, c_apply):
        """numpy simulator, C simulator and the dense matrix agree on a random state"""
        want = U @ v
        assert np.abs(U.conj().T @ U - np.eye(2 ** nq)).max() < 1e-13          # unitary
        got = v.copy()
        got = np_apply(got)
        assert np.abs(got - want).max() < 1e-14
        r = cref.RefState(nq)
        r.state[:] = v
        c_apply(r)
        assert np.abs(r.state - want).max() < 1e-14

    named = dict(sv.MATS)                                       # h x y z s sdg t tdg sx sxdg ...
    named.update({"rz": sv.rz(0.37), "rx": sv.rx(-1.1), "ry": sv.ry(2.2), "p": sv.phase(0.9), "u3": sv.u3(0.3, 1.2, -0.7)})
    for name, M in named.items():
        for t in {0, nq // 2, nq - 1}:
            U = _kron_embed(nq, {t: M})
            both(U, lambda s, t=t, M=M: sv.apply_1q(s, t, M), lambda r, t=t, M=M: r.apply_1q(t, M))
            assert np.abs(sv.dense_unitary_1q(nq, t, M) - U).max() < 1e-15
    if nq >= 2:
        for trial in range(6):                                   # cx / ccx / mcx with +- flags, controlled 2x2, cp / mcphase
            k = int(rs.randint(1, min(nq - 1, 3) + 1))
            qs = [int(x) for x in rs.permutation(nq)[:k + 1]]
            vals = [int(x) for x in rs.randint(0, 2, size=k)]
            X = sv.MATS["x"]
            U = _kron_controlled(nq, qs[-1], X, qs[:-1], vals)
            both(U, lambda s: sv.apply_mcx(s, qs[:-1], qs[-1], vals), lambda r: r.apply_mcx(qs[:-1], qs[-1], vals))
            M = ru()
            U = _kron_controlled(nq, qs[-1], M, qs[:-1], vals)
            both(U, lambda s: sv.apply_1q(s, qs[-1], M, qs[:-1], vals), lambda r: r.apply_1q(qs[-1], M, qs[:-1], vals))
            assert np.abs(sv.dense_unitary_1q(nq, qs[-1], M, qs[:-1], vals) - U).max() < 1e-15
            lam = float(rs.uniform(-3, 3))
            allv = vals + [1]
            U = _kron_diag(nq, qs, [np.exp(1j * lam) if j == sum(b << e for e, b in enumerate(allv)) else 1.0 for j in range(2 ** (k + 1))])
            both(U, lambda s: sv.apply_mcphase(s, qs, lam, allv), lambda r: r.apply_mcphase(qs, lam, allv))
            tab = np.exp(1j * rs.uniform(-3, 3, size=2 ** (k + 1)))
            U = _kron_diag(nq, qs, tab)
            both(U, lambda s: sv.apply_diag(s, qs, tab), lambda r: r.apply_diag(qs, tab))
            mats = np.array([ru() for _ in range(2 ** k)])
            U = _kron_mux(nq, qs[:-1], qs[-1], mats)
            both(U, lambda s: sv.apply_mux(s, qs[:-1], qs[-1], mats), lambda r: r.apply_mux(qs[:-1], qs[-1], mats))
        # dense k-qubit gate (numpy only: the C oracle has none): U_k embedded by index arithmetic of
        # the TEST, not of apply_kq
        for k in range(1, min(nq, 4) + 1):
            qs = [int(x) for x in rs.permutation(nq)[:k]]
            Uk = np.linalg.qr(rs.randn(2 ** k, 2 ** k) + 1j * rs.randn(2 ** k, 2 ** k))[0]
            U = np.zeros((2 ** nq, 2 ** nq), dtype=np.complex128)
            for col in range(2 ** nq):
                jc = sum(((col >> q) & 1) << b for b, q in enumerate(qs))
                rest = col & ~sum(1 << q for q in qs)
                for jr in range(2 ** k):
                    U[rest | sum(((jr >> b) & 1) << q for b, q in enumerate(qs)), col] = Uk[jr, jc]
            got = sv.apply_kq(v.copy(), qs, Uk)
            assert np.abs(got - U @ v).max() < 1e-14


def test_gate_inverse_and_diagonal_commutation_properties():
    """SURVEY.md 4.3 property checks on the numpy simulator: gate . inverse = identity, diagonals
    commute with each other and with controls, norm is preserved"""
    nq = 7
    rs = np.random.RandomState(11)
    v = rs.randn(2 ** nq) + 1j * rs.randn(2 ** nq)
    v /= np.linalg.norm(v)
    s = v.copy()
    M = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2))[0]
    s = sv.apply_1q(s, 3, M, [0, 5], [1, 0])
    s = sv.apply_mcx(s, [1, 2], 6, [0, 1])
    assert abs(np.linalg.norm(s) - 1) < 1e-14
    s = sv.apply_mcx(s, [1, 2], 6, [0, 1])
    s = sv.apply_1q(s, 3, M.conj().T, [0, 5], [1, 0])
    assert np.abs(s - v).max() < 1e-14
    t1, t2 = np.exp(1j * rs.randn(8)), np.exp(1j * rs.randn(4))
    a = sv.apply_diag(sv.apply_diag(v.copy(), [0, 3, 6], t1), [3, 4], t2)
    b = sv.apply_diag(sv.apply_diag(v.copy(), [3, 4], t2), [0, 3, 6], t1)
    assert np.abs(a - b).max() < 1e-15
    a = sv.apply_diag(sv.apply_mcx(v.copy(), [0, 3], 5), [0, 3, 6], t1)       # a diagonal on the CONTROLS commutes with the gate
    b = sv.apply_mcx(sv.apply_diag(v.copy(), [0, 3, 6], t1), [0, 3], 5)
    assert np.abs(a - b).max() < 1e-15


def test_c_oracle_under_address_and_ub_sanitizers():
    """make -C oracle asan: every entry point of qsv_ref.c on small random states under ASan + UBSan
    (CPU only -- the GPU pool has no sanitizer)"""
    import os, shutil, subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    if shutil.which("make") is None or shutil.which("gcc") is None:
        pytest.skip("needs make + gcc")
    r = subprocess.run(["make", "-C", here, "-B", "asan"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "asan driver ok" in r.stdout
