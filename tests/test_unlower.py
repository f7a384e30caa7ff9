"""CPU: symbolic re-assembly of basis-gate circuits (qcmrf_amd/unlower.py) -- what run_experiment.py:52 feeds the
simulator.  Exactness against dense matrices on random circuits, and full re-assembly (one multiplexer per clique,
no dense k-qubit gate left) for all seven reference graphs (run_experiment.py:20), incl. the 3- and 4-variable cliques."""
import numpy as np
import pytest

from conftest import random_theta
from _qiskit_shapes import lower_like_qiskit
from oracle import closed_form as cf
from qcmrf_amd import QCMRF, ir, passes, unlower, workloads
from qcmrf_amd.backend import QsvBackend
from qcmrf_amd.circuit import QuantumCircuit
from qcmrf_amd.ingest import ingest
from qcmrf_amd.transpile import transpile
from test_host_logic import run_numpy


def unitary_of(ops, n):
    """dense 2^n x 2^n of an op list (passes._op_on_rows applies each gate to the rows of U)"""
    U = np.eye(2 ** n, dtype=np.complex128)
    pos = {q: q for q in range(n)}
    for o in ops:
        if o.kind == "init":
            continue
        passes._op_on_rows(U, o, pos)
    return U


def random_structured(n, depth, seed):
    """gates a QCMRF-like circuit is made of, at random: h, x, rz, p, cx, ccx with +-flags, cp, mcx"""
    rs = np.random.RandomState(seed)
    qc = QuantumCircuit(n)
    for _ in range(depth):
        r = rs.randint(9)
        q = [int(x) for x in rs.permutation(n)]
        if r == 0:
            qc.h(q[0])
        elif r == 1:
            qc.x(q[0])
        elif r == 2:
            qc.rz(float(rs.uniform(-3, 3)), q[0])
        elif r == 3:
            qc.p(float(rs.uniform(-3, 3)), q[0])
        elif r == 4:
            qc.cx(q[0], q[1])
        elif r == 5:
            qc.ccx(q[0], q[1], q[2])
        elif r == 6:
            qc.cp(float(rs.uniform(-3, 3)), q[0], q[1])
        elif r == 7 and n >= 4:
            qc.mcx(q[:3], q[3])
        else:                                     # H . (monomial) . H: the shape unlower closes as a multiplexer
            qc.h(q[0]); qc.cp(float(rs.uniform(-3, 3)), q[1], q[0]); qc.cx(q[2], q[0]); qc.rz(0.3, q[0]); qc.h(q[0])
    return qc


@pytest.mark.parametrize("seed", range(12))
def test_unlower_is_exact_on_random_basis_gate_circuits(seed):
    n = 4 + seed % 3
    qc = random_structured(n, 30, 100 + seed)
    for t in (transpile(qc), lower_like_qiskit(qc), lower_like_qiskit(qc, merge=False)):
        ing = ingest(t, peephole=True)
        out, n_raw = unlower.unlower(ing.ops)
        want = unitary_of(ing.ops, n)
        got = unitary_of(out, n)
        assert np.abs(got - want).max() < 1e-11, (seed, n_raw)
        # and through the whole pass pipeline, sharded or not
        ref = unitary_of(ingest(t).ops, n)[:, 0] * np.exp(1j * ingest(t).global_phase)
        for shards in (1, 2):
            amp, _, _, _ = run_numpy(t, fusion=3, shards=shards)
            assert np.abs(amp - ref).max() < 1e-11


def test_every_reference_graph_comes_back_as_one_multiplexer_per_clique():
    """run_experiment.py:20: all seven graphs, hand-lowered and transpiler-shaped: no dense gate is left, every clique
    block is ONE multiplexer (fold_fresh off) / one factor of the initial state (default)"""
    be = QsvBackend()
    for j, C in enumerate(workloads.REFERENCE_GRAPHS):
        th = random_theta(workloads.dimension(C), seed=j)
        qc = QCMRF(C, th)
        n, m, W, dim = cf.model_shape(C)
        for t in (transpile(qc), lower_like_qiskit(qc)):
            ing, pl = be.compile(t, fold_fresh=False)
            kinds = [o.kind for o in pl.ops]
            assert kinds.count("mux") == m and "kq" not in kinds and set(kinds) <= {"init", "mux", "diag"}, (j, kinds)
            ing, pl = be.compile(t)
            kinds = [o.kind for o in pl.ops]
            assert set(kinds) <= {"init", "diag"} and bin(pl.ops[0].mask).count("1") == n + m, (j, kinds)
            amp, _, _, _ = run_numpy(t, fusion=3)
            assert np.abs(amp - cf.amplitudes(C, th)).max() < 2e-12, j


def test_config5_lowered_compiles_to_the_generator_form():
    """BASELINE configs[4] lowered (7.4 k gates): init + 19 diagonal factors + a global phase, nothing parked"""
    name, C = workloads.baseline_config(4)
    t = transpile(QCMRF(C, workloads.theta_halfnorm(workloads.dimension(C))))
    ing = ingest(t, peephole=True)
    assert ing.flat is not None and len(ing.ops) < len(t.data)
    ops = passes.optimise(ing.ops, level=3, fresh=True, flat=ing.flat)
    kinds = [o.kind for o in ops]
    assert kinds[0] == "init" and kinds.count("diag") <= 21 and set(kinds) == {"init", "diag"}


def test_unlower_passes_on_what_it_cannot_hold():
    """a non-unitary 'u', a 3-qubit diagonal, a bracket that never closes: emitted as they came, result still exact"""
    rs = np.random.RandomState(5)
    q, _ = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2))
    ops = [ir.op_u(0, ir.FIXED_1Q["h"]), ir.op_x(1, [0]), ir.op_diag([0, 1, 2], np.exp(1j * rs.randn(8))), ir.op_u(1, q),
           ir.op_x(2, [1]), ir.op_u(0, 2.0 * ir.FIXED_1Q["h"]), ir.op_x(0, [2]), ir.op_u(2, ir.FIXED_1Q["h"])]
    out, n_raw = unlower.unlower(ops)
    assert n_raw >= 1
    assert np.abs(unitary_of(out, 3) - unitary_of(ops, 3)).max() < 1e-12


@pytest.mark.parametrize("seed", range(8))
def test_flat_ingest_run_algebra_matches_matrix_products(seed):
    """ingest._walk_flat keeps a run of rz / sx / x as angles between Hadamards (no complex arithmetic): the op it emits,
    materialised, times the global phase it books, equals the product of the gate matrices -- for random runs incl. the
    folds H D(0) H = 1, H D(pi) H = X, runs that multiply out to a number, and runs that keep three Hadamards"""
    rs = np.random.RandomState(seed)
    SX = 0.5 * np.array([[1 + 1j, 1 - 1j], [1 - 1j, 1 + 1j]])
    X = np.array([[0, 1], [1, 0]], dtype=complex)
    special = [0.0, np.pi, -np.pi, np.pi / 2, -np.pi / 2, np.pi / 4, 2 * np.pi]
    for trial in range(60):
        qc = QuantumCircuit(2)
        U = np.eye(2, dtype=complex)
        for _ in range(int(rs.randint(1, 9))):
            r = rs.randint(4)
            if r == 0:
                lam = float(special[rs.randint(len(special))] if rs.rand() < 0.6 else rs.uniform(-4, 4))
                qc.rz(lam, 0)
                U = np.diag([np.exp(-0.5j * lam), np.exp(0.5j * lam)]) @ U
            elif r == 1:
                qc.sx(0)
                U = SX @ U
            elif r == 2:
                qc.x(0)
                U = X @ U
            else:
                qc.id(0)
        qc.cx(0, 1)                                   # closes the run
        ing = ingest(qc, peephole=True)
        assert ing.flat is not None
        run = [o for o in ing.ops if not (o.kind == "x" and o.ctrls)]
        assert len(run) <= 1
        got = np.eye(2, dtype=complex)
        if run:
            o = run[0]
            got = X if o.kind == "x" else np.diag(o.table) if o.kind == "diag" else o.mat
        got = np.exp(1j * ing.global_phase) * got
        assert np.abs(got - U).max() < 1e-12, (seed, trial, [ci.operation.name for ci in qc.data])


def test_merge_diagonals_is_exact_and_lets_scattered_phases_meet():
    """passes.merge_diagonals: diagonals travel forward past everything that is diagonal on their qubits (multiplexer selects,
    other diagonals) and multiply into a waiting one on the same qubits or a superset; a dense gate on one of their qubits
    stops them.  Exact on random lists; a T on a variable wire and its inverse three gates later cancel."""
    rs = np.random.RandomState(3)

    def rmux(ctrls, t):
        mats = []
        for _ in range(2 ** len(ctrls)):
            q, _ = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2))
            mats.append(q)
        return ir.op_mux(ctrls, t, np.array(mats))
    n = 5
    for trial in range(20):
        ops = []
        for _ in range(12):
            r = rs.randint(4)
            q = [int(x) for x in rs.permutation(n)]
            if r == 0:
                k = int(rs.randint(1, 4))
                ops.append(ir.op_diag(q[:k], np.exp(1j * rs.uniform(-3, 3, size=2 ** k))))
            elif r == 1:
                ops.append(ir.op_mcphase(q[:rs.randint(1, 4)], float(rs.uniform(-3, 3))))
            elif r == 2:
                ops.append(rmux(q[:2], q[2]))
            else:
                ops.append(ir.op_x(q[0], q[1:2]))
        out = passes.merge_diagonals(ops)
        assert np.abs(unitary_of(out, n) - unitary_of(ops, n)).max() < 1e-12
        assert sum(o.kind in ("diag", "mcphase") for o in out) <= sum(o.kind in ("diag", "mcphase") for o in ops)
    t = np.exp(0.25j * np.pi)
    ops = [ir.op_diag([0], [1, t]), rmux([0, 1], 3), ir.op_diag([1, 0], [1, 1, 1, -1]), rmux([0], 4), ir.op_diag([0], [1, t.conjugate()])]
    out = passes.merge_diagonals(ops)
    assert [o.kind for o in out] == ["mux", "mux", "diag"] and set(out[2].qubits) == {0, 1}
    assert np.abs(unitary_of(out, 5) - unitary_of(ops, 5)).max() < 1e-14
    # fold_fresh: a constant table is a number, not a device factor
    folded = passes.fold_fresh([ir.op_init(0b11), ir.op_diag([0], [t, t]), ir.op_diag([1], [1, t])])
    assert [o.kind for o in folded] == ["init", "diag"] and np.allclose(folded[1].table, [t, t * t])


@pytest.mark.parametrize("seed", range(8))
def test_compact_gate_records_are_the_same_circuit(seed):
    """ingest(compact=True) hands unlower plain tuples instead of ir.Op objects (what backend.compile asks for at
    fusion 3): same unitary record by record, same re-assembly, and the passes fall back to ops below level 3"""
    n = 4 + seed % 3
    qc = random_structured(n, 30, 300 + seed)
    for t in (transpile(qc), lower_like_qiskit(qc)):
        a, b = ingest(t, peephole=True), ingest(t, peephole=True, compact=True)
        assert b.flat["compact"] and not a.flat["compact"] and all(type(r) is tuple for r in b.ops)
        assert abs(a.global_phase - b.global_phase) < 1e-14 and len(a.ops) == len(b.ops)
        conv = [unlower.rec_to_op(r) for r in b.ops]
        want = unitary_of(a.ops, n)
        assert np.abs(unitary_of(conv, n) - want).max() < 1e-13
        out, _ = unlower.unlower(b.ops)
        assert all(isinstance(o, ir.Op) for o in out)
        assert np.abs(unitary_of(out, n) - want).max() < 1e-11
        for level in (1, 2, 3):
            oa = passes.optimise(a.ops, level=level, fresh=True, flat=a.flat)
            ob = passes.optimise(b.ops, level=level, fresh=True, flat=b.flat)
            assert [o.kind for o in oa] == [o.kind for o in ob], level
            assert oa[0].mask == ob[0].mask
            assert np.abs(unitary_of(oa, n) - unitary_of(ob, n)).max() < 1e-11


def test_compact_records_keep_mid_circuit_measures_out():
    """keep_measures wants ir.Op('measure') in the stream: the flat walk stays with ops then"""
    qc = QuantumCircuit(3, 3)
    qc.h(0); qc.cx(0, 1); qc.cx(1, 2); qc.measure(1, 1)
    t = transpile(qc)
    ing = ingest(t, peephole=True, keep_measures=True, compact=True)
    assert ing.flat is None or not ing.flat["compact"]
    assert all(isinstance(o, ir.Op) for o in ing.ops)
