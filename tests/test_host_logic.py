"""CPU: ingest, fusion passes, planner and program encoding, executed on the numpy stand-in
engine (oracle/sharded_numpy.py) and checked against the closed form / the gate-level oracle."""
import numpy as np
import pytest

from conftest import random_theta
from oracle import closed_form as cf, gate_stream as gs, sv_numpy as sv
from oracle.sharded_numpy import NumpyEngine
from qcmrf_amd import QCMRF, ir, passes, planner, program, workloads
from qcmrf_amd.backend import QsvBackend
from qcmrf_amd.ingest import ingest


def logical_amplitudes(eng, layout, W):
    amp = eng.amplitudes()
    p = np.arange(2 ** W, dtype=np.int64)
    l = np.zeros_like(p)
    for q, pos in enumerate(layout):
        l |= ((p >> pos) & 1) << q
    out = np.empty_like(amp)
    out[l] = amp
    return out


def run_numpy(qc, fusion=2, shards=1, layout="auto"):
    be = QsvBackend()
    ing, pl = be.compile(qc, shards, fusion=fusion, layout=layout)
    eng = NumpyEngine(ing.num_qubits, shards)
    program.run_stepwise(eng, pl.ops)
    return logical_amplitudes(eng, pl.layout, ing.num_qubits), ing, pl, eng


def test_ingest_flattens_to_reference_stream(models):
    """fusion 0: the ingested primitive list is the reference's gate stream, op for op"""
    for j, C in enumerate(models["0.5"]["GRAPHS"]):
        th = models["0.5"]["THETAS"][str(j)][3]
        ing = ingest(QCMRF(C, th))
        ref = [o for o in gs.reference_stream(C, th) if o[0] != "measure"]
        assert len(ing.ops) == len(ref)
        for op, r in zip(ing.ops, ref):
            if r[0] == "h":
                assert op.kind == "u" and op.target == r[1] and op.label == "h"
            elif r[0] == "x":
                assert op.kind == "x" and op.target == r[1] and not op.ctrls
            elif r[0] == "mcx":
                assert op.kind == "x" and op.ctrls == tuple(r[1]) and op.target == r[2] and all(op.vals)
            elif r[0] == "cp":
                assert op.kind == "mcphase" and op.qubits == (r[2], r[3]) and op.angle == r[1]
        n, m, W, dim = cf.model_shape(C)
        assert ing.measure == {**{n + 1 + i: n + 1 + i for i in range(m)}, **{q: q for q in range(n)}}
        assert ing.num_clbits == W and n not in ing.measure


@pytest.mark.parametrize("fusion", [0, 1, 2])
def test_fusion_levels_are_exact(models, fusion):
    for j, C in enumerate(models["0.25"]["GRAPHS"]):
        th = models["0.25"]["THETAS"][str(j)][1]
        amp, ing, pl, _ = run_numpy(QCMRF(C, th), fusion=fusion)
        assert np.abs(amp - cf.amplitudes(C, th)).max() < 1e-13


def test_fused_program_shape():
    C = workloads.grid(2, 6, drop_last=1)
    qc = QCMRF(C, random_theta(60))
    be = QsvBackend()
    ing, pl = be.compile(qc, fusion=2)
    kinds = [o.kind for o in pl.ops]
    assert kinds == ["init"] + ["mux"] * 15                  # one sweep per clique
    assert all(len(o.ctrls) == 3 for o in pl.ops[1:])       # two variables + the AND scratch qubit
    ing, pl = be.compile(qc, fusion=1)
    assert [o.kind for o in pl.ops] == ["init"] + ["u", "diag", "u"] * 15      # H . diag . H per clique
    ing, pl = be.compile(qc, fusion=0)
    assert len(pl.ops) == 1 + 12 + 15 * 60


def test_gamma_close_to_zero_is_skipped():
    C = [[0, 1], [1, 2]]
    th = [0.0, -0.3, -1e-20, -0.7, -0.2, 0.0, -0.9, -0.4]
    amp, ing, pl, _ = run_numpy(QCMRF(C, th), fusion=0)
    n_cp = sum(1 for o in ing.ops if o.kind == "mcphase")
    assert n_cp == 2 * 5                                      # three parameters emit no gates at all
    assert np.abs(amp - cf.amplitudes(C, th)).max() < 1e-13


def test_with_barriers_and_without_measurements():
    C = [[0, 1], [1, 2]]
    th = random_theta(8)
    a1, ing1, _, _ = run_numpy(QCMRF(C, th, with_barriers=True))
    a2, ing2, _, _ = run_numpy(QCMRF(C, th, with_measurements=False))
    assert np.abs(a1 - cf.amplitudes(C, th)).max() < 1e-13 and np.abs(a2 - a1).max() < 1e-15
    assert ing2.measure == {}


def rand_circuit(nq, n_ops, seed):
    """random circuit over the engine's primitive set, as (container circuit, oracle stream)"""
    from qcmrf_amd.circuit import QuantumCircuit
    rs = np.random.RandomState(seed)
    qc = QuantumCircuit(nq, nq)
    for _ in range(n_ops):
        k = rs.randint(0, 10)
        q = rs.permutation(nq)[:3].tolist()
        a = float(rs.uniform(-3, 3))
        if k == 0: qc.h(q[0])
        elif k == 1: qc.x(q[0])
        elif k == 2: qc.cx(q[0], q[1])
        elif k == 3: qc.ccx(q[0], q[1], q[2])
        elif k == 4: qc.rz(a, q[0])
        elif k == 5: qc.cp(a, q[0], q[1])
        elif k == 6: qc.sx(q[0])
        elif k == 7: qc.t(q[0]); qc.s(q[1]); qc.z(q[2])
        elif k == 8: qc.cz(q[0], q[1]); qc.y(q[2])
        else: qc.ry(a, q[0]); qc.swap(q[1], q[2])
    return qc


def oracle_state_of(qc):
    """independent evaluation of a container circuit with the numpy oracle"""
    s = sv.zero_state(qc.num_qubits)
    for ci in qc.data:
        name, p = ci.operation.name, ci.operation.params
        q = [qc.find_bit(b).index for b in ci.qubits]
        if name in sv.MATS: sv.apply_1q(s, q[0], sv.MATS[name])
        elif name == "cx": sv.apply_mcx(s, [q[0]], q[1])
        elif name == "ccx": sv.apply_mcx(s, q[:2], q[2])
        elif name == "rz": sv.apply_1q(s, q[0], sv.rz(p[0]))
        elif name == "ry": sv.apply_1q(s, q[0], sv.ry(p[0]))
        elif name == "cp": sv.apply_mcphase(s, q, p[0])
        elif name == "cz": sv.apply_mcphase(s, q, np.pi)
        elif name == "swap":
            sv.apply_mcx(s, [q[0]], q[1]); sv.apply_mcx(s, [q[1]], q[0]); sv.apply_mcx(s, [q[0]], q[1])
        else: raise AssertionError(name)
    return s


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("fusion", [0, 1, 2])
def test_random_circuits_all_fusion_levels(seed, fusion):
    qc = rand_circuit(7, 60, seed)
    amp, ing, pl, _ = run_numpy(qc, fusion=fusion)
    assert np.abs(amp - oracle_state_of(qc)).max() < 1e-12
    if fusion:
        assert len(pl.ops) <= len(ing.ops) + 1


@pytest.mark.parametrize("layout", ["auto", "reference"])
@pytest.mark.parametrize("shards", [2, 4, 8])
@pytest.mark.parametrize("fusion", [0, 2])
def test_sharded_plan_matches_closed_form(shards, layout, fusion):
    C = workloads.grid(2, 3)
    th = random_theta(cf.model_shape(C)[3], seed=shards)
    amp, ing, pl, eng = run_numpy(QCMRF(C, th), fusion=fusion, shards=shards, layout=layout)
    assert np.abs(amp - cf.amplitudes(C, th)).max() < 1e-13
    assert eng.n_exchanges == pl.n_exchanges
    if layout == "auto" and fusion == 2:
        assert pl.n_exchanges == 0
        g = shards.bit_length() - 1
        n = cf.model_shape(C)[0]
        shard_logical = [q for q, p in enumerate(pl.layout) if p >= ing.num_qubits - g]
        assert all(q < n for q in shard_logical)             # variable qubits became the shard bits
    if layout == "reference":
        assert pl.layout != [] and pl.initial_layout == list(range(ing.num_qubits))
        assert pl.n_exchanges > 0


@pytest.mark.parametrize("seed", range(4))
def test_sharded_random_circuits(seed):
    qc = rand_circuit(8, 50, 100 + seed)
    want = oracle_state_of(qc)
    for shards in (2, 4):
        for layout in ("auto", "reference"):
            amp, _, pl, _ = run_numpy(qc, fusion=2, shards=shards, layout=layout)
            assert np.abs(amp - want).max() < 1e-12


def test_program_encoding_roundtrip():
    from qcmrf_amd import _lib
    C = workloads.chain(4)
    qc = QCMRF(C, random_theta(12))
    be = QsvBackend()
    for fusion in (0, 1, 2):
        ing, pl = be.compile(qc, fusion=fusion)
        rec, data = program.encode(pl.ops)
        assert rec.dtype == _lib.OP_DTYPE and len(rec) == len(pl.ops)
        for r, op in zip(rec, pl.ops):
            if op.kind == "mux":
                assert r["kind"] == _lib.OP_MUX and r["target"] == op.target
                off, cnt = int(r["data_off"]), 8 * 2 ** int(r["n"])
                got = data[off: off + cnt].view(np.complex128).reshape(-1, 2, 2)
                assert np.array_equal(got, op.mats)
            if op.kind == "mcphase":
                assert r["angle"] == op.angle and list(r["vals"][: r["n"]]) == list(op.vals)


def test_ingest_rejects_what_it_cannot_do():
    from qcmrf_amd.circuit import QuantumCircuit, Instruction
    qc = QuantumCircuit(2, 2)
    qc.h(0); qc.measure(0, 0); qc.x(0)
    with pytest.raises(ValueError, match="after it was measured"):
        ingest(qc)
    qc = QuantumCircuit(2, 2)
    qc.append(Instruction("frobnicate", 2), [0, 1])
    with pytest.raises(ValueError, match="frobnicate"):
        ingest(qc)
    qc = QuantumCircuit(1, 1)
    qc.h(0)
    qc.data[-1].operation.condition = ("c", 1)
    with pytest.raises(ValueError, match="conditioned"):
        ingest(qc)


def test_open_controls_and_ctrl_state():
    from qcmrf_amd.circuit import QuantumCircuit, Instruction
    qc = QuantumCircuit(3, 0)
    qc.h(0); qc.h(1)
    op = Instruction("ccx_o1", 3)
    op.ctrl_state = 1                      # control 0 fires on 1, control 1 fires on 0
    qc.append(op, [0, 1, 2])
    amp, _, _, _ = run_numpy(qc, fusion=0)
    s = sv.zero_state(3)
    sv.apply_1q(s, 0, sv.MATS["h"]); sv.apply_1q(s, 1, sv.MATS["h"])
    sv.apply_mcx(s, [0, 1], 2, [1, 0])
    assert np.abs(amp - s).max() < 1e-15


# ---- circuits lowered to the reference's basis (run_experiment.py:52) ----------------------------
def test_transpile_stand_in_is_exact_and_in_basis(models):
    from qcmrf_amd.transpile import transpile, BASIS
    for j in (0, 1, 2, 4, 6):
        C = models["0.5"]["GRAPHS"][j]
        th = models["0.5"]["THETAS"][str(j)][0]
        t = transpile(QCMRF(C, th), basis_gates=['cx', 'id', 'rz', 'sx', 'x'])
        assert set(ci.operation.name for ci in t.data) <= set(BASIS) | {"measure"}
        for fusion in (0, 3):
            amp, ing, pl, _ = run_numpy(t, fusion=fusion)
            assert np.abs(amp - cf.amplitudes(C, th)).max() < 5e-12     # incl. the global phase
        assert ing.measure == ingest(QCMRF(C, th)).measure
    with pytest.raises(ValueError):
        transpile(QCMRF([[0]], [-.1, -.2]), basis_gates=["u3", "cx"])


def test_lowered_qcmrf_is_reassembled_into_one_multiplexer_per_clique():
    """dense windows + structure recovery: ~390 basis gates per 2-clique collapse back to the
    same init + one mux per clique the nested form gives (plus phase-only diagonals)"""
    from qcmrf_amd.transpile import transpile
    C = workloads.grid(2, 3)                                   # 7 cliques, W = 14
    th = random_theta(cf.model_shape(C)[3])
    t = transpile(QCMRF(C, th))
    be = QsvBackend()
    ing, pl = be.compile(t, fold_fresh=False)
    kinds = [o.kind for o in pl.ops]
    assert len(t.data) > 2000 and kinds[0] == "init"
    assert kinds.count("mux") == 7 and kinds.count("kq") == 0 and set(kinds) <= {"init", "mux", "diag"}
    n = cf.model_shape(C)[0]
    assert bin(be.compile(t, layout="reference", fold_fresh=False)[1].ops[0].mask).count("1") == n     # variables folded into init
    # default: every multiplexer acts on a fresh ancilla and becomes a factor of the initial state
    ing, pl = be.compile(t)
    assert [o.kind for o in pl.ops].count("mux") == 0 and bin(pl.ops[0].mask).count("1") == n + 7
    amp, _, _, _ = run_numpy(t)
    assert np.abs(amp - cf.amplitudes(C, th)).max() < 1e-11


@pytest.mark.parametrize("seed", range(5))
def test_random_lowered_circuits_dense_fusion(seed):
    from qcmrf_amd.transpile import transpile
    qc = rand_circuit(7, 70, 40 + seed)
    t = transpile(qc)
    want = oracle_state_of(qc)
    for fusion in (0, 2, 3):
        for shards in (1, 4):
            amp, ing, pl, _ = run_numpy(t, fusion=fusion, shards=shards)
            assert np.abs(amp - want).max() < 1e-11
    assert len(pl.ops) < len(ing.ops) / 4


def test_dense_window_recovery_kinds():
    """_recover: block-diagonal windows come back as diag / u / mux, the rest as kq"""
    rs = np.random.RandomState(0)
    def ru(k):
        q, _ = np.linalg.qr(rs.randn(2 ** k, 2 ** k) + 1j * rs.randn(2 ** k, 2 ** k))
        return q
    ops = [ir.op_kq([1, 4, 6], ru(3)), ir.op_u(4, ru(1)), ir.op_x(6, [1])]
    out = passes.fuse_dense(ops)
    assert [o.kind for o in out] == ["kq"] and set(out[0].qubits) == {1, 4, 6}
    ops = [ir.op_u(2, ru(1), [5], [0]), ir.op_mcphase([5, 2], 0.3), ir.op_x(2, [5, 0])]
    out = passes.fuse_dense(ops)
    assert [o.kind for o in out] == ["mux"] and out[0].target == 2 and set(out[0].ctrls) == {5, 0}
    ops = [ir.op_x(3, [1]), ir.op_mcphase([3, 1], 0.4), ir.op_x(3, [1])]
    out = passes.fuse_dense(ops)
    assert [o.kind for o in out] == ["diag"]
    for o_in, o_out in ((ops, out),):
        a = sv.zero_state(6); a[:] = rs.randn(64) + 1j * rs.randn(64)
        b = a.copy()
        e1, e2 = NumpyEngine(6), NumpyEngine(6)
        e1.sh[0][:], e2.sh[0][:] = a, b
        program.run_stepwise(e1, o_in); program.run_stepwise(e2, o_out)
        assert np.abs(e1.sh[0] - e2.sh[0]).max() < 1e-14


# ---- trajectory mode (mid-circuit measurements taken when they occur, qubits released) -----------
def _numpy_factory(n, devices=(0,), **kw):
    return NumpyEngine(n, 1)


def test_trajectory_compile_live_width_and_segments():
    from qcmrf_amd import trajectory
    C = workloads.chain(6)                                    # n = 6, m = 5, W = 12
    segs, width, final, nclb, cregs, nsrc = trajectory.compile_trajectory(QCMRF(C, random_theta(20)))
    assert width == 6 + 2                                     # variables + scratch + ONE recycled ancilla
    assert len(segs) == 5 and nclb == 12
    assert [s.measure_clbit for s in segs[:-1]] == [7, 8, 9, 10] and all(s.release for s in segs[:-1])
    assert all(s.measure_slot == segs[0].measure_slot for s in segs[:-1])       # the same slot every time
    assert sorted(c for _, c in final) == [0, 1, 2, 3, 4, 5, 11]
    assert [s.n_ops for s in segs] == [2, 1, 1, 1, 1]         # init + mux, then one mux per clique
    # every later segment carries, per outcome of the previous measurement, ONE record list: the 0/1 projection (and the X
    # that hands the released slot back after outcome 1) in front of its own ops
    assert segs[0].prog is None
    for sg in segs[1:]:
        assert len(sg.prog[0][0]) == sg.n_ops + 1 and len(sg.prog[1][0]) == sg.n_ops + 2


@pytest.mark.parametrize("fusion", [0, 3])
def test_trajectory_counts_follow_the_closed_form(models, fusion):
    from qcmrf_amd import trajectory
    for j in (2, 5):
        C = models["0.5"]["GRAPHS"][j]
        th = models["0.5"]["THETAS"][str(j)][3]
        shots = 60000
        vals, cnts, nclb, cregs, meta = trajectory.run_trajectories(QCMRF(C, th), shots, 11, fusion=fusion,
                                                                    engine_factory=_numpy_factory)
        n, m, W, dim = cf.model_shape(C)
        assert meta["live_qubits"] == n + 2 and nclb == W and cnts.sum() == shots
        p = cf.probabilities(C, th)
        obs = np.zeros(p.size)
        for v, c in zip(vals.tolist(), cnts.tolist()):
            obs[v] += c
        assert obs[p == 0].sum() == 0
        sel = p * shots > 5
        chi = ((obs[sel] - p[sel] * shots) ** 2 / (p[sel] * shots)).sum() / (sel.sum() - 1)
        assert 0.8 < chi < 1.25


def test_trajectory_through_the_backend_and_seeded():
    be = QsvBackend(method="trajectory")
    be._engine_factory = _numpy_factory
    C = [[0, 1], [1, 2]]
    qc = QCMRF(C, random_theta(8))
    r1 = be.run(qc, shots=3000, seed_simulator=5).result()
    r2 = be.run(qc, shots=3000, seed_simulator=5).result()
    c1 = r1.get_counts()
    assert c1 == r2.get_counts() and sum(c1.values()) == 3000 and all(len(k) == 6 for k in c1)
    assert r1.metadata(0)["method"] == "trajectory" and r1.metadata(0)["live_qubits"] == 5
    p = cf.probabilities(C, qc.theta)
    assert all(p[int(k, 2)] > 0 for k in c1)


@pytest.mark.parametrize("seed", range(12))
def test_fold_fresh_random_circuits(seed):
    """passes.fold_fresh: gates whose target nothing has touched yet become diagonal factors of
    the initial product state.  Random circuits open with plenty of those (on |0>, on |+> after a
    folded H, with controls on qubits that are still |0>); the rewritten program must give the
    same state as the oracle, with and without the pass, sharded or not."""
    nq = 8 + seed % 3
    qc = rand_circuit(nq, 40 + 5 * seed, 1000 + seed)
    want = oracle_state_of(qc)
    be = QsvBackend()
    for shards in (1, 2):
        for fresh in (True, False):
            ing, pl = be.compile(qc, shards, fusion=3, fold_fresh=fresh)
            eng = NumpyEngine(ing.num_qubits, shards)
            rec, data = program.encode(pl.ops)
            eng.exec(rec, data)
            assert np.abs(logical_amplitudes(eng, pl.layout, ing.num_qubits) - want).max() < 1e-12, (shards, fresh)


def test_fold_fresh_rules():
    """the individual rules, on hand-made op lists (logical qubits, no planner)"""
    rs = np.random.RandomState(11)

    def ru():
        q, _ = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2))
        return q

    def state(ops, n):
        eng = NumpyEngine(n)
        program.run_stepwise(eng, ops)
        return eng.amplitudes()

    n = 6
    mats = np.array([ru() for _ in range(4)])
    base = [ir.op_init(0b000011),                                   # q0, q1 in |+>
            ir.op_mux([0, 1], 2, mats),                            # fresh |0> target, populated selects
            ir.op_u(3, ru(), ctrls=[4], vals=[0]),                 # control on |0> with value 0: always fires
            ir.op_u(5, ru(), ctrls=[4], vals=[1]),                 # control on |0> with value 1: never fires, dropped
            ir.op_diag([0, 4], np.exp(1j * rs.randn(4))),          # diagonal over a populated and a |0> qubit
            ir.op_u(1, ru()),                                      # q1 selected the multiplexer: no longer |+>, cannot fold
            ir.op_mcphase([2, 3], 0.7),
            ir.op_x(4, ctrls=[2]),                                 # now qubit 4 gets populated, controlled by a folded target
            ir.op_u(0, ru()),                                      # q0 has been used by factors: cannot fold
            ir.op_u(2, ru(), ctrls=[0]),                           # touches blocked q0: cannot fold
            ir.op_diag([3, 5], np.exp(1j * rs.randn(4)))]          # disjoint from everything blocked: still folds
    out = passes.fold_fresh(base)
    kinds = [o.kind for o in out]
    assert kinds[0] == "init" and kinds.count("u") == 3 and kinds[-3:] == ["u", "u", "u"]
    assert out[0].mask == 0b011111                                  # q2, q3, q4 populated; q5 never left |0>
    assert all(o.kind == "diag" for o in out[1:-3])
    assert all(((out[0].mask >> q) & 1) for o in out[1:-3] for q in o.qubits)   # no factor reads an unpopulated bit
    # an untouched |+> target does fold (row sums)
    plus = [ir.op_init(0b11), ir.op_u(0, ru()), ir.op_mux([0], 1, np.array([ru(), ru()]))]
    outp = passes.fold_fresh(plus)
    assert [o.kind for o in outp] == ["init", "diag", "diag"] and np.abs(state(outp, 2) - state(plus, 2)).max() < 1e-13
    assert np.abs(state(out, n) - state(base, n)).max() < 1e-13
    # nothing to fold after a dense gate has touched every qubit
    blocked = [ir.op_init(0), ir.op_kq([0, 1, 2], np.linalg.qr(rs.randn(8, 8) + 1j * rs.randn(8, 8))[0]), ir.op_u(1, ru())]
    assert [o.kind for o in passes.fold_fresh(blocked)] == ["init", "kq", "u"]
    # a program that does not start with init is left alone
    assert passes.fold_fresh(base[1:]) == base[1:]


def test_dense_windows_never_outgrow_the_local_qubits():
    """fusion 3 builds dense windows of up to 5 qubits; on 4 shards of a 6-qubit circuit only 4
    qubits are local, and a window as wide as that would leave the planner no local qubit to swap a
    shard-bit target in with (it used to raise 'leaves no local qubit free').  backend.compile caps
    the window width at L, and the planner may evict a gate's own controls / selects to shard bits."""
    for seed in range(12):
        for nq in (5, 6):
            qc = rand_circuit(nq, 60, 900 + seed)
            want = oracle_state_of(qc)
            for shards in (2, 4):
                for layout in ("auto", "reference"):
                    amp, ing, pl, eng = run_numpy(qc, fusion=3, shards=shards, layout=layout)
                    assert np.abs(amp - want).max() < 1e-12, (seed, nq, shards, layout)
                    L = nq - (shards.bit_length() - 1)
                    assert all(len(o.qubits) <= L for o in pl.ops if o.kind == "kq")


def test_engine_options_are_per_run():
    """run(..., engine_options={...}) must not leak into later runs on the cached engine"""
    calls = []

    class Eng(NumpyEngine):
        def set_option(self, name, value):
            calls.append((name, value))

    be = QsvBackend()
    be._engine_factory = lambda n, devices=(0,), rank=None, world_size=None: Eng(n, len(devices))
    qc = QCMRF([[0, 1], [1, 2]], random_theta(8))
    be.run(qc, shots=10, seed_simulator=1, engine_options={"zero_tracking": 1, "multi_r": 3})
    assert calls == [("zero_tracking", 1), ("multi_r", 3)]
    del calls[:]
    be.run(qc, shots=10, seed_simulator=1, engine_options={"multi_r": 2})
    assert calls == [("zero_tracking", 0), ("multi_r", 2)]           # the override of the first run is undone
    del calls[:]
    be.run(qc, shots=10, seed_simulator=1)
    assert calls == [("multi_r", 5)]
    del calls[:]
    be.run(qc, shots=10, seed_simulator=1)
    assert calls == []


@pytest.mark.parametrize("P", [2, 4, 8])
def test_planner_batches_shard_bit_swaps_into_one_exchange(P):
    """reference layout of a fused QCMRF circuit: the last log2(P) ancillas -- dense targets -- sit on
    the shard bits (QCMRF.py:231, qubit n+1+ii).  The planner brings them ALL in with one batched
    swap when the first of them is needed (one all-to-all), not one half-shard exchange each."""
    C = workloads.grid(2, 3)                          # n = 6, m = 7, W = 14
    th = random_theta(cf.model_shape(C)[3], seed=P)
    qc = QCMRF(C, th)
    g = P.bit_length() - 1
    amp, ing, pl, eng = run_numpy(qc, fusion=2, shards=P, layout="reference")
    assert np.abs(amp - cf.amplitudes(C, th)).max() < 1e-13
    swaps = [o for o in pl.ops if o.kind == "swap"]
    assert len(swaps) == 1 and len(swaps[0].a) == g and pl.n_exchanges == 1 == eng.n_exchanges
    L = 14 - g
    assert all(x >= L for x in swaps[0].a) and all(x < L for x in swaps[0].b)
    # one at a time (batch_swaps off) is the same state with g exchange steps
    ops = passes.optimise(ing.ops, level=2)
    pl1 = planner.plan(ops, 14, P, "reference", batch_swaps=False)
    assert pl1.n_exchanges == g and all(len(o.a) == 1 for o in pl1.ops if o.kind == "swap")
    e1 = NumpyEngine(14, P)
    program.run_stepwise(e1, pl1.ops)
    assert np.abs(logical_amplitudes(e1, pl1.layout, 14) - cf.amplitudes(C, th)).max() < 1e-13


@pytest.mark.parametrize("shards", [1, 4])
def test_expectation_hamiltonian_host_path(models, shards):
    """QCMRF.expectation_hamiltonian / expectation_sufficient_statistic through the backend's layout
    mapping (numpy stand-in engine): <H> in the post-selected state is the Gibbs average of
    H[x] = -sum_C theta_{C,x_C}; unconditioned it is the plain mean (the variables stay uniform)."""
    be = QsvBackend(devices=(0,) * shards)
    be._engine_factory = lambda n, devices=(0,), rank=None, world_size=None: NumpyEngine(n, len(devices))
    for j in (2, 5):
        C = models["0.5"]["GRAPHS"][j]
        th = models["0.5"]["THETAS"][str(j)][4]
        qc = QCMRF(C, th)
        p, Z = cf.gibbs_pmf(C, th)
        H = -(np.log(p) + np.log(Z))                      # p = exp(-H) / Z, independent of the product's own H
        assert np.abs(H - qc.hamiltonian_diagonal()).max() < 1e-12
        n = qc.num_vertices
        for opts in ({}, {"fold_fresh": False}, {"fusion": 0}):
            val, prob = qc.expectation_hamiltonian(be, **opts)
            assert abs(val - float((p * H).sum())) < 1e-12 and abs(prob - Z / 2 ** n) < 1e-12
            val, prob = qc.expectation_hamiltonian(be, post_selected=False, **opts)
            assert abs(val - H.mean()) < 1e-12 and abs(prob - 1.0) < 1e-12
        y = (1, 0, 1)[:len(C[-1])]
        val, prob = qc.expectation_sufficient_statistic(be, C[-1], y)
        assert abs(val - float((p * qc.sufficient_statistic_diagonal(C[-1], y)).sum())) < 1e-12


def _kinds(ops):
    k = {}
    for o in ops:
        k[o.kind] = k.get(o.kind, 0) + 1
    return k


def test_qiskit_shaped_lowering_helpers_are_exact():
    """tests/_qiskit_shapes.py: ZSX re-synthesis of one-qubit runs and CX cancellation preserve the
    unitary including the global phase (checked through the gate-level oracle, fusion 0)"""
    from _qiskit_shapes import lower_like_qiskit, zsx_synthesis
    rs = np.random.RandomState(5)
    for _ in range(50):
        U = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2))[0]
        seq, g = zsx_synthesis(U)
        assert [n for n, _ in seq] == ["rz", "sx", "rz", "sx", "rz"]
    for seed in range(4):
        qc = rand_circuit(7, 60, 300 + seed)
        t = lower_like_qiskit(qc, extra_phase=0.37)
        assert set(t.count_ops()) <= {"cx", "rz", "sx", "x", "measure", "barrier"}
        amp, ing, pl, _ = run_numpy(t, fusion=0)
        assert np.abs(amp - np.exp(0.37j) * oracle_state_of(qc)).max() < 1e-12


@pytest.mark.parametrize("graph", [2, 3, "chain7", "grid23"])
def test_qiskit_shaped_lowered_circuits_come_back_as_one_multiplexer_per_clique(models, graph):
    """What run_experiment.py:52 really feeds the simulator has been through a transpiler's clean-up
    passes: one-qubit runs re-synthesised as rz sx rz (sx rz) and placed wherever the wire's next CX
    is, X gates of neighbouring AND blocks merged or cancelled, the H that closes one CCX on the
    scratch qubit cancelled against the H that opens the next ACROSS clique blocks, a global phase
    (tests/_qiskit_shapes.py).  Structure recovery must not depend on the tidy gate order of
    qcmrf_amd.transpile: still ONE multiplexer per (pairwise) clique, variables never dense, the
    opening H gates folded into the init write; with fold_fresh nothing but init x diagonal factors.
    (Gate-for-gate equality with a particular Qiskit version's output is parity unpinned: Qiskit is
    not installable here.)"""
    from _qiskit_shapes import lower_like_qiskit
    C = {2: workloads.REFERENCE_GRAPHS[2], 3: workloads.REFERENCE_GRAPHS[3], "chain7": workloads.chain(7),
         "grid23": workloads.grid(2, 3)}[graph]
    n, m, W, dim = cf.model_shape(C)
    th = random_theta(dim, seed=7)
    qc = QCMRF(C, th)
    want = cf.amplitudes(C, th)
    for variant in (dict(merge=True, cancel=True), dict(merge=True, cancel=False), dict(merge=False, cancel=True)):
        t = lower_like_qiskit(qc, extra_phase=-1.1, **variant)
        assert len(t.data) > 150 * m and abs(t.global_phase) > 1
        for fusion in (0, 2, 3):
            amp, ing, pl, _ = run_numpy(t, fusion=fusion)
            assert np.abs(amp - np.exp(-1.1j) * want).max() < 1e-11, (variant, fusion)
        be = QsvBackend()
        ing, pl = be.compile(t, fold_fresh=False, layout="reference")               # physical = logical qubits
        k = _kinds(pl.ops)
        assert k.get("mux", 0) == m and "kq" not in k and "u" not in k and "x" not in k, (variant, k)
        assert pl.ops[0].kind == "init" and pl.ops[0].mask == (1 << n) - 1          # every variable's H folded into the init write
        assert all(len(o.ctrls) == 3 for o in pl.ops if o.kind == "mux")            # two variables + the AND scratch qubit
        ing, pl = be.compile(t)                                                       # fold_fresh: init x diagonal factors only
        assert set(_kinds(pl.ops)) == {"init", "diag"}, _kinds(pl.ops)


def test_gate_by_gate_programs_keep_masked_targets_off_the_lane_bits():
    """planner.choose_layout, general policy: in a program made mostly of controlled X / phase / 2x2 ops (the
    reference's stream at fusion 0) the lane bits (physical 0..5) go to the qubits that are the target of the
    FEWEST such ops -- a controlled X costs 31 us on a register bit and 74 us on a lane bit of a general pass,
    an uncontrolled X nothing anywhere -- and the fused forms of the same circuit keep the round-1 layout."""
    from qcmrf_amd import QCMRF, workloads as wl
    from qcmrf_amd.backend import QsvBackend
    for W in (20, 28, 34):
        C = wl.for_width(W)
        qc = QCMRF(C, wl.theta_halfnorm(wl.dimension(C)))
        ing, pl = QsvBackend().compile(qc, fusion=0)
        ccx = [o for o in pl.ops if o.kind == "x" and len(o.ctrls) > 0]
        assert ccx and all(o.target >= 6 for o in ccx), W
        # the AND scratch qubit, target of every CCX, sits right above the lanes
        assert len({o.target for o in ccx}) == 1 and ccx[0].target <= 8
        # the pure controls (the MRF variables) select whole workgroups: they sit on the highest bits below the scratch
        # qubit's known-zero... i.e. above every target qubit; the lane bits carry ancillas (two dense gates each)
        n = qc.num_vertices
        var_pos = sorted(pl.layout[q] for q in range(n))
        tgt_pos = sorted(pl.layout[q] for q in range(n, qc.num_qubits))
        assert var_pos[0] > tgt_pos[-1] and tgt_pos[:6] == [0, 1, 2, 3, 4, 5], (W, var_pos, tgt_pos)
        # a permutation of the qubits, whatever the policy
        assert sorted(pl.layout) == list(range(qc.num_qubits))
        ing3, pl3 = QsvBackend().compile(qc, fusion=3)
        assert all(o.kind in ("init", "diag") for o in pl3.ops)
