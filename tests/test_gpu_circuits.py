"""GPU parity of whole QCMRF circuits: engine (through the C ABI) vs the oracle.

Tolerances: amplitudes 1e-12 absolute, probabilities 1e-10 (the bar BASELINE.json states for
fp64); counts are compared statistically (Aer's own committed counts are unseeded samples)."""
import numpy as np
import pytest

from conftest import random_theta
from oracle import closed_form as cf, gate_stream as gs, sv_numpy as sv

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    from qcmrf_amd.backend import QsvBackend
    b = QsvBackend()
    yield b
    b.close()


def logical_index(layout, n_qubits):
    p = np.arange(2 ** n_qubits, dtype=np.uint64)
    l = np.zeros_like(p)
    for q, pos in enumerate(layout):
        l |= ((p >> np.uint64(pos)) & np.uint64(1)) << np.uint64(q)
    return l


def run_state(be, qc, **opts):
    """evolve only (shots=0) and return amplitudes in LOGICAL index order + metadata"""
    res = be.run(qc, shots=0, **opts).result()
    meta = res.metadata(0)
    out = be.statevector()
    return out, meta


@pytest.mark.parametrize("fusion", [0, 1, 2, 3])
def test_reference_graphs_amplitudes(be, models, fusion):
    from qcmrf_amd import QCMRF
    for s in ("0.1", "0.5"):
        for j, C in enumerate(models[s]["GRAPHS"]):
            for rep in (0, 7):
                th = models[s]["THETAS"][str(j)][rep]
                amp, meta = run_state(be, QCMRF(C, th), fusion=fusion)
                want = cf.amplitudes(C, th)
                assert np.abs(amp - want).max() < 1e-12
                assert np.abs(np.abs(amp) ** 2 - cf.probabilities(C, th)).max() < 1e-10


def test_config1_golden_vector(be, config1):
    """BASELINE config 1 against tests/golden/config1.json.  That file holds the ORACLE's output
    (closed form, written by tests/golden/make_golden.py) -- the reference commits no amplitude
    vector, so this is HIP-vs-oracle at 1e-12, not HIP-vs-reference; the reference's own outputs
    (10 000-shot Aer counts) pin the same circuit statistically in test_counts_vs_closed_form_and_vs_aer."""
    from qcmrf_amd import QCMRF
    amp, _ = run_state(be, QCMRF(config1["cliques"], config1["theta"]))
    want = np.array(config1["amp_re"]) + 1j * np.array(config1["amp_im"])
    assert np.abs(amp - want).max() < 1e-12
    assert np.abs(np.abs(amp) ** 2 - np.array(config1["probabilities"])).max() < 1e-10


def test_exec_and_stepwise_agree_with_gate_level_oracle(be):
    """fusion 0 = the reference's own gate stream, gate by gate; checked against the numpy
    gate-level simulator (not the closed form) at W = 14."""
    from qcmrf_amd import QCMRF, _lib, program
    C = gs.chain_cliques(7)
    th = random_theta(cf.model_shape(C)[3])
    qc = QCMRF(C, th)
    W = qc.num_qubits
    want = sv.run_stream(gs.reference_stream(C, th), W)
    amp, meta = run_state(be, qc, fusion=0)
    assert meta["n_device_ops"] == meta["n_source_ops"] - (W - 1) + 1     # every gate, + init, - measures
    assert np.abs(amp - want).max() < 1e-12
    ing, pl = be.compile(qc, fusion=0)
    with _lib.Engine(W) as e:
        program.run_stepwise(e, pl.ops)
        got = np.empty(2 ** W, dtype=np.complex128)
        got[logical_index(pl.layout, W).astype(np.int64)] = e.amplitudes()
        assert np.abs(got - want).max() < 1e-12


@pytest.mark.parametrize("fusion", [0, 2])
def test_config2_chain_w20(be, fusion):
    from qcmrf_amd import QCMRF
    C = gs.chain_cliques(10)
    th = random_theta(36)
    amp, meta = run_state(be, QCMRF(C, th), fusion=fusion)
    assert meta["n_qubits"] == 20
    assert np.abs(amp - cf.amplitudes(C, th)).max() < 1e-12


@pytest.mark.parametrize("layout", ["auto", "reference"])
@pytest.mark.parametrize("P", [2, 4, 8])
@pytest.mark.parametrize("fusion", [0, 2])
def test_virtual_shards_match_single_shard(be, P, layout, fusion):
    """P shards on one device: shard-bit controls / table slicing / exchanges are the real
    library code; only the transport (device copy instead of RCCL) differs from multi-GPU."""
    from qcmrf_amd import QCMRF
    C = gs.grid_cliques(2, 3)                      # n=6, m=7, W=14
    th = random_theta(cf.model_shape(C)[3], seed=P)
    amp, meta = run_state(be, QCMRF(C, th), fusion=fusion, layout=layout, devices=(0,) * P)
    assert meta["n_shards"] == P
    if layout == "auto" and fusion == 2:
        assert meta["n_exchanges"] == 0
    if layout == "reference":
        assert meta["n_exchanges"] > 0
    assert np.abs(amp - cf.amplitudes(C, th)).max() < 1e-12


def test_counts_vs_closed_form_and_vs_aer(be, models, aer_counts):
    from qcmrf_amd import QCMRF
    m = models["0.5"]
    circs, dists = [], []
    for j, C in enumerate(m["GRAPHS"]):
        for rep in range(10):
            circs.append(QCMRF(C, m["THETAS"][str(j)][rep]))
            dists.append((C, m["THETAS"][str(j)][rep]))
    counts = be.run(circs, shots=10000, seed_simulator=1984).result().get_counts()
    assert isinstance(counts, list) and len(counts) == 70
    chis, chis2 = [], []
    for k, (C, th) in enumerate(dists):
        n, mm, W, dim = cf.model_shape(C)
        assert all(len(key) == W and isinstance(v, int) for key, v in counts[k].items())
        assert sum(counts[k].values()) == 10000
        p = cf.probabilities(C, th)
        obs = np.zeros(2 ** W)
        for key, v in counts[k].items():
            obs[int(key, 2)] += v
        assert obs[p == 0].sum() == 0                       # support, incl. classical bit n == '0'
        sel = p * 10000 > 5
        chis.append(((obs[sel] - p[sel] * 1e4) ** 2 / (p[sel] * 1e4)).sum() / max(sel.sum() - 1, 1))
        aer = np.zeros(2 ** W)
        for key, v in aer_counts["0.5"][k].items():
            aer[int(key, 2)] += v
        both = (obs + aer) > 10                             # two-sample chi^2 against Aer's own counts
        chis2.append(((obs[both] - aer[both]) ** 2 / (obs[both] + aer[both])).sum() / max(both.sum() - 1, 1))
    assert 0.85 < np.mean(chis) < 1.15
    assert 0.85 < np.mean(chis2) < 1.15


def test_extract_probs_pipeline(be, models):
    """run_experiment.py + eval.py arithmetic end to end: fidelity with the exact Gibbs pmf."""
    from qcmrf_amd import QCMRF, extract_probs, fidelity
    C = models["0.5"]["GRAPHS"][3]
    th = models["0.5"]["THETAS"]["3"][0]
    qc = QCMRF(C, th)
    counts = be.run(qc, shots=100000, seed_simulator=7).result().get_counts()
    q, delta = extract_probs(counts, qc.num_vertices, qc.num_cliques + 1)
    p, Z = cf.gibbs_pmf(C, th)
    assert fidelity(p, q) > 0.999
    assert abs(delta - Z / 2 ** qc.num_vertices) < 0.01


@pytest.mark.parametrize("fusion,fold", [(2, True), (3, False), (1, True), (0, True)])
def test_init_passes_never_read_a_table_they_did_not_stage(be, fusion, fold):
    """Workgroups of an init pass that start from zeros skip staging the gate tables into LDS.  LDS is not
    cleared between kernels, so a read of an unstaged table is 'usually fine' (finite garbage times zero) --
    an early version did exactly that in the scalar shortcut of the first round and produced NaN once in a
    while.  With quiet NaNs left in every compute unit's LDS first (qsv_poison_lds) it fails every time."""
    from qcmrf_amd import QCMRF
    C = gs.chain_cliques(10)
    th = random_theta(36)
    be.run(QCMRF([[0, 1]], [-0.1] * 4), shots=0)                 # make sure the engine of this width exists below
    amp, meta = run_state(be, QCMRF(C, th), fusion=fusion, fold_fresh=fold)
    for _ in range(3):
        be.last_engine.poison_lds()
        amp, meta = run_state(be, QCMRF(C, th), fusion=fusion, fold_fresh=fold)
        assert not np.isnan(amp).any()
        assert np.abs(amp - cf.amplitudes(C, th)).max() < 1e-12


def _full_size_properties(be, C, **opts):
    """size-independent checks at sizes no oracle can hold: norm = 1; P(all ancillas 0) = Z/2^n;
    P(x | ancillas 0) = Gibbs pmf (1e-10); random amplitude slices (incl. the very top of the
    index range: 64-bit addressing) equal the closed form (1e-12); counts only on the support."""
    from qcmrf_amd import QCMRF
    n, m, W, dim = cf.model_shape(C)
    th = random_theta(dim)
    qc = QCMRF(C, th)
    res = be.run(qc, shots=4096, seed_simulator=1984, **opts).result()
    meta = res.metadata(0)
    lay = meta["layout"]
    eng = be.last_engine
    assert abs(eng.norm() - 1.0) < 1e-12
    var_phys = [lay[q] for q in range(n)]
    fix_mask = 0
    for q in range(n, W):
        fix_mask |= 1 << lay[q]
    px = eng.probabilities(var_phys, fix_mask, 0)          # index bit q <-> variable qubit q
    p, Z = cf.gibbs_pmf(C, th)                              # index has x_0 as MSB; variable v on qubit n-1-v
    assert abs(px.sum() - Z / 2 ** n) < 1e-10
    assert np.abs(px / px.sum() - p).max() < 1e-10
    rs = np.random.RandomState(0)
    starts = [int(x) for x in rs.randint(0, 2 ** 31 - 1, size=12).astype(np.int64) * (2 ** W // 2 ** 31)] + [0, 2 ** W - 4096]
    for start in starts:
        start = min(max(start, 0), 2 ** W - 4096)
        got = eng.amplitudes(start, 4096)
        pidx = np.arange(start, start + 4096, dtype=np.uint64)
        lidx = np.zeros_like(pidx)
        for q, pos in enumerate(lay):
            lidx |= ((pidx >> np.uint64(pos)) & np.uint64(1)) << np.uint64(q)
        assert np.abs(got - cf.amplitudes_at(C, th, lidx)).max() < 1e-12
    counts = res.get_counts()
    assert sum(counts.values()) == 4096
    ok = sum(v for k, v in counts.items() if int(k, 2) < 2 ** n)
    delta = Z / 2 ** n
    assert abs(ok / 4096 - delta) < 5 * np.sqrt(delta * (1 - delta) / 4096) + 1e-3
    assert all(k[W - 1 - n] == "0" for k in counts)         # classical bit n is never written
    pr = cf.probabilities_at(C, th, np.array([int(k, 2) for k in counts], dtype=np.uint64))
    assert (pr > 0).all()
    return meta


def test_full_size_w28_properties(be):
    """BASELINE config 3 (2x6 grid minus last edge, W = 28, 4 GiB state)"""
    C = gs.grid_cliques(2, 6, drop_last=1)
    assert cf.model_shape(C) == (12, 15, 28, 60)
    meta = _full_size_properties(be, C)
    assert meta["n_device_ops"] == 16
    _full_size_properties(be, C, engine_options={"zero_tracking": 1})
    _full_size_properties(be, C, fusion=1, engine_options={"zero_tracking": 0})
    # the reference's own stream gate by gate: general k_multi passes (tile updated in place, X frame with
    # bits pending across passes) at a size where the workgroups of a pass no longer run all at once
    meta = _full_size_properties(be, C, fusion=0, profile=True)
    assert meta["stats"]["kinds"]["multi"]["launches"] >= 5
    _full_size_properties(be, C, fusion=0, engine_options={"xframe": 0})


@pytest.mark.parametrize("P", [2, 4])
def test_full_size_w31_config4(be, P):
    """BASELINE config 4 (random graph G(10,20), W = 31, 32 GiB state) on 2 then 4 shards of one
    device: 16 / 8 GiB shards, so the shard-bit exchanges move real 8 / 4 GiB half-shards.
    layout='reference' keeps the ancillas -- the dense targets -- on the top bits (QCMRF.py:231-236,
    qubit n+1+ii), i.e. ON the shard bits: with fold_fresh off their multiplexers must sweep, so the
    planner has to exchange; layout='auto' shards by variable qubits and needs none."""
    from qcmrf_amd import workloads
    name, C = workloads.baseline_config(3)
    assert cf.model_shape(C) == (10, 20, 31, 80)
    devs = (0,) * P
    meta = _full_size_properties(be, C, devices=devs)                                   # default path: generator per shard
    assert meta["n_shards"] == P and meta["n_exchanges"] == 0
    meta = _full_size_properties(be, C, devices=devs, fold_fresh=False)                 # sweeps, exchange-free layout
    assert meta["n_exchanges"] == 0
    meta = _full_size_properties(be, C, devices=devs, layout="reference", fold_fresh=False)
    assert meta["n_exchanges"] >= 1
    st = be.last_engine.stats()
    assert st["exchanges"] == meta["n_exchanges"] and "exchange" in st["kinds"]
    be.run(__import__("qcmrf_amd").QCMRF([[0, 1]], [-0.1] * 4), shots=1, devices=(0,))  # frees the 32 GiB


def test_full_size_w32_64bit_addressing(be):
    """a 64 GiB shard (the size class of the 8-GPU config's 32 GiB shards and beyond): every byte
    offset above 2^32 and every index above 2^31 is live here"""
    from qcmrf_amd import workloads
    C = workloads.for_width(32)
    assert cf.model_shape(C)[2] == 32
    _full_size_properties(be, C)
    _full_size_properties(be, C, engine_options={"zero_tracking": 1})
    be.run(__import__("qcmrf_amd").QCMRF([[0, 1]], [-0.1] * 4), shots=1, engine_options={"zero_tracking": 0})   # frees the 64 GiB


@pytest.mark.parametrize("zero_tracking", [0, 1])
@pytest.mark.parametrize("multi_r", [0, 1, 2, 3, 4, 5, 6])
def test_exec_sweep_blocking_random_circuits(be, multi_r, zero_tracking):
    """qsv_exec groups consecutive gates into register-tiled k_multi passes; every register-tile
    width must reproduce the gate-by-gate result (controls on register / lane / block bits,
    tables with register-bit selects, X, phases, diagonals)."""
    from test_host_logic import rand_circuit, oracle_state_of
    for seed, nq in ((1, 9), (2, 12), (3, 13)):
        qc = rand_circuit(nq, 150, seed)
        want = oracle_state_of(qc)
        for fusion in (0, 2):
            amp, meta = run_state(be, qc, fusion=fusion,
                                  engine_options={"multi_r": multi_r, "zero_tracking": zero_tracking})
            assert np.abs(amp - want).max() < 1e-12, (seed, nq, fusion)
    be.run(qc, shots=0, engine_options={"multi_r": 5, "zero_tracking": 0})          # restore the defaults


@pytest.mark.parametrize("zero_tracking", [0, 1])
@pytest.mark.parametrize("multi_r", [0, 3, 5])
def test_exec_sweep_blocking_sharded(be, multi_r, zero_tracking):
    from qcmrf_amd import QCMRF
    C = gs.grid_cliques(2, 3)
    th = random_theta(cf.model_shape(C)[3], seed=3)
    for layout in ("auto", "reference"):
        for fusion in (0, 2):
            amp, meta = run_state(be, QCMRF(C, th), fusion=fusion, layout=layout, devices=(0,) * 4,
                                  engine_options={"multi_r": multi_r, "zero_tracking": zero_tracking})
            assert np.abs(amp - cf.amplitudes(C, th)).max() < 1e-12
    be.run(QCMRF(C, th), shots=0, devices=(0,) * 4, engine_options={"multi_r": 5, "zero_tracking": 0})


def test_zero_tracking_qcmrf_all_graphs(be, models):
    """opt-in zero tracking (skip the provably-zero part of the shard) is exact"""
    from qcmrf_amd import QCMRF
    for j, C in enumerate(models["0.5"]["GRAPHS"]):
        th = models["0.5"]["THETAS"][str(j)][5]
        for fusion in (0, 1, 2):
            amp, meta = run_state(be, QCMRF(C, th), fusion=fusion, engine_options={"zero_tracking": 1})
            assert np.abs(amp - cf.amplitudes(C, th)).max() < 1e-12
    C = gs.chain_cliques(10)
    th = random_theta(36)
    amp, meta = run_state(be, QCMRF(C, th), engine_options={"zero_tracking": 1})
    assert np.abs(amp - cf.amplitudes(C, th)).max() < 1e-12
    counts = be.run(QCMRF(C, th), shots=2000, seed_simulator=3).result().get_counts()
    p = cf.probabilities(C, th)
    assert all(p[int(k, 2)] > 0 for k in counts)
    be.run(QCMRF(C, th), shots=0, engine_options={"zero_tracking": 0})


@pytest.mark.parametrize("fusion", [0, 2, 3])
def test_lowered_basis_gate_circuits(be, models, fusion):
    """the form run_experiment.py:52-56 actually feeds the simulator: {cx, id, rz, sx, x}.
    fusion 3 re-assembles the blocks (dense windows, f64 MFMA for what stays dense)."""
    from qcmrf_amd import QCMRF
    from qcmrf_amd.transpile import transpile
    from test_host_logic import rand_circuit, oracle_state_of
    for j in range(7):                                        # every graph of run_experiment.py:20, the 4-variable clique included
        C = models["0.25"]["GRAPHS"][j]
        th = models["0.25"]["THETAS"][str(j)][2]
        amp, meta = run_state(be, transpile(QCMRF(C, th)), fusion=fusion)
        assert np.abs(amp - cf.amplitudes(C, th)).max() < 5e-12
        if fusion == 3:                                       # re-assembled completely: no dense k-qubit gate reaches the device
            assert "kq" not in [o.kind for o in be.last_plan.ops], j
    for seed in (1, 2, 3):
        qc = rand_circuit(11, 120, 70 + seed)
        amp, meta = run_state(be, transpile(qc), fusion=fusion)
        assert np.abs(amp - oracle_state_of(qc)).max() < 5e-12
    C = gs.chain_cliques(8)                                   # W = 16: one mux per clique again
    th = random_theta(28)
    t = transpile(QCMRF(C, th))
    amp, meta = run_state(be, t, fusion=fusion)
    assert np.abs(amp - cf.amplitudes(C, th)).max() < 5e-12
    if fusion == 3:
        assert meta["n_device_ops"] < 20 < meta["n_source_ops"] / 100
    counts = be.run(t, shots=3000, seed_simulator=5, fusion=fusion).result().get_counts()
    p = cf.probabilities(C, th)
    assert sum(counts.values()) == 3000 and all(p[int(k, 2)] > 1e-12 for k in counts)


def test_run_experiment_driver_end_to_end(tmp_path, models):
    """python -m qcmrf_amd.run_experiment: same files as run_experiment.py:35-38,59-61, then
    python -m qcmrf_amd.eval on its output (eval.py --mode file)."""
    import json
    from qcmrf_amd import run_experiment, eval as ev
    counts = run_experiment.main(["--scale", "0.5", "--shots", "10000", "--outdir", str(tmp_path),
                                  "--seed-simulator", "1984"])
    m = json.load(open(tmp_path / "models_0.5.json"))
    assert m["GRAPHS"] == models["0.5"]["GRAPHS"]
    assert m["THETAS"] == models["0.5"]["THETAS"]                 # bit-identical theta draws
    saved = json.load(open(tmp_path / "result_simulation_0.5.json"))
    assert saved == counts and len(saved) == 70 and all(sum(c.values()) == 10000 for c in saved)
    rows = ev.main(["--results", "result_simulation_0.5.json", "--scale", "0.5", "--resdir", str(tmp_path)])
    assert all(r[1] > 0.99 for r in rows)                          # mean fidelity with the exact Gibbs pmf
    deltas = [cf.success_probability(C, models["0.5"]["THETAS"][str(j)][0]) for j, C in enumerate(m["GRAPHS"])]
    assert all(abs(r[4] - np.mean([cf.success_probability(C, th) for th in models["0.5"]["THETAS"][str(j)]])) < 0.02
               for j, (C, r) in enumerate(zip(m["GRAPHS"], rows)))


@pytest.mark.parametrize("zero_tracking", [0, 1])
@pytest.mark.parametrize("multi_r", [2, 5, 6])
def test_fused_tile_sums_sampling(be, multi_r, zero_tracking):
    """the last k_multi pass leaves per-tile |amp|^2 sums; sampling from them (tile-order locate
    kernel) must follow the same distribution as the separate read pass, reproducibly"""
    from qcmrf_amd import QCMRF
    C = gs.chain_cliques(9)                                    # W = 18
    th = random_theta(32)
    qc = QCMRF(C, th)
    p = cf.probabilities(C, th)
    shots = 60000
    res = {}
    for fused in (1, 0):
        opts = {"multi_r": multi_r, "zero_tracking": zero_tracking, "fused_sums": fused}
        c1 = be.run(qc, shots=shots, seed_simulator=42, engine_options=opts).result().get_counts()
        c2 = be.run(qc, shots=shots, seed_simulator=42, engine_options=opts).result().get_counts()
        assert c1 == c2
        assert abs(be.last_engine.norm() - 1.0) < 1e-12
        obs = np.zeros(p.size)
        for k, v in c1.items():
            obs[int(k, 2)] += v
        assert obs[p == 0].sum() == 0
        sel = p * shots > 5
        chi = ((obs[sel] - p[sel] * shots) ** 2 / (p[sel] * shots)).sum() / (sel.sum() - 1)
        assert 0.85 < chi < 1.15, (fused, chi)
        res[fused] = obs
    be.run(qc, shots=0, engine_options={"multi_r": 5, "zero_tracking": 0, "fused_sums": 1})


def test_trajectory_mode_small_and_beyond_statevector_reach(be, models):
    """method='trajectory': mid-circuit measurements taken when they occur, ancilla slot recycled.
    (a) small graph: full chi^2 against the closed form; (b) chain of 22 variables: the circuit
    has W = 44 qubits -- no statevector of that width exists -- but only 24 are ever live."""
    from qcmrf_amd import QCMRF, extract_probs, fidelity, mrf
    from qcmrf_amd.backend import QsvBackend
    tb = QsvBackend(method="trajectory")
    C = models["0.5"]["GRAPHS"][3]
    th = models["0.5"]["THETAS"]["3"][1]
    shots = 50000
    res = tb.run(QCMRF(C, th), shots=shots, seed_simulator=3).result()
    counts = res.get_counts()
    n, m, W, dim = cf.model_shape(C)
    assert res.metadata(0)["live_qubits"] == n + 2 and sum(counts.values()) == shots
    p = cf.probabilities(C, th)
    obs = np.zeros(p.size)
    for k, v in counts.items():
        obs[int(k, 2)] += v
    assert obs[p == 0].sum() == 0
    sel = p * shots > 5
    assert 0.8 < ((obs[sel] - p[sel] * shots) ** 2 / (p[sel] * shots)).sum() / (sel.sum() - 1) < 1.25
    # (b)
    C = gs.chain_cliques(22)
    th = random_theta(4 * 21, scale=0.25)
    qc = QCMRF(C, th)
    assert qc.num_qubits == 44
    shots = 4096
    res = tb.run(qc, shots=shots, seed_simulator=9).result()
    meta = res.metadata(0)
    assert meta["live_qubits"] == 24 and meta["n_segments"] == 21
    counts = res.get_counts()
    assert sum(counts.values()) == shots and all(len(k) == 44 and k[44 - 1 - 22] == "0" for k in counts)
    pg, lnZ = mrf.gibbs_pmf(C, th)
    delta = np.exp(lnZ) / 2 ** 22
    ok = sum(v for k, v in counts.items() if int(k, 2) < 2 ** 22)
    assert abs(ok / shots - delta) < 5 * np.sqrt(delta * (1 - delta) / shots) + 1e-3
    # samples that passed every ancilla follow the Gibbs distribution: compare single-site and
    # nearest-neighbour marginals (the full 2^22 pmf cannot be resolved by 4096 shots)
    xs = np.array([[int(b) for b in k[-22:]] for k, v in counts.items() if int(k, 2) < 2 ** 22 for _ in range(v)])
    idx = np.arange(2 ** 22)
    for v0 in (0, 7, 21):
        bit = (idx >> (21 - v0)) & 1
        assert abs(xs[:, v0].mean() - pg[bit == 1].sum()) < 5 * 0.5 / np.sqrt(len(xs)) + 1e-3
    for v0 in (3, 12):
        both = ((idx >> (21 - v0)) & 1) & ((idx >> (20 - v0)) & 1)
        assert abs((xs[:, v0] * xs[:, v0 + 1]).mean() - pg[both == 1].sum()) < 5 * 0.5 / np.sqrt(len(xs)) + 1e-3
    tb.close()


def test_trajectory_mode_is_reproducible_for_a_seed():
    """branch probabilities are summed in a fixed order (no atomics): the same seed_simulator walks the same tree of
    outcomes and returns the same counts, run after run (BASELINE seeds the sampler: seed_simulator=1984)"""
    from qcmrf_amd import QCMRF, workloads
    from qcmrf_amd.backend import QsvBackend
    tb = QsvBackend(method="trajectory")
    C = workloads.chain(14)
    qc = QCMRF(C, workloads.theta_halfnorm(workloads.dimension(C), scale=0.25))
    runs = [tb.run(qc, shots=2048, seed_simulator=1984).result() for _ in range(3)]
    assert runs[0].get_counts() == runs[1].get_counts() == runs[2].get_counts()
    assert len({r.metadata(0)["branch_nodes"] for r in runs}) == 1
    assert tb.run(qc, shots=2048, seed_simulator=7).result().get_counts() != runs[0].get_counts()
    tb.close()


@pytest.mark.parametrize("fusion", [0, 3])
def test_random_circuit_w20_against_c_oracle(be, fusion):
    """a 20-qubit random circuit (all gate kinds, targets on lane / register / block bits) against
    the plain-C gate-level oracle: the general k_multi paths at a size where every lane of every
    wavefront is live"""
    from test_host_logic import rand_circuit
    from oracle import cref
    nq = 20
    qc = rand_circuit(nq, 160, 2024)
    ref = cref.RefState(nq)
    for ci in qc.data:
        name, p = ci.operation.name, ci.operation.params
        q = [qc.find_bit(b).index for b in ci.qubits]
        if name == "x": ref.apply_mcx([], q[0])
        elif name in sv.MATS: ref.apply_1q(q[0], sv.MATS[name])
        elif name == "cx": ref.apply_mcx([q[0]], q[1])
        elif name == "ccx": ref.apply_mcx(q[:2], q[2])
        elif name == "rz": ref.apply_1q(q[0], sv.rz(p[0]))
        elif name == "ry": ref.apply_1q(q[0], sv.ry(p[0]))
        elif name == "cp": ref.apply_mcphase(q, p[0])
        elif name == "cz": ref.apply_mcphase(q, np.pi)
        elif name == "swap":
            ref.apply_mcx([q[0]], q[1]); ref.apply_mcx([q[1]], q[0]); ref.apply_mcx([q[0]], q[1])
        else: raise AssertionError(name)
    # multi_nt / init_prod_nt = 1: the non-temporal kernel forms, which a state this small would not select by itself
    for opts in ({}, {"zero_tracking": 1}, {"lane_targets": 0}, {"multi_r": 3}, {"multi_nt": 1, "init_prod_nt": 1},
                 {"multi_nt": 1, "zero_tracking": 1}, {"xframe": 0}):
        amp, meta = run_state(be, qc, fusion=fusion, engine_options=opts)
        assert np.abs(amp - ref.state).max() < 1e-12, (fusion, opts)
    be.run(qc, shots=0, engine_options={"zero_tracking": 0, "lane_targets": 1, "multi_r": 5})


def _lane_program(W, regs, lanes, rx_like, seed, with_diag=False, init=True, hints=()):
    """init + one multiplexed 2x2 per target (selects on two quiet high bits): the shape of a fused
    QCMRF circuit, with as many distinct targets as one k_multi pass can be made to hold"""
    from qcmrf_amd import ir
    rs = np.random.RandomState(seed)

    def mat():
        if rx_like:
            a = rs.rand() * 3
            return np.array([[np.cos(a), -1j * np.sin(a)], [-1j * np.sin(a), np.cos(a)]])
        q, _ = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2))
        return q
    sel = [W - 1, W - 2]
    quiet = (1 << W) - 1
    for q in regs + lanes:
        quiet &= ~(1 << q)
    ops = [ir.op_init(quiet)] if init else []
    for t in regs + lanes:
        ops.append(ir.op_mux(sel, t, np.array([mat() for _ in range(4)])))
        if t in hints:
            ops[-1].new_pass = True
        if with_diag and t == regs[-1]:
            ops.append(ir.op_diag([3, W - 3], np.exp(1j * rs.randn(4))))
    return ops


@pytest.mark.parametrize("rx_like", [True, False])
def test_borrowed_lane_bits_and_lane_map(rx_like):
    """k_multi lends lane bits 3..5 to targets anywhere below bit 28 (a wave load becomes 8 x 128 B)
    and may remap lane bit 5 purely for the access pattern: both change only WHERE a lane's
    amplitudes live, so every setting must reproduce the numpy engine to 1e-12 -- amplitudes and
    the tile-order sampling that follows the last pass."""
    from qcmrf_amd import _lib, program
    from oracle.sharded_numpy import NumpyEngine
    W = 17
    cases = [
        ([6, 7, 8, 9, 10], [11, 12, 13, 0, 1, 2], ()),          # 5 registers + 3 borrowed + 3 static lanes
        ([6, 7, 8, 9, 10], [11, 4, 13, 2], ()),                 # a static lane target on bit 4 keeps its lane
        ([9, 7, 12, 6, 14], [8, 13, 10], ()),                   # scattered registers, borrowed bits between them
        ([6, 7, 8, 9, 10], [], ()),                             # lane_map = 1 pattern (lane bit 5 -> bit 11)
        ([6, 7, 8, 9, 10], [0, 3, 4], ()),                      # ... with static lane gates beside it
        ([6, 7, 8, 9, 10], [11, 12, 13, 14, 0], (14,)),         # planner hint: bit 14 opens the second pass
    ]
    with _lib.Engine(W) as eng:
        for ci, (regs, lanes, hints) in enumerate(cases):
            for with_diag in (False, True):
                for init in (True, False):
                    ops = _lane_program(W, regs, lanes, rx_like, 100 + ci, with_diag, init, hints)
                    rec, data = program.encode(ops)
                    ref = NumpyEngine(W)
                    if not init:
                        ref.init_uniform((1 << W) - 1)
                    ref.exec(rec, data)
                    want = ref.amplitudes()
                    for dyn, lmap in ((3, 1), (0, 0), (1, 0), (2, 1), (3, 11 << 10 | 12 << 5)):
                        eng.set_option("dyn_lanes", dyn)
                        eng.set_option("lane_map", lmap)
                        if not init:
                            eng.init_uniform((1 << W) - 1)
                        eng.reset_stats()
                        eng.exec(rec, data)
                        got = eng.amplitudes()
                        assert np.abs(got - want).max() < 1e-12, (ci, with_diag, init, dyn, lmap)
                        launches = sum(v["launches"] for v in eng.stats()["kinds"].values())
                        if dyn == 3 and not with_diag and not hints and len(regs) + len(lanes) <= 11:
                            assert launches == 1, (ci, launches)      # everything rode in ONE pass
                        if hints and dyn == 3:
                            assert launches == 2
        # sampling straight from the tile sums of a pass that used borrowed lanes
        ops = _lane_program(W, [6, 7, 8, 9, 10], [11, 12, 13, 0, 1, 2], rx_like, 7)
        rec, data = program.encode(ops)
        eng.set_option("dyn_lanes", 3)
        eng.exec(rec, data)
        p = np.abs(eng.amplitudes()) ** 2
        shots = 200000
        idx = eng.sample(shots, 11)
        assert (eng.sample(shots, 11) == idx).all()
        obs = np.bincount(idx.astype(np.int64), minlength=p.size)
        assert obs[p == 0].sum() == 0
        # 2^17 outcomes are too many for a per-outcome chi^2 at this shot count: bin by 64
        pb, ob = p.reshape(-1, 64).sum(1), obs.reshape(-1, 64).sum(1)
        keep = pb * shots > 5
        chi = ((ob[keep] - pb[keep] * shots) ** 2 / (pb[keep] * shots)).sum() / (keep.sum() - 1)
        assert 0.85 < chi < 1.15, chi


def test_init_product_generator_against_numpy_engine():
    """init + diagonal factors only -> k_init_prod (one write-only pass).  Factors on no / one /
    several of the generator's register bits (6..10), unpopulated bits in and outside the tile,
    sampling from its tile sums; same program through the ordinary k_multi path (init_prod = 0)."""
    from qcmrf_amd import _lib, ir, program
    from oracle.sharded_numpy import NumpyEngine
    W = 16
    rs = np.random.RandomState(5)
    with _lib.Engine(W) as eng:
        for case in range(6):
            mask = (1 << W) - 1
            for q in rs.choice(W, size=case % 4, replace=False):      # some qubits stay |0>
                mask &= ~(1 << int(q))
            ops = [ir.op_init(mask)]
            for _ in range(12):
                k = int(rs.randint(1, 5))
                qs = [int(q) for q in rs.choice(W, size=k, replace=False)]
                ops.append(ir.op_diag(qs, rs.randn(2 ** k) + 1j * rs.randn(2 ** k)))
            ops.append(ir.op_diag([6, 7, 8, 9, 10], np.exp(1j * rs.randn(32))))      # all register bits at once
            rec, data = program.encode(ops)
            ref = NumpyEngine(W)
            ref.exec(rec, data)
            want = ref.amplitudes()
            scale = np.abs(want).max()
            for ip in (1, 0, 2):
                eng.set_option("init_prod", 1 if ip else 0)
                eng.set_option("init_prod_nt", 1 if ip == 2 else 0)          # 2: the generator with non-temporal stores
                eng.reset_stats()
                eng.exec(rec, data)
                kinds = {k: v["launches"] for k, v in eng.stats()["kinds"].items()}
                assert ("init_prod" in kinds) == bool(ip), kinds
                if ip:
                    assert kinds == {"init_prod": 1}
                assert np.abs(eng.amplitudes() - want).max() < 1e-12 * max(1.0, scale), (case, ip)
        eng.set_option("init_prod", 1)
        eng.set_option("init_prod_nt", -1)
        # a normalised product state: sample from the generator's tile sums
        ops = [ir.op_init((1 << W) - 1 - (1 << 3))]          # bit 7 populated by the folded-gate-like factor below
        for q in range(0, W - 1, 2):
            th = rs.rand(4) * 3
            ops.append(ir.op_diag([q, q + 1], np.exp(1j * th)))
        a = rs.rand() * 3
        ops.append(ir.op_diag([0, 7], np.sqrt(2.0) * np.array([np.cos(a), np.cos(a / 2), -1j * np.sin(a), -1j * np.sin(a / 2)])))
        rec, data = program.encode(ops)
        eng.exec(rec, data)
        p = np.abs(eng.amplitudes()) ** 2
        assert abs(p.sum() - 1.0) < 1e-12 and abs(eng.norm() - 1.0) < 1e-12
        shots = 100000
        idx = eng.sample(shots, 3)
        obs = np.bincount(idx.astype(np.int64), minlength=p.size)
        assert obs[p == 0].sum() == 0
        pb, ob = p.reshape(-1, 64).sum(1), obs.reshape(-1, 64).sum(1)
        keep = pb * shots > 5
        chi = ((ob[keep] - pb[keep] * shots) ** 2 / (pb[keep] * shots)).sum() / (keep.sum() - 1)
        assert 0.8 < chi < 1.2, chi


def test_rccl_binding_selftest():
    """RCCL refuses two ranks on one device, so the 2-GPU exchange cannot run on the one-GPU test
    box; what can: the dlopen binding, a 1-rank communicator and the grouped ncclSend/ncclRecv +
    stream sequence the exchange uses, rank 0 to itself (8 MiB)."""
    from qcmrf_amd import _lib
    _lib.rccl_selftest(0, 1 << 20)


def test_full_size_w34_on_one_gpu(be):
    """BASELINE config 5 (2x7 grid, W = 34): 256 GiB of amplitudes on ONE MI355X (288 GiB) -- the
    workload bench.py times at every N.  Default path (generator) and full-width sweeps."""
    from qcmrf_amd import _lib
    free, total = _lib.device_memory(0)
    if free < 16 * 2 ** 34 + (6 << 30):
        pytest.skip("needs 262 GiB of free HBM, device has %.0f GiB free" % (free / 2 ** 30))
    C = gs.grid_cliques(2, 7)
    assert cf.model_shape(C) == (14, 19, 34, 76)
    be.close()
    meta = _full_size_properties(be, C)
    assert meta["n_device_ops"] == 20
    _full_size_properties(be, C, fold_fresh=False)
    _full_size_properties(be, C, fusion=0)                                     # 1154 gates one by one, 1.2 s
    be.run(__import__("qcmrf_amd").QCMRF([[0, 1]], [-0.1] * 4), shots=1)      # frees the 256 GiB


@pytest.mark.parametrize("P", [2, 4, 8])
def test_virtual_shards_default_path_generator(be, P):
    """the default (folded) path on P shards: every factor with a qubit on a shard bit is sliced
    per shard and each shard runs its own k_init_prod (L >= 14 local qubits); no exchanges"""
    from qcmrf_amd import QCMRF
    C = gs.chain_cliques(10)                       # n = 10, m = 9, W = 20
    th = random_theta(cf.model_shape(C)[3], seed=10 + P)
    for layout in ("auto", "reference"):
        amp, meta = run_state(be, QCMRF(C, th), layout=layout, devices=(0,) * P)
        assert meta["n_shards"] == P and meta["n_exchanges"] == 0
        assert np.abs(amp - cf.amplitudes(C, th)).max() < 1e-12
        kinds = be.last_engine.stats()["kinds"]
    counts = be.run(QCMRF(C, th), shots=4000, seed_simulator=2, devices=(0,) * P).result().get_counts()
    p = cf.probabilities(C, th)
    assert sum(counts.values()) == 4000 and all(p[int(k, 2)] > 0 for k in counts)
    be.run(QCMRF([[0, 1]], [-0.1] * 4), shots=1)


@pytest.mark.parametrize("cliques", [
    [[0, 1, 2], [2, 3, 4], [4, 5, 6], [6, 7, 8]],            # |C| = 3: W = 14, four-qubit factors
    [[0, 1, 2, 3], [3, 4, 5, 6], [6, 7, 8, 9]],              # |C| = 4: W = 14, five-qubit factors
    [[0, 1], [1, 2, 3], [3, 4, 5, 6], [6, 7], [7, 8, 9]],    # mixed sizes: W = 16
])
def test_larger_cliques_all_paths(be, cliques):
    """cliques of 3 and 4 variables (run_experiment.py:20 has them at toy width) at a width where
    the wide kernels run: default path (generator), multiplexed sweeps, gate by gate"""
    from qcmrf_amd import QCMRF
    dim = cf.model_shape(cliques)[3]
    th = random_theta(dim, seed=dim)
    want = cf.amplitudes(cliques, th)
    for opts in ({}, {"fold_fresh": False}, {"fusion": 2}, {"fusion": 0}, {"devices": (0, 0)}):
        amp, meta = run_state(be, QCMRF(cliques, th), **opts)
        assert np.abs(amp - want).max() < 1e-12, opts
    kinds = None
    be.run(QCMRF(cliques, th), shots=0)
    kinds = be.last_engine.stats()["kinds"]
    assert set(kinds) == {"init_prod"}, kinds


def test_expectation_hamiltonian_on_device(be):
    """f4 (QCMRF.py:159-193): <H> and <Phi_{C,y}> evaluated by qsv_expect_diag on the resident state.
    W = 20: against the Gibbs average from the closed-form oracle (H = -log p - log Z, not the
    product's own diagonal), every execution path, 1 and 4 shards; W = 28 (4 GiB): the same
    through the size-independent identity <H>_post = sum_x p(x) H(x) with p, Z from the exact MRF sum."""
    from qcmrf_amd import QCMRF, workloads
    C = gs.chain_cliques(10)                                   # n = 10, m = 9, W = 20
    th = random_theta(36, seed=4)
    qc = QCMRF(C, th)
    p, Z = cf.gibbs_pmf(C, th)
    H = -(np.log(p) + np.log(Z))
    for opts in ({}, {"fold_fresh": False}, {"fusion": 0}, {"devices": (0,) * 4}, {"devices": (0,) * 4, "layout": "reference", "fold_fresh": False}):
        val, prob = qc.expectation_hamiltonian(be, **opts)
        assert abs(val - float((p * H).sum())) < 1e-11 and abs(prob - Z / 2 ** 10) < 1e-12, opts
        val, prob = qc.expectation_hamiltonian(be, post_selected=False, **opts)
        assert abs(val - H.mean()) < 1e-11 and abs(prob - 1.0) < 1e-12, opts
    for Cq, y in ((C[0], (0, 1)), (C[5], (1, 1))):
        val, _ = qc.expectation_sufficient_statistic(be, Cq, y, devices=(0,))
        assert abs(val - float((p * qc.sufficient_statistic_diagonal(Cq, y)).sum())) < 1e-12
    name, C = workloads.baseline_config(2)                     # W = 28
    th = random_theta(60, seed=5)
    qc = QCMRF(C, th)
    p, Z = cf.gibbs_pmf(C, th)
    H = -(np.log(p) + np.log(Z))
    for opts in ({}, {"fold_fresh": False}):
        val, prob = qc.expectation_hamiltonian(be, **opts)
        assert abs(val - float((p * H).sum())) < 1e-10 and abs(prob - Z / 2 ** 12) < 1e-12
        st = be.last_engine.stats()["kinds"]["prob"]
        assert st["launches"] == 1 and st["bytes"] == 16.0 * 2 ** 28      # one read pass, nothing else


def test_qiskit_shaped_lowered_circuits_on_device(be):
    """run_experiment.py:52 input as a transpiler's clean-up passes leave it (tests/_qiskit_shapes.py:
    merged / re-synthesised one-qubit runs in DAG order, cancelled CX and cross-block H pairs, global
    phase): amplitudes against the closed form on the device at every fusion level, one multiplexer per
    clique at fusion 3, the generator path with fold_fresh, counts on the support"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _qiskit_shapes import lower_like_qiskit
    from qcmrf_amd import QCMRF, workloads
    for C in (workloads.REFERENCE_GRAPHS[3], workloads.chain(8), workloads.grid(2, 4)):        # W = 10, 16, 20
        n, m, W, dim = cf.model_shape(C)
        th = random_theta(dim, seed=W)
        t = lower_like_qiskit(QCMRF(C, th), extra_phase=0.6)
        want = np.exp(0.6j) * cf.amplitudes(C, th)
        for opts in ({"fusion": 0}, {"fusion": 2}, {"fusion": 3, "fold_fresh": False}, {"fusion": 3}):
            if W > 16 and opts["fusion"] == 0:
                continue
            amp, meta = run_state(be, t, **opts)
            assert np.abs(amp - want).max() < 5e-12, (W, opts)
            if opts["fusion"] == 3:
                kinds = be.last_engine.stats()["kinds"]
                assert "kq" not in kinds and meta["n_device_ops"] <= 2 * m + 4, (kinds, meta["n_device_ops"])
                if opts.get("fold_fresh", True) and W >= 16:
                    assert set(kinds) == {"init_prod"}, kinds
        counts = be.run(t, shots=3000, seed_simulator=5).result().get_counts()
        p = cf.probabilities(C, th)
        assert sum(counts.values()) == 3000 and all(p[int(k, 2)] > 1e-12 for k in counts)
