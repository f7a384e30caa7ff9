"""GPU parity: every HIP kernel against the numpy oracle on the same seeded state (bit-level
index conventions + fp64 arithmetic; tolerance 1e-13 absolute on amplitudes of a unit vector)."""
import numpy as np
import pytest

from oracle import sv_numpy as sv

pytestmark = pytest.mark.gpu
TOL = 1e-13


@pytest.fixture(scope="module")
def lib():
    from qcmrf_amd import _lib
    _lib.load()
    assert _lib.device_count() >= 1
    return _lib


def rand_state(n, seed):
    rs = np.random.RandomState(seed)
    v = rs.randn(2 ** n) + 1j * rs.randn(2 ** n)
    return v / np.linalg.norm(v)


def rand_u(k, seed):
    rs = np.random.RandomState(seed)
    a = rs.randn(2 ** k, 2 ** k) + 1j * rs.randn(2 ** k, 2 ** k)
    q, _ = np.linalg.qr(a)
    return q


@pytest.mark.parametrize("n", [3, 6, 9, 13, 16])
def test_dense_1q_every_target(lib, n):
    ref = rand_state(n, n)
    with lib.Engine(n) as e:
        e.set_amplitudes(0, ref)
        for t in range(n):
            m = rand_u(1, 100 + t)
            e.apply_1q(t, m)
            sv.apply_1q(ref, t, m)
        got = e.amplitudes()
    assert np.abs(got - ref).max() < TOL


@pytest.mark.parametrize("shuffle", [0, 1])
def test_lowt_shuffle_matches_pair_kernel(lib, shuffle):
    n = 14
    ref = rand_state(n, 5)
    with lib.Engine(n) as e:
        e.set_option("lowt_shuffle", shuffle)
        e.set_amplitudes(0, ref)
        for t in range(6):
            m = rand_u(1, t)
            e.apply_1q(t, m)
            sv.apply_1q(ref, t, m)
        assert np.abs(e.amplitudes() - ref).max() < TOL


@pytest.mark.parametrize("n", [4, 10, 15])
def test_controlled_1q_and_mcx_with_flags(lib, n):
    rs = np.random.RandomState(n)
    ref = rand_state(n, 7)
    with lib.Engine(n) as e:
        e.set_amplitudes(0, ref)
        for trial in range(24):
            nc = rs.randint(0, min(4, n - 1) + 1)
            qs = rs.permutation(n)[: nc + 1].tolist()
            vals = rs.randint(0, 2, size=nc).tolist()
            if trial % 2:
                e.apply_mcx(qs[:-1], qs[-1], vals)
                sv.apply_mcx(ref, qs[:-1], qs[-1], vals)
            else:
                m = rand_u(1, trial)
                e.apply_1q(qs[-1], m, qs[:-1], vals)
                sv.apply_1q(ref, qs[-1], m, qs[:-1], vals)
        assert np.abs(e.amplitudes() - ref).max() < TOL


@pytest.mark.parametrize("mask,nt,swz", [(1, 0, 1), (1, 1, 2), (0, 0, 1), (1, 0, 2)])
def test_controls_inside_a_cache_line_as_a_mask(lib, mask, nt, swz):
    """controlled X / 2x2 with controls on address bits 0..2: the masked full-line sweep (k_pair_m, default) and the
    enumerating form, plain and non-temporal, targets below, between and above the controls, +-control values; with the
    index swizzle forced (2) the sixth free address bit changes places with bit 11 -- whichever bit that is when controls
    or the target sit on bits 5..10, and not at all when bit 11 is taken"""
    n = 15
    rs = np.random.RandomState(5)
    ref = rand_state(n, 13)
    with lib.Engine(n) as e:
        e.set_option("lowctl_mask", mask)
        e.set_option("nontemporal", nt)
        e.set_option("swizzle", swz)
        e.set_amplitudes(0, ref)
        cases = [([1], 12), ([1, 5], 12), ([0, 2, 9], 4), ([2], 0), ([0, 1, 2], 14), ([1, 13], 2), ([2, 7, 11], 1), ([0], 1),
                 ([1, 5, 6], 7), ([2, 3, 4, 5], 8), ([0, 5], 11), ([1, 6, 8], 5)]
        for trial, (cs, t) in enumerate(cases * 2):
            vals = rs.randint(0, 2, size=len(cs)).tolist()
            if trial % 2:
                e.apply_mcx(cs, t, vals)
                sv.apply_mcx(ref, cs, t, vals)
            else:
                m = rand_u(1, 90 + trial)
                e.apply_1q(t, m, cs, vals)
                sv.apply_1q(ref, t, m, cs, vals)
        assert np.abs(e.amplitudes() - ref).max() < TOL


@pytest.mark.parametrize("n", [3, 9, 14])
def test_diag_and_mcphase(lib, n):
    rs = np.random.RandomState(n)
    ref = rand_state(n, 11)
    with lib.Engine(n) as e:
        e.set_amplitudes(0, ref)
        for trial in range(12):
            k = rs.randint(1, min(n, 6) + 1)
            qs = rs.permutation(n)[:k].tolist()
            tab = np.exp(1j * rs.uniform(-3, 3, size=2 ** k))
            e.apply_diag(qs, tab)
            sv.apply_diag(ref, qs, tab)
            vals = rs.randint(0, 2, size=k).tolist()
            ang = rs.uniform(-3, 3)
            e.apply_mcphase(qs, ang, vals)
            sv.apply_mcphase(ref, qs, ang, vals)
        assert np.abs(e.amplitudes() - ref).max() < TOL


def test_diag_wide_table_global_path(lib):
    n, k = 14, 13
    rs = np.random.RandomState(3)
    ref = rand_state(n, 1)
    qs = rs.permutation(n)[:k].tolist()
    tab = np.exp(1j * rs.uniform(-3, 3, size=2 ** k))
    with lib.Engine(n) as e:
        e.set_amplitudes(0, ref)
        e.apply_diag(qs, tab)
        sv.apply_diag(ref, qs, tab)
        assert np.abs(e.amplitudes() - ref).max() < TOL


@pytest.mark.parametrize("n", [4, 9, 15])
def test_mux(lib, n):
    rs = np.random.RandomState(n)
    ref = rand_state(n, 13)
    with lib.Engine(n) as e:
        e.set_amplitudes(0, ref)
        for trial in range(10):
            k = rs.randint(0, min(n - 1, 5) + 1)
            qs = rs.permutation(n)[: k + 1].tolist()
            mats = np.array([rand_u(1, 1000 * trial + j) for j in range(2 ** k)])
            e.apply_mux(qs[:-1], qs[-1], mats)
            sv.apply_mux(ref, qs[:-1], qs[-1], mats)
        assert np.abs(e.amplitudes() - ref).max() < TOL


@pytest.mark.parametrize("mfma", [0, 1])
@pytest.mark.parametrize("n,k", [(5, 1), (6, 2), (7, 3), (9, 3), (8, 4), (12, 4), (8, 5), (13, 5), (16, 5)])
def test_kq_dense(lib, n, k, mfma):
    """dense k-qubit unitary: VALU register kernel (mfma=0) and the f64 matrix-core kernel
    (mfma=1, k >= 3) against the numpy oracle, targets in random order at random positions"""
    rs = np.random.RandomState(n * 10 + k)
    ref = rand_state(n, 17)
    with lib.Engine(n) as e:
        e.set_option("kq_mfma", mfma)
        e.set_amplitudes(0, ref)
        for trial in range(4):
            qs = rs.permutation(n)[:k].tolist()
            u = rand_u(k, trial)
            e.apply_kq(qs, u)
            sv.apply_kq(ref, qs, u)
        assert np.abs(e.amplitudes() - ref).max() < 1e-12


@pytest.mark.parametrize("variant,tile3,nt,chunked", [(0, 0, 0, 0), (1, 0, 0, 0), (2, 0, 0, 0), (2, 1, 0, 0), (1, 2, 1, 0), (2, 0, 1, 1),
                                                    (2, 2, 1, 0), (3, 0, 0, 0), (3, 1, 1, 1), (4, 0, 0, 0), (4, 0, 1, 0), (5, 0, 0, 0), (5, 1, 1, 0), (6, 0, 0, 0), (6, 0, 1, 0), (7, 0, 0, 0), (8, 0, 1, 0)])
@pytest.mark.parametrize("n,k", [(9, 3), (14, 3), (12, 4), (13, 5), (16, 5)])
def test_kq_kernel_variants(lib, n, k, variant, tile3, nt, chunked):
    """every form of the dense k-qubit gate: four real matrix-core products per complex one (0), three (Gauss, 1), three
    with the next batch prefetched (2), the A fragments in LDS (3 with, 4 without prefetch), two batches per step loaded 32 groups wide with lane swaps (5), the batch staged through LDS and moved in memory order (6; 7, 8: two, three batches of loads in flight per wave); K = 3 on the vector units
    (k_kq_tile: 1 when no target sits inside a 128-byte line, 2 always) or embedded in a 16 x 16 matrix-core tile; plain and non-temporal / index-swizzled access; grid-stride and contiguous walks -- against numpy"""
    rs = np.random.RandomState(n * 31 + k)
    ref = rand_state(n, 23)
    with lib.Engine(n) as e:
        e.set_option("kq_variant", variant)
        e.set_option("kq3_tile", tile3)
        e.set_option("nontemporal", nt)
        e.set_option("swizzle", 2 if nt else 1)
        e.set_option("kq_chunked", chunked)
        e.set_amplitudes(0, ref)
        for trial in range(5):
            qs = rs.permutation(n)[:k].tolist() if trial else list(range(k))       # incl. the k lowest bits
            if trial == 4 and n > 11 and 11 not in qs:
                qs[-1] = 11                                # a target on address bit 11: it becomes index bit 0 (kq_order 1)
            u = rand_u(k, 40 + trial)
            e.apply_kq(qs, u)
            sv.apply_kq(ref, qs, u)
        assert np.abs(e.amplitudes() - ref).max() < 1e-12


@pytest.mark.parametrize("order", [0, 2, 3, 4])
@pytest.mark.parametrize("variant", [3, 5, 8])
@pytest.mark.parametrize("n,k", [(14, 3), (13, 4), (16, 5)])
def test_kq_target_order_is_the_kernels_choice(lib, n, k, variant, order):
    """which target is index bit 0, 1, ... of the matrix is an access-pattern choice of the matrix-core kernels (kq_order:
    as given / ascending / descending / nearest address bit 11 first; the default, a target on bit 11 first, runs in every other test): the matrix is re-indexed to match, the gate is the same"""
    rs = np.random.RandomState(n * 7 + k + order)
    ref = rand_state(n, 29)
    with lib.Engine(n) as e:
        e.set_option("kq_variant", variant)
        e.set_option("kq3_tile", 0)
        e.set_option("kq_order", order)
        e.set_amplitudes(0, ref)
        for trial in range(4):
            qs = rs.permutation(n)[:k].tolist()
            if trial == 3:
                qs[1] = 11 if 11 not in qs else qs[1]             # a target on bit 11 somewhere in the middle
            u = rand_u(k, 60 + trial)
            e.apply_kq(qs, u)
            sv.apply_kq(ref, qs, u)
        assert np.abs(e.amplitudes() - ref).max() < 1e-12


def test_copy_state_has_read_the_source_before_the_source_moves_on(lib):
    """qsv_copy_state runs on the destination's stream; the source's next kernels (a trajectory run projects it on the
    sibling outcome at once) wait for it -- without that wait the copy of a 1 GiB state picked up amplitudes the
    projection had already zeroed, and a seeded trajectory run walked a different tree every time"""
    n = 26
    with lib.Engine(n) as a, lib.Engine(n) as b:
        a.init_uniform((1 << n) - 1)
        a.apply_diag([0, n - 1], np.exp(1j * np.arange(4)))
        a.sync()
        windows = [0, (1 << (n - 1)) - 2048, (1 << (n - 1)), (1 << n) - 4096]
        want = [a.amplitudes(w, 4096) for w in windows]
        for rep in range(4):
            b.copy_from(a)
            a.apply_diag([n - 1], np.array([0.0, 1.0]))       # zeroes the first half of the source at once
            a.apply_mcx([], n - 1)
            got = [b.amplitudes(w, 4096) for w in windows]
            for g, w in zip(got, want):
                assert np.array_equal(g, w), rep
            a.copy_from(b)                                    # ... and back, the same way round
            b.apply_diag([n - 1], np.array([1.0, 0.0]))
            got = [a.amplitudes(w, 4096) for w in windows]
            for g, w in zip(got, want):
                assert np.array_equal(g, w), rep


@pytest.mark.parametrize("n,swz", [(12, 1), (16, 2)])
def test_uncontrolled_x_on_every_target(lib, n, swz):
    """X without controls: targets 0..4 go through the wave-shuffle sweep as the 2x2 they are, target 5 and up through
    the pair sweep (with the swizzle forced: bit 6 in bit 5's place for target 5) -- every target, against numpy, exactly"""
    ref = rand_state(n, 53)
    with lib.Engine(n) as e:
        e.set_option("swizzle", swz)
        e.set_amplitudes(0, ref)
        for t in list(range(n)) + [0, 5, 3]:
            e.apply_mcx([], t)
            sv.apply_mcx(ref, [], t, [])
            m = rand_u(1, 300 + t)
            e.apply_1q(t, m)
            sv.apply_1q(ref, t, m)
        assert np.abs(e.amplitudes() - ref).max() < 1e-13


def test_measurement_knob_needs_the_opt_in(lib, monkeypatch):
    """kq_debug times the LDS-staged dense-gate kernel with a part of its work left out (scripts/kq_variants.py): the
    state it leaves is wrong by design, so the library takes it only with QSV_MEASUREMENT_KNOBS in the environment"""
    monkeypatch.delenv("QSV_MEASUREMENT_KNOBS", raising=False)
    with lib.Engine(10) as e:
        with pytest.raises(ValueError, match="QSV_MEASUREMENT_KNOBS"):
            e.set_option("kq_debug", 1)
        e.set_option("kq_debug", 0)


def test_nontemporal_kernel_forms(lib):
    """every one-gate kernel in its non-temporal form (loads and stores that bypass the caches: what
    states of >= 2^26 amplitudes select by themselves) against the numpy oracle at a size where the
    plain form would be chosen: dense 1q on every target, MCX / controlled 2x2 with controls above
    and inside a 128-byte line, diag, mux, dense 3..5-qubit gates on the matrix cores, norm"""
    n = 14
    rs = np.random.RandomState(77)
    ref = rand_state(n, 41)
    with lib.Engine(n) as e:
        e.set_option("nontemporal", 1)
        e.set_option("multi_nt", 1)
        e.set_amplitudes(0, ref)
        for t in range(n):
            m = rand_u(1, 500 + t)
            e.apply_1q(t, m)
            sv.apply_1q(ref, t, m)
        for trial in range(12):
            nc = int(rs.randint(0, 4))
            qs = [int(x) for x in rs.permutation(n)[: nc + 1]]
            vals = [int(x) for x in rs.randint(0, 2, size=nc)]
            e.apply_mcx(qs[:-1], qs[-1], vals)
            sv.apply_mcx(ref, qs[:-1], qs[-1], vals)
            m = rand_u(1, 600 + trial)
            e.apply_1q(qs[-1], m, qs[:-1], vals)
            sv.apply_1q(ref, qs[-1], m, qs[:-1], vals)
            k = int(rs.randint(1, 5))
            qs = [int(x) for x in rs.permutation(n)[:k + 1]]
            tab = np.exp(1j * rs.uniform(-3, 3, size=2 ** (k + 1)))
            e.apply_diag(qs, tab)
            sv.apply_diag(ref, qs, tab)
            mats = np.array([rand_u(1, 700 + 16 * trial + j) for j in range(2 ** k)])
            e.apply_mux(qs[:-1], qs[-1], mats)
            sv.apply_mux(ref, qs[:-1], qs[-1], mats)
        for k in (3, 4, 5):
            qs = [int(x) for x in rs.permutation(n)[:k]]
            u = rand_u(k, 800 + k)
            e.apply_kq(qs, u)
            sv.apply_kq(ref, qs, u)
        assert np.abs(e.amplitudes() - ref).max() < 1e-12
        assert abs(e.norm() - 1.0) < 1e-12
        tab = rs.randn(8)
        idx = np.arange(2 ** n)
        j = ((idx >> 2) & 1) | (((idx >> 9) & 1) << 1) | (((idx >> 13) & 1) << 2)
        got = e.expect_diag([2, 9, 13], tab)
        assert abs(got[0] - (np.abs(ref) ** 2 * tab[j]).sum()) < 1e-12


@pytest.mark.parametrize("swizzle", [2, 0])          # 2: also below 2^26 amplitudes, where 1 leaves it off
def test_index_swizzle_of_the_one_gate_kernels(lib, swizzle):
    """The one-gate sweep kernels index their amplitudes with bits 5 and 11 exchanged (a wave access = two
    512-byte runs 32 KiB apart); a sweep that enumerates the indices with its controls and target left out
    exchanges bit 11 with whichever address bit lane bit 5 lands on (the sixth free one), and leaves the swizzle out when
    bit 11 is taken.  Gates whose target / controls / selects sit on bits 5, 11 or crowd the bits below, both ways, against numpy."""
    n = 16
    rs = np.random.RandomState(5 + swizzle)
    ref = rand_state(n, 43)
    with lib.Engine(n) as e:
        e.set_option("swizzle", swizzle)
        e.set_amplitudes(0, ref)
        for t in (0, 4, 5, 6, 10, 11, 12, 15):
            m = rand_u(1, 900 + t)
            e.apply_1q(t, m)
            sv.apply_1q(ref, t, m)
        for ctrls, t in (([5], 11), ([11], 5), ([5, 11], 3), ([3], 12), ([12, 4], 5), ([6], 11), ([10, 2], 14), ([], 5), ([], 11),
                         ([3, 5, 6], 7), ([3, 4, 5, 6, 7], 9), ([0, 1, 2, 3, 4], 5), ([5, 6, 7, 8, 9], 10)):
            vals = [int(x) for x in rs.randint(0, 2, size=len(ctrls))]
            e.apply_mcx(ctrls, t, vals)
            sv.apply_mcx(ref, ctrls, t, vals)
            m = rand_u(1, 950 + t + len(ctrls))
            e.apply_1q(t, m, ctrls, vals)
            sv.apply_1q(ref, t, m, ctrls, vals)
            qs = ctrls + [t]
            ang = float(rs.uniform(-3, 3))
            e.apply_mcphase(qs, ang, [1] * len(qs))
            sv.apply_mcphase(ref, qs, ang, [1] * len(qs))
            tab = np.exp(1j * rs.uniform(-3, 3, size=2 ** len(qs)))
            e.apply_diag(qs, tab)
            sv.apply_diag(ref, qs, tab)
            mats = np.array([rand_u(1, 1000 + 8 * t + j) for j in range(2 ** len(ctrls))])
            e.apply_mux(ctrls, t, mats)
            sv.apply_mux(ref, ctrls, t, mats)
        assert np.abs(e.amplitudes() - ref).max() < 1e-12
        assert abs(e.norm() - 1.0) < 1e-12


def test_init_uniform_and_zero(lib):
    n = 11
    with lib.Engine(n) as e:
        e.init_zero()
        a = e.amplitudes()
        assert a[0] == 1 and np.count_nonzero(a) == 1
        mask = 0b10110100101
        e.init_uniform(mask)
        a = e.amplitudes()
        idx = np.arange(2 ** n)
        want = np.where((idx & ~mask) == 0, 2.0 ** (-0.5 * bin(mask).count("1")), 0.0)
        assert np.array_equal(a, want.astype(np.complex128))


def test_swap_layout_local(lib):
    n = 10
    ref = rand_state(n, 19)
    with lib.Engine(n) as e:
        e.set_amplitudes(0, ref)
        e.swap_layout([1, 7], [8, 3])
        got = e.amplitudes()
    idx = np.arange(2 ** n)

    def sw(i, a, b):
        ba, bb = (i >> a) & 1, (i >> b) & 1
        return (i & ~((1 << a) | (1 << b))) | (bb << a) | (ba << b)
    want = ref[sw(sw(idx, 1, 8), 7, 3)]
    assert np.array_equal(got, want)


@pytest.mark.parametrize("P", [2, 4, 8])
def test_swap_layout_batched_shard_bits_is_one_all_to_all(lib, P):
    """several (shard bit, local bit) pairs in ONE qsv_swap_layout call = one batched exchange (an
    all-to-all inside each group of shards): same permutation as the pairs one after the other
    (numpy engine), counted as ONE exchange; subsets of the shard bits; mixed with a local swap"""
    from oracle.sharded_numpy import NumpyEngine
    W = 15
    g = P.bit_length() - 1
    L = W - g
    rs = np.random.RandomState(40 + P)
    cases = []
    shard_bits = list(range(L, W))
    for k in range(1, g + 1):
        for trial in range(3):
            G = [int(x) for x in rs.choice(shard_bits, size=k, replace=False)]
            J = [int(x) for x in rs.choice(L, size=k, replace=False)]
            cases.append((G, J))
    with lib.Engine(W, devices=(0,) * P) as e:
        for G, J in cases:
            ref = NumpyEngine(W, P)
            st = rand_state(W, len(G) * 7 + J[0])
            for s in range(P):
                ref.sh[s][:] = st[s << L:(s + 1) << L]
            e.set_amplitudes(0, st)
            e.reset_stats()
            # caller's pair order is arbitrary: (a, b) with the shard bit on either side
            a = [G[i] if i % 2 == 0 else J[i] for i in range(len(G))]
            b = [J[i] if i % 2 == 0 else G[i] for i in range(len(G))]
            e.swap_layout(a, b)
            ref.swap_layout(a, b)
            assert np.array_equal(e.amplitudes(), ref.amplitudes()), (G, J)
            assert e.stats()["exchanges"] == 1, (G, J)
        # a local pair in the same call, and a call whose pairs share a qubit (executed one by one)
        st = rand_state(W, 99)
        ref = NumpyEngine(W, P)
        for s in range(P):
            ref.sh[s][:] = st[s << L:(s + 1) << L]
        e.set_amplitudes(0, st)
        a, b = [W - 1, 2, 5], [4, 9, W - 1] if g >= 1 else [4, 9, 6]
        e.swap_layout(a, b)
        ref.swap_layout(a, b)
        assert np.array_equal(e.amplitudes(), ref.amplitudes())


def test_rccl_exchange_pipeline_selftest(lib):
    """the RCCL side of a batched exchange -- staging, two-stream double buffering, grouped send/recv
    to two 'peers' -- on a 1-rank communicator (RCCL refuses two ranks on one device): many small
    chunks (pipeline depth), one chunk, and a chunk size that does not divide the block"""
    lib.rccl_exchange_selftest(0, 22, 12)       # 2^20-amplitude blocks in 256 chunks
    lib.rccl_exchange_selftest(0, 18, 30)       # everything in one chunk
    lib.rccl_exchange_selftest(0, 20, 17)


def test_direct_shard_bit_swap_after_exec_drops_the_cached_tile_sums(lib):
    """qsv_exec leaves per-tile |amp|^2 sums for sampling; a qsv_swap_layout across a shard bit
    issued directly through the C ABI afterwards (Engine.swap_layout) moves amplitudes between
    the shards without a launch on every one of them -- norm and sample must then see the NEW
    state, not the sums and tile order of the old one (2 virtual shards, vs the numpy engine)."""
    from qcmrf_amd import ir, program
    from oracle.sharded_numpy import NumpyEngine
    W, P = 16, 2
    rs = np.random.RandomState(3)
    # an unevenly weighted state: shard 0 and shard 1 carry different mass before and after the swap
    ops = [ir.op_init((1 << W) - 1 - (1 << 9))]
    for t, a in ((9, 0.4), (6, 1.1), (7, 0.3)):
        ops.append(ir.op_mux([W - 1, 2], t, np.array([[[np.cos(a * k), -1j * np.sin(a * k)], [-1j * np.sin(a * k), np.cos(a * k)]]
                                                       for k in (0.5, 1.0, 1.7, 2.9)])))
    rec, data = program.encode(ops)
    ref = NumpyEngine(W, P)
    ref.exec(rec, data)
    with lib.Engine(W, devices=(0,) * P) as e:
        e.exec(rec, data)
        m0 = e.norm()
        assert abs(m0 - 1.0) < 1e-12
        e.sample(2000, 5)                                    # fetches and caches the tile sums of the pre-swap state
        e.swap_layout([W - 1], [9])                         # shard bit <-> local bit, straight through the C ABI
        ref.swap_layout([W - 1], [9])
        want = ref.amplitudes()
        assert np.abs(e.amplitudes() - want).max() < 1e-13
        assert abs(e.norm() - 1.0) < 1e-12
        p = np.abs(want) ** 2
        shots = 100000
        idx = e.sample(shots, 5)
        obs = np.bincount(idx.astype(np.int64), minlength=p.size)
        assert obs[p == 0].sum() == 0                        # a stale tile order lands on unpopulated indices
        pb, ob = p.reshape(-1, 64).sum(1), obs.reshape(-1, 64).sum(1)
        keep = pb * shots > 5
        chi = ((ob[keep] - pb[keep] * shots) ** 2 / (pb[keep] * shots)).sum() / (keep.sum() - 1)
        assert 0.8 < chi < 1.2, chi
        # per-shard mass after the swap (the numbers a multi-rank sampling merge would use)
        half = p.size // 2
        got = e.probabilities([W - 1])
        assert np.abs(got - np.array([p[:half].sum(), p[half:].sum()])).max() < 1e-12


def test_probabilities_norm_and_conditional(lib):
    n = 12
    ref = rand_state(n, 23)
    p = np.abs(ref) ** 2
    idx = np.arange(2 ** n)
    with lib.Engine(n) as e:
        e.set_amplitudes(0, ref)
        assert abs(e.norm() - 1.0) < 1e-13
        qs = [9, 0, 4]
        got = e.probabilities(qs)
        j = ((idx >> 9) & 1) | (((idx >> 0) & 1) << 1) | (((idx >> 4) & 1) << 2)
        want = np.bincount(j, weights=p, minlength=8)
        assert np.abs(got - want).max() < 1e-13
        fm, fv = (1 << 11) | (1 << 2), (1 << 2)
        got = e.probabilities(qs, fm, fv)
        sel = (idx & fm) == fv
        want = np.bincount(j[sel], weights=p[sel], minlength=8)
        assert np.abs(got - want).max() < 1e-13
        full = e.probabilities(list(range(n)))
        assert np.abs(full - p).max() < 1e-15


@pytest.mark.parametrize("P", [1, 4])
def test_expect_diag(lib, P):
    """qsv_expect_diag vs numpy: LDS tables (k <= 12) and wide ones through L2, qubits on shard bits,
    conditioning mask, deterministic"""
    n = 16
    ref = rand_state(n, 31)
    p = np.abs(ref) ** 2
    idx = np.arange(2 ** n)
    rs = np.random.RandomState(7)
    with lib.Engine(n, devices=(0,) * P) as e:
        e.set_amplitudes(0, ref)
        for k in (1, 3, 9, 12, 13, 16):
            qs = [int(x) for x in rs.permutation(n)[:k]]
            tab = rs.randn(2 ** k)
            j = np.zeros_like(idx)
            for b, q in enumerate(qs):
                j |= ((idx >> q) & 1) << b
            got = e.expect_diag(qs, tab)
            assert abs(got[0] - (p * tab[j]).sum()) < 1e-13 and abs(got[1] - 1.0) < 1e-13
            assert e.expect_diag(qs, tab) == got
            fm, fv = (1 << 15) | (1 << 4) | (1 << 9), (1 << 15) | (1 << 9)
            sel = (idx & fm) == fv
            got = e.expect_diag(qs, tab, fm, fv)
            assert abs(got[0] - (p[sel] * tab[j[sel]]).sum()) < 1e-13 and abs(got[1] - p[sel].sum()) < 1e-13
        with pytest.raises(ValueError):
            e.expect_diag([1, 1], np.zeros(4))

@pytest.mark.parametrize("P", [1, 2])
@pytest.mark.parametrize("variant", [6, 2])
def test_expect_diag_tiled_and_grid_stride_walks(lib, P, variant):
    """from 2^20 amplitudes per shard on, qsv_expect_diag runs one workgroup per 64 KiB tile and folds the tiles' partial
    sums on the device in a fixed order (blocksum_variant & 4, the default); without the bit, a grid-stride walk: both
    against numpy at 22 qubits, with and without a conditioning mask, narrow and wide tables, reproducibly"""
    n = 22
    ref = rand_state(n, 37)
    p = np.abs(ref) ** 2
    idx = np.arange(2 ** n)
    rs = np.random.RandomState(11)
    with lib.Engine(n, devices=(0,) * P) as e:
        e.set_option("blocksum_variant", variant)
        e.set_amplitudes(0, ref)
        for k in (1, 7, 12, 14):
            qs = [int(x) for x in rs.permutation(n)[:k]]
            if k == 7:
                qs[0] = n - 1                              # a shard bit (P = 2) among the table's qubits
            tab = rs.randn(2 ** k)
            j = np.zeros_like(idx)
            for b, q in enumerate(qs):
                j |= ((idx >> q) & 1) << b
            got = e.expect_diag(qs, tab)
            assert abs(got[0] - (p * tab[j]).sum()) < 1e-12 and abs(got[1] - 1.0) < 1e-12
            assert e.expect_diag(qs, tab) == got
            fm, fv = (1 << 21) | (1 << 4) | (1 << 13), (1 << 21) | (1 << 13)
            sel = (idx & fm) == fv
            got = e.expect_diag(qs, tab, fm, fv)
            assert abs(got[0] - (p[sel] * tab[j[sel]]).sum()) < 1e-12 and abs(got[1] - p[sel].sum()) < 1e-12


@pytest.mark.parametrize("n,P", [(10, 4), (11, 4), (9, 8), (7, 8), (11, 2)])
def test_expect_diag_shards_smaller_than_one_step(lib, n, P):
    """shards of fewer than 1024 amplitudes (L < 10): qubits and fix bits on the shard bits in [L, 10) come from the
    shard number, not from the thread's low offset (they were read as 0)"""
    ref = rand_state(n, 41)
    p = np.abs(ref) ** 2
    idx = np.arange(2 ** n)
    rs = np.random.RandomState(3)
    L = n - (P.bit_length() - 1)
    with lib.Engine(n, devices=(0,) * P) as e:
        e.set_amplitudes(0, ref)
        for trial in range(12):
            k = int(rs.randint(1, n + 1))
            qs = [int(x) for x in rs.permutation(n)[:k]]
            if trial < 4:                                  # every shard bit among the table's qubits
                qs = list(range(L, n)) + [q for q in qs if q < L][:max(0, k - (n - L))]
            tab = rs.randn(2 ** len(qs))
            j = np.zeros_like(idx)
            for b, q in enumerate(qs):
                j |= ((idx >> q) & 1) << b
            got = e.expect_diag(qs, tab)
            assert abs(got[0] - (p * tab[j]).sum()) < 1e-13 and abs(got[1] - 1.0) < 1e-13, (trial, qs)
            fm = int(rs.randint(0, 2 ** n)) | (1 << (n - 1)) | (1 << L)          # fix bits on shard bits and local bits
            fv = int(rs.randint(0, 2 ** n)) & fm
            sel = (idx & fm) == fv
            got = e.expect_diag(qs, tab, fm, fv)
            assert abs(got[0] - (p[sel] * tab[j[sel]]).sum()) < 1e-13 and abs(got[1] - p[sel].sum()) < 1e-13, (trial, qs, fm, fv)



def test_sampling_matches_distribution(lib):
    n = 10
    ref = rand_state(n, 29)
    ref[(np.arange(2 ** n) & 0b1000) != 0] = 0          # a structurally empty half
    ref /= np.linalg.norm(ref)
    p = np.abs(ref) ** 2
    shots = 200000
    with lib.Engine(n) as e:
        e.set_amplitudes(0, ref)
        s1 = e.sample(shots, 1234)
        s2 = e.sample(shots, 1234)
        s3 = e.sample(shots, 99)
    assert np.array_equal(s1, s2) and not np.array_equal(s1, s3)       # seeded, reproducible
    cnt = np.bincount(s1.astype(np.int64), minlength=2 ** n)
    assert cnt[p == 0].sum() == 0                                        # never outside the support
    sel = p * shots > 5
    chi2 = ((cnt[sel] - p[sel] * shots) ** 2 / (p[sel] * shots)).sum() / (sel.sum() - 1)
    assert 0.8 < chi2 < 1.25
    # measured-qubit packing
    with lib.Engine(n) as e:
        e.set_amplitudes(0, ref)
        raw = e.sample(1000, 5)
        packed = e.sample(1000, 5, [7, 1, 3])
    want = ((raw >> 7) & 1) | (((raw >> 1) & 1) << 1) | (((raw >> 3) & 1) << 2)
    assert np.array_equal(packed, want)


def test_errors_are_loud(lib):
    with lib.Engine(5) as e:
        with pytest.raises(ValueError):
            e.apply_1q(5, np.eye(2))
        with pytest.raises(ValueError):
            e.apply_mcx([1, 1], 2)
        with pytest.raises(ValueError):
            e.apply_mcx([2], 2)
        with pytest.raises(ValueError):
            e.init_uniform(1 << 5)
    with pytest.raises(ValueError):
        lib.Engine(5, devices=(0, 0, 0))            # not a power of two
    with lib.Engine(6, devices=(0, 0)) as e:        # target on the shard bit must be refused
        with pytest.raises(RuntimeError):
            e.apply_1q(5, np.eye(2))


@pytest.mark.parametrize("seed", [7, 21])
def test_random_programs_differential(seed):
    """scripts/stress_random_programs.py: 200 random programs per seed (table ops with RX-like or
    general matrices, controlled gates, X, phases, diagonals, leading init or not, random pass
    hints, 1/2/4 virtual shards, random engine options: tile width, borrowed lanes, lane map, zero
    tracking, generator on/off) through qsv_exec against the numpy engine, amplitudes 1e-12."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("stress_random_programs", os.path.join(ROOT, "scripts", "stress_random_programs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(200, seed, verbose=False) == 0


def test_random_programs_differential_wide_states():
    """the same at 21-23 qubits, where the workgroups of a pass run in several waves: an in-place
    permutation that crossed workgroups (as the first X frame did on block bits) is invisible below"""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("stress_random_programs", os.path.join(ROOT, "scripts", "stress_random_programs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(40, 3, verbose=False, widths=(21, 22, 23)) == 0


@pytest.mark.parametrize("P", [1, 2])
def test_x_frame_on_block_bits_of_a_state_wider_than_one_wave_of_workgroups(lib, P):
    """An uncontrolled X rides in a pass as an XOR on the STORE addresses -- race-free only for bits inside
    one wavefront's own tile (register + lane bits).  On a bit that selects another workgroup the X has to
    stay pending across passes (or run as a swap): at 2^24 amplitudes the workgroups of a pass no longer
    run all at once, so a store through such a bit would land in a tile nobody has read yet.  Program:
    X gates on lane, wave, block and top bits interleaved with gates that force read+write passes
    (general and table-op ones), one X left pending at the end; amplitudes against the numpy engine."""
    from qcmrf_amd import ir, program
    from oracle.sharded_numpy import NumpyEngine
    W = 24
    L = W - (P.bit_length() - 1)
    rs = np.random.RandomState(5)
    ops = [ir.op_init((1 << W) - 1)]
    ops.append(ir.op_diag([0, 7, 13, L - 1], np.exp(1j * rs.randn(16))))           # make every amplitude distinct
    ops.append(ir.op_diag([2, 6, 20, 9], np.exp(1j * rs.randn(16))))
    for xq, (t, c) in zip([L - 1, 20, 7, 6, 2, 13, L - 1, 17], [(10, 3), (11, L - 1), (9, 20), (12, 7), (10, 13), (8, 2), (14, 6), (10, 17)]):
        ops.append(ir.op_x(xq))
        ops.append(ir.op_u(t, rand_u(1, 100 + t), [c], [1]))                      # general pass, control on the flipped bit
        ops.append(ir.op_mux([xq], (t + 5) % 16 + 6 if (t + 5) % 16 + 6 != xq else 15, np.array([rand_u(1, 7), rand_u(1, 8)])))
    ops.append(ir.op_x(19))                                                         # left pending at the end of the program
    rec, data = program.encode(ops)
    ref = NumpyEngine(W, P)
    ref.exec(rec, data)
    want = ref.amplitudes()
    for opts in ({}, {"general_r": 3, "multi_r": 3}, {"xframe": 0}):
        with lib.Engine(W, devices=(0,) * P) as eng:
            for k, v in opts.items():
                eng.set_option(k, v)
            eng.exec(rec, data)
            got = eng.amplitudes()
            assert np.abs(got - want).max() < 1e-12, opts
            assert abs(eng.norm() - 1.0) < 1e-10
