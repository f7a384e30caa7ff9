"""Sharded numpy stand-in for the device engine.  TEST INFRASTRUCTURE ONLY.

Same method names and physical-qubit semantics as ``qcmrf_amd._lib.Engine`` (which binds the
HIP library), so the host-side planner / program logic can be exercised on CPU and so the
N>1 path can be run across real processes over gloo (tests/test_multiproc_gloo.py).
The shipped package never imports this.
"""
from __future__ import annotations

import numpy as np

from . import sv_numpy as sv

_X = sv.MATS["x"]


class NumpyEngine:
    def __init__(self, n_qubits, n_shards=1, owned=None, exchange=None):
        """owned: shard numbers held by this process (default: all).
        exchange(my_shard_no, peer_shard_no, send_array) -> received array, for shards not owned."""
        self.n_qubits = n_qubits
        self.world = n_shards
        self.g = n_shards.bit_length() - 1
        self.L = n_qubits - self.g
        self.local_qubits = self.L
        self.owned = list(range(n_shards)) if owned is None else list(owned)
        self.sh = {s: np.zeros(2 ** self.L, dtype=np.complex128) for s in self.owned}
        self._exchange = exchange
        self.n_exchanges = 0

    # ---- helpers
    def _bit(self, s, q):
        return (s >> (q - self.L)) & 1

    def _local_ctrl(self, s, ctrls, vals):
        lq, lv = [], []
        for c, v in zip(ctrls, vals):
            if c >= self.L:
                if self._bit(s, c) != v:
                    return None
            else:
                lq.append(c)
                lv.append(v)
        return lq, lv

    def _slice(self, s, qubits, table, ent_shape):
        table = np.asarray(table, dtype=np.complex128).reshape((-1,) + ent_shape)
        lq, lpos, gfix = [], [], 0
        for b, q in enumerate(qubits):
            if q >= self.L:
                gfix |= self._bit(s, q) << b
            else:
                lq.append(q)
                lpos.append(b)
        sub = []
        for jl in range(2 ** len(lq)):
            j = gfix
            for b, pos in enumerate(lpos):
                j |= ((jl >> b) & 1) << pos
            sub.append(table[j])
        return lq, np.array(sub)

    # ---- state preparation
    def init_zero(self):
        self.init_uniform(0)

    def init_uniform(self, mask):
        val = 2.0 ** (-0.5 * bin(mask).count("1"))
        idx = np.arange(2 ** self.L, dtype=np.int64)
        for s in self.owned:
            g = (s << self.L) | idx
            self.sh[s][:] = np.where((g & ~mask) == 0, val, 0.0)

    # ---- gates
    def apply_1q(self, t, m, ctrls=(), ctrl_vals=None):
        assert t < self.L, "dense target on a shard bit"
        vals = [1] * len(ctrls) if ctrl_vals is None else list(ctrl_vals)
        for s in self.owned:
            lc = self._local_ctrl(s, ctrls, vals)
            if lc is not None:
                sv.apply_1q(self.sh[s], t, np.asarray(m).reshape(2, 2), lc[0], lc[1])

    def apply_mcx(self, ctrls, t, ctrl_vals=None):
        self.apply_1q(t, _X, ctrls, ctrl_vals)

    def apply_mcphase(self, qubits, angle, vals=None):
        vals = [1] * len(qubits) if vals is None else list(vals)
        for s in self.owned:
            lc = self._local_ctrl(s, qubits, vals)
            if lc is not None:
                if lc[0]:
                    sv.apply_mcphase(self.sh[s], lc[0], angle, lc[1])
                else:
                    self.sh[s] *= np.exp(1j * angle)

    def apply_diag(self, qubits, table):
        for s in self.owned:
            lq, sub = self._slice(s, qubits, table, ())
            if lq:
                sv.apply_diag(self.sh[s], lq, sub)
            else:
                self.sh[s] *= sub[0]

    def apply_mux(self, ctrls, t, mats):
        assert t < self.L
        for s in self.owned:
            lq, sub = self._slice(s, ctrls, mats, (2, 2))
            sv.apply_mux(self.sh[s], lq, t, sub)

    def apply_kq(self, qubits, u):
        assert all(q < self.L for q in qubits)
        for s in self.owned:
            sv.apply_kq(self.sh[s], list(qubits), u)

    def swap_layout(self, a, b):
        # like libqsv: all shard-bit pairs of ONE call (on distinct qubits) are one exchange step;
        # pair by pair here, which is the same permutation
        flat = [q for x, y in zip(a, b) if x != y for q in (x, y)]
        batched = len(set(flat)) == len(flat)
        n_before = self.n_exchanges
        self._swap_pairs(a, b)
        if batched and self.n_exchanges > n_before:
            self.n_exchanges = n_before + 1

    def _swap_pairs(self, a, b):
        for x, y in zip(a, b):
            lo, hi = min(x, y), max(x, y)
            if lo == hi:
                continue
            if hi < self.L:
                for s in self.owned:
                    st = self.sh[s]
                    idx = np.arange(st.size)
                    sel = idx[(((idx >> lo) & 1) == 1) & (((idx >> hi) & 1) == 0)]
                    other = sel ^ (1 << lo) ^ (1 << hi)
                    st[sel], st[other] = st[other].copy(), st[sel].copy()
            else:
                assert lo < self.L, "swap of two shard bits"
                self._exchange_bit(hi, lo)

    def _exchange_bit(self, G, j):
        self.n_exchanges += 1
        gb = G - self.L
        idx = np.arange(2 ** self.L)
        done = set()
        for s in self.owned:
            if s in done:
                continue
            u = (s >> gb) & 1
            peer = s ^ (1 << gb)
            region = idx[((idx >> j) & 1) == (1 - u)]        # my entries that travel
            if peer in self.sh:
                pregion = idx[((idx >> j) & 1) == u]
                a, b = self.sh[s][region].copy(), self.sh[peer][pregion].copy()
                self.sh[s][region], self.sh[peer][pregion] = b, a
                done.update((s, peer))
            else:
                self.sh[s][region] = self._exchange(s, peer, self.sh[s][region].copy())
                done.add(s)

    # ---- measurement
    def norm(self):
        return float(sum((np.abs(v) ** 2).sum() for v in self.sh.values()))

    def amplitudes(self, start=0, count=None):
        full = np.concatenate([self.sh[s] for s in sorted(self.sh)])
        base = min(self.sh) << self.L
        count = full.size - (start - base) if count is None else count
        return full[start - base: start - base + count].copy()

    def sample(self, shots, seed, meas_qubits=None):
        owned = sorted(self.sh)
        p = np.concatenate([np.abs(self.sh[s]) ** 2 for s in owned])
        gidx = np.concatenate([(s << self.L) | np.arange(2 ** self.L) for s in owned]).astype(np.uint64)
        rng = np.random.RandomState(seed % (2 ** 32))
        pick = gidx[rng.choice(p.size, size=shots, p=p / p.sum())]
        if meas_qubits is None:
            return pick
        out = np.zeros(shots, dtype=np.uint64)
        for b, q in enumerate(meas_qubits):
            if q >= 0:                                     # -1: an output bit nothing is measured into
                out |= ((pick >> np.uint64(q)) & np.uint64(1)) << np.uint64(b)
        return out

    # ---- the rest of the qcmrf_amd._lib.Engine surface the backend touches
    def exec(self, rec, data):
        """decode qsv_op records (include/qsv.h) -- the same bytes libqsv's qsv_exec consumes"""
        K = ("init_zero", "init_uniform", "1q", "mcx", "diag", "mcphase", "mux", "kq", "swap")
        for r in rec:
            kind, n = K[int(r["kind"])], int(r["n"])
            q, v = [int(x) for x in r["qubits"][:n]], [int(x) for x in r["vals"][:n]]
            off = int(r["data_off"])

            def cdata(count):
                return np.asarray(data[off: off + 2 * count]).view(np.complex128)
            if kind == "init_zero":
                self.init_zero()
            elif kind == "init_uniform":
                self.init_uniform(int(r["mask"]))
            elif kind == "1q":
                self.apply_1q(int(r["target"]), cdata(4).reshape(2, 2), q, v)
            elif kind == "mcx":
                self.apply_mcx(q, int(r["target"]), v)
            elif kind == "diag":
                self.apply_diag(q, cdata(2 ** n))
            elif kind == "mcphase":
                self.apply_mcphase(q, float(r["angle"]), v)
            elif kind == "mux":
                self.apply_mux(q, int(r["target"]), cdata(4 * 2 ** n).reshape(-1, 2, 2))
            elif kind == "kq":
                self.apply_kq(q, cdata(4 ** n).reshape(2 ** n, 2 ** n))
            elif kind == "swap":
                self.swap_layout(q, v)

    def probabilities(self, qubits, fix_mask=0, fix_val=0):
        out = np.zeros(2 ** len(qubits))
        for s in self.owned:
            g = (s << self.L) | np.arange(2 ** self.L)
            sel = (g & fix_mask) == fix_val
            j = np.zeros_like(g)
            for b, q in enumerate(qubits):
                j |= ((g >> q) & 1) << b
            out += np.bincount(j[sel], weights=np.abs(self.sh[s][sel]) ** 2, minlength=out.size)
        return out

    def expect_diag(self, qubits, table, fix_mask=0, fix_val=0):
        table = np.asarray(table, dtype=np.float64)
        s0 = s1 = 0.0
        for s in self.owned:
            g = (s << self.L) | np.arange(2 ** self.L)
            sel = (g & fix_mask) == fix_val
            j = np.zeros_like(g)
            for b, q in enumerate(qubits):
                j |= ((g >> q) & 1) << b
            p = np.abs(self.sh[s][sel]) ** 2
            s0 += float((p * table[j[sel]]).sum())
            s1 += float(p.sum())
        return s0, s1

    def copy_from(self, other):
        for s in self.owned:
            self.sh[s][:] = other.sh[s]

    def sync(self): pass
    def close(self): pass
    def reset_stats(self): pass
    def set_profiling(self, on): pass
    def set_option(self, name, value): pass
    def comm_init(self, uid): pass
    def comm_bootstrap(self, comm, device=0, transport="auto"): pass
    def stats(self): return {"kinds": {}, "exchanges": self.n_exchanges, "exchange_bytes": 0.0, "fused_gates": 0}
