"""Exact circuit-level fusion passes (operator identities only -- no knowledge of the state
beyond the |0..0> start, and never the closed-form output distribution).

What Aer does with its "fusion" transpile stage before sweeping the vector, re-thought for a
bandwidth-bound GPU: every fused op is ONE HBM sweep, so the goal is fewest sweeps.

pass 1  fold_init       leading H gates on untouched qubits of |0..0>  ->  one ``init`` write
                        (QCMRF.py:204-205).
pass 2  fuse_monomial   runs of permutation/phase gates (x, cx, ccx, mcx, p, cp, rz, z, s, t ...)
                        are tracked as a (permutation, phase) pair over <= ``kmax`` qubits; whenever
                        the permutation returns to the identity the run IS a diagonal and is emitted
                        as one ``diag``.  ``AND . cp . AND`` repeated over all clique states y
                        (QCMRF.py:221-228) collapses to one (|C|+2)-qubit diagonal this way.
pass 3  fuse_mux        runs of {dense 2x2 on t, X on t, diagonals touching t} with everything
                        else acting as controls  ->  one uniformly controlled 2x2 (``mux``).
                        H . cU . X . cU^dg . X . H (QCMRF.py:231-236) becomes one sweep.
"""
from __future__ import annotations

import numpy as np

from . import ir
from .ir import Op

# --------------------------------------------------------------------------------------------
# pass 1
# --------------------------------------------------------------------------------------------


_X2 = np.array([[0, 1], [1, 0]], dtype=np.complex128)
SQH = 1.0 / np.sqrt(2.0)
_SQRT2 = np.sqrt(2.0)


def _1q_qubit(op):
    """the qubit if op is an uncontrolled one-qubit gate, else None (no matrix built)"""
    k = op.kind
    if k == "u" or k == "x":
        return None if op.ctrls else op.target
    if k == "diag" or k == "mcphase":
        return op.qubits[0] if len(op.qubits) == 1 else None
    return None


def _as_1q(op):
    """(qubit, 2x2) if op is an uncontrolled one-qubit gate, else None"""
    if op.kind == "u" and not op.ctrls:
        return op.target, op.mat
    if op.kind == "x" and not op.ctrls:
        return op.target, _X2
    if op.kind == "diag" and len(op.qubits) == 1:
        return op.qubits[0], np.diag(op.table)
    if op.kind == "mcphase" and len(op.qubits) == 1:
        d = np.ones(2, dtype=np.complex128)
        d[op.vals[0]] = np.exp(1j * op.angle)
        return op.qubits[0], np.diag(d)
    return None


def split_leading(ops):
    """lead[q] = product of the uncontrolled one-qubit gates that are the first things to touch q
    (``rz sx rz`` of a lowered H, or a literal H); rest = everything else, order kept."""
    lead, closed, rest = {}, set(), []
    for op in ops:
        q = _1q_qubit(op)
        if q is not None and q not in closed:
            m = _as_1q(op)[1]
            lead[q] = m @ lead[q] if q in lead else np.array(m, dtype=np.complex128)
            continue
        closed.update(op.support())
        rest.append(op)
    return lead, rest


def _is_hlike(m):
    return abs(abs(m[0, 0]) - SQH) < 1e-15 and abs(abs(m[1, 0]) - SQH) < 1e-15


def fold_init(ops, hold=None, split=None):
    """[init(mask)] + remaining ops.

    A leading one-qubit gate that maps |0> to an equal-weight superposition (H, or its lowered
    form) is folded into the initial state: the qubit goes into the uniform ``init`` mask, its
    relative phase becomes a one-qubit diagonal, its common phase a global phase.  Qubits in
    ``hold`` are exempt.  Default ``hold``: every qubit that is a dense target later on -- the MRF
    variable qubits fold (QCMRF.py:204-205), an ancilla's opening H (QCMRF.py:231) does not: it
    fuses into the ancilla's own multiplexer anyway, and keeping the ancilla in |0> until then is
    what lets the engine skip the still-empty part of the vector (zero tracking)."""
    lead, rest = split if split is not None else split_leading(ops)
    if hold is None:
        hold = set(q for op in rest for q in op.dense_targets())
    fold = set(q for q, m in lead.items() if q not in hold and _is_hlike(m))
    mask, gphase = 0, 1.0 + 0.0j
    front = []
    for q in sorted(fold):
        m = lead[q]
        mask |= 1 << q
        a, b = m[0, 0] / abs(m[0, 0]), m[1, 0] / abs(m[1, 0])
        gphase *= a
        if abs(b / a - 1.0) > 1e-15:
            front.append(ir.op_diag([q], [1.0, b / a]))
    # gates of folded qubits' leading runs disappear; everything else keeps its place
    closed, body = set(), []
    for op in ops:
        q = _1q_qubit(op)
        if q is not None and q not in closed and q in fold:
            continue
        closed.update(op.support())
        body.append(op)
    out = [ir.op_init(mask)] + front + body
    if abs(gphase - 1.0) > 1e-15:
        out.append(ir.op_diag([0], [gphase, gphase]))
    return out


# --------------------------------------------------------------------------------------------
# pass 2: monomial (permutation x phase) windows
# --------------------------------------------------------------------------------------------
def _is_monomial_u(op):
    nz = np.abs(op.mat) > 0
    return nz.sum() == 2 and nz.sum(axis=0).tolist() == [1, 1]


def _monomial_ok(op):
    if op.kind in ("x", "diag", "mcphase"):
        return True
    return op.kind == "u" and _is_monomial_u(op)


class _Window:
    def __init__(self):
        self.q = []                       # window qubits; window bit b <-> logical qubit q[b]
        self.pos = {}                     # logical qubit -> window bit
        self.perm = np.zeros(1, dtype=np.int64)
        self.phase = np.ones(1, dtype=np.complex128)
        self.ops = []                     # source ops absorbed so far
        self.is_id = True                 # perm is the identity right now
        self.mark = 0                     # ops[:mark] compose to the diagonal ``mark_phase``
        self.mark_phase = self.phase
        self.mark_nq = 0

    def bit(self, q):
        b = self.pos.get(q)
        if b is None:
            b = self.pos[q] = len(self.q)
            self.q.append(q)
            n = self.perm.size
            self.perm = np.concatenate([self.perm, self.perm + n])
            self.phase = np.concatenate([self.phase, self.phase])
        return b

    def _fire(self, idx, qubits, vals):
        cmask = cval = 0
        for q, v in zip(qubits, vals):
            b = self.bit(q)
            cmask |= 1 << b
            cval |= v << b
        return cmask, cval

    def add(self, op):
        k = op.kind
        if k == "x" or k == "u":
            tb = self.bit(op.target)
            cmask, cval = self._fire(None, op.ctrls, op.vals)
            idx = self.perm                              # current images (after any growth)
            if k == "x":
                if cmask:
                    self.perm = idx ^ (((idx & cmask) == cval) << tb)
                else:
                    self.perm = idx ^ (1 << tb)
                self.is_id = False
            else:
                m = op.mat
                tv = (idx >> tb) & 1
                anti = m[0, 0] == 0
                ph = np.where(tv == 0, m[1, 0], m[0, 1]) if anti else np.where(tv == 0, m[0, 0], m[1, 1])
                if cmask:
                    fire = (idx & cmask) == cval
                    self.phase = self.phase * np.where(fire, ph, 1.0)
                    if anti:
                        self.perm = idx ^ (fire << tb)
                else:
                    self.phase = self.phase * ph
                    if anti:
                        self.perm = idx ^ (1 << tb)
                if anti:
                    self.is_id = False
        elif k == "mcphase":
            cmask, cval = self._fire(None, op.qubits, op.vals)
            idx = self.perm
            self.phase = self.phase * np.where((idx & cmask) == cval, np.exp(1j * op.angle), 1.0)
        else:                                            # diag
            if list(op.qubits) == self.q:                # table index == window index: no bit shuffling
                self.phase = self.phase * (op.table if self.is_id else op.table[self.perm])
            else:
                bits = [self.bit(q) for q in op.qubits]
                idx = self.perm
                j = (idx >> bits[0]) & 1
                for e in range(1, len(bits)):
                    j |= ((idx >> bits[e]) & 1) << e
                self.phase = self.phase * op.table[j]
        self.ops.append(op)
        if not self.is_id and (k == "x" or k == "u"):
            self.is_id = bool((self.perm == np.arange(self.perm.size)).all())
        if self.is_id:
            self.mark = len(self.ops)
            self.mark_phase = self.phase
            self.mark_nq = len(self.q)


def _reduce_diag(qubits, table):
    """drop qubits the table does not depend on; None if the table is the identity"""
    qubits = list(qubits)
    table = np.asarray(table)
    b = 0
    while b < len(qubits):
        t = table.reshape(-1, 2, 2 ** b)                 # axis1 = bit b
        if np.array_equal(t[:, 0, :], t[:, 1, :]):
            table = t[:, 0, :].reshape(-1)
            qubits.pop(b)
        else:
            b += 1
    if not qubits:
        return None if table[0] == 1.0 else ir.op_diag([0], [table[0], table[0]])
    return ir.op_diag(qubits, table)


def fuse_monomial(ops, kmax=10, level_split=True):
    out = []
    pending = list(ops)
    pos = 0
    win = None

    def flush():
        """emit the identity-permutation prefix as one diag; hand back the unfused remainder"""
        nonlocal win
        if win is None:
            return []
        rem = win.ops[win.mark:]
        if win.mark == 1:
            out.append(win.ops[0])                        # a lone gate: keep its cheaper native form
        elif win.mark > 1:
            d = _reduce_diag(win.q[:win.mark_nq], win.mark_phase[:2 ** win.mark_nq])
            if d is not None:
                out.append(d)
        win = None
        return rem

    while pos < len(pending):
        op = pending[pos]
        if not _monomial_ok(op):
            rem = flush()
            if rem:                                       # could not close: emit raw, keep order
                out.extend(rem)
            out.append(op)
            pos += 1
            continue
        if win is None:
            win = _Window()
        newq = [q for q in op.support() if q not in win.pos]
        if not win.ops and len(newq) > kmax:              # a single gate wider than any window
            win = None
            out.append(op)
            pos += 1
            continue
        if newq and win.ops and win.is_id and win.mark == len(win.ops) and len(win.ops) >= 1 and level_split:
            # the run so far is a finished diagonal and the next gate reaches for a new qubit: emit
            # it now, so diagonals do not straddle the blocks of a lowered circuit
            flush()
            continue
        if len(win.q) + len(newq) > kmax:
            rem = flush()
            if rem:
                # re-feed the tail: emit its first op raw so progress is guaranteed
                out.append(rem[0])
                pending[pos:pos] = rem[1:]
            continue
        win.add(op)
        pos += 1
    rem = flush()
    out.extend(rem)
    return out


# --------------------------------------------------------------------------------------------
# pass 3: uniformly controlled 2x2 windows
# --------------------------------------------------------------------------------------------
class _MuxWindow:
    def __init__(self, target):
        self.t = target
        self.c = []                                       # control qubits, LSB first
        self.mats = np.eye(2, dtype=np.complex128)[None].copy()
        self.ops = []

    def _ctrl_bit(self, q):
        if q not in self.c:
            self.c.append(q)
            self.mats = np.concatenate([self.mats, self.mats])
        return self.c.index(q)

    def can_take(self, op, smax):
        if op.kind in ("u", "x"):
            if op.target != self.t:
                return False
            extra = [q for q in op.ctrls if q not in self.c]
        elif op.kind in ("diag", "mcphase"):
            if self.t not in op.qubits and not set(op.qubits) & set(self.c):
                return False
            extra = [q for q in op.qubits if q != self.t and q not in self.c]
        else:
            return False
        return len(self.c) + len(extra) <= smax

    def take(self, op):
        if op.kind in ("u", "x"):
            bits = [self._ctrl_bit(q) for q in op.ctrls]
            j = np.arange(self.mats.shape[0])
            fire = np.ones(j.shape, dtype=bool)
            for b, v in zip(bits, op.vals):
                fire &= ((j >> b) & 1) == v
            g = op.mat if op.kind == "u" else np.array([[0, 1], [1, 0]], dtype=np.complex128)
            self.mats[fire] = g @ self.mats[fire]
        else:
            qs = list(op.qubits)
            if op.kind == "mcphase":
                tab = np.ones(2 ** len(qs), dtype=np.complex128)
                jj = sum(v << b for b, v in enumerate(op.vals))
                tab[jj] = np.exp(1j * op.angle)
            else:
                tab = op.table
            if qs[-1] == self.t and qs[:-1] == self.c[:len(qs) - 1] and (len(qs) - 1 == len(self.c) or not self.c):
                # table index = (select bits in window order, target as MSB): a reshape, no gather
                for q in qs[:-1]:
                    self._ctrl_bit(q)
                d = np.asarray(tab).reshape(2, -1).T
            else:
                bits = [(-1 if q == self.t else self._ctrl_bit(q)) for q in qs]
                j = np.arange(self.mats.shape[0])
                d = np.empty((j.size, 2), dtype=np.complex128)
                for tv in (0, 1):
                    k = np.zeros(j.shape, dtype=np.int64)
                    for pos, b in enumerate(bits):
                        bitval = tv if b < 0 else ((j >> b) & 1)
                        k |= bitval << pos
                    d[:, tv] = tab[k]
            self.mats = d[:, :, None] * self.mats          # diag(d0, d1) @ M
        self.ops.append(op)

    def emit(self):
        n_dense = sum(1 for o in self.ops if o.kind in ("u", "x"))
        if len(self.ops) == 1:
            return [self.ops[0]]
        if n_dense == 0:
            return list(self.ops)
        if not self.c:
            return [ir.op_u(self.t, self.mats[0], label="fused")]
        off = np.abs(self.mats[:, 0, 1]).max() + np.abs(self.mats[:, 1, 0]).max()
        if off == 0.0:                                     # the run multiplied out to a diagonal
            tab = np.empty(2 ** (len(self.c) + 1), dtype=np.complex128)
            tab[0::2], tab[1::2] = self.mats[:, 0, 0], self.mats[:, 1, 1]
            d = _reduce_diag([self.t] + self.c, tab)
            return [d] if d is not None else []
        return [ir.op_mux(self.c, self.t, ir.snap(self.mats))]


def fuse_mux(ops, smax=8):
    out = []
    win = None
    deferred = []          # diagonals met inside a window that do not touch its target: they commute
                           # with everything the window can still absorb, so they slide behind it
    for op in ops:
        if win is not None:
            if op.kind in ("diag", "mcphase") and win.t not in op.qubits:
                deferred.append(op)
                continue
            if win.can_take(op, smax):
                win.take(op)
                continue
            out.extend(win.emit())
            out.extend(deferred)
            deferred = []
            win = None
        if op.kind in ("u", "x") and len(op.ctrls) <= smax:
            win = _MuxWindow(op.target)
            # diagonals directly in front of the window that touch its target belong to it too
            back = []
            sel = set(op.ctrls)
            while out and out[-1].kind in ("diag", "mcphase") and op.target in out[-1].qubits:
                more = sel | (set(out[-1].qubits) - {op.target})
                if len(more) > smax:
                    break
                sel = more
                back.insert(0, out.pop())
            for o in back:
                win.take(o)
            win.take(op)
        else:
            out.append(op)
    if win is not None:
        out.extend(win.emit())
        out.extend(deferred)
    return out


# --------------------------------------------------------------------------------------------
# pass 4: dense windows (<= 5 qubits) + structure recovery, for circuits lowered to a basis
# --------------------------------------------------------------------------------------------
_ZERO = 1e-13


_ROWS = {}


def _pair_rows(n, tb, cbits, cvals):
    """row indices (r0, r1 = r0 | 1<<tb) of an n-row matrix on which a controlled 2x2 acts; cached:
    a lowered circuit asks for the same few patterns thousands of times"""
    key = (n, tb, cbits, cvals)
    hit = _ROWS.get(key)
    if hit is None:
        rows = np.arange(n)
        fire = ((rows >> tb) & 1) == 0
        for c, v in zip(cbits, cvals):
            fire &= ((rows >> c) & 1) == v
        r0 = rows[fire]
        hit = _ROWS[key] = (r0, r0 | (1 << tb))
    return hit


def _op_on_rows(U, op, pos):
    """U <- G U for gate ``op`` acting on the row index of U; pos maps logical qubit -> row bit.
    The gates a basis-gate circuit is made of (rz, sx, x, cx) go through reshaped VIEWS of U --
    row index = (hi, bit, lo) -- without building index arrays: a few microseconds each."""
    k = op.kind
    n = U.shape[0]
    if k == "diag" and len(op.qubits) == 1:
        V = U.reshape(n >> (pos[op.qubits[0]] + 1), 2, -1)
        V *= op.table.reshape(1, 2, 1)
        return U
    if k in ("u", "x"):
        nc = len(op.ctrls)
        tb = pos[op.target]
        if nc == 0:
            V = U.reshape(n >> (tb + 1), 2, -1)
            if k == "x":
                a = V[:, 0].copy()
                V[:, 0] = V[:, 1]
                V[:, 1] = a
            else:
                V[...] = np.matmul(op.mat, V)              # (2, 2) @ (hi, 2, lo): one call instead of eight
            return U
        if nc == 1 and k == "x":
            cb = pos[op.ctrls[0]]
            hb, lb = (cb, tb) if cb > tb else (tb, cb)
            V = U.reshape(n >> (hb + 1), 2, 1 << (hb - lb - 1), 2, -1)
            v = op.vals[0]
            if cb > tb:
                a = V[:, v, :, 0].copy()
                V[:, v, :, 0] = V[:, v, :, 1]
                V[:, v, :, 1] = a
            else:
                a = V[:, 0, :, v].copy()
                V[:, 0, :, v] = V[:, 1, :, v]
                V[:, 1, :, v] = a
            return U
        r0, r1 = _pair_rows(n, tb, tuple(pos[c] for c in op.ctrls), tuple(op.vals))
        a, b = U[r0], U[r1]
        if k == "x":
            U[r0], U[r1] = b, a
        else:
            m = op.mat
            U[r0], U[r1] = m[0, 0] * a + m[0, 1] * b, m[1, 0] * a + m[1, 1] * b
        return U
    rows = np.arange(U.shape[0])
    if k == "mcphase":
        fire = np.ones(rows.shape, dtype=bool)
        for q, v in zip(op.qubits, op.vals):
            fire &= ((rows >> pos[q]) & 1) == v
        U[fire] *= np.exp(1j * op.angle)
    elif k == "diag":
        j = np.zeros_like(rows)
        for e, q in enumerate(op.qubits):
            j |= ((rows >> pos[q]) & 1) << e
        U *= op.table[j][:, None]
    elif k == "mux":
        tb = pos[op.target]
        r0 = rows[((rows >> tb) & 1) == 0]
        r1 = r0 | (1 << tb)
        j = np.zeros_like(r0)
        for e, q in enumerate(op.ctrls):
            j |= ((r0 >> pos[q]) & 1) << e
        m = op.mats[j]
        a, b = U[r0], U[r1]
        U[r0] = m[:, 0, 0, None] * a + m[:, 0, 1, None] * b
        U[r1] = m[:, 1, 0, None] * a + m[:, 1, 1, None] * b
    else:                                                  # kq
        kk = len(op.qubits)
        offs = np.zeros(2 ** kk, dtype=np.int64)
        qm = 0
        for e, q in enumerate(op.qubits):
            qm |= 1 << pos[q]
            offs |= ((np.arange(2 ** kk) >> e) & 1) << pos[q]
        base = rows[(rows & qm) == 0]
        idx = base[None, :] | offs[:, None]
        U[idx] = np.einsum("rc,cbn->rbn", op.mat, U[idx])
    return U


def _recover(qubits, U, n_ops):
    """A dense window that is block diagonal in some of its qubits is really a diagonal / a
    multiplexed 2x2: emit that (it rides in k_multi passes) instead of a dense gate."""
    k = len(qubits)
    idx = np.arange(2 ** k)
    big = np.abs(U) > _ZERO
    sel = []                                               # window bits in which U is block diagonal
    for b in range(k):
        rb = (idx >> b) & 1
        if not big[rb[:, None] != rb[None, :]].any():
            sel.append(b)
    dense = [b for b in range(k) if b not in sel]
    if not dense:
        return [d for d in [_reduce_diag(qubits, np.diag(U).copy())] if d is not None]
    if len(dense) == 1:
        t = dense[0]
        mats = np.zeros((2 ** len(sel), 2, 2), dtype=np.complex128)
        for j in range(2 ** len(sel)):
            r = 0
            for e, b in enumerate(sel):
                r |= ((j >> e) & 1) << b
            for x in (0, 1):
                for y in (0, 1):
                    mats[j, x, y] = U[r | (x << t), r | (y << t)]
        if not sel:
            return [ir.op_u(qubits[t], mats[0], label="fused")]
        # (diagonal on the selects) x (one 2x2 on t)?  then it is a plain gate plus a diagonal
        m0 = mats[0]
        r = np.einsum("xy,jxy->j", m0.conj(), mats) / np.vdot(m0, m0)
        if np.abs(mats - r[:, None, None] * m0[None]).max() < _ZERO:
            d = _reduce_diag([qubits[b] for b in sel], r)
            return [ir.op_u(qubits[t], m0, label="fused")] + ([d] if d is not None else [])
        return [ir.op_mux([qubits[b] for b in sel], qubits[t], ir.snap(mats))]
    if n_ops < 3:
        return None
    return [ir.op_kq(qubits, U)]


_RC = {}


def _dense_mask(U):
    """window bits in which U is NOT block diagonal, as a bit mask: bit b is set iff some nonzero
    entry sits at (r, c) with bit b of r != bit b of c, i.e. iff bit b of OR(r xor c) is set"""
    n = U.shape[0]
    rc = _RC.get(n)
    if rc is None:
        idx = np.arange(n)
        rc = _RC[n] = (idx[:, None] ^ idx[None, :])
    return int(np.bitwise_or.reduce(rc[np.abs(U) > _ZERO], initial=0))


def _n_dense_bits(U):
    return bin(_dense_mask(U)).count("1")


def _factor_left(U, mask, order=None):
    """U = (G on window bit b) . M with M block diagonal in b, for one of the (two) dense bits of
    ``mask``: returns (b, G, M) with M's only dense bit the other one, else None.

    This is what the end of a block looks like in a circuit that went through a transpiler's
    clean-up passes: the H that closes one CCX on the AND scratch qubit and the H that opens the
    next have cancelled ACROSS the block boundary (QCMRF.py:225-227 lowered, then merged one-qubit
    runs), so each block's own gates multiply out to (one-qubit gate) x (multiplexed 2x2) rather than
    to the multiplexer itself."""
    n = U.shape[0]
    for b in (order if order is not None else range(n.bit_length() - 1)):
        if not (mask >> b) & 1:
            continue
        V = U.reshape(n >> (b + 1), 2, 1 << b, n >> (b + 1), 2, 1 << b)      # row (hi, r_b, lo), column (hi', c_b, lo')
        cols = []
        for j in (0, 1):                                   # column block j: [g0j M_j ; g1j M_j]
            A0, A1 = V[:, 0, :, :, j, :], V[:, 1, :, :, j, :]
            s0, s1 = float(np.vdot(A0, A0).real), float(np.vdot(A1, A1).real)
            ref, oth, sr = (A0, A1, s0) if s0 >= s1 else (A1, A0, s1)
            r = np.vdot(ref, oth) / sr
            if np.abs(oth - r * ref).max() > 1e-12:
                cols = None
                break
            c = np.array([1.0, r]) if s0 >= s1 else np.array([r, 1.0])
            cols.append(c / np.linalg.norm(c))
        if cols is None:
            continue
        G = np.array(cols).T                               # columns g_.0, g_.1
        if abs(np.vdot(G[:, 0], G[:, 1])) > 1e-12:
            continue
        M = U.copy()
        W = M.reshape(n >> (b + 1), 2, -1)
        a0, a1 = W[:, 0].copy(), W[:, 1].copy()
        Gh = G.conj().T
        W[:, 0] = Gh[0, 0] * a0 + Gh[0, 1] * a1
        W[:, 1] = Gh[1, 0] * a0 + Gh[1, 1] * a1
        rest = _dense_mask(M)
        if not (rest >> b) & 1 and bin(rest).count("1") <= 1:
            return b, G, M
    return None


class _DenseWindow:
    def __init__(self):
        self.q, self.pos, self.U, self.ops = [], {}, np.eye(1, dtype=np.complex128), []
        self.mark = None                 # (n_ops, U or M, qubits, carried 1q op or None, target qubit) at the remembered structured point
        self.ndense = None               # cached _n_dense_bits(U); None = stale
        self.busy = set()                # wires some multi-qubit gate of the window touches

    def add(self, op):
        sup = op.support()
        if len(sup) > 1:
            self.busy.update(sup)
        for q in sup:
            if q not in self.pos:
                self.pos[q] = len(self.q)
                self.q.append(q)
                n = self.U.shape[0]
                grown = np.zeros((2 * n, 2 * n), dtype=np.complex128)
                grown[:n, :n] = self.U
                grown[n:, n:] = self.U
                self.U = grown
                self.ndense = None
        _op_on_rows(self.U, op, self.pos)
        self.ops.append(op)
        if op.kind not in ("diag", "mcphase"):
            self.ndense = None

    def consider_mark(self):
        """Called when the window is about to reach for another qubit -- where one block of a lowered
        circuit ends and the next begins -- and when it closes.  If what it holds is exactly a
        diagonal / multiplexed 2x2 right now (possibly times a one-qubit gate on one wire that
        cancelled across the block boundary, _factor_left), remember the point; should the window
        overflow later it is cut at the remembered point.  A later point replaces an earlier one
        unless it only ADDS selects to the same target: that is the next block's first gates
        leaking in (its CCZ core is diagonal and would ride along), not a bigger block."""
        if len(self.ops) < 2:
            return
        cand = self._candidate(self.U, [])
        if cand is None:
            # ... up to the X gates that close its wires (the X that restores a negated control after a
            # block's last AND, QCMRF.py:224-227): they are the last thing on their wire in the
            # window, so they commute to behind it
            last = {}
            for i, o in enumerate(self.ops):
                for x in o.support():
                    last[x] = i
            tail = []
            for i in sorted(set(last.values())):
                o = self.ops[i]
                if o.kind == "x" and not o.ctrls:
                    tail.append(i)
                elif o.kind == "u" and not o.ctrls and abs(o.mat[0, 0]) < 1e-15 and abs(o.mat[1, 1]) < 1e-15:
                    tail.append(i)
            if tail and len(tail) < len(self.ops) - 1:
                U2 = self.U.copy()
                for i in tail:
                    qq, m = _as_1q(self.ops[i])
                    _op_on_rows(U2, ir.op_u(qq, np.asarray(m).conj().T), self.pos)
                cand = self._candidate(U2, tail)
        if cand is None:
            return
        old = self.mark
        if old is not None and old[4] is not None and old[4] == cand[4] and len(cand[2]) > len(old[2]) and len(old[2]) >= 3:
            return
        self.mark = cand

    def _candidate(self, U, peeled):
        """(n_ops, M, qubits, carried op, target qubit, peeled op indices) if U is a diagonal / multiplexed
        2x2, possibly times one one-qubit gate that is carried on to the next window; else None"""
        dm = _dense_mask(U)
        nd = bin(dm).count("1")
        if not peeled:
            self.ndense = nd
        if nd <= 1:
            return (len(self.ops), U.copy() if U is self.U else U, list(self.q), None, self.q[dm.bit_length() - 1] if dm else None, list(peeled))
        if nd == 2:
            cnt = [0] * len(self.q)                        # the busier wire first: the AND scratch qubit carries most of a block's gates
            for o in self.ops:
                for x in o.support():
                    cnt[self.pos[x]] += 1
            f = _factor_left(U, dm, sorted(range(len(self.q)), key=lambda b: -cnt[b]))
            if f is not None:
                b, G, M = f
                rest = dm & ~(1 << b)
                return (len(self.ops), M, list(self.q), ir.op_u(self.q[b], G, label="carry"), self.q[rest.bit_length() - 1], list(peeled))
        return None


_DENSE_KINDS = ("u", "x", "diag", "mcphase", "mux", "kq")


def _components(ops):
    """ops grouped by the connected components of their qubit supports (order kept inside a group,
    groups in order of their first op); groups act on disjoint qubits, so they commute"""
    parent = {}

    def find(x):
        while parent.setdefault(x, x) != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for o in ops:
        sup = o.support()
        for q in sup[1:]:
            parent[find(q)] = find(sup[0])
        if sup:
            find(sup[0])
    groups, order = {}, []
    for o in ops:
        sup = o.support()
        r = find(sup[0]) if sup else None
        if r not in groups:
            groups[r] = []
            order.append(r)
        groups[r].append(o)
    return [groups[r] for r in order]


def fuse_dense(ops, kmax=5):
    # nothing to gain unless two neighbouring gates fit one window together
    for a, b in zip(ops, ops[1:]):
        if a.kind in _DENSE_KINDS and b.kind in _DENSE_KINDS and len(set(a.support()) | set(b.support())) <= kmax:
            break
    else:
        return list(ops)
    out = []
    pending = list(ops)
    pos = 0
    win = None

    def emit(ops_, q, U):
        """one window: recovered form if it has one.  A window whose gates fall into several groups
        of qubits that never interact (say two leftover one-qubit runs, or the tail of one block and
        the head of the next) is a tensor product: each group is recovered on its own, never
        multiplied out into one dense gate over all of them"""
        comps = _components(ops_)
        if len(comps) > 1:
            for part in comps:
                w2 = _DenseWindow()
                for o in part:
                    w2.add(o)
                rec = _recover(w2.q, w2.U, len(part)) if len(part) >= 2 else None
                out.extend(rec if rec is not None else part)
            return
        rec = _recover(q, U, len(ops_)) if len(ops_) >= 2 else None
        out.extend(rec if rec is not None else ops_)

    def flush(final=False):
        """emit the window (up to its remembered structured point, if it has one); return what is left"""
        nonlocal win
        if win is None or not win.ops:
            win = None
            return []
        w, win = win, None
        w.consider_mark()                  # the window as it stands: a block that fills it exactly, or the end of the circuit
        if final and len(w.ops) >= 2 and (w.mark is None or w.mark[0] < len(w.ops)):
            if True:
                # ... up to SOME of the non-diagonal one-qubit gates that close its wires (the X that
                # restores a negated control after the block's last AND commutes to behind the window;
                # the H that closes the AND scratch qubit must stay): smallest subsets first
                last = {}
                for i, o in enumerate(w.ops):
                    for x in o.support():
                        last[x] = i
                tail = sorted(set(i for x, i in last.items() if w.ops[i].kind in ("u", "x") and not w.ops[i].ctrls))
                if tail and len(tail) < len(w.ops) - 1:
                    import itertools
                    done = False
                    for r in range(1, len(tail) + 1):
                        for sub in itertools.combinations(tail, r):
                            U2 = w.U.copy()
                            for i in sub:
                                qq, m = _as_1q(w.ops[i])
                                _op_on_rows(U2, ir.op_u(qq, np.asarray(m).conj().T), w.pos)
                            if bin(_dense_mask(U2)).count("1") <= 1:
                                core = [o for i, o in enumerate(w.ops) if i not in sub]
                                rec = _recover(w.q, U2, len(core))
                                out.extend(rec if rec is not None else core)
                                out.extend(w.ops[i] for i in sub)
                                done = True
                                break
                        if done:
                            return []
        if w.mark is not None and (w.mark[0] < len(w.ops) or w.mark[3] is not None or w.mark[5]):
            n, U, q, carry, _, peeled = w.mark
            if carry is not None or peeled:               # U has the carried / peeled one-qubit gates divided out already
                rec = _recover(q, U, n - len(peeled))
                out.extend(rec if rec is not None else [ir.op_kq(q, U)] if len(q) > 1 else [ir.op_u(q[0], U)])
                return ([carry] if carry is not None else []) + [w.ops[i] for i in peeled] + w.ops[n:]
            emit(w.ops[:n], q, U)
            return w.ops[n:]
        emit(w.ops, w.q, w.U)
        return []

    # One-qubit gates on a wire the window does not hold yet wait outside it (``held``) until a
    # multi-qubit gate brings their wire in -- then they enter first, in order.  If that never happens
    # (a leftover X of an earlier block, a wire's closing gate) they commute with everything in the
    # window and go straight to the output when it closes: they would only hide its structure.
    held = {}

    def release(keep=()):
        for q in sorted(held):
            if q not in keep:
                out.extend(held.pop(q))

    while pos < len(pending):
        op = pending[pos]
        sup = op.support()
        if op.kind not in ("u", "x", "diag", "mcphase", "mux", "kq") or len(set(sup)) > kmax:
            rem = flush()
            if rem:
                pending[pos:pos] = rem
                continue
            release()
            out.append(op)
            pos += 1
            continue
        if len(sup) == 1 and (win is None or sup[0] not in win.pos):
            held.setdefault(sup[0], []).append(op)
            pos += 1
            continue
        if win is not None:
            new = [q for q in sup if q not in win.pos]
            if new:
                win.consider_mark()
            if len(win.q) + len(new) > kmax:
                rem = flush()
                release(keep=sup)
                pending[pos:pos] = rem
                continue
        if win is None:
            win = _DenseWindow()
        for q in sup:
            if q not in win.pos:
                for h in held.pop(q, ()):
                    win.add(h)
        win.add(op)
        pos += 1
    while True:
        rem = flush(final=True)
        if not rem:
            break
        win = _DenseWindow()
        for op in rem:
            win.add(op)
    release()
    return out


# --------------------------------------------------------------------------------------------
# pass 4: gates on fresh qubits become factors of the initial product state
# --------------------------------------------------------------------------------------------
def _as_mux(op):
    """(selects, target, mats[2^k,2,2]) of a 2x2-type op, controls spelled out as table entries"""
    if op.kind == "mux":
        return tuple(op.ctrls), op.target, op.mats
    if op.kind in ("u", "x"):
        m = op.mat if op.kind == "u" else _X2
        k = len(op.ctrls)
        mats = np.tile(np.eye(2, dtype=np.complex128), (1 << k, 1, 1))
        hit = sum(int(v) << e for e, v in enumerate(op.vals))
        mats[hit] = m
        return tuple(op.ctrls), op.target, mats
    return None


def _as_diag(op):
    if op.kind == "diag":
        return tuple(op.qubits), op.table
    if op.kind == "mcphase":
        k = len(op.qubits)
        t = np.ones(1 << k, dtype=np.complex128)
        t[sum(int(v) << e for e, v in enumerate(op.vals))] = np.exp(1j * op.angle)
        return tuple(op.qubits), t
    return None


def _slice_zero(qubits, table, populated, ent=1):
    """drop every qubit outside ``populated`` from a table index: such a qubit is known |0>, so
    only its 0-slice is ever read.  ``table`` has 2^k entries of ``ent`` values each."""
    qubits = list(qubits)
    t = np.asarray(table).reshape((2,) * len(qubits) + (ent,)) if qubits else np.asarray(table).reshape(1, ent)
    # numpy axis a <-> qubits[k-1-a] (index bit e is qubit e, little-endian)
    for e in range(len(qubits) - 1, -1, -1):
        if not (populated >> qubits[e]) & 1:
            t = np.take(t, 0, axis=len(qubits) - 1 - e)
            del qubits[e]
    return tuple(qubits), t.reshape(-1, ent)


def fold_fresh(ops):
    """[init(mask)] + ops  ->  [init(mask')] + diagonal factors + the ops that could not fold.

    The state right after ``init`` is a product state known in closed form: amplitude ``val`` on
    every index whose bits outside ``mask`` are 0.  A 2x2-type gate (u, x, mux; any controls)
    whose TARGET nothing has touched yet does not need a sweep: applied to |0> it writes column 0
    of its matrix, amp'(.., t=b, ..) = M_sel[b,0] amp(.., t=0, ..); applied to an untouched |+> it
    writes the row sums.  Either way the result is again "``val`` times a table looked up from the
    index", i.e. an ``init`` over mask + {t} followed by a DIAGONAL on (selects, t) -- which libqsv
    evaluates while it writes the initial state, at no HBM traffic of its own.  Diagonal gates
    multiply into the same product.  Selects on qubits still |0> are sliced away on the spot, so
    a factor never depends on an unpopulated bit.

    Exact and structural (which qubits an op touches, the |0..0> start) -- the rule any simulator
    may apply to a gate on a freshly allocated qubit; for a QCMRF circuit every clique block ends up
    here, because each ancilla is touched by exactly one fused multiplexer (QCMRF.py:231-236).
    Folding stops, per qubit, at the first op that cannot fold: everything after it on those
    qubits is emitted unchanged, in order."""
    if not ops or ops[0].kind != "init":
        return list(ops)
    populated = ops[0].mask          # bits whose |1> half carries amplitude
    plus = ops[0].mask               # populated AND still exactly |+>, untouched by any factor
    blocked = set()                  # qubits an emitted (unfolded) op has touched
    factors, emitted = [], []
    scalar = 1.0 + 0.0j
    for op in ops[1:]:
        sup = set(op.support())
        done = False
        if not (sup & blocked):
            dg = _as_diag(op)
            mx = _as_mux(op) if dg is None else None
            if dg is not None:
                q, t = _slice_zero(dg[0], dg[1], populated)
                if q and not (t != t.ravel()[0]).any():
                    q = ()                                  # a constant table is a number (a global phase), not a factor
                if q:
                    factors.append(ir.op_diag(q, t.ravel()))
                    for x in q:
                        plus &= ~(1 << x)
                else:
                    scalar *= t.ravel()[0]
                done = True
            elif mx is not None:
                sel, tg, mats = mx
                in_pop = (populated >> tg) & 1
                if not in_pop or (plus >> tg) & 1:
                    smask = 0
                    for x in sel:
                        smask |= 1 << x
                    if smask & ~populated:                  # selects on qubits still |0>: only their 0-slice is read
                        sq, st = _slice_zero(sel, np.asarray(mats).reshape(-1, 4), populated, ent=4)
                        st = st.reshape(-1, 2, 2)
                    else:
                        sq, st = sel, mats
                    if not in_pop and not st[:, 1, 0].any() and np.array_equal(st, np.broadcast_to(np.eye(2), st.shape)):
                        continue            # e.g. a control that needs a |0> qubit to be 1: the gate never fires
                    # table index: bits 0..k-1 the surviving selects, bit k the target
                    tab = np.empty(2 * st.shape[0], dtype=np.complex128)
                    half = tab.reshape(2, -1)
                    if in_pop:
                        np.add(st[:, :, 0].T, st[:, :, 1].T, out=half)
                    else:
                        np.multiply(st[:, :, 0].T, _SQRT2, out=half)
                    fq = tuple(sq) + (tg,)
                    factors.append(Op("diag", qubits=fq, table=tab))
                    populated |= 1 << tg
                    plus &= ~(smask | (1 << tg))
                    done = True
        if not done:
            emitted.append(op)
            blocked |= sup
            for x in sup:
                plus &= ~(1 << x)
    if abs(scalar - 1.0) > 1e-15:
        if factors:
            factors[0] = ir.op_diag(factors[0].qubits, factors[0].table * scalar)
        else:
            factors.append(ir.op_diag([0], [scalar, scalar]))
    return [ir.op_init(populated)] + factors + emitted


# --------------------------------------------------------------------------------------------
# peephole: 1q . diag . X . diag . X . 1q on one qubit  ->  one multiplexed 2x2
# --------------------------------------------------------------------------------------------
def fuse_sandwich(ops):
    """``U2 . X . D2 . X . D1 . U1`` with U1, U2 uncontrolled one-qubit gates and the two X on the same
    qubit a, D1 and D2 diagonals over one qubit list that contains a: since X diag(d0,d1) X =
    diag(d1,d0), the run is U2 diag(D2[s,1] D1[s,0], D2[s,0] D1[s,1]) U1 for every value s of the
    other qubits -- one uniformly controlled 2x2.  This is the real-part-extraction block of the
    reference (QCMRF.py:231-236: H, cU, X, cU^dg, X, H) once ingest has emitted cU as a diagonal;
    the monomial and multiplexer windows reach the same op the long way (they stay for
    everything that does not have exactly this shape).

    Matching is per op; the arithmetic of all matches with the same table shape runs as ONE batched
    product (a 34-qubit QCMRF circuit has 19 of them: 19 einsum/snap calls cost more than the rest
    of the pass pipeline together)."""
    out, hits, i, n = [], [], 0, len(ops)
    while i < n:
        o = ops[i]
        if i + 5 < n and o.kind == "u" and not o.ctrls:
            d1, x1, d2, x2, u2 = ops[i + 1:i + 6]
            a = o.target
            if (d1.kind == "diag" and d2.kind == "diag" and x1.kind == "x" and x2.kind == "x" and u2.kind == "u"
                    and not x1.ctrls and not x2.ctrls and not u2.ctrls
                    and x1.target == a and x2.target == a and u2.target == a
                    and d1.qubits == d2.qubits and a in d1.qubits and 2 <= len(d1.qubits) <= 9):
                hits.append((len(out), o, d1, d2, u2))
                out.append(None)
                i += 6
                continue
        out.append(o)
        i += 1
    groups = {}
    for h in hits:
        qs = h[2].qubits
        groups.setdefault((len(qs), qs.index(h[1].target)), []).append(h)
    for (k, e), hs in groups.items():
        ax = k - e                                             # numpy axis of table-index bit e is k-1-e (+1: batch axis)
        t1 = np.moveaxis(np.stack([h[2].table for h in hs]).reshape((len(hs),) + (2,) * k), ax, -1).reshape(len(hs), -1, 2)
        t2 = np.moveaxis(np.stack([h[3].table for h in hs]).reshape((len(hs),) + (2,) * k), ax, -1).reshape(len(hs), -1, 2)
        ed = t2[:, :, ::-1] * t1                               # (batch, select, target value)
        u1 = np.stack([h[1].mat for h in hs])                  # (batch, 2, 2)
        u2 = np.stack([h[4].mat for h in hs])
        mats = ir.snap(np.matmul(u2[:, None] * ed[:, :, None, :], u1[:, None]))
        for b, (pos, o, d1, d2, _) in enumerate(hs):
            a = o.target
            out[pos] = Op("mux", ctrls=tuple([q for q in d1.qubits if q != a]), target=a, mats=mats[b])
    return out


# --------------------------------------------------------------------------------------------
# helpers for circuits lowered to a basis by a transpiler (run_experiment.py:52)
# --------------------------------------------------------------------------------------------
def hoist_leading(ops, split=None, lone_h=False):
    """A transpiler emits gates in some topological order of the circuit's DAG: the one-qubit run
    that opens a wire (the lowered H of QCMRF.py:204-205, merged with whatever one-qubit gates
    follow it) may sit anywhere before the wire's first two-qubit gate, in the middle of another
    block's gates.  Nothing has touched the wire before it, so the run commutes to the very front:
    [one fused 2x2 per hoisted wire] + everything else in its order.

    Hoisted are the wires whose first multi-qubit gate uses them as a CONTROL (or diagonally): a
    superposed select qubit -- an MRF variable -- whose opening H would otherwise make every window
    that contains it dense in that qubit.  A wire that starts as a TARGET (the AND scratch qubit, an
    ancilla) keeps its opening gate where it is: it is part of its block."""
    lead, rest = split if split is not None else split_leading(ops)
    starts_as_target = set()
    seen = set()
    for op in rest:
        sup = op.support()
        fresh = [q for q in sup if q not in seen]
        if fresh:
            dense = op.dense_targets()
            starts_as_target.update(q for q in fresh if q in dense)
            seen.update(fresh)
    if lone_h and starts_as_target:
        # ... unless that opening gate is the ONLY non-monomial one-qubit gate the wire ever sees: then the wire is a
        # select all the same (an MRF variable whose first two-qubit gate happens to be the cx tail of a lowered CCX,
        # which targets a control); a scratch or ancilla wire meets a second Hadamard-like gate that closes its block
        again = set()
        for op in rest:
            if op.kind == "u" and not op.ctrls and op.target in starts_as_target:
                m = op.mat
                if min(abs(m[0, 0]), abs(m[0, 1])) > 1e-12:
                    again.add(op.target)
        starts_as_target &= again
    front, keep = [], {}
    for q in sorted(lead):
        m = lead[q]
        if abs(m[0, 1]) < 1e-15 and abs(m[1, 0]) < 1e-15:
            o = None if (abs(m[0, 0] - 1.0) <= 1e-15 and abs(m[1, 1] - 1.0) <= 1e-15) else ir.op_diag([q], [m[0, 0], m[1, 1]])
        else:
            o = ir.op_u(q, m, label="lead")
        if o is None:
            continue
        if q in starts_as_target:
            keep[q] = o
        else:
            front.append(o)
    if keep:                                   # put a kept opening gate right in front of its wire's first gate
        body = []
        for op in rest:
            for q in op.support():
                o = keep.pop(q, None)
                if o is not None:
                    body.append(o)
            body.append(op)
        body.extend(keep[q] for q in sorted(keep))
        rest = body
    return front, rest


def merge_1q_runs(ops):
    """consecutive uncontrolled one-qubit gates on a wire (no other gate on that wire in between)
    -> one 2x2, placed where the run BEGINS (right behind the previous gate on that wire): a
    transpiler's topological order tends to leave such runs wherever the wire's next two-qubit gate
    happens to be, far from the block they close.  Exact; a lowered circuit shrinks by a third and
    the dense windows that follow multiply far fewer matrices."""
    out, pend, slot = [], {}, {}

    def flush(q):
        m = pend.pop(q, None)
        if m is None:
            return
        i = slot.pop(q)
        single = keep.pop(q, None)
        if single is not None:
            out[i] = single
            return
        if abs(m[0, 1]) < 1e-15 and abs(m[1, 0]) < 1e-15:
            if abs(m[0, 0] - 1.0) > 1e-15 or abs(m[1, 1] - 1.0) > 1e-15:
                out[i] = Op("diag", qubits=(q,), table=np.array([m[0, 0], m[1, 1]]))
        elif abs(m[0, 0]) < 1e-15 and abs(m[1, 1]) < 1e-15 and abs(m[0, 1] - 1.0) < 1e-15 and abs(m[1, 0] - 1.0) < 1e-15:
            out[i] = Op("x", target=q)
        else:
            out[i] = Op("u", target=q, mat=m, label="run")
    keep = {}                              # wire -> the single op of a run of length one (re-emitted untouched)
    for op in ops:
        q = _1q_qubit(op)
        if q is not None:
            k = op.kind
            cur = pend.get(q)
            if cur is None:
                slot[q] = len(out)
                out.append(None)
                keep[q] = op
            else:
                keep.pop(q, None)
            if k == "diag":
                t = op.table
                pend[q] = np.array([[t[0], 0], [0, t[1]]], dtype=np.complex128) if cur is None else \
                    np.array([[t[0] * cur[0, 0], t[0] * cur[0, 1]], [t[1] * cur[1, 0], t[1] * cur[1, 1]]])
            else:
                m = _as_1q(op)[1]
                pend[q] = np.array(m, dtype=np.complex128) if cur is None else m @ cur
            continue
        if pend:
            for x in op.support():
                if x in pend:
                    flush(x)
        out.append(op)
    for q in sorted(pend):
        flush(q)
    return [o for o in out if o is not None]


def defer_1q(ops):
    """every uncontrolled one-qubit gate moves FORWARD to just in front of the next gate on its wire
    (exact: nothing in between touches the wire) -- the opposite of ``merge_1q_runs``' placement,
    used after the blocks have been re-assembled so that a wire's opening gate (hoisted to the front
    by ``hoist_leading``) meets the multiplexer it belongs to and ``fuse_mux`` joins them."""
    out, pend = [], {}
    for op in ops:
        q = _1q_qubit(op)
        if q is not None and q not in pend:
            pend[q] = op
            continue
        for x in op.support():
            o = pend.pop(x, None)
            if o is not None:
                out.append(o)
        out.append(op)
    out.extend(pend[q] for q in sorted(pend))
    return out


def _conj_x(op, F):
    """X_F op X_F for the set F of qubits: control values and table selects on them flip, a 2x2 on
    such a target has rows and columns swapped.  Returns a new op (or the same if untouched)."""
    sup = op.support()
    if not F.intersection(sup):
        return op
    k = op.kind
    if k in ("u", "x"):
        vals = tuple(v ^ 1 if c in F else v for c, v in zip(op.ctrls, op.vals))
        if k == "x":
            return Op("x", target=op.target, ctrls=op.ctrls, vals=vals)
        m = op.mat[::-1, ::-1] if op.target in F else op.mat
        return Op("u", target=op.target, ctrls=op.ctrls, vals=vals, mat=np.ascontiguousarray(m), label=op.label)
    if k == "mcphase":
        return Op("mcphase", qubits=op.qubits, vals=tuple(v ^ 1 if q in F else v for q, v in zip(op.qubits, op.vals)), angle=op.angle)
    if k == "diag":
        flip = sum(1 << e for e, q in enumerate(op.qubits) if q in F)
        return Op("diag", qubits=op.qubits, table=op.table[np.arange(op.table.size) ^ flip])
    if k == "mux":
        flip = sum(1 << e for e, q in enumerate(op.ctrls) if q in F)
        mats = op.mats[np.arange(op.mats.shape[0]) ^ flip]
        if op.target in F:
            mats = mats[:, ::-1, ::-1]
        return Op("mux", ctrls=op.ctrls, target=op.target, mats=np.ascontiguousarray(mats))
    if k == "kq":
        flip = sum(1 << e for e, q in enumerate(op.qubits) if q in F)
        idx = np.arange(op.mat.shape[0]) ^ flip
        return Op("kq", qubits=op.qubits, mat=np.ascontiguousarray(op.mat[np.ix_(idx, idx)]))
    return None


def absorb_x(ops):
    """Plain X gates travel BACKWARDS through the circuit as a frame -- every op they pass is
    conjugated (controls / selects on the qubit flip) -- until they meet an uncontrolled one-qubit
    gate on their wire, which swallows them (M -> X M), or another plain X, which they cancel.  A
    lowered QCMRF circuit is full of them: the +-flag X gates around each AND (QCMRF.py:224-227) are
    merged and re-ordered by a transpiler so that a block's closing X ends up far from its opening
    one, which would make the MRF variable qubits look like dense targets.  Afterwards they are what
    they were in the nested circuit: control values."""
    F = set()
    out = []
    for op in reversed(ops):
        k = op.kind
        if k == "x" and not op.ctrls:
            F ^= {op.target}
            continue
        if k == "u" and not op.ctrls and abs(op.mat[0, 0]) < 1e-15 and abs(op.mat[1, 1]) < 1e-15:
            # anti-diagonal 2x2 = diag(m01, m10) . X: the diagonal stays here, the X travels on
            t = op.target
            d = Op("diag", qubits=(t,), table=np.array([op.mat[0, 1], op.mat[1, 0]]))
            d = _conj_x(d, F) if F else d
            if abs(d.table[0] - 1.0) > 1e-15 or abs(d.table[1] - 1.0) > 1e-15:
                out.append(d)
            F ^= {t}
            continue
        if F:
            if k == "u" and not op.ctrls and op.target in F:
                F.discard(op.target)
                op = Op("u", target=op.target, mat=np.ascontiguousarray(op.mat[::-1, :]), label=op.label)    # X . M: rows swapped
            new = _conj_x(op, F) if F else op
            if new is None:                                   # an op kind the frame cannot pass: drop the frame here
                out.extend(Op("x", target=q) for q in sorted(F))
                F = set()
            else:
                op = new
        out.append(op)
    out.extend(Op("x", target=q) for q in sorted(F))
    out.reverse()
    return out


# --------------------------------------------------------------------------------------------
def _fuse_body(ops, level, kmax, smax, lowered=False, dense_kmax=5):
    head, body = ops[:1], ops[1:]
    if level >= 2:
        body = fuse_sandwich(body)
    body = fuse_monomial(body, kmax=kmax)
    if level >= 2:
        body = fuse_mux(body, smax=smax)
    if level >= 3 and not lowered:
        body = fuse_dense(body, kmax=dense_kmax)
    return head + body


def reassemble(body, dense_kmax=5):
    """basis-gate input: re-assemble the blocks first, while the gate order is still pristine -- symbolically
    (unlower: integer arithmetic per gate, exact); what that has to pass on as raw gates goes through the
    numeric <= 5-qubit windows"""
    from .unlower import unlower
    body, n_raw = unlower(body)
    if n_raw:
        body = fuse_dense(body, kmax=dense_kmax)
    return defer_1q(body)


def merge_diagonals(ops):
    """Diagonal gates commute with each other and with everything that is diagonal on their qubits (a multiplexer on
    its selects): each one travels forward to the next gate that is DENSE on one of its qubits (or to the end) and
    multiplies into a diagonal already waiting there on the same qubits or a superset of them.  What is left of the
    phases a lowered circuit scatters over its variable wires (a T of a hoisted opening run here, its inverse out of
    the last Toffoli network there) meets and cancels."""
    out, pend = [], []                                   # pend: [qubits tuple, table] in arrival order

    def emit(entry):
        d = _reduce_diag(entry[0], entry[1])
        if d is not None:
            out.append(d)

    for op in ops:
        if op.kind in ("diag", "mcphase"):
            q, t = _as_diag(op)
            qs = set(q)
            host = next((e for e in pend if qs <= set(e[0])), None)
            if host is None:
                grown = [e for e in pend if set(e[0]) < qs]     # the newcomer is the superset: it takes the smaller ones in
                host = [tuple(q), np.array(t, dtype=np.complex128)]
                for e in grown:
                    pend.remove(e)
                    _mul_into(host, e[0], e[1])
                pend.append(host)
            else:
                _mul_into(host, q, t)
            continue
        dense = set(op.dense_targets())
        if dense and pend:
            keep = []
            for e in pend:
                if dense.intersection(e[0]):
                    emit(e)
                else:
                    keep.append(e)
            pend = keep
        out.append(op)
    for e in pend:
        emit(e)
    return out


def _mul_into(host, qubits, table):
    """host[1] *= table, the table's index bits re-read from the host's qubit order"""
    hq = host[0]
    j = np.arange(host[1].size)
    k = np.zeros_like(j)
    for e, q in enumerate(qubits):
        k |= ((j >> hq.index(q)) & 1) << e
    host[1] = host[1] * np.asarray(table)[k]


def optimise(ops, level=3, kmax=10, smax=8, fresh=True, dense_kmax=5, flat=None):
    """level 0: gate by gate as ingested (|0..0> init prepended).
    level 1: + init folding + diagonal (monomial) fusion.   level 2: + multiplexer fusion.
    level 3: + dense <= 5-qubit windows with structure recovery (for basis-gate circuits)
             + (``fresh``) gates on untouched qubits folded into the initial product state.
    ``flat``: what ingest's walk of a basis-gate circuit already knows about each wire (Ingested.flat): the run
    that opens it, how its first two-qubit gate uses it, whether it is a dense target later -- saves three walks
    over thousands of ops; one-qubit runs are merged already."""
    out = _optimise(ops, level, kmax, smax, dense_kmax, flat)
    return fold_fresh(out) if (fresh and level >= 3) else out


def _optimise(ops, level, kmax, smax, dense_kmax=5, flat=None):
    if level <= 0:
        return [ir.op_init(0)] + list(ops)
    compact = flat is not None and flat.get("compact")
    if compact:
        # ops are unlower's gate records: the wires that open with a Hadamard-like run are those whose record says so
        from .unlower import rec_to_op
        lead_ops = flat["lead"]
        cands = set(q for q, r in lead_ops.items() if r[0] == "h")         # an 'h' record IS e^{ig} D(b) H D(a)
        hold0 = flat["dense"] & cands
        if level < 3 or hold0 != cands or not cands:         # not the lowered case after all: the passes below want ir.Op objects
            conv = {}
            for r in ops:
                if id(r) not in conv:
                    conv[id(r)] = rec_to_op(r) if type(r) is tuple else r
            flat = dict(flat, compact=False, lead={q: conv[id(r)] for q, r in lead_ops.items()})
            return _optimise([conv[id(r)] for r in ops], level, kmax, smax, dense_kmax, flat)
        skip = set(map(id, lead_ops.values()))
        lead = rest = None
    elif flat is not None:
        lead_ops = flat["lead"]
        lead = {q: np.array(_as_1q(o)[1], dtype=np.complex128) for q, o in lead_ops.items()}
        skip = set(map(id, lead_ops.values()))
        rest = [o for o in ops if id(o) not in skip]
        cands = set(q for q, m in lead.items() if _is_hlike(m))
        hold0 = flat["dense"] & cands
    else:
        lead, rest = split_leading(ops)
        cands = set(q for q, m in lead.items() if _is_hlike(m))
        hold0 = set(q for op in rest for q in op.dense_targets()) & cands
    if level < 3 or hold0 != cands or not cands:
        return _fuse_body(fold_init(ops, hold=hold0, split=(lead, rest)), level, kmax, smax, dense_kmax=dense_kmax)
    # every candidate looks dense in the raw stream: a circuit lowered to basis gates, where even
    # pure select qubits are CX targets inside decompositions (and a CCX opens with rz-sx-rz on
    # its own target).  Re-assemble the blocks first with nothing folded; in THAT op list the
    # variable qubits' opening gates stand alone in front and are never dense again, so the
    # ordinary rule applies to it.
    if flat is not None:
        # hoist_leading's rule, read off the wire facts: an opening run goes to the front unless its wire starts as a
        # TARGET and meets another Hadamard-like gate later (scratch, ancilla: the run is part of its block and sits
        # where the run began -- right in front of the wire's first gate -- already)
        role, hrest = flat["role"], flat["hrest"]
        front, keep = [], set()
        for q in sorted(lead_ops):
            if role.get(q) == "t" and hrest.get(q, 0) > 0:
                keep.add(id(lead_ops[q]))
            else:
                front.append(rec_to_op(lead_ops[q]) if compact else lead_ops[q])
        body = [o for o in ops if id(o) not in skip or id(o) in keep]
    else:
        front, rest = hoist_leading(ops, split=(lead, rest), lone_h=True)
        body = merge_1q_runs(rest)
    # the X gates the blocks leave behind (the +-flag X of QCMRF.py:224-227, merged and re-ordered by a transpiler) go back
    # to being control values BEFORE the multiplexer pass sees them as dense gates on the variable qubits
    body = absorb_x(front + reassemble(body, dense_kmax))
    fused = _fuse_body([ir.op_init(0)] + body, level, kmax, smax, lowered=True, dense_kmax=dense_kmax)
    refolded = fold_init(fused[1:])
    refolded[0].mask |= fused[0].mask
    return refolded[:1] + merge_diagonals(refolded[1:])
