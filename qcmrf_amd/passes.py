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
_H = ir.FIXED_1Q["h"]


def fold_init(ops):
    """[init(mask)] + remaining ops.  A plain H on a qubit nothing has touched yet is folded into
    the initial state -- but only if the qubit is never a dense target afterwards (it then stays
    a pure select in uniform superposition, like the MRF variable qubits, QCMRF.py:204-205).
    An ancilla's opening H (QCMRF.py:231) is NOT folded: it fuses into the ancilla's own
    multiplexer anyway, and keeping the ancilla in |0> until then is what lets the engine skip
    the still-empty part of the vector (zero tracking)."""
    last_dense = {}
    for k, op in enumerate(ops):
        for q in op.dense_targets():
            last_dense[q] = k
    mask = 0
    touched = set()
    rest = []
    for k, op in enumerate(ops):
        if (op.kind == "u" and not op.ctrls and op.target not in touched and last_dense.get(op.target) == k
                and np.array_equal(op.mat, _H)):
            mask |= 1 << op.target          # commutes past every deferred op: none of them touches it
            touched.add(op.target)
            continue
        touched.update(op.support())
        rest.append(op)
    return [ir.op_init(mask)] + rest


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


def fuse_monomial(ops, kmax=10):
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
        return [ir.op_mux(self.c, self.t, self.mats)]


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
            while out and out[-1].kind in ("diag", "mcphase") and op.target in out[-1].qubits:
                trial = _MuxWindow(op.target)
                for o in [out[-1]] + back + [op]:
                    if not trial.can_take(o, smax):
                        trial = None
                        break
                    trial.take(o)
                if trial is None:
                    break
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
def optimise(ops, level=2, kmax=10, smax=8):
    """level 0: gate by gate as ingested (|0..0> init prepended).
    level 1: + init folding + diagonal (monomial) fusion.   level 2: + mux fusion."""
    if level <= 0:
        return [ir.op_init(0)] + list(ops)
    ops = fold_init(ops)
    head, body = ops[:1], ops[1:]
    body = fuse_monomial(body, kmax=kmax)
    if level >= 2:
        body = fuse_mux(body, smax=smax)
    return head + body
