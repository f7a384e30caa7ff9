"""Symbolic re-assembly of circuits that were lowered to a {cx, rz, sx, x} basis.

/root/reference/run_experiment.py:52 transpiles every circuit to ``['cx','id','rz','sx','x']`` before it
reaches the simulator (run_experiment.py:56): a QCMRF circuit of 19 two-variable cliques arrives as
5.7-7.4 k basis gates.  Aer's fusion stage multiplies such gates into dense <= 5-qubit matrices; this
module instead recovers what the gates ARE, exactly, with integer arithmetic per gate and numpy only
per block:

* every gate but a Hadamard-like one-qubit gate is MONOMIAL (a permutation of basis states times a
  phase): a window of <= 8 variables tracks, per wire, the Boolean function of the window's input
  assignment that the wire currently carries -- a truth table held in ONE Python integer (256 bits),
  so ``cx`` is one XOR -- and the phase as a sum of ``angle x [truth table]`` terms in a dict, so
  ``rz`` is one dict update;
* a Hadamard-like gate  g D(a) H D(b)  on a wire OPENS a bracket: the wire now carries a fresh
  variable y and the term  pi [y and previous value]  joins the phase;
* the next Hadamard-like gate on that wire CLOSES it: summing y out of  H . (monomial) . H  gives
  either a monomial again -- the wire then carries  S(x) = [phase difference between y=1 and y=0 is pi]
  (how every lowered CCX / MCX returns, whatever one-qubit runs a transpiler merged into it) -- or,
  when the difference is a generic angle (the real-part-extraction sandwich H cU X cU^dg X H of
  QCMRF.py:231-236), one uniformly controlled 2x2 on that wire, which is emitted and the wire retired;
* closing arithmetic (the only per-bracket numpy) is memoised on the bracket's CONTENT -- its phase
  terms and truth tables in window coordinates; windows restart at every block boundary with wires
  numbered in order of appearance, so the 304 Toffoli networks of the 34-qubit circuit are 4 distinct
  contents.

Exact and state-independent (operator identities only).  Whatever does not fit -- a one-qubit gate
that is not Hadamard-like, a bracket that cannot close, more than 8 live variables -- is emitted as
the gates it came as, for ``passes.fuse_dense`` (the numeric <= 5-qubit windows) to look at.
"""
from __future__ import annotations

import cmath
import math
import threading

import numpy as np

from . import ir
from .ir import Op

PI = math.pi
TWO_PI = 2.0 * math.pi
_EPS = 1e-9
_CACHES = {}                            # K -> the caches keyed by truth tables of that width


def _configure(k):
    """window size: K variables (wires + open brackets), truth tables of N = 2^K bits.  8 holds every block of a
    pairwise MRF (two variables, scratch, ancilla, two brackets) with room to spare; the nested brackets of a lowered
    4-controlled X need 10"""
    global K, N, ALL, VAR, LOW, SH, _ARANGE, _BASE, _BITS, _BMAT, _CLOSE
    K = k
    N = 1 << K
    ALL = (1 << N) - 1
    VAR = []
    for i in range(K):
        period = 1 << (i + 1)
        block = ((1 << (1 << i)) - 1) << (1 << i)
        v = 0
        for j in range(0, N, period):
            v |= block << j
        VAR.append(v)                    # truth table of "variable i"
    LOW = [ALL ^ v for v in VAR]
    SH = [1 << i for i in range(K)]
    _ARANGE = np.arange(N)
    _BASE = [_ARANGE & ~(1 << i) for i in range(K)]
    _BITS, _BMAT, _CLOSE = _CACHES.setdefault(k, ({}, {}, {}))


_configure(8)



def _bits(T):
    """truth table (int) -> float64 0/1 vector over the N assignments (cached: windows are numbered
    canonically, so the same tables recur in every block)"""
    b = _BITS.get(T)
    if b is None:
        if len(_BITS) > 8192:
            _BITS.clear()
        b = np.unpackbits(np.frombuffer(T.to_bytes(N // 8, "little"), dtype=np.uint8), bitorder="little").astype(np.float64)
        b.setflags(write=False)
        _BITS[T] = b
    return b


def _to_int(boolvec):
    return int.from_bytes(np.packbits(boolvec.astype(np.uint8), bitorder="little").tobytes(), "little")


_C1, _C2 = TWO_PI, 2.4492935982947064e-16          # 2 pi = _C1 + _C2 to twice the precision of a double


def _mod2pi(x):
    """x mod 2 pi into (-pi, pi], reduced against 2 pi in two pieces: a global phase that has collected thousands of
    radians keeps its last bits"""
    k = round(x / _C1)
    return (x - k * _C1) - k * _C2


def _wrap(a):
    return a - TWO_PI * np.round(a / TWO_PI)


def _phase_vector(terms):
    """sum of angle x [table] over the N assignments"""
    if not terms:
        return np.zeros(N)
    key = tuple(t for t, _ in terms)
    B = _BMAT.get(key)
    if B is None:
        if len(_BMAT) > 1024:
            _BMAT.clear()
        B = _BMAT[key] = np.stack([_bits(t) for t in key])
    return np.array([a for _, a in terms]) @ B


_WHT = {}


def _parity_terms(psi):
    """a real function over the N assignments as  c0 + sum_S c_S [parity_S(x)]  over the variables it
    depends on: [(table of parity_S, c_S)], c0.  Terms in this form cancel syntactically against the
    phases later gates put on wires that carry parities."""
    sup = [v for v in range(K) if np.abs(_wrap(psi[_BASE[v]] - psi[_BASE[v] | SH[v]])).max() > 1e-14]
    k = len(sup)
    j = np.arange(1 << k)
    idx = np.zeros(1 << k, dtype=np.int64)
    for e, v in enumerate(sup):
        idx |= ((j >> e) & 1) << v
    vals = _wrap(psi[idx])
    H = _WHT.get(k)
    if H is None:
        H = np.ones((1, 1))
        for _ in range(k):
            H = np.block([[H, H], [H, -H]])
        _WHT[k] = H
    hat = H @ vals / (1 << k)
    terms = []
    for s in range(1, 1 << k):
        c = -2.0 * hat[s]
        c = c - TWO_PI * round(c / TWO_PI)
        if abs(c) > 1e-14:
            T = 0
            for e, v in enumerate(sup):
                if (s >> e) & 1:
                    T ^= VAR[v]
            terms.append((T, float(c)))
    return terms, float(vals[0])




def _close_monomial(dep, y, h):
    """sum_y of  exp(i Phi_y(x)) (-1)^(z (y xor h(x)))  for the phase terms ``dep`` (those that depend on
    the bracket variable y): if Phi_1 - Phi_0 is 0 or pi everywhere, exactly one z survives per x --
    returns (S, residual terms, constant) with z = S(x), else None.  Memoised on content."""
    key = (y, h, tuple(sorted(dep)))
    hit = _CLOSE.get(key, 0)
    if hit != 0:
        return hit
    phi = _phase_vector(dep)
    base = _BASE[y]
    phi0, phi1 = phi[base], phi[base | SH[y]]
    delta = _wrap(phi1 - phi0)
    is1 = np.abs(np.abs(delta) - PI) < _EPS
    if not np.all(is1 | (np.abs(delta) < _EPS)):
        res = None
    else:
        psi = phi0 + PI * (_bits(h)[base] * is1)
        terms, const = _parity_terms(psi)
        res = (_to_int(is1), terms, const)
    if len(_CLOSE) > 4096:
        _CLOSE.clear()
    _CLOSE[key] = res
    return res


# ---- compact gate records ----------------------------------------------------------------------------------------
# ingest's flat walk can hand over plain tuples instead of ir.Op objects (thousands of gates per circuit; an Op costs
# several times a tuple, and the window reads only these fields):
#     ('c', control, target)        cx               ('x', q)                     x
#     ('d', q, g, a)                e^{ig} D(a)      ('a', q, g, a)               X e^{ig} D(a)
#     ('h', q, g, al, be)           e^{ig} D(al) H D(be)
#     ('g', q, g, th, b1, a2)       e^{ig} D(a2) H D(th) H D(b1)
def rec_to_op(rec):
    """the ir.Op a record stands for (fallbacks, leading runs, raw emission)"""
    tag = rec[0]
    if tag == "c":
        return Op("x", target=rec[2], ctrls=(rec[1],), vals=(1,))
    if tag == "x":
        return Op("x", target=rec[1])
    if tag == "d":
        return Op("diag", qubits=(rec[1],), cls=("D", rec[2], rec[3]))
    if tag == "a":
        return Op("u", target=rec[1], cls=("A", rec[2], rec[3]), label="run")
    if tag == "h":
        return Op("u", target=rec[1], cls=("h", rec[2], rec[3], rec[4]), label="run")
    return Op("u", target=rec[1], cls=("G", rec[2], rec[3], rec[4], rec[5]), label="run")


def _support(item):
    if type(item) is tuple:
        return (item[1], item[2]) if item[0] == "c" else (item[1],)
    return item.support()


def _cls_of(op):
    """the angle form of a one-qubit gate (ir.classify_1q), cached on the op; ('?',) if it has none"""
    c = op.cls
    if c is None:
        k = op.kind
        if k == "u" and not op.ctrls:
            m = op.mat.tolist()
            c = ir.classify_1q(m[0][0], m[0][1], m[1][0], m[1][1])
        elif k == "diag" and len(op.qubits) == 1:
            t = op.table.tolist()
            c = ir.classify_1q(t[0], 0.0, 0.0, t[1])
        elif k == "mcphase" and len(op.qubits) == 1:
            c = ("D", 0.0, op.angle) if op.vals[0] else ("D", op.angle, -op.angle)
        op.cls = c = c if c is not None else ("?",)
    return c


OK, BOUNDARY, STUCK, BARRIER = 0, 1, 2, 3
_OVERFLOW = [False]                     # a window ran out of variables during the current run
_GIVE_UP = [False]                      # ... and the run is to stop right there (the wider windows take over)


class _Overflow(Exception):
    """a window of 8 variables ran out of room"""



class _Frame:
    """What has left the windows of one run and commutes with everything still to come, kept symbolic until the end:

    pool   the diagonal phases, as coefficients of AND-monomials over the wires' REFERENCE values,
           F(x) = sum_S a_S prod_{i in S} x_i  -- the one basis in which "equal up to multiples of 2 pi" is
           coefficient-wise (Moebius inversion has integer weights), so the residues the lowered Toffoli networks of
           neighbouring blocks leave on the variable wires cancel term by term instead of piling up as diagonals;
    flip   the wires that carry a pending X.

    A wire that a window releases is at rest -- it carries its own input, plain or complemented -- so by induction
    every wire outside a window carries (reference value) xor (flip), and a window variable IS the reference value.
    Multiplexers only read such wires (their targets retire), so pool and frame commute past them to wherever the
    run next has to emit a raw gate, or to its end."""
    __slots__ = ("pool", "flip")

    def __init__(self):
        self.pool, self.flip = {}, set()

    def snapshot(self):
        return dict(self.pool), set(self.flip)

    def restore(self, snap):
        self.pool, self.flip = dict(snap[0]), set(snap[1])

    def add_table(self, wires, angles):
        """angles[j], j = sum_e x_{wires[e]} << e  ->  pool coefficients (returns the constant term)"""
        k = len(wires)
        t = np.array(angles, dtype=np.float64)
        for e in range(k):
            v = t.reshape(-1, 2, 1 << e)
            v[:, 1, :] -= v[:, 0, :]
        t -= TWO_PI * np.round(t / TWO_PI)
        pool = self.pool
        for s in np.nonzero(np.abs(t) > 1e-13)[0].tolist():
            if s == 0:
                continue
            key = tuple(sorted([wires[e] for e in range(k) if (s >> e) & 1]))
            a = pool.get(key, 0.0) + float(t[s])
            a -= TWO_PI * round(a / TWO_PI)
            if abs(a) < 1e-13:
                pool.pop(key, None)
            else:
                pool[key] = a
        return float(t[0])

    def drain(self, out):
        for key in sorted(self.pool, key=lambda k: (len(k), k)):
            out.append(Op("mcphase", qubits=key, vals=(1,) * len(key), angle=self.pool[key]))
        for q in sorted(self.flip):
            out.append(Op("x", target=q))
        self.pool, self.flip = {}, set()


class _Window:
    __slots__ = ("ctx", "slot", "owner", "free", "tt", "ph", "moved", "br", "emitted", "retired", "gph", "nops", "block_done", "last")

    def __init__(self, ctx):
        self.ctx = ctx
        self.slot = {}                   # wire -> variable (= the wire's reference value, see _Frame)
        self.owner = {}                  # variable -> wire (wire variables only)
        self.free = list(range(K - 1, -1, -1))
        self.tt = {}                     # wire -> truth table of its current value
        self.ph = {}                     # truth table -> angle
        self.moved = set()               # wires whose value is not their own input variable
        self.br = {}                     # wire -> bracket variable
        self.emitted = []
        self.retired = set()
        self.gph = 0.0
        self.nops = 0
        self.block_done = False          # a multiplexer was emitted: flush at the next clean point
        self.last = {}                   # wire -> order of joining (eviction: longest in the window first)

    # ---- phase bookkeeping: tables are kept with assignment 0 unset ([not T] = 1 - [T]) --------
    def padd(self, T, a):
        if T & 1:
            T ^= ALL
            g = self.gph + a
            self.gph = g if -64.0 < g < 64.0 else _mod2pi(g)
            a = -a
        if T == 0:
            return
        ph = self.ph
        v = ph.get(T)
        if v is None:
            ph[T] = a
        else:
            v += a
            if abs(v - TWO_PI * round(v / TWO_PI)) < 1e-13:
                del ph[T]
            else:
                ph[T] = v

    def clean(self):
        return not self.moved and not self.br

    def join(self, q, held, keep=()):
        """wire q enters the window; the one-qubit gates that were waiting on it follow in order"""
        if not self.free and not self.evict(1, keep):
            return STUCK
        v = self.free.pop()
        self.slot[q] = v
        self.owner[v] = q
        self.tt[q] = VAR[v] ^ ALL if q in self.ctx.flip else VAR[v]
        self.nops += 1
        self.last[q] = self.nops
        for op in held.pop(q, ()):
            r = self.one_rec(op, q) if type(op) is tuple else self.one_qubit(op, q)
            if r != OK:
                return r
        return OK

    def touch(self, q):
        """moved = carries anything but its own input, plain or complemented (a pending X is no obstacle: it is
        emitted as the X it is when the window closes)"""
        d = self.tt[q] ^ VAR[self.slot[q]]
        if d == 0 or d == ALL:
            self.moved.discard(q)
        else:
            self.moved.add(q)

    def one_rec(self, rec, q):
        """a one-qubit record on a wire of the window"""
        tag = rec[0]
        if tag == "d":
            self.gph += rec[2]
            if rec[3]:
                self.padd(self.tt[q], rec[3])
            return OK
        if tag == "h":
            self.gph += rec[2]
            return self.hadamard_like(q, rec[3], rec[4])
        if tag == "x":
            self.tt[q] ^= ALL
            self.touch(q)
            return OK
        if tag == "a":
            self.gph += rec[2]
            if rec[3]:
                self.padd(self.tt[q], rec[3])
            self.tt[q] ^= ALL
            self.touch(q)
            return OK
        # 'g': two Hadamard-like gates in a row
        self.gph += rec[2]
        r = self.hadamard_like(q, rec[3], rec[4])
        if r != OK:
            return r
        if q not in self.tt:
            e = cmath.exp(1j * rec[5])
            self.emitted.append(ir.op_u(q, [[ir.SQ2, ir.SQ2], [ir.SQ2 * e, -ir.SQ2 * e]], label="run"))
            return OK
        return self.hadamard_like(q, rec[5], 0.0)

    def feed_rec(self, rec, held):
        """``feed`` for a compact record"""
        tag = rec[0]
        slot = self.slot
        if tag == "c":
            c, t = rec[1], rec[2]
            if c in slot and t in slot:
                tt = self.tt
                b = tt[t] ^ tt[c]
                tt[t] = b
                b ^= VAR[slot[t]]
                if b == 0 or b == ALL:
                    self.moved.discard(t)
                else:
                    self.moved.add(t)
                return OK
            wires = (c, t)
            new = [w for w in wires if w not in slot]
            if slot and self.clean():
                return BOUNDARY
            if any(w in self.retired for w in new):
                return STUCK
            if len(new) > len(self.free) and not self.evict(len(new), wires):
                return STUCK
            for w in new:
                if self.join(w, held, wires) != OK:
                    return STUCK
            if c not in slot or t not in slot:
                return STUCK
            tt = self.tt
            tt[t] ^= tt[c]
            self.touch(t)
            return OK
        q = rec[1]
        if q in slot:
            if tag == "d":                                   # (one_rec and padd, in line: the most frequent record after cx)
                a = rec[3]
                g = self.gph + rec[2]
                T = self.tt[q]
                if T & 1:
                    T ^= ALL
                    g += a
                    a = -a
                self.gph = g if -64.0 < g < 64.0 else _mod2pi(g)
                if T and a:
                    ph = self.ph
                    v = ph.get(T)
                    if v is None:
                        ph[T] = a
                    else:
                        v += a
                        if abs(v - TWO_PI * round(v / TWO_PI)) < 1e-13:
                            del ph[T]
                        else:
                            ph[T] = v
                return OK
            return self.one_rec(rec, q)
        if q in self.retired:
            return BOUNDARY if self.clean() else STUCK
        held.setdefault(q, []).append(rec)                  # commutes with the window: waits for its wire
        return OK

    def one_qubit(self, op, q):
        if op.kind == "x":
            self.tt[q] ^= ALL
            self.touch(q)
            return OK
        c = op.cls or _cls_of(op)
        tag = c[0]
        if tag == "D":
            self.gph += c[1]
            if c[2]:
                self.padd(self.tt[q], c[2])
            return OK
        if tag == "h":
            self.gph += c[1]
            return self.hadamard_like(q, c[2], c[3])
        if tag == "A":                                   # X times a phase gate
            self.gph += c[1]
            if c[2]:
                self.padd(self.tt[q], c[2])
            self.tt[q] ^= ALL
            self.touch(q)
            return OK
        if tag != "G":
            return BARRIER
        # two Hadamard-like gates in a row (what a ZSX re-synthesis writes as rz sx rz sx rz)
        self.gph += c[1]
        r = self.hadamard_like(q, c[2], c[3])
        if r != OK:
            return r
        if q not in self.tt:                             # the first half closed a bracket into a multiplexer and the wire left:
            e = cmath.exp(1j * c[4])                     # the second half follows it out as the gate it is
            self.emitted.append(ir.op_u(q, [[ir.SQ2, ir.SQ2], [ir.SQ2 * e, -ir.SQ2 * e]], label="run"))
            return OK
        return self.hadamard_like(q, c[4], 0.0)

    def hadamard_like(self, q, alpha, beta):
        """D(alpha) H D(beta) on wire q: opens a bracket, or closes the one that is open"""
        cur = self.tt[q]
        if beta:
            self.padd(cur, beta)
        y = self.br.get(q)
        if y is None:                                    # open: the wire carries a fresh variable
            if not self.free and not self.evict(1, (q,)):
                return STUCK
            y = self.free.pop()
            self.padd(VAR[y] & cur, PI)
            self.tt[q] = VAR[y]
            self.br[q] = y
            if alpha:
                self.padd(VAR[y], alpha)
            return OK
        return self.close(q, y, cur, alpha)

    def close(self, q, y, cur, alpha):
        sy, ly = SH[y], LOW[y]
        h = cur & ly
        h |= h << sy                                      # cofactor of the wire's value at y = 0
        if cur != VAR[y] ^ h:
            return STUCK
        for w, T in self.tt.items():
            if w != q and ((T >> sy) ^ T) & ly:
                return STUCK                              # another wire still depends on y
        ph = self.ph
        dep = [(T, a) for T, a in ph.items() if ((T >> sy) ^ T) & ly]
        res = _close_monomial(dep, y, h)
        if res is not None:
            S, terms, const = res
            for T, _ in dep:
                del ph[T]
            for T, a in terms:
                self.padd(T, a)
            self.gph += const
            del self.br[q]
            self.free.append(y)
            self.tt[q] = S
            if alpha:
                self.padd(S, alpha)
            self.touch(q)
            return OK
        return self.close_mux(q, y, h, alpha)

    def close_mux(self, q, y, h, alpha):
        """H . (monomial, generic angles) . H on wire q: one uniformly controlled 2x2 on q, selected by
        the window's input values of the other wires; q leaves the window"""
        vq = self.slot[q]                                 # (other brackets may be open: the selects found below must be wire inputs)
        sq, lq = SH[vq], LOW[vq]
        for w, T in self.tt.items():
            if w != q and ((T >> sq) ^ T) & lq:
                return STUCK
        sy, ly = SH[y], LOW[y]
        ph = self.ph
        mine = [(T, a) for T, a in ph.items() if (((T >> sy) ^ T) & ly) or (((T >> sq) ^ T) & lq)]
        # the selects: the other variables any of these terms (or h) reads -- they must be wire inputs
        tabs = [T for T, _ in mine]
        tabs.append(h)
        sel = []
        for v in range(K):
            if v == y or v == vq:
                continue
            sv, lv = SH[v], LOW[v]
            for T in tabs:
                if ((T >> sv) ^ T) & lv:
                    if v not in self.owner:
                        return STUCK
                    sel.append(v)
                    break
        j = np.arange(1 << len(sel))
        idx = np.zeros(j.shape, dtype=np.int64)
        for e, v in enumerate(sel):
            idx |= ((j >> e) & 1) << v
        phi = _phase_vector(mine)
        hb = _bits(h)
        # A[z, a_in](selects) = 1/2 sum_y exp(i phi(y, a_in, .)) (-1)^(z (y xor h))
        i00, i10 = idx, idx | SH[y]                      # (y, a_in) = (0, 0), (1, 0)
        i01, i11 = i00 | SH[vq], i10 | SH[vq]
        e = 0.5 * np.exp(1j * phi[np.stack([i00, i10, i01, i11])])       # rows: y0a0, y1a0, y0a1, y1a1
        hs = 1.0 - 2.0 * hb[np.stack([i00, i01])]         # (-1)^h at a_in = 0, 1 (h does not depend on y)
        mats = np.empty((idx.size, 2, 2), dtype=np.complex128)
        mats[:, 0, 0] = e[0] + e[1]
        mats[:, 0, 1] = e[2] + e[3]
        mats[:, 1, 0] = (e[0] - e[1]) * hs[0]
        mats[:, 1, 1] = (e[2] - e[3]) * hs[1]
        if alpha:
            mats[:, 1, :] *= cmath.exp(1j * alpha)
        keep = [e_ for e_, v in enumerate(sel)              # a listed variable may cancel out of the matrices
                if np.abs(mats[((j >> e_) & 1) == 0] - mats[((j >> e_) & 1) == 1]).max() > 1e-13]
        if len(keep) != len(sel):
            jj = np.arange(1 << len(keep))
            pick = np.zeros(jj.shape, dtype=np.int64)
            for e2, e_ in enumerate(keep):
                pick |= ((jj >> e2) & 1) << e_
            mats = mats[pick]
            sel = [sel[e_] for e_ in keep]
        for T, _ in mine:
            del ph[T]
        pool = self.ctx.pool                              # phases parked earlier that read this wire go out in front of its multiplexer
        for key in [k for k in pool if q in k]:
            self.emitted.append(Op("mcphase", qubits=key, vals=(1,) * len(key), angle=pool.pop(key)))
        if sel:
            self.emitted.append(ir.op_mux([self.owner[v] for v in sel], q, ir.snap(mats)))
        else:
            self.emitted.append(ir.op_u(q, mats[0], label="fused"))
        del self.br[q], self.tt[q], self.slot[q], self.owner[vq]
        self.ctx.flip.discard(q)
        self.moved.discard(q)
        self.free.append(y)
        self.free.append(vq)
        self.retired.add(q)
        self.block_done = True
        return OK

    def feed(self, op, held):
        k = op.kind
        slot = self.slot
        if k == "x":
            cs = op.ctrls
            if len(cs) == 1:                             # cx: most of a lowered circuit
                c, t = cs[0], op.target
                if c in slot and t in slot:
                    tt = self.tt
                    a = tt[c]
                    b = tt[t] ^ (a if op.vals[0] else a ^ ALL)
                    tt[t] = b
                    b ^= VAR[slot[t]]
                    if b == 0 or b == ALL:
                        self.moved.discard(t)
                    else:
                        self.moved.add(t)
                    return OK
                wires = (c, t)
            elif cs:
                wires = cs + (op.target,)
            else:
                wires = None
                q = op.target
        elif k == "u":
            if op.ctrls:
                return BARRIER
            wires = None
            q = op.target
        elif k == "diag" or k == "mcphase":
            wires = op.qubits
            if len(wires) == 1:
                q = wires[0]
                wires = None
                c = op.cls
                if c is not None and c[0] == "D" and q in slot:        # a phase gate on a wire of the window: rz, t, ...
                    a = c[2]
                    g = self.gph + c[1]
                    T = self.tt[q]
                    if T & 1:                            # (padd, in line)
                        T ^= ALL
                        g += a
                        a = -a
                    self.gph = g if -64.0 < g < 64.0 else _mod2pi(g)
                    if T:
                        ph = self.ph
                        v = ph.get(T)
                        if v is None:
                            ph[T] = a
                        else:
                            v += a
                            if abs(v - TWO_PI * round(v / TWO_PI)) < 1e-13:
                                del ph[T]
                            else:
                                ph[T] = v
                    return OK
            elif k == "diag":
                return BARRIER
        else:
            return BARRIER
        if wires is None:                                # a one-qubit gate
            if q in slot:
                return self.one_qubit(op, q)
            if q in self.retired:
                return BOUNDARY if self.clean() else STUCK
            if k != "x" and (op.cls or _cls_of(op))[0] == "?":
                return BARRIER
            held.setdefault(q, []).append(op)             # commutes with the window: waits for its wire
            return OK
        new = [w for w in wires if w not in slot]
        if new:
            if slot and self.clean():
                return BOUNDARY
            if any(w in self.retired for w in new):
                return STUCK
            if len(new) > len(self.free) and not self.evict(len(new), wires):
                return STUCK
            for w in new:
                if self.join(w, held, wires) != OK:
                    return STUCK                          # (the window is rebuilt from its last clean point)
            if any(w not in slot for w in wires):
                return STUCK                              # a waiting run closed its own bracket into a multiplexer: the wire left again
        tt = self.tt
        if k == "x":
            t = op.target
            f = ALL
            for c, v in zip(op.ctrls, op.vals):
                f &= tt[c] if v else tt[c] ^ ALL
            tt[t] ^= f
            self.touch(t)
        else:
            f = ALL
            for c, v in zip(op.qubits, op.vals):
                f &= tt[c] if v else tt[c] ^ ALL
            self.padd(f, op.angle)
        return OK

    def park(self, terms):
        """phase terms leave the window for the run's pool (they are a function of the wires' reference values);
        False if they still depend on an open bracket's variable"""
        phi = _phase_vector(terms)
        sup = [v for v in range(K) if np.abs(phi[_BASE[v]] - phi[_BASE[v] | SH[v]]).max() > 1e-13]
        if any(v not in self.owner for v in sup):
            return False
        j = np.arange(1 << len(sup))
        idx = np.zeros(j.shape, dtype=np.int64)
        for e, v in enumerate(sup):
            idx |= ((j >> e) & 1) << v
        self.gph += self.ctx.add_table([self.owner[v] for v in sup], phi[idx])
        return True

    def evict(self, need, keep):
        """make room: a wire at rest (carrying its own input, plain or complemented, that no other wire's value
        depends on) leaves the window -- the phase terms that involve it join the run's pool (they are a function of
        the wires' reference values, _Frame), the X it may carry joins the frame.  Longest in the window first."""
        for q in sorted((q for q in self.slot if q not in keep and q not in self.moved and q not in self.br),
                        key=lambda q: self.last.get(q, 0)):
            v = self.slot[q]
            sv, lv = SH[v], LOW[v]
            if any(w != q and ((T >> sv) ^ T) & lv for w, T in self.tt.items()):
                continue
            ph = self.ph
            terms = [(T, a) for T, a in ph.items() if ((T >> sv) ^ T) & lv]
            if terms:
                if not self.park(terms):
                    continue
                for T, _ in terms:
                    del ph[T]
            if self.tt[q] != VAR[v]:
                self.ctx.flip.add(q)
            else:
                self.ctx.flip.discard(q)
            del self.slot[q], self.tt[q], self.owner[v]
            self.free.append(v)
            if len(self.free) >= need:
                return True
        _OVERFLOW[0] = True
        if _GIVE_UP[0]:
            raise _Overflow()
        return False

    def flush(self, out):
        """a clean window ends: the multiplexers it emitted go out, its phase joins the pool, the X gates of the
        wires left complemented join the frame"""
        out.extend(self.emitted)
        if self.ph:
            ok = self.park(list(self.ph.items()))
            assert ok
        flip = self.ctx.flip
        for q, T in self.tt.items():
            if T != VAR[self.slot[q]]:
                flip.add(q)
            else:
                flip.discard(q)
        return self.gph


_LOCK = threading.Lock()


_WIDE = {}                                                # programs (by length and first gates) that needed windows of 10 variables


def unlower(ops):
    """see ``_run``; windows of 8 variables first; the moment one of them runs out of room the run is dropped and
    windows of 10 take over (what they cannot hold either goes out raw).  A program that needed the wide windows is
    remembered by its shape (length, opening gates): the next one like it -- the reference runs ten circuits per graph,
    `run_experiment.py:44-47` -- starts with them.  (Both widths are exact; this only decides which is tried first.)"""
    key = (len(ops), tuple((o if o[0] in "cx" else o[0]) if type(o) is tuple else o.kind for o in ops[:24]))
    with _LOCK:
        if not _WIDE.get(key):
            _OVERFLOW[0] = False
            _GIVE_UP[0] = True
            try:
                return _run(ops)
            except _Overflow:
                if len(_WIDE) > 256:
                    _WIDE.clear()
                _WIDE[key] = True
            finally:
                _GIVE_UP[0] = False
        _configure(10)
        try:
            return _run(ops)
        finally:
            _configure(8)


def _run(ops):
    """ops of a basis-gate circuit (one-qubit ``diag`` / ``x`` / ``u``, ``x`` with controls, ``mcphase``)
    -> (ops, n_raw): the same operator as multiplexers and diagonals wherever the gates compose to them;
    ``n_raw`` counts the multi-qubit gates that had to be passed on as they came."""
    out, held = [], {}
    n = len(ops)
    gph = 0.0
    n_raw = 0
    ctx = _Frame()

    def release(wires=None):
        for q in (sorted(held) if wires is None else wires):
            out.extend(rec_to_op(o) if type(o) is tuple else o for o in held.pop(q, ()))

    def mark():
        return {q: list(v) for q, v in held.items()}, ctx.snapshot()

    def rewind(m):
        held.clear()
        held.update({q: list(v) for q, v in m[0].items()})
        ctx.restore(m[1])

    def run(lo, hi):
        """ops[lo:hi] through a fresh window that is known to end clean"""
        nonlocal gph
        w = _Window(ctx)
        for op in ops[lo:hi]:
            r = w.feed_rec(op, held) if type(op) is tuple else w.feed(op, held)
            assert r == OK, r
        gph = _mod2pi(gph + w.flush(out))

    def raw(lo, hi):
        """gates that go out as they came: pool and frame first (they may be dense where those are not)"""
        nonlocal n_raw
        ctx.drain(out)
        for op in ops[lo:hi]:
            sup = _support(op)
            release(sup)
            if type(op) is tuple:
                op = rec_to_op(op)
            out.append(op)
            n_raw += len(sup) > 1 or op.kind == "u"

    recs = n > 0 and type(ops[0]) is tuple                # compact records (all of them) or ir.Op objects (all of them)
    i = w0 = last_clean = 0
    win = _Window(ctx)
    m0 = mark()
    while i <= n:
        r = OK
        if i < n:
            feed, moved, br = (win.feed_rec if recs else win.feed), win.moved, win.br   # the common case in a tight loop: gates the window takes
            while i < n:
                r = feed(ops[i], held)
                if r:
                    break
                i += 1
                if not moved and not br:
                    last_clean = i
                    if win.block_done:                    # a block just ended: restart, so the next one is numbered like it
                        gph = _mod2pi(gph + win.flush(out))
                        win, w0, m0 = _Window(ctx), i, mark()
                        break
        else:
            r = BOUNDARY if win.clean() else STUCK
        if r == OK:
            continue
        if r == BOUNDARY:
            gph = _mod2pi(gph + win.flush(out))
            if i < n and any(q in win.retired for q in _support(ops[i])):
                ctx.drain(out)                           # a wire that left through a multiplexer comes back: it is no reference value any more
            win, w0, last_clean, m0 = _Window(ctx), i, i, mark()
            if i == n:
                break
            continue
        # BARRIER / STUCK: back to the last clean point of this window, emit up to there, go on from there
        if not (r == BARRIER and win.clean()):
            rewind(m0)
            if last_clean > w0:
                run(w0, last_clean)
                i = w0 = last_clean
                win, m0 = _Window(ctx), mark()
                continue
            # no clean point: everything the window took in, and the gate it failed on, go out as they came
            hi = min(i + 1, n)
            raw(w0, hi)
            i = w0 = last_clean = hi
            win, m0 = _Window(ctx), mark()
            if i == n:
                break
            continue
        gph = _mod2pi(gph + win.flush(out))               # a gate the window cannot hold, met at a clean point
        raw(i, i + 1)
        i = w0 = last_clean = i + 1
        win, m0 = _Window(ctx), mark()
    ctx.drain(out)
    release()
    if abs(gph - TWO_PI * round(gph / TWO_PI)) > 1e-15:
        g = complex(math.cos(gph), math.sin(gph))
        out.append(ir.op_diag([0], [g, g]))
    return out, n_raw
