"""Gate-level intermediate representation the engine executes.

Qubits are LOGICAL until ``planner.plan`` maps them onto physical amplitude-index positions.
Little-endian throughout: qubit q is bit q of the basis index (Qiskit convention).

kinds
-----
init     mask                       |0..0> with H on every qubit of ``mask`` (written directly)
u        target, ctrls, vals, mat   (multi-controlled) dense 2x2
x        target, ctrls, vals        (multi-controlled) X: pure permutation
diag     qubits, table              table[j], j = sum_b bit(qubits[b]) << b
mcphase  qubits, vals, angle        e^{i angle} where every qubit matches its value
mux      ctrls, target, mats        uniformly controlled 2x2, mats[j] with j from ctrls (LSB first)
kq       qubits, mat                dense 2^k x 2^k, index bit b <-> qubits[b]
swap     a, b                       physical layout swap (planner output only)
"""
from __future__ import annotations

import cmath
import math

import numpy as np

SQ2 = 1.0 / np.sqrt(2.0)


def _m(rows):
    return np.array(rows, dtype=np.complex128)


FIXED_1Q = {
    "h": _m([[SQ2, SQ2], [SQ2, -SQ2]]),
    "y": _m([[0, -1j], [1j, 0]]),
    "sx": 0.5 * _m([[1 + 1j, 1 - 1j], [1 - 1j, 1 + 1j]]),
    "sxdg": 0.5 * _m([[1 - 1j, 1 + 1j], [1 + 1j, 1 - 1j]]),
}
FIXED_PHASE = {          # diag(1, e^{i lam})
    "z": np.pi, "s": np.pi / 2, "sdg": -np.pi / 2, "t": np.pi / 4, "tdg": -np.pi / 4,
}


def rx(th):
    c, s = np.cos(th / 2), np.sin(th / 2)
    return _m([[c, -1j * s], [-1j * s, c]])


def ry(th):
    c, s = np.cos(th / 2), np.sin(th / 2)
    return _m([[c, -s], [s, c]])


def u3(th, ph, lam):
    c, s = np.cos(th / 2), np.sin(th / 2)
    return _m([[c, -np.exp(1j * lam) * s], [np.exp(1j * ph) * s, np.exp(1j * (ph + lam)) * c]])


def snap(m, tol=4e-16):
    """Zero the real / imaginary parts of a fused matrix (stack) that are rounding residue of the
    products that built it (|part| <= tol x the largest entry of its matrix).  The exact product of
    e.g. H . diag . X . diag^dg . X . H has a real diagonal and an imaginary off-diagonal; the
    residues are ~1e-17 and dropping them is an improvement, and it lets the device kernel see the
    structure (half the flops)."""
    m = np.array(m, dtype=np.complex128)
    scale = np.abs(m).reshape(m.shape[:-2] + (-1,)).max(axis=-1)[..., None, None] if m.ndim >= 2 else np.abs(m).max()
    re, im = m.real.copy(), m.imag.copy()
    re[np.abs(re) <= tol * scale] = 0.0
    im[np.abs(im) <= tol * scale] = 0.0
    return re + 1j * im


def classify_1q(m00, m01, m10, m11):
    """A unitary 2x2, entries as Python numbers, as one of (angles only, D(t) = diag(1, e^{it})):
        ('D', g, a)            e^{ig} D(a)                         diagonal
        ('A', g, a)            X e^{ig} D(a)                       anti-diagonal
        ('h', g, al, be)       e^{ig} D(al) H D(be)                every entry of modulus 1/sqrt2
        ('G', g, th, b1, a2)   e^{ig} D(a2) H D(th) H D(b1)        anything else (H D(th) H = e^{i th/2} Rx(th))
    or None if it is not unitary to 1e-12.  Tolerances, not exact zeros: products of merged one-qubit runs carry
    rounding residue where an exact zero belongs."""
    a00, a01 = abs(m00), abs(m01)
    if a01 < 1e-15 and abs(m10) < 1e-15:
        if abs(a00 - 1.0) > 1e-12 or abs(abs(m11) - 1.0) > 1e-12:
            return None
        return ("D", cmath.phase(m00), cmath.phase(m11 / m00))
    if a00 < 1e-15 and abs(m11) < 1e-15:
        if abs(a01 - 1.0) > 1e-12 or abs(abs(m10) - 1.0) > 1e-12:
            return None
        return ("A", cmath.phase(m10), cmath.phase(m01 / m10))
    if abs(a00 - SQ2) < 1e-12 and abs(a01 - SQ2) < 1e-12:
        return ("h", cmath.phase(m00), cmath.phase(m10 / m00), cmath.phase(m01 / m00))
    if not 1e-12 < a00 < 1.0 - 1e-13:
        return None
    th = 2.0 * math.acos(a00)
    p = m00 / a00
    js = -1j * math.sin(0.5 * th)
    b1, a2 = cmath.phase(m01 / (p * js)), cmath.phase(m10 / (p * js))
    if abs(m11 - p * a00 * cmath.exp(1j * (a2 + b1))) > 1e-12:
        return None
    return ("G", cmath.phase(p) - 0.5 * th, th, b1, a2)


_H = None


def _mat_of_cls(c):
    k = c[0]
    g = cmath.exp(1j * c[1])
    if k == "D":
        return np.array([[g, 0.0], [0.0, g * cmath.exp(1j * c[2])]], dtype=np.complex128)
    if k == "A":
        return np.array([[0.0, g * cmath.exp(1j * c[2])], [g, 0.0]], dtype=np.complex128)
    if k == "h":
        a, b = cmath.exp(1j * c[2]), cmath.exp(1j * c[3])
        return (g * SQ2) * np.array([[1.0, b], [a, -a * b]], dtype=np.complex128)
    th, b1, a2 = cmath.exp(1j * c[2]), cmath.exp(1j * c[3]), cmath.exp(1j * c[4])
    # D(a2) H D(th) H D(b1) = 1/2 [[1 + th, (1 - th) b1], [(1 - th) a2, (1 + th) a2 b1]]
    return (0.5 * g) * np.array([[1.0 + th, (1.0 - th) * b1], [(1.0 - th) * a2, (1.0 + th) * a2 * b1]], dtype=np.complex128)


class Op:
    __slots__ = ("kind", "target", "ctrls", "vals", "qubits", "_mat", "_table", "mats", "angle", "mask",
                 "a", "b", "label", "new_pass", "_sup", "cls")

    def __init__(self, kind, target=None, ctrls=(), vals=(), qubits=(), mat=None, table=None, mats=None,
                 angle=0.0, mask=0, a=(), b=(), label="", cls=None):
        self.kind = kind
        self.target = target
        self.ctrls, self.vals, self.qubits = ctrls, vals, qubits
        self._mat, self._table, self.mats = mat, table, mats
        self.angle = angle
        self.mask = mask
        self.a, self.b = a, b
        self.label = label
        self.new_pass = False        # planner hint: this gate opens a new multi-gate pass
        self._sup = None             # cached support(); whoever re-targets a copied op resets it
        # one-qubit gates met while walking a basis-gate circuit carry what kind of 2x2 they are as a few angles
        # (classify_1q); the arrays are built on first use
        self.cls = cls

    @property
    def mat(self):
        m = self._mat
        if m is None and self.cls is not None and self.kind == "u":
            m = self._mat = _mat_of_cls(self.cls)
        return m

    @mat.setter
    def mat(self, m):
        self._mat = m

    @property
    def table(self):
        t = self._table
        if t is None and self.cls is not None and self.kind == "diag":
            g = cmath.exp(1j * self.cls[1])
            t = self._table = np.array([g, g * cmath.exp(1j * self.cls[2])], dtype=np.complex128)
        return t

    @table.setter
    def table(self, t):
        self._table = t

    def support(self):
        """every logical qubit the op reads or writes (cached: the passes ask for it constantly)"""
        s = self._sup
        if s is None:
            k = self.kind
            if k in ("u", "x", "mux"):
                s = tuple(self.ctrls) + (self.target,)
            elif k in ("diag", "mcphase", "kq"):
                s = tuple(self.qubits)
            elif k == "swap":
                s = tuple(self.a) + tuple(self.b)
            else:
                s = ()
            self._sup = s
        return s

    def dense_targets(self):
        """qubits on which the op is NOT diagonal (these must be local to a shard)"""
        if self.kind in ("u", "x", "mux"):
            return (self.target,)
        if self.kind == "kq":
            return tuple(self.qubits)
        return ()

    def __repr__(self):
        if self.kind in ("u", "x"):
            return "%s(t=%s, c=%s%s)" % (self.kind if not self.label else self.label, self.target,
                                        list(self.ctrls), "" if all(self.vals) else " v=%s" % list(self.vals))
        if self.kind == "mux":
            return "mux(t=%s, c=%s)" % (self.target, list(self.ctrls))
        if self.kind == "diag":
            return "diag(%s)" % (list(self.qubits),)
        if self.kind == "mcphase":
            return "mcphase(%s, %.6g)" % (list(self.qubits), self.angle)
        if self.kind == "kq":
            return "kq(%s)" % (list(self.qubits),)
        if self.kind == "init":
            return "init(mask=%#x)" % self.mask
        if self.kind == "swap":
            return "swap(%s,%s)" % (list(self.a), list(self.b))
        return self.kind


def op_u(target, mat, ctrls=(), vals=None, label=""):
    vals = tuple(1 for _ in ctrls) if vals is None else tuple(int(v) for v in vals)
    return Op("u", target=int(target), mat=np.asarray(mat, dtype=np.complex128).reshape(2, 2),
              ctrls=tuple(int(c) for c in ctrls), vals=vals, label=label)


def op_x(target, ctrls=(), vals=None):
    vals = tuple(1 for _ in ctrls) if vals is None else tuple(int(v) for v in vals)
    return Op("x", target=int(target), ctrls=tuple(int(c) for c in ctrls), vals=vals)


def op_diag(qubits, table):
    return Op("diag", qubits=tuple(int(q) for q in qubits),
              table=np.asarray(table, dtype=np.complex128).ravel())


def op_phase1(qubit, lam):
    return op_diag([qubit], [1.0, np.exp(1j * lam)])


def op_mcphase(qubits, angle, vals=None):
    vals = tuple(1 for _ in qubits) if vals is None else tuple(int(v) for v in vals)
    return Op("mcphase", qubits=tuple(int(q) for q in qubits), vals=vals, angle=float(angle))


def op_mux(ctrls, target, mats):
    return Op("mux", ctrls=tuple(int(c) for c in ctrls), target=int(target),
              mats=np.asarray(mats, dtype=np.complex128).reshape(-1, 2, 2))


def op_kq(qubits, mat):
    k = len(qubits)
    return Op("kq", qubits=tuple(int(q) for q in qubits),
              mat=np.asarray(mat, dtype=np.complex128).reshape(2 ** k, 2 ** k))


def op_init(mask):
    return Op("init", mask=int(mask))
