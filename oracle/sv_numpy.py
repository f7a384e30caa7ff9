"""Gate-level fp64 statevector simulator in numpy.  TEST INFRASTRUCTURE ONLY.

Plays the role Qiskit Aer's CPU statevector plays behind
/root/reference/run_experiment.py:54-57 -- Aer is a third-party dependency that is not
vendored in the reference and not installed here, so its published algorithm is restated:
amplitude vector of 2^W complex128, qubit q = bit q of the index (little endian), each
gate a sweep over the vector.  Written with boolean masks over ``arange(2^W)`` on purpose:
the index arithmetic shares nothing with the HIP kernels it checks.

Gate matrices follow the Qiskit circuit-library definitions (h, x, sx, rz, p, cp, u, ...).
"""
from __future__ import annotations

import numpy as np

SQ2 = 1.0 / np.sqrt(2.0)
MATS = {
    "h": np.array([[SQ2, SQ2], [SQ2, -SQ2]], dtype=np.complex128),
    "x": np.array([[0, 1], [1, 0]], dtype=np.complex128),
    "y": np.array([[0, -1j], [1j, 0]], dtype=np.complex128),
    "z": np.array([[1, 0], [0, -1]], dtype=np.complex128),
    "s": np.array([[1, 0], [0, 1j]], dtype=np.complex128),
    "sdg": np.array([[1, 0], [0, -1j]], dtype=np.complex128),
    "t": np.array([[1, 0], [0, np.exp(0.25j * np.pi)]], dtype=np.complex128),
    "tdg": np.array([[1, 0], [0, np.exp(-0.25j * np.pi)]], dtype=np.complex128),
    "sx": 0.5 * np.array([[1 + 1j, 1 - 1j], [1 - 1j, 1 + 1j]], dtype=np.complex128),
    "sxdg": 0.5 * np.array([[1 - 1j, 1 + 1j], [1 + 1j, 1 - 1j]], dtype=np.complex128),
    "id": np.eye(2, dtype=np.complex128),
}


def rz(lam):
    return np.array([[np.exp(-0.5j * lam), 0], [0, np.exp(0.5j * lam)]], dtype=np.complex128)


def rx(th):
    c, s = np.cos(th / 2), np.sin(th / 2)
    return np.array([[c, -1j * s], [-1j * s, c]], dtype=np.complex128)


def ry(th):
    c, s = np.cos(th / 2), np.sin(th / 2)
    return np.array([[c, -s], [s, c]], dtype=np.complex128)


def phase(lam):
    return np.array([[1, 0], [0, np.exp(1j * lam)]], dtype=np.complex128)


def u3(th, ph, lam):
    c, s = np.cos(th / 2), np.sin(th / 2)
    return np.array([[c, -np.exp(1j * lam) * s],
                     [np.exp(1j * ph) * s, np.exp(1j * (ph + lam)) * c]], dtype=np.complex128)


_IDX = {}


def _idx(nq):
    if nq not in _IDX:
        _IDX.clear()
        _IDX[nq] = np.arange(2 ** nq, dtype=np.int64)
    return _IDX[nq]


def nqubits(state):
    return int(state.size).bit_length() - 1


def zero_state(nq):
    s = np.zeros(2 ** nq, dtype=np.complex128)
    s[0] = 1.0
    return s


def _ctrl_mask(nq, ctrls, vals):
    idx = _idx(nq)
    ok = np.ones(idx.shape, dtype=bool)
    for c, v in zip(ctrls, vals):
        ok &= ((idx >> c) & 1) == v
    return ok


def apply_1q(state, t, M, ctrls=(), ctrl_vals=None):
    """(Multi-)controlled 2x2 on qubit t; control c fires when its bit equals ctrl_vals[c]."""
    nq = nqubits(state)
    vals = [1] * len(ctrls) if ctrl_vals is None else ctrl_vals
    idx = _idx(nq)
    sel = _ctrl_mask(nq, ctrls, vals) & (((idx >> t) & 1) == 0)
    i0 = idx[sel]
    i1 = i0 | (1 << t)
    a0, a1 = state[i0], state[i1]
    state[i0] = M[0, 0] * a0 + M[0, 1] * a1
    state[i1] = M[1, 0] * a0 + M[1, 1] * a1
    return state


def apply_mcx(state, ctrls, t, ctrl_vals=None):
    return apply_1q(state, t, MATS["x"], ctrls, ctrl_vals)


def apply_mcphase(state, qubits, lam, vals=None):
    """e^{i lam} on the subspace where every listed qubit matches vals (default all ones)."""
    nq = nqubits(state)
    vals = [1] * len(qubits) if vals is None else vals
    state[_ctrl_mask(nq, qubits, vals)] *= np.exp(1j * lam)
    return state


def apply_diag(state, qubits, table):
    """table[j], j = sum_b bit(qubits[b]) << b."""
    idx = _idx(nqubits(state))
    j = np.zeros_like(idx)
    for b, q in enumerate(qubits):
        j |= ((idx >> q) & 1) << b
    state *= np.asarray(table, dtype=np.complex128)[j]
    return state


def apply_mux(state, ctrls, t, mats):
    """Uniformly controlled 2x2: mats[j] applied to qubit t where j = control bits (ctrls[0] = LSB)."""
    idx = _idx(nqubits(state))
    mats = np.asarray(mats, dtype=np.complex128).reshape(-1, 2, 2)
    i0 = idx[((idx >> t) & 1) == 0]
    i1 = i0 | (1 << t)
    j = np.zeros_like(i0)
    for b, q in enumerate(ctrls):
        j |= ((i0 >> q) & 1) << b
    a0, a1 = state[i0], state[i1]
    state[i0] = mats[j, 0, 0] * a0 + mats[j, 0, 1] * a1
    state[i1] = mats[j, 1, 0] * a0 + mats[j, 1, 1] * a1
    return state


def apply_kq(state, qubits, U):
    """Dense 2^k x 2^k unitary; row/column index bit b <-> qubits[b]."""
    nq = nqubits(state)
    k = len(qubits)
    idx = _idx(nq)
    qmask = 0
    for q in qubits:
        qmask |= 1 << q
    base = idx[(idx & qmask) == 0]
    offs = np.zeros(2 ** k, dtype=np.int64)
    for j in range(2 ** k):
        for b, q in enumerate(qubits):
            if (j >> b) & 1:
                offs[j] |= 1 << q
    gathered = state[base[None, :] | offs[:, None]]            # (2^k, 2^(nq-k))
    state[base[None, :] | offs[:, None]] = np.asarray(U, dtype=np.complex128) @ gathered
    return state


def run_stream(ops, nq, state=None):
    """Execute oracle.gate_stream primitives (plus a few extras used by tests)."""
    s = zero_state(nq) if state is None else state
    for op in ops:
        k = op[0]
        if k in MATS:
            apply_1q(s, op[1], MATS[k])
        elif k == "mcx":
            apply_mcx(s, list(op[1]), op[2])
        elif k == "cx":
            apply_mcx(s, [op[1]], op[2])
        elif k == "cp":
            apply_mcphase(s, [op[2], op[3]], op[1])
        elif k == "rz":
            apply_1q(s, op[2], rz(op[1]))
        elif k == "p":
            apply_1q(s, op[2], phase(op[1]))
        elif k in ("measure", "barrier"):
            pass
        else:
            raise ValueError("oracle: unknown primitive %r" % (k,))
    return s


def probabilities(state):
    return state.real ** 2 + state.imag ** 2


def dense_unitary_1q(nq, t, M, ctrls=(), ctrl_vals=None):
    """Kronecker-style dense matrix of a controlled 1q gate (W <= ~10), for kernel checks."""
    dim = 2 ** nq
    U = np.zeros((dim, dim), dtype=np.complex128)
    for col in range(dim):
        e = np.zeros(dim, dtype=np.complex128)
        e[col] = 1
        U[:, col] = apply_1q(e, t, M, ctrls, ctrl_vals)
    return U
