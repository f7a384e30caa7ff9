"""The reference's gate stream, restated as a flat primitive list.  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/QCMRF.py:199-243 statement by statement.  The composite
instructions the reference appends are expanded with the documented semantics of the
Qiskit circuit library (Qiskit itself is not installed here):

* ``AND(k, flags)`` on ``var = ctrl_1..ctrl_k, result`` -- X on every variable whose flag
  is negative, a k-controlled X onto ``result``, the same X gates again
  (QCMRF.py:224-225,227).
* ``cp(lam, c, t)`` -- diag(1, 1, 1, e^{i lam}) (QCMRF.py:226).
* ``circuit.inverse()`` -- reversed order, every gate inverted (QCMRF.py:234): ``AND`` is
  self-inverse, ``cp(lam)`` -> ``cp(-lam)``.

Primitive tuples:  ("h", q)  ("x", q)  ("mcx", (ctrl, ...), target)
                   ("cp", lam, ctrl, target)  ("measure", qubit, clbit)  ("barrier",)
"""
from __future__ import annotations

import itertools
import numpy as np

from .closed_form import model_shape, gamma_of_theta


def _and_block(var, flags):
    ctrls, result = var[:-1], var[-1]
    flips = [("x", q) for q, f in zip(ctrls, flags) if f < 0]
    return flips + [("mcx", tuple(ctrls), result)] + flips


def _cu_block(C, gammas, n, anc):
    """One ``cU_C`` sub-circuit (QCMRF.py:218-228), already mapped onto global qubits."""
    var = [(n - 1) - v for v in C] + [n]
    ops = []
    for g, y in zip(gammas, itertools.product([0, 1], repeat=len(C))):
        if not np.isclose(g, 0):
            flags = [2 * b - 1 for b in y]
            ops += _and_block(var, flags)
            ops.append(("cp", 2 * g, n, anc))
            ops += _and_block(var, flags)
    return ops


def _inverse(ops):
    out = []
    for op in reversed(ops):
        if op[0] == "cp":
            out.append(("cp", -op[1], op[2], op[3]))
        else:                      # h, x, mcx are self-inverse
            out.append(op)
    return out


def reference_stream(cliques, theta=None, gamma=None, beta=1.0,
                     with_measurements=True, with_barriers=False):
    """Flat primitive list in exactly the order QCMRF._build emits (QCMRF.py:204-243)."""
    n, m, W, dim = model_shape(cliques)
    if gamma is None:
        gamma = gamma_of_theta(theta, beta)
    ops = [("h", q) for q in range(n)]
    if with_barriers:
        ops.append(("barrier",))
    i = 0
    for ii, C in enumerate(cliques):
        anc = n + 1 + ii
        cu = _cu_block(C, gamma[i:i + 2 ** len(C)], n, anc)
        i += 2 ** len(C)
        ops.append(("h", anc))
        ops += cu
        ops.append(("x", anc))
        ops += _inverse(cu)
        ops.append(("x", anc))
        ops.append(("h", anc))
        if with_measurements:
            ops.append(("measure", anc, anc))
        if with_barriers:
            ops.append(("barrier",))
    if with_measurements:
        ops += [("measure", q, q) for q in range(n)]
    return ops


def grid_cliques(rows, cols, drop_last=0):
    """Pairwise cliques of a rows x cols grid: horizontal edges row-major, then vertical
    edges column-major; ``drop_last`` edges removed from the end."""
    vid = lambda r, c: r * cols + c
    E = [[vid(r, c), vid(r, c + 1)] for r in range(rows) for c in range(cols - 1)]
    E += [[vid(r, c), vid(r + 1, c)] for c in range(cols) for r in range(rows - 1)]
    return E[:len(E) - drop_last] if drop_last else E


def chain_cliques(n):
    return [[i, i + 1] for i in range(n - 1)]
