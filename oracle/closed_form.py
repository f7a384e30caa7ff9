"""Exact output state / distribution of a QCMRF circuit.  TEST INFRASTRUCTURE ONLY.

Derivation (each step cites the reference line it follows):

* register: ``n`` variable qubits, one scratch qubit ``n``, one ancilla per clique at
  qubit ``n+1+ii``; ``W = n+m+1`` qubits and as many classical bits
  (/root/reference/QCMRF.py:52-57,78,202,231).
* variable ``v`` sits on qubit ``n-1-v`` (QCMRF.py:219).
* clique state ``y`` runs over ``itertools.product([0,1], repeat=|C|)`` so the first
  clique member is the most significant bit of the parameter index (QCMRF.py:221,228).
* ``AND . cp(2 gamma) . AND`` multiplies ``|x>|0>_n|a>`` by ``exp(2 i gamma a [x_C = y])``
  (QCMRF.py:225-227); the sandwich ``H cU X cU^dagger X H`` on an ancilla in ``|0>``
  gives ``cos(2 gamma)|0> - i sin(2 gamma)|1>`` (QCMRF.py:231-236).
* ``gamma = 1/2 arccos(exp(beta theta / 2))`` (QCMRF.py:154), gates skipped when
  ``np.isclose(gamma, 0)`` (QCMRF.py:223).
* every measured qubit ``q`` goes to classical bit ``q``; bit ``n`` is never written
  (QCMRF.py:239,243).  Qiskit prints classical bit ``W-1`` leftmost.

Validated against the reference's committed Aer counts in tests/test_golden_aer.py.
"""
from __future__ import annotations

import itertools
import numpy as np


def model_shape(cliques):
    """(n, m, W, dim) exactly as QCMRF.py:50-65,78 computes them."""
    n = max(v for C in cliques for v in C) + 1
    m = len(cliques)
    dim = sum(2 ** len(C) for C in cliques)
    return n, m, n + m + 1, dim


def param_offsets(cliques):
    off, acc = [], 0
    for C in cliques:
        off.append(acc)
        acc += 2 ** len(C)
    return off


def gamma_of_theta(theta, beta=1.0):
    """QCMRF.py:154, element by element, in fp64."""
    return [0.5 * np.arccos(np.exp(beta * 0.5 * w)) for w in theta]


def _clique_factors(theta, beta):
    """Per-parameter (c, s): ancilla amplitude is c on |0>, -i*s on |1>."""
    g = np.asarray(gamma_of_theta(theta, beta), dtype=np.float64)
    c = np.cos(2.0 * g)
    s = np.sin(2.0 * g)
    skip = np.isclose(g, 0)          # QCMRF.py:223: block not emitted at all
    c = np.where(skip, 1.0, c)
    s = np.where(skip, 0.0, s)
    return c, s


def _yindex(idx, C, n):
    """Parameter sub-index of clique C for basis-state indices ``idx`` (QCMRF.py:219,221)."""
    k = len(C)
    y = np.zeros_like(idx)
    for j, v in enumerate(C):
        y |= ((idx >> np.uint64(n - 1 - v)) & np.uint64(1)) << np.uint64(k - 1 - j)
    return y


def amplitudes_at(cliques, theta, idx, beta=1.0):
    """Final-state amplitudes at the given basis indices (qubit q = bit q of the index)."""
    n, m, W, dim = model_shape(cliques)
    assert len(theta) == dim
    idx = np.asarray(idx, dtype=np.uint64)
    c, s = _clique_factors(theta, beta)
    amp = np.full(idx.shape, 2.0 ** (-n / 2.0), dtype=np.complex128)
    amp[((idx >> np.uint64(n)) & np.uint64(1)) == 1] = 0.0        # scratch qubit ends in |0>
    for ii, (C, off) in enumerate(zip(cliques, param_offsets(cliques))):
        y = _yindex(idx, C, n).astype(np.int64) + off
        a = (idx >> np.uint64(n + 1 + ii)) & np.uint64(1)
        amp *= np.where(a == 0, c[y].astype(np.complex128), -1j * s[y])
    return amp


def amplitudes(cliques, theta, beta=1.0):
    n, m, W, dim = model_shape(cliques)
    return amplitudes_at(cliques, theta, np.arange(2 ** W, dtype=np.uint64), beta)


def probabilities_at(cliques, theta, idx, beta=1.0):
    """P(outcome) with P = 2^-n prod_C [a_C = 0 ? e^{beta theta} : 1 - e^{beta theta}]."""
    n, m, W, dim = model_shape(cliques)
    idx = np.asarray(idx, dtype=np.uint64)
    e = np.exp(beta * np.asarray(theta, dtype=np.float64))
    g = np.asarray(gamma_of_theta(theta, beta))
    e = np.where(np.isclose(g, 0), 1.0, e)
    p = np.full(idx.shape, 2.0 ** (-n), dtype=np.float64)
    p[((idx >> np.uint64(n)) & np.uint64(1)) == 1] = 0.0
    for ii, (C, off) in enumerate(zip(cliques, param_offsets(cliques))):
        y = _yindex(idx, C, n).astype(np.int64) + off
        a = (idx >> np.uint64(n + 1 + ii)) & np.uint64(1)
        p *= np.where(a == 0, e[y], 1.0 - e[y])
    return p


def probabilities(cliques, theta, beta=1.0):
    n, m, W, dim = model_shape(cliques)
    return probabilities_at(cliques, theta, np.arange(2 ** W, dtype=np.uint64), beta)


def key_of(index, width):
    """Qiskit counts key: classical bit ``width-1`` leftmost, no spaces (single creg)."""
    return format(int(index), "0{}b".format(width))


def distribution(cliques, theta, beta=1.0, tol=0.0):
    """{key: probability} over the analytic support."""
    n, m, W, dim = model_shape(cliques)
    p = probabilities(cliques, theta, beta)
    return {key_of(i, W): float(p[i]) for i in np.nonzero(p > tol)[0]}


def gibbs_pmf(cliques, theta, beta=1.0):
    """Exact MRF pmf p(x) ~ exp(beta sum_C theta_{C,x_C}); index has x_0 as MSB
    (the ``int(key, 2)`` convention of /root/reference/eval.py:119-121)."""
    n, m, W, dim = model_shape(cliques)
    theta = np.asarray(theta, dtype=np.float64)
    p = np.zeros(2 ** n)
    offs = param_offsets(cliques)
    for xid, x in enumerate(itertools.product([0, 1], repeat=n)):
        e = 0.0
        for C, off in zip(cliques, offs):
            yi = 0
            for v in C:
                yi = (yi << 1) | x[v]
            e += theta[off + yi]
        p[xid] = np.exp(beta * e)
    Z = p.sum()
    return p / Z, Z


def success_probability(cliques, theta, beta=1.0):
    """delta = Z / 2^n: probability that every ancilla reads 0."""
    n = model_shape(cliques)[0]
    return gibbs_pmf(cliques, theta, beta)[1] / 2.0 ** n
