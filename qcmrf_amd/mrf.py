"""Exact (classical) quantities of the Markov random field behind a QCMRF circuit.

The reference's evaluation scripts obtain these from the closed-source ``kiopto_native``
(/root/reference/eval.py:84-93: ``px.backend``, ``px.weights``, ``px.infer(task='partition')``,
``px.logpot``).  For the binary MRFs used there they are elementary: the log-potential of a
joint state x is ``sum_C theta[offset(C) + index(x_C)]`` with the clique state read MSB-first
(the parameter order of /root/reference/QCMRF.py:221,228), so they are computed directly.
"""
from __future__ import annotations

import numpy as np


def num_vertices(cliques):
    return max(v for C in cliques for v in C) + 1


def dimension(cliques):
    """number of parameters = len(px.weights(backend)) of run_experiment.py:26-27"""
    return sum(2 ** len(C) for C in cliques)


def log_potentials(cliques, theta):
    """logpot[xid] for every joint state; xid = int(''.join(x_0 x_1 ... x_{n-1}), 2)
    (the key -> index convention of eval.py:100-101,119)."""
    n = num_vertices(cliques)
    theta = np.asarray(theta, dtype=np.float64)
    if theta.size != dimension(cliques):
        raise ValueError("theta has %d entries, the model has %d parameters" % (theta.size, dimension(cliques)))
    xid = np.arange(2 ** n)
    lp = np.zeros(2 ** n)
    off = 0
    for C in cliques:
        y = np.zeros_like(xid)
        for v in C:
            y = (y << 1) | ((xid >> (n - 1 - v)) & 1)
        lp += theta[off + y]
        off += 2 ** len(C)
    return lp


def gibbs_pmf(cliques, theta, beta=1.0):
    """(p, lnZ): p[xid] = exp(beta*logpot - lnZ)   (eval.py:88-93)"""
    lp = beta * log_potentials(cliques, theta)
    m = lp.max()
    lnZ = m + np.log(np.exp(lp - m).sum())
    return np.exp(lp - lnZ), lnZ


def success_probability(cliques, theta, beta=1.0):
    """probability that every real-part-extraction ancilla reads 0: Z / 2^n"""
    p, lnZ = gibbs_pmf(cliques, theta, beta)
    return float(np.exp(lnZ) / 2 ** num_vertices(cliques))
