"""Synthetic MRF structures of the BASELINE.json configs and the reference's
parameter law (theta = -halfnorm.rvs(scale), np.random.seed(1984): run_experiment.py:3,18,30)."""
from __future__ import annotations

import numpy as np

REFERENCE_GRAPHS = [[[0]], [[0, 1]], [[0, 1], [1, 2], [2, 3]], [[0, 1], [1, 2], [2, 3], [3, 4]],
                    [[0, 1, 2]], [[0, 1, 2], [2, 3, 4]], [[0, 1, 2, 3]]]      # run_experiment.py:20


def chain(n):
    return [[i, i + 1] for i in range(n - 1)]


def grid(rows, cols, drop_last=0):
    """pairwise cliques: horizontal edges row-major, then vertical edges column-major"""
    vid = lambda r, c: r * cols + c
    E = [[vid(r, c), vid(r, c + 1)] for r in range(rows) for c in range(cols - 1)]
    E += [[vid(r, c), vid(r + 1, c)] for c in range(cols) for r in range(rows - 1)]
    return E[:len(E) - drop_last] if drop_last else E


def random_graph(n, m, seed=1984):
    """m distinct edges of K_n, seeded; every vertex 0..n-1 is guaranteed to appear"""
    rs = np.random.RandomState(seed)
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    while True:
        pick = rs.choice(len(pairs), size=m, replace=False)
        E = [list(pairs[k]) for k in sorted(pick)]
        if len({v for e in E for v in e}) == n:
            return E


def width(cliques):
    return max(v for C in cliques for v in C) + 1 + len(cliques) + 1


def dimension(cliques):
    return sum(2 ** len(C) for C in cliques)


def theta_halfnorm(dim, scale=0.5, seed=1984):
    from scipy.stats import halfnorm
    rs = np.random.RandomState(seed)
    return (-halfnorm.rvs(loc=0, scale=scale, size=dim, random_state=rs)).tolist()


def baseline_config(i):
    """(name, cliques) of BASELINE.json configs[i], i = 1..4"""
    if i == 1:
        return "config2: 20-qubit chain MRF (n=10, m=9)", chain(10)
    if i == 2:
        return "config3: 28-qubit 2x6-grid MRF minus last edge (n=12, m=15)", grid(2, 6, drop_last=1)
    if i == 3:
        return "config4: 31-qubit random-graph MRF G(10,20) seed 1984 (n=10, m=20)", random_graph(10, 20)
    if i == 4:
        return "config5: 34-qubit 2x7-grid MRF (n=14, m=19)", grid(2, 7)
    raise ValueError("configs[1..4] only")


def for_width(W):
    """a pairwise grid MRF (>= 2 rows) whose circuit has exactly W qubits: the smallest full
    grid that is wide enough, minus trailing vertical edges (every vertex stays attached)"""
    best = None
    for r in range(2, 8):
        for c in range(r, 40):
            full = 3 * r * c - r - c + 1
            n_vertical = c * (r - 1)
            if full >= W and full - W <= n_vertical:
                if best is None or full < best[0]:
                    best = (full, r, c)
                break
    if best is None:
        raise ValueError("no grid for width %d" % W)
    full, r, c = best
    E = grid(r, c, drop_last=full - W)
    assert width(E) == W, (W, r, c, width(E))
    return E
