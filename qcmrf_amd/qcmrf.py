"""``QCMRF(cliques, theta, ...)`` -- the reference's circuit model behind the same constructor.

Mirrors the public surface of /root/reference/QCMRF.py:13-157,199-284 (constructor signature,
properties, ValueError texts, gate order, default-theta RNG draws, post-processing helpers) so
that ``run_experiment.py`` style scripts run unchanged.  When Qiskit is importable the class
derives from the real ``qiskit.QuantumCircuit`` (and uses the real ``AND``); otherwise from the
container in ``qcmrf_amd.circuit``.  Either way the object is what the engine's ingest reads.

``sufficient_statistic`` / ``Hamiltonian`` (QCMRF.py:159-197) exist here as the diagonals they are
(``hamiltonian_diagonal``, ``sufficient_statistic_diagonal``: opflow was removed from Qiskit), with
their expectation values evaluated on the device (``expectation_hamiltonian``,
``expectation_sufficient_statistic`` -> libqsv ``qsv_expect_diag``).
"""
from __future__ import annotations

import itertools

import numpy as np

try:                                             # pragma: no cover - Qiskit absent in this image
    from qiskit import QuantumCircuit
    from qiskit.circuit.library import AND
    HAVE_QISKIT = True
except Exception:                                # ModuleNotFoundError here
    from .circuit import QuantumCircuit, AND
    HAVE_QISKIT = False

DEFAULT_BASIS = ['cx', 'id', 'rz', 'sx', 'x']


class QCMRF(QuantumCircuit):
    """Quantum circuit Markov random field (Gibbs-state preparation by real-part extraction)."""

    def __init__(self, cliques=None, theta=None, gamma=None, beta: float = 1, name: str = "QCMRF",
                 with_measurements=True, with_barriers=False, basis_gates=DEFAULT_BASIS):
        self._cliques, self._theta, self._gamma = cliques, theta, gamma
        self._beta, self._name = beta, name
        self._with_measurements, self._with_barriers = with_measurements, with_barriers
        self.basis_gates = basis_gates

        # same acceptance test and message as QCMRF.py:45-48 (only the first element is inspected)
        ok = type(cliques) == list and type(cliques[0]) == list and type(cliques[0][0]) == int
        if not ok:
            raise ValueError("The set of clique is not set properly. Type must be list of list of int.")

        self._num_cliques = len(cliques)
        self._n = max(cliques[0][0], max(v for C in cliques for v in C)) + 1      # QCMRF.py:52-57
        sizes = [len(C) for C in cliques]
        self._dim = sum(2 ** k for k in sizes)                                     # QCMRF.py:59-65
        self._c_max = max([0] + sizes)

        if theta is not None and len(theta) != self._dim:                          # QCMRF.py:68-71
            raise ValueError("The parameter vector has an incorrect dimension. Expected: " + str(self._dim))
        if gamma is not None and len(gamma) != self._dim:                          # QCMRF.py:73-76
            raise ValueError("The QCMRF parameter vector has an incorrect dimension. Expected: " + str(self._dim))

        width = self._n + self._num_cliques + 1                                    # QCMRF.py:78
        super().__init__(width, width, name=name)
        self._build()

    # ---- read-only model facts (QCMRF.py:82-127) ------------------------------------------
    @property
    def dimension(self):
        """Number of parameters, sum over cliques of 2^|C|."""
        return self._dim

    @property
    def cliques(self):
        return self._cliques

    @property
    def num_vertices(self):
        return self._n

    num_nodes = num_vertices

    @property
    def num_cliques(self):
        return self._num_cliques

    @property
    def max_clique(self):
        return self._c_max

    # ---- parameters: lazily converted both ways (QCMRF.py:129-157) --------------------------
    @property
    def theta(self):
        if self._theta is None:
            self._theta = [2 * np.log(np.cos(2 * g)) / self._beta for g in self._gamma]
        return self._theta

    @property
    def gamma(self):
        if self._gamma is None:
            self._gamma = [0.5 * np.arccos(np.exp(self._beta * 0.5 * w)) for w in self._theta]
        return self._gamma

    def hamiltonian_diagonal(self):
        """H = -sum theta_{C,y} Phi_{C,y} is diagonal: H[x] = -sum_C theta_{C,x_C}; index has
        variable v on bit n-1-v (QCMRF.py:159-193 restated without opflow)."""
        n = self._n
        idx = np.arange(2 ** n)
        H = np.zeros(2 ** n)
        off = 0
        th = np.asarray(self.theta, dtype=np.float64)
        for C in self._cliques:
            y = np.zeros_like(idx)
            for j, v in enumerate(C):
                y |= ((idx >> (n - 1 - v)) & 1) << (len(C) - 1 - j)
            H -= th[off + y]
            off += 2 ** len(C)
        return H

    def sufficient_statistic_diagonal(self, C, y):
        """Phi_{C,y} = prod_{v in C} |y_v><y_v| is a diagonal projector: returns its 0/1 diagonal
        over the 2^n variable states (QCMRF.py:159-179 restated without opflow; same index
        convention as ``hamiltonian_diagonal``)."""
        n = self._n
        idx = np.arange(2 ** n)
        d = np.ones(2 ** n)
        for v, b in zip(C, y):
            d *= (((idx >> (n - 1 - v)) & 1) == int(bool(b)))
        return d

    def expectation_hamiltonian(self, backend, post_selected=True, **run_options):
        """<H> of ``Hamiltonian()`` (QCMRF.py:181-193) in the state this circuit prepares, evaluated on
        the device: the circuit is run on ``backend`` (no shots) and the diagonal H is averaged over
        the resident amplitudes in one read pass.  post_selected=True: in the state conditioned on
        every ancilla (and the AND scratch qubit) reading 0 -- the Gibbs state the construction
        targets.  Returns (<H>, probability of that condition)."""
        backend.run(self, shots=0, **run_options)
        n, W = self._n, self._n + self._num_cliques + 1
        fixed = {q: 0 for q in range(n, W)} if post_selected else None
        s0, s1 = backend.expectation_diagonal(self.hamiltonian_diagonal(), list(range(n)), fixed)
        return s0 / s1, s1

    def expectation_sufficient_statistic(self, backend, C, y, post_selected=True, **run_options):
        """<Phi_{C,y}> of ``sufficient_statistic(C, y)`` (QCMRF.py:159-179): the probability that the
        variables of C read y, same conventions as ``expectation_hamiltonian``"""
        backend.run(self, shots=0, **run_options)
        n, W = self._n, self._n + self._num_cliques + 1
        fixed = {q: 0 for q in range(n, W)} if post_selected else None
        # Phi only looks at the clique's own qubits: variable v lives on qubit n-1-v (QCMRF.py:219)
        qs = [n - 1 - v for v in C]
        tab = np.zeros(2 ** len(qs))
        tab[sum(int(bool(b)) << e for e, b in enumerate(y))] = 1.0
        s0, s1 = backend.expectation_diagonal(tab, qs, fixed)
        return s0 / s1, s1

    # ---- circuit construction (QCMRF.py:199-243) ---------------------------------------------
    def _clique_unitary(self, index, C, first_param):
        """cU_C: for every clique state y, AND . cp(2 gamma_y) . AND on (variables, scratch, ancilla)."""
        n = self._n
        sub = QuantumCircuit(n + 2, name='cU_C' + str(index))
        wires = [(n - 1) - v for v in C] + [n]
        for off, y in enumerate(itertools.product([0, 1], repeat=len(C))):
            g = self.gamma[first_param + off]
            if np.isclose(g, 0):
                continue
            flags = (np.array(y) * 2 - 1).tolist()
            # a fresh AND per append, compute and uncompute alike, exactly as QCMRF.py:225,227: the engine's
            # ingest recognises the blocks by content, however many distinct objects spell them
            sub.append(AND(len(C), flags), wires)
            sub.cp(2 * g, n, n + 1)
            sub.append(AND(len(C), flags), wires)
        return sub

    def _build(self):
        n = self._n
        for q in range(n):
            self.h(q)
        if self._with_barriers:
            self.barrier()

        if self._theta is None and self._gamma is None:
            # one scalar draw per parameter from the global numpy RNG, as QCMRF.py:210-213
            self._theta = [np.random.uniform(low=-5.0, high=0) for _ in range(self._dim)]

        main = list(range(n + 1))
        first = 0
        for ii, C in enumerate(self._cliques):
            anc = n + 1 + ii
            cu = self._clique_unitary(ii, C, first)
            first += 2 ** len(C)
            # real-part extraction: H cU X cU^dagger X H on the clique's ancilla
            self.h(anc)
            self.append(cu, main + [anc])
            self.x([anc])
            self.append(cu.inverse(), main + [anc])
            self.x([anc])
            self.h(anc)
            if self._with_measurements:
                self.measure(anc, anc)       # success <=> ancilla reads 0
            if self._with_barriers:
                self.barrier()
        if self._with_measurements:
            self.measure(range(n), range(n))


# ---- post-processing (QCMRF.py:247-284) ---------------------------------------------------------

def fidelity(P, Q):
    """Squared Bhattacharyya coefficient of two pmfs (entries that are <= 0 in either are skipped)."""
    P, Q = np.asarray(P, dtype=np.float64), np.asarray(Q, dtype=np.float64)
    both = (P > 0) & (Q > 0)
    return float(np.sum(np.sqrt(P[both] * Q[both]))) ** 2


def KL(P, Q):
    """Kullback-Leibler divergence sum P log(P/Q) over the common support."""
    P, Q = np.asarray(P, dtype=np.float64), np.asarray(Q, dtype=np.float64)
    both = (P > 0) & (Q > 0)
    return float(np.sum(P[both] * np.log(P[both] / Q[both])))


def extract_probs(R, n, a):
    """Counts dict -> (conditional pmf over the n variable bits given all ``a`` high bits are
    '0', success rate).  Key layout: ``'0'*a + x_0 x_1 ... x_{n-1}``."""
    P = np.zeros(2 ** n)
    head = '0' * a
    for i in range(2 ** n):
        key = head + format(i, '0{}b'.format(n)) if n > 0 else head
        if key in R:
            P[i] += R[key]
    z = np.sum(P)
    z0 = 0
    for key in R:
        z0 += R[key]
    if z == 0:
        return P, 0
    return P / z, z / z0
