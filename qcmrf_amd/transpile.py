"""Lower a circuit to the reference's basis ``['cx', 'id', 'rz', 'sx', 'x']``.

/root/reference/run_experiment.py:52 hands its circuits to ``qiskit.transpile(CIRCS,
basis_gates=['cx','id','rz','sx','x'])`` before simulating them.  Qiskit is not installed here,
so this module provides a small stand-in with the same call shape: textbook decompositions
(h -> rz sx rz; cp -> 2 cx + 3 rz; ccx -> 6 cx + 9 one-qubit gates; n-controlled X by the
Barenco recursion through controlled phases), exact including the global phase, which is
accumulated in ``circuit.global_phase``.  The gate sequence is NOT claimed to equal Qiskit's
transpiler output gate for gate -- only to be a circuit of that basis with the same unitary;
it is what the engine's dense-fusion path is exercised with.  With Qiskit installed, use
``qiskit.transpile``; the backend ingests either.
"""
from __future__ import annotations

import math

from .circuit import QuantumCircuit

BASIS = ['cx', 'id', 'rz', 'sx', 'x']
PI = math.pi


class _Emitter:
    def __init__(self, out):
        self.c = out

    # --- basis
    def rz(self, lam, q):
        self.c.rz(lam, q)

    def sx(self, q):
        self.c.sx(q)

    def x(self, q):
        self.c.x(q)

    def cx(self, a, b):
        self.c.cx(a, b)

    # --- one-qubit, exact with phase
    def p(self, lam, q):                       # p(lam) = e^{i lam/2} rz(lam)
        self.rz(lam, q)
        self.c.global_phase += lam / 2

    def h(self, q):                            # h = e^{i pi/4} rz(pi/2) sx rz(pi/2)
        self.rz(PI / 2, q)
        self.sx(q)
        self.rz(PI / 2, q)
        self.c.global_phase += PI / 4

    def u(self, th, ph, lam, q):               # u(th,ph,lam) = e^{i(ph+lam+pi)/2} rz(ph+pi) sx rz(th+pi) sx rz(lam)
        self.rz(lam, q)
        self.sx(q)
        self.rz(th + PI, q)
        self.sx(q)
        self.rz(ph + PI, q)
        self.c.global_phase += (ph + lam + PI) / 2

    # --- two-qubit and beyond
    def cp(self, lam, a, b):
        self.p(lam / 2, a)
        self.cx(a, b)
        self.p(-lam / 2, b)
        self.cx(a, b)
        self.p(lam / 2, b)

    def ccx(self, a, b, t):
        self.h(t)
        self.cx(b, t); self.p(-PI / 4, t)
        self.cx(a, t); self.p(PI / 4, t)
        self.cx(b, t); self.p(-PI / 4, t)
        self.cx(a, t); self.p(PI / 4, b); self.p(PI / 4, t)
        self.h(t)
        self.cx(a, b); self.p(PI / 4, a); self.p(-PI / 4, b)
        self.cx(a, b)

    def mcp(self, lam, ctrls, t):              # phase e^{i lam} iff all ctrls and t are 1
        if not ctrls:
            self.p(lam, t)
        elif len(ctrls) == 1:
            self.cp(lam, ctrls[0], t)
        else:
            last, rest = ctrls[-1], ctrls[:-1]
            self.cp(lam / 2, last, t)
            self.mcx(rest, last)
            self.cp(-lam / 2, last, t)
            self.mcx(rest, last)
            self.mcp(lam / 2, rest, t)

    def mcx(self, ctrls, t):
        if not ctrls:
            self.x(t)
        elif len(ctrls) == 1:
            self.cx(ctrls[0], t)
        elif len(ctrls) == 2:
            self.ccx(ctrls[0], ctrls[1], t)
        else:
            self.h(t)
            self.mcp(PI, list(ctrls), t)
            self.h(t)


_FIXED_PHASE = {"z": PI, "s": PI / 2, "sdg": -PI / 2, "t": PI / 4, "tdg": -PI / 4}


def _lower(circuit, qmap, cmap, em, out):
    out.global_phase += float(getattr(circuit, "global_phase", 0.0) or 0.0)
    for ci in circuit.data:
        op = ci.operation
        q = [qmap[circuit.find_bit(b).index] for b in ci.qubits]
        name, pr = op.name, [float(x) for x in op.params]
        if name == "measure":
            out.measure(q[0], cmap[circuit.find_bit(ci.clbits[0]).index])
        elif name == "barrier":
            out.barrier(*q)
        elif name == "id":
            out.id(q[0])
        elif name == "x":
            em.x(q[0])
        elif name == "sx":
            em.sx(q[0])
        elif name == "rz":
            em.rz(pr[0], q[0])
        elif name == "cx":
            em.cx(q[0], q[1])
        elif name == "h":
            em.h(q[0])
        elif name in _FIXED_PHASE:
            em.p(_FIXED_PHASE[name], q[0])
        elif name in ("p", "u1"):
            em.p(pr[0], q[0])
        elif name == "y":                                   # y = u(pi, pi/2, pi/2)
            em.u(PI, PI / 2, PI / 2, q[0])
        elif name == "sxdg":                                # sx^-1 = e^{-i pi/4} rx(-pi/2)
            em.u(-PI / 2, -PI / 2, PI / 2, q[0])
            out.global_phase -= PI / 4
        elif name == "rx":
            em.u(pr[0], -PI / 2, PI / 2, q[0])
        elif name == "ry":
            em.u(pr[0], 0.0, 0.0, q[0])
        elif name in ("u", "u3"):
            em.u(pr[0], pr[1], pr[2], q[0])
        elif name in ("cp", "cu1"):
            em.cp(pr[0], q[0], q[1])
        elif name == "cz":
            em.cp(PI, q[0], q[1])
        elif name == "swap":
            em.cx(q[0], q[1]); em.cx(q[1], q[0]); em.cx(q[0], q[1])
        elif name in ("ccx", "mcx", "mcx_gray", "c3x", "c4x"):
            em.mcx(q[:-1], q[-1])
        elif op.definition is not None:
            c = [cmap[circuit.find_bit(b).index] for b in ci.clbits]
            _lower(op.definition, q, c, em, out)
        else:
            raise ValueError("transpile: no rule for %r" % name)


def transpile(circuits, basis_gates=None, circuit_class=None):
    """``transpile(circuit | [circuits], basis_gates=['cx','id','rz','sx','x'])``; the result is built through
    the public gate methods (``rz sx x cx id measure barrier``, ``global_phase``) of ``circuit_class``
    (default: the in-tree container; any Qiskit-style QuantumCircuit class works)"""
    QC = circuit_class or QuantumCircuit
    if basis_gates is not None and sorted(basis_gates) != sorted(BASIS):
        raise ValueError("this stand-in lowers to %s only" % BASIS)
    single = not isinstance(circuits, (list, tuple))
    outs = []
    for c in ([circuits] if single else circuits):
        out = QC(c.num_qubits, c.num_clbits, name=getattr(c, "name", "circuit"))
        _lower(c, list(range(c.num_qubits)), list(range(c.num_clbits)), _Emitter(out), out)
        outs.append(out)
    return outs[0] if single else outs
