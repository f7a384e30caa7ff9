"""A small, Qiskit-free circuit container with Qiskit's *data model*.

The reference builds its circuits with ``qiskit.QuantumCircuit`` and ``qiskit.circuit.library.AND``
(/root/reference/QCMRF.py:7,9,13).  Qiskit is not installed in this image, so ``QCMRF(...)`` needs
something to derive from.  This module provides exactly the surface QCMRF.py touches -- ``h``,
``x``, ``cp``, ``measure``, ``barrier``, ``append``, ``inverse``, nested instructions with a
``definition`` -- using the attribute names the engine's duck-typed ingest reads on real Qiskit
objects too (``circuit.data[i].operation.name/.params/.definition``, ``.qubits``, ``.clbits``,
``circuit.find_bit(bit).index``, ``circuit.global_phase``).  When Qiskit *is* importable,
``qcmrf_amd.qcmrf`` derives from the real class instead and this file is not used.

It is a container, not a simulator: no matrices, no state.
"""
from __future__ import annotations

from collections import namedtuple

BitLocations = namedtuple("BitLocations", ("index", "registers"))

# name -> (num_qubits, num_params, inverse name or None for "negate params")
_STANDARD = {
    "h": (1, 0, "h"), "x": (1, 0, "x"), "y": (1, 0, "y"), "z": (1, 0, "z"), "id": (1, 0, "id"),
    "s": (1, 0, "sdg"), "sdg": (1, 0, "s"), "t": (1, 0, "tdg"), "tdg": (1, 0, "t"),
    "sx": (1, 0, "sxdg"), "sxdg": (1, 0, "sx"),
    "rx": (1, 1, None), "ry": (1, 1, None), "rz": (1, 1, None), "p": (1, 1, None),
    "cx": (2, 0, "cx"), "cz": (2, 0, "cz"), "swap": (2, 0, "swap"), "cp": (2, 1, None),
    "crz": (2, 1, None), "ccx": (3, 0, "ccx"),
}


class Bit:
    __slots__ = ("_register", "_index")

    def __init__(self, register=None, index=None):
        self._register, self._index = register, index

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, self._index)


class Qubit(Bit):
    pass


class Clbit(Bit):
    pass


class Register:
    def __init__(self, size, name, bit_type):
        self.name, self.size = name, size
        self._bits = [bit_type(self, i) for i in range(size)]

    def __len__(self):
        return self.size

    def __iter__(self):
        return iter(self._bits)

    def __getitem__(self, i):
        return self._bits[i]


class Instruction:
    """Counterpart of qiskit.circuit.Instruction (name, params, definition, condition)."""

    def __init__(self, name, num_qubits, num_clbits=0, params=(), definition=None):
        self.name = name
        self.num_qubits = num_qubits
        self.num_clbits = num_clbits
        self.params = list(params)
        self.definition = definition
        self.condition = None

    def inverse(self):
        if self.definition is not None:
            name = self.name[:-3] if self.name.endswith("_dg") else self.name + "_dg"
            inv = self.definition.inverse()
            inv.name = name
            return Instruction(name, self.num_qubits, self.num_clbits, self.params, inv)
        if self.name.startswith("mcx"):
            return Instruction(self.name, self.num_qubits)
        if self.name in _STANDARD:
            nq, npar, inv = _STANDARD[self.name]
            if inv is None:
                return Instruction(self.name, nq, 0, [-p for p in self.params])
            if self.name == "u":
                th, ph, lam = self.params
                return Instruction("u", 1, 0, [-th, -lam, -ph])
            return Instruction(inv, nq)
        if self.name == "u":
            th, ph, lam = self.params
            return Instruction("u", 1, 0, [-th, -lam, -ph])
        raise ValueError("cannot invert instruction %r" % self.name)

    def copy(self):
        return Instruction(self.name, self.num_qubits, self.num_clbits, list(self.params), self.definition)

    def __repr__(self):
        return "Instruction(name=%r, num_qubits=%d, params=%r)" % (self.name, self.num_qubits, self.params)


class CircuitInstruction:
    """(operation, qubits, clbits) triple; also unpacks like the legacy tuple form."""
    __slots__ = ("operation", "qubits", "clbits")

    def __init__(self, operation, qubits=(), clbits=()):
        self.operation, self.qubits, self.clbits = operation, tuple(qubits), tuple(clbits)

    def __iter__(self):
        return iter((self.operation, list(self.qubits), list(self.clbits)))

    def __repr__(self):
        return "CircuitInstruction(%r, qubits=%r, clbits=%r)" % (self.operation, self.qubits, self.clbits)


class QuantumCircuit:
    """Counterpart of qiskit.QuantumCircuit for the calls QCMRF.py makes (QCMRF.py:78,205-243)."""

    def __init__(self, num_qubits=0, num_clbits=0, name=None, global_phase=0.0):
        self.name = name if name is not None else "circuit"
        self.global_phase = global_phase
        self._qreg = Register(int(num_qubits), "q", Qubit)
        self._creg = Register(int(num_clbits), "c", Clbit)
        self.qregs = [self._qreg] if num_qubits else []
        self.cregs = [self._creg] if num_clbits else []
        self.qubits = list(self._qreg)
        self.clbits = list(self._creg)
        self._qindex = {id(b): i for i, b in enumerate(self.qubits)}
        self._cindex = {id(b): i for i, b in enumerate(self.clbits)}
        self.data = []

    # ---- structure queries ---------------------------------------------------------
    @property
    def num_qubits(self):
        return len(self.qubits)

    @property
    def num_clbits(self):
        return len(self.clbits)

    def find_bit(self, bit):
        if id(bit) in self._qindex:
            return BitLocations(self._qindex[id(bit)], [(self._qreg, self._qindex[id(bit)])])
        if id(bit) in self._cindex:
            return BitLocations(self._cindex[id(bit)], [(self._creg, self._cindex[id(bit)])])
        raise ValueError("bit %r is not in this circuit" % (bit,))

    def size(self):
        return sum(1 for ci in self.data if ci.operation.name != "barrier")

    def count_ops(self):
        out = {}
        for ci in self.data:
            out[ci.operation.name] = out.get(ci.operation.name, 0) + 1
        return out

    def __len__(self):
        return len(self.data)

    # ---- argument plumbing ---------------------------------------------------------
    def _q(self, spec):
        """int | Qubit | iterable thereof -> list of Qubit"""
        if isinstance(spec, Qubit):
            return [spec]
        if isinstance(spec, (int,)) or hasattr(spec, "__index__"):
            return [self.qubits[int(spec)]]
        return [q for s in spec for q in self._q(s)]

    def _c(self, spec):
        if isinstance(spec, Clbit):
            return [spec]
        if isinstance(spec, (int,)) or hasattr(spec, "__index__"):
            return [self.clbits[int(spec)]]
        return [c for s in spec for c in self._c(s)]

    def _add(self, op, qubits, clbits=()):
        if len(set(id(q) for q in qubits)) != len(qubits):
            raise ValueError("duplicate qubit arguments in %s" % op.name)
        self.data.append(CircuitInstruction(op, qubits, clbits))
        return self.data[-1]

    def _broadcast_1q(self, name, qubit, params=()):
        for q in self._q(qubit):
            self._add(Instruction(name, 1, 0, params), [q])

    # ---- gates -----------------------------------------------------------------------
    def h(self, qubit): self._broadcast_1q("h", qubit)
    def x(self, qubit): self._broadcast_1q("x", qubit)
    def y(self, qubit): self._broadcast_1q("y", qubit)
    def z(self, qubit): self._broadcast_1q("z", qubit)
    def s(self, qubit): self._broadcast_1q("s", qubit)
    def sdg(self, qubit): self._broadcast_1q("sdg", qubit)
    def t(self, qubit): self._broadcast_1q("t", qubit)
    def tdg(self, qubit): self._broadcast_1q("tdg", qubit)
    def sx(self, qubit): self._broadcast_1q("sx", qubit)
    def sxdg(self, qubit): self._broadcast_1q("sxdg", qubit)
    def id(self, qubit): self._broadcast_1q("id", qubit)
    def rx(self, theta, qubit): self._broadcast_1q("rx", qubit, [theta])
    def ry(self, theta, qubit): self._broadcast_1q("ry", qubit, [theta])
    def rz(self, phi, qubit): self._broadcast_1q("rz", qubit, [phi])
    def p(self, lam, qubit): self._broadcast_1q("p", qubit, [lam])
    def u(self, theta, phi, lam, qubit): self._broadcast_1q("u", qubit, [theta, phi, lam])

    def cx(self, control, target):
        self._add(Instruction("cx", 2), self._q(control) + self._q(target))

    def cz(self, control, target):
        self._add(Instruction("cz", 2), self._q(control) + self._q(target))

    def swap(self, a, b):
        self._add(Instruction("swap", 2), self._q(a) + self._q(b))

    def cp(self, theta, control, target):
        self._add(Instruction("cp", 2, 0, [theta]), self._q(control) + self._q(target))

    def crz(self, theta, control, target):
        self._add(Instruction("crz", 2, 0, [theta]), self._q(control) + self._q(target))

    def ccx(self, c1, c2, target):
        self._add(Instruction("ccx", 3), self._q(c1) + self._q(c2) + self._q(target))

    def mcx(self, control_qubits, target_qubit):
        ctrls = self._q(control_qubits)
        tgt = self._q(target_qubit)
        name = {0: "x", 1: "cx", 2: "ccx"}.get(len(ctrls), "mcx")
        self._add(Instruction(name, len(ctrls) + 1), ctrls + tgt)

    def measure(self, qubit, cbit):
        qs, cs = self._q(qubit), self._c(cbit)
        if len(qs) != len(cs):
            raise ValueError("measure: %d qubits but %d clbits" % (len(qs), len(cs)))
        for q, c in zip(qs, cs):
            self._add(Instruction("measure", 1, 1), [q], [c])

    def barrier(self, *qargs):
        qs = self._q(qargs) if qargs else list(self.qubits)
        self._add(Instruction("barrier", len(qs)), qs)

    # ---- composition -----------------------------------------------------------------
    def to_instruction(self):
        return Instruction(self.name, self.num_qubits, self.num_clbits, [], self)

    to_gate = to_instruction

    def append(self, instruction, qargs=None, cargs=None):
        if isinstance(instruction, QuantumCircuit):
            instruction = instruction.to_instruction()
        qs = self._q(qargs or [])
        cs = self._c(cargs or [])
        if len(qs) != instruction.num_qubits:
            raise ValueError("instruction %s acts on %d qubits, %d given"
                             % (instruction.name, instruction.num_qubits, len(qs)))
        return self._add(instruction, qs, cs)

    def inverse(self):
        """same bits, name + '_dg', negated global phase, every operation inverted on its own in reverse
        order (qiskit.QuantumCircuit.inverse takes no arguments either)"""
        inv = QuantumCircuit(self.num_qubits, self.num_clbits, name=self.name + "_dg",
                             global_phase=-self.global_phase)
        for ci in reversed(self.data):
            qs = [inv.qubits[self._qindex[id(q)]] for q in ci.qubits]
            cs = [inv.clbits[self._cindex[id(c)]] for c in ci.clbits]
            inv._add(ci.operation.inverse(), qs, cs)
        return inv


def AND(num_variable_qubits, flags=None):
    """Counterpart of qiskit.circuit.library.AND(num_variable_qubits, flags) (QCMRF.py:9,225):
    a circuit on ``num_variable_qubits + 1`` qubits that XORs the conjunction of the flagged
    variables onto the last qubit.  flag > 0: variable as is; flag < 0: negated; 0: ignored."""
    flags = list(flags) if flags is not None else [1] * num_variable_qubits
    if len(flags) != num_variable_qubits:
        raise ValueError("AND: %d flags for %d variables" % (len(flags), num_variable_qubits))
    inner = QuantumCircuit(num_variable_qubits + 1, name="and")
    ctrl = [q for q, f in enumerate(flags) if f != 0]
    flip = [q for q, f in enumerate(flags) if f < 0]
    if flip:
        inner.x(flip)
    inner.mcx(ctrl, num_variable_qubits)
    if flip:
        inner.x(flip)
    # nested as Qiskit nests it: the AND circuit holds ONE gate "and" whose definition is x.. mcx x..
    circ = QuantumCircuit(num_variable_qubits + 1, name="and")
    circ.append(inner.to_gate(), list(range(num_variable_qubits + 1)))
    return circ
