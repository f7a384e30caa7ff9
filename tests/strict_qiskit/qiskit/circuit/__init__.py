"""TEST INFRASTRUCTURE -- the circuit data model of the strict Qiskit double (see ../__init__.py).

Signatures follow qiskit-terra 0.45/0.46 (the window SURVEY.md 8c infers for the reference):
``qiskit.circuit.{Bit, Qubit, Clbit, Register, QuantumRegister, ClassicalRegister, AncillaRegister,
Instruction, Gate, ControlledGate, CircuitInstruction, QuantumCircuit, Measure, Barrier}``.
Nothing here simulates anything.
"""
from __future__ import annotations

import copy as _copy
import itertools
from collections import namedtuple

from ..exceptions import QiskitError


class CircuitError(QiskitError):
    pass


BitLocations = namedtuple("BitLocations", ("index", "registers"))


# ---- bits and registers ------------------------------------------------------------------------
class Bit:
    """old-style bit: equal (and hashing alike) when (register name, register size, index) agree,
    as in Qiskit < 1.0 -- two circuits with a register 'q' of the same size hold EQUAL but DISTINCT bits"""
    __slots__ = ("_register", "_index", "_hash", "_repr")

    def __init__(self, register=None, index=None):
        if (register, index) != (None, None):
            if not isinstance(index, int) or index < -register.size or index >= register.size:
                raise CircuitError("index must be an integer inside the register")
        self._register, self._index = register, index
        if register is None:
            self._repr, self._hash = None, object.__hash__(self)
        else:
            self._repr = "%s(%r, %d)" % (type(self).__name__, register, index)
            self._hash = hash(self._repr)

    def __repr__(self):
        return object.__repr__(self) if self._repr is None else self._repr

    def __hash__(self):
        return self._hash

    def __eq__(self, other):
        if self._repr is None or getattr(other, "_repr", None) is None:
            return other is self
        return self._repr == other._repr


class Qubit(Bit):
    __slots__ = ()


class AncillaQubit(Qubit):
    __slots__ = ()


class Clbit(Bit):
    __slots__ = ()


class Register:
    bit_type = None
    prefix = "reg"
    _counter = itertools.count()

    def __init__(self, size=None, name=None, bits=None):
        if (size, bits) == (None, None) or (size is not None and bits is not None):
            raise CircuitError("Exactly one of the size or bits arguments can be provided")
        if bits is not None:
            size = len(bits)
        if int(size) != size or size < 0:
            raise CircuitError("Register size must be a non-negative integer")
        self._size = int(size)
        self._name = name if name is not None else "%s%d" % (self.prefix, next(self._counter))
        self._repr = "%s(%d, '%s')" % (type(self).__name__, self._size, self._name)
        self._bits = list(bits) if bits is not None else [self.bit_type(self, i) for i in range(self._size)]

    name = property(lambda self: self._name)
    size = property(lambda self: self._size)

    def __repr__(self):
        return self._repr

    def __len__(self):
        return self._size

    def __iter__(self):
        return iter(self._bits)

    def __getitem__(self, key):
        if not isinstance(key, (int, slice, list)):
            raise CircuitError("expected integer or slice index into register")
        if isinstance(key, list):
            return [self._bits[i] for i in key]
        return self._bits[key]

    def index(self, bit):
        return self._bits.index(bit)

    def __eq__(self, other):
        return type(self) is type(other) and self._repr == other._repr

    def __hash__(self):
        return hash(self._repr)


class QuantumRegister(Register):
    bit_type, prefix = Qubit, "q"


class AncillaRegister(QuantumRegister):
    bit_type, prefix = AncillaQubit, "a"


class ClassicalRegister(Register):
    bit_type, prefix = Clbit, "c"


# ---- operations --------------------------------------------------------------------------------
class Operation:
    pass


class Instruction(Operation):
    """qiskit.circuit.Instruction(name, num_qubits, num_clbits, params, duration=None, unit='dt', label=None)"""

    def __init__(self, name, num_qubits, num_clbits, params, duration=None, unit="dt", label=None):
        if not isinstance(num_qubits, int) or not isinstance(num_clbits, int):
            raise CircuitError("num_qubits and num_clbits must be integer.")
        if num_qubits < 0 or num_clbits < 0:
            raise CircuitError("bad instruction dimensions")
        self._name = name
        self._num_qubits, self._num_clbits = num_qubits, num_clbits
        self._params = list(params)
        self._definition = None
        self._label = label
        self.condition = None
        self._duration, self._unit = duration, unit

    # name / sizes / params are properties in Qiskit too (ControlledGate overrides name)
    @property
    def name(self):
        return self._name

    @name.setter
    def name(self, v):
        self._name = v

    num_qubits = property(lambda self: self._num_qubits)
    num_clbits = property(lambda self: self._num_clbits)
    label = property(lambda self: self._label)
    duration = property(lambda self: self._duration)
    unit = property(lambda self: self._unit)
    mutable = True

    @property
    def params(self):
        return self._params

    @params.setter
    def params(self, v):
        self._params = list(v)

    @property
    def definition(self):
        if self._definition is None:
            self._define()
        return self._definition

    @definition.setter
    def definition(self, circ):
        self._definition = circ

    def _define(self):
        pass

    def is_parameterized(self):
        return False

    def copy(self, name=None):
        cpy = _copy.copy(self)
        cpy._params = list(self._params)
        if name is not None:
            cpy._name = name
        return cpy

    def to_mutable(self):
        return self.copy()

    def c_if(self, classical, val):
        self.condition = (classical, int(val))
        return self

    def inverse(self):
        """composite instruction: a new gate / instruction named ``<name>_dg`` (``_dg`` stripped if it
        is already there) whose definition holds the inverses of the definition's instructions in
        reverse order -- every one a NEW object"""
        if self.definition is None:
            raise CircuitError("inverse() not implemented for %s." % self.name)
        name = self.name[:-3] if self.name.endswith("_dg") else self.name + "_dg"
        if self.num_clbits:
            inv = Instruction(name, self.num_qubits, self.num_clbits, list(self.params))
        else:
            inv = Gate(name, self.num_qubits, list(self.params))
        d = self._definition.copy_empty_like()
        d.global_phase = -self.definition.global_phase
        for inst in reversed(self._definition.data):
            d._append(CircuitInstruction(inst.operation.inverse(), inst.qubits, inst.clbits))
        inv.definition = d
        return inv

    def broadcast_arguments(self, qargs, cargs):
        if len(qargs) != self.num_qubits:
            raise CircuitError("The amount of qubit arguments %d does not match the instruction expectation (%d)."
                               % (len(qargs), self.num_qubits))
        if any(len(a) != 1 for a in qargs) and len(set(len(a) for a in qargs)) != 1:
            raise CircuitError("broadcast of unequal argument lengths is not modelled")
        if not qargs:
            yield [], []
            return
        for k in range(len(qargs[0])):
            yield [a[k if len(a) > 1 else 0] for a in qargs], [a[k if len(a) > 1 else 0] for a in cargs]

    def __repr__(self):
        return "Instruction(name='%s', num_qubits=%d, num_clbits=%d, params=%r)" % (
            self.name, self.num_qubits, self.num_clbits, self.params)


class Gate(Instruction):
    """qiskit.circuit.Gate(name, num_qubits, params, label=None, duration=None, unit='dt')"""

    def __init__(self, name, num_qubits, params, label=None, duration=None, unit="dt"):
        super().__init__(name, num_qubits, 0, params, duration=duration, unit=unit, label=label)

    def to_matrix(self):
        if hasattr(self, "__array__"):
            return self.__array__(dtype=complex)
        raise CircuitError("to_matrix not defined for this %s" % type(self))

    def broadcast_arguments(self, qargs, cargs):
        if len(qargs) != self.num_qubits or cargs:
            raise CircuitError("The amount of qubit(%d)/clbit(%d) arguments does not match the gate expectation (%d)."
                               % (len(qargs), len(cargs), self.num_qubits))
        if any(not a for a in qargs):
            raise CircuitError("One or more of the arguments are empty")
        if len(qargs) == 1:
            for q in qargs[0]:
                yield [q], []
        elif len(qargs) == 2 and len(qargs[0]) != len(qargs[1]):
            a, b = qargs
            if len(a) == 1:
                for q in b:
                    yield [a[0], q], []
            elif len(b) == 1:
                for q in a:
                    yield [q, b[0]], []
            else:
                raise CircuitError("Not sure how to combine these two-qubit arguments")
        else:
            if len(set(len(a) for a in qargs)) != 1:
                raise CircuitError("Not sure how to combine these qubit arguments")
            for tup in zip(*qargs):
                yield list(tup), []

    def __repr__(self):
        return "Instruction(name='%s', num_qubits=%d, num_clbits=0, params=%r)" % (self.name, self.num_qubits, self.params)


class ControlledGate(Gate):
    """qiskit.circuit.ControlledGate(name, num_qubits, params, label=None, num_ctrl_qubits=1,
    definition=None, ctrl_state=None, base_gate=None, duration=None, unit='dt')"""

    def __init__(self, name, num_qubits, params, label=None, num_ctrl_qubits=1, definition=None,
                 ctrl_state=None, base_gate=None, duration=None, unit="dt"):
        self.base_gate = None if base_gate is None else base_gate.copy()
        super().__init__(name, num_qubits, params, label=label, duration=duration, unit=unit)
        self._num_ctrl_qubits = 1
        self.num_ctrl_qubits = num_ctrl_qubits
        self.definition = _copy.deepcopy(definition)
        self._ctrl_state = None
        self._open_ctrl = None
        self.ctrl_state = ctrl_state

    @property
    def name(self):
        """open controls are part of the NAME: ``ccx_o1`` (as Qiskit does)"""
        return "%s_o%d" % (self._name, self.ctrl_state) if self._open_ctrl else self._name

    @name.setter
    def name(self, v):
        self._name = v

    @property
    def num_ctrl_qubits(self):
        return self._num_ctrl_qubits

    @num_ctrl_qubits.setter
    def num_ctrl_qubits(self, v):
        if v != int(v) or not 1 <= v <= self.num_qubits:
            raise CircuitError("The number of control qubits must be in `[1, num_qubits]`.")
        self._num_ctrl_qubits = int(v)

    @property
    def ctrl_state(self):
        return self._ctrl_state

    @ctrl_state.setter
    def ctrl_state(self, state):
        full = 2 ** self.num_ctrl_qubits - 1
        if state is None:
            state = full
        elif isinstance(state, str):
            if len(state) != self.num_ctrl_qubits:
                raise CircuitError("invalid control bit string: " + state)
            state = int(state, 2)
        if not isinstance(state, int) or not 0 <= state <= full:
            raise CircuitError("invalid control state specification: %r" % (state,))
        self._ctrl_state = state
        self._open_ctrl = state != full

    @property
    def definition(self):
        """closed-control definition conjugated by X on the open controls"""
        closed = Instruction.definition.fget(self)
        if not self._open_ctrl or closed is None:
            return closed
        qc = QuantumCircuit(QuantumRegister(self.num_qubits, "q"))
        flip = [i for i in range(self.num_ctrl_qubits) if not (self.ctrl_state >> i) & 1]
        for i in flip:
            qc.x(i)
        twin = self.copy()
        twin.ctrl_state = None
        qc.append(twin, list(range(self.num_qubits)), [])
        for i in flip:
            qc.x(i)
        return qc

    @definition.setter
    def definition(self, circ):
        self._definition = circ


class Measure(Instruction):
    def __init__(self):
        super().__init__("measure", 1, 1, [])

    def broadcast_arguments(self, qargs, cargs):
        q, c = qargs[0], cargs[0]
        if len(q) == len(c):
            for a, b in zip(q, c):
                yield [a], [b]
        elif len(q) == 1 and c:
            for b in c:
                yield q, [b]
        else:
            raise CircuitError("register size error")


class Barrier(Instruction):
    def __init__(self, num_qubits, label=None):
        super().__init__("barrier", num_qubits, 0, [], label=label)

    def inverse(self):
        return Barrier(self.num_qubits)

    def broadcast_arguments(self, qargs, cargs):
        yield [q for a in qargs for q in a], []


class CircuitInstruction:
    """qiskit.circuit.CircuitInstruction(operation, qubits=(), clbits=()) -- no tuple unpacking"""
    __slots__ = ("operation", "qubits", "clbits")

    def __init__(self, operation, qubits=(), clbits=()):
        self.operation = operation
        self.qubits = tuple(qubits)
        self.clbits = tuple(clbits)

    def copy(self):
        return CircuitInstruction(self.operation, self.qubits, self.clbits)

    def replace(self, operation=None, qubits=None, clbits=None):
        return CircuitInstruction(self.operation if operation is None else operation,
                                  self.qubits if qubits is None else qubits,
                                  self.clbits if clbits is None else clbits)

    def __eq__(self, other):
        return (isinstance(other, CircuitInstruction) and self.operation is other.operation
                and self.qubits == other.qubits and self.clbits == other.clbits)

    def __repr__(self):
        return "CircuitInstruction(operation=%r, qubits=%r, clbits=%r)" % (self.operation, self.qubits, self.clbits)


class InstructionSet:
    def __init__(self):
        self._instructions = []

    def add(self, ci):
        self._instructions.append(ci)

    def __len__(self):
        return len(self._instructions)

    def __getitem__(self, i):
        return self._instructions[i]

    def c_if(self, classical, val):
        for ci in self._instructions:
            ci.operation = ci.operation.c_if(classical, val)
        return self


class QuantumCircuitData:
    """read view ``circuit.data``: len / iteration / indexing, as in Qiskit"""

    def __init__(self, items):
        self._items = items

    def __len__(self):
        return len(self._items)

    def __iter__(self):
        return iter(list(self._items))

    def __getitem__(self, i):
        return self._items[i]

    def __reversed__(self):
        return reversed(list(self._items))

    def copy(self):
        return list(self._items)


class QuantumCircuit:
    """qiskit.QuantumCircuit(*regs, name=None, global_phase=0, metadata=None)"""
    _instances = itertools.count()

    def __init__(self, *regs, name=None, global_phase=0, metadata=None):
        if any(not isinstance(r, (Register, list, tuple)) for r in regs):
            try:
                regs = tuple(int(r) for r in regs)
            except Exception:
                raise CircuitError("Circuit args must be Registers or integers.")
        if name is None:
            name = "circuit-%d" % next(self._instances)
        elif not isinstance(name, str):
            raise CircuitError("The circuit name should be a string (or None to auto-generate a name).")
        self.name = name
        self._data = []
        self.qregs, self.cregs = [], []
        self._qubits, self._clbits = [], []
        self._qubit_indices, self._clbit_indices = {}, {}
        self._global_phase = 0.0
        self.global_phase = global_phase
        self.metadata = {} if metadata is None else metadata
        self.add_register(*regs)

    # ---- registers and bits -------------------------------------------------------------
    def add_register(self, *regs):
        if not regs:
            return
        if all(isinstance(r, int) for r in regs):
            if len(regs) == 1:
                regs = (QuantumRegister(regs[0], "q"),)
            elif len(regs) == 2:
                regs = (QuantumRegister(regs[0], "q"), ClassicalRegister(regs[1], "c"))
            else:
                raise CircuitError("QuantumCircuit parameters can be Registers or Integers; at most two integers.")
        for r in regs:
            if isinstance(r, (list, tuple)):
                self.add_bits(r)
                continue
            if r.name in [x.name for x in self.qregs + self.cregs]:
                raise CircuitError('register name "%s" already exists' % r.name)
            if isinstance(r, QuantumRegister):
                self.qregs.append(r)
                bits, index = self._qubits, self._qubit_indices
            elif isinstance(r, ClassicalRegister):
                self.cregs.append(r)
                bits, index = self._clbits, self._clbit_indices
            else:
                raise CircuitError("expected a register")
            for i, b in enumerate(r):
                if b in index:
                    index[b].registers.append((r, i))
                else:
                    index[b] = BitLocations(len(bits), [(r, i)])
                    bits.append(b)

    def add_bits(self, bits):
        for b in bits:
            if isinstance(b, Qubit):
                lst, index = self._qubits, self._qubit_indices
            elif isinstance(b, Clbit):
                lst, index = self._clbits, self._clbit_indices
            else:
                raise CircuitError("Expected an instance of Qubit, Clbit, or AncillaQubit")
            if b in index:
                raise CircuitError("Attempted to add bits found already in circuit")
            index[b] = BitLocations(len(lst), [])
            lst.append(b)

    qubits = property(lambda self: list(self._qubits))
    clbits = property(lambda self: list(self._clbits))
    num_qubits = property(lambda self: len(self._qubits))
    num_clbits = property(lambda self: len(self._clbits))
    data = property(lambda self: QuantumCircuitData(self._data))

    @property
    def global_phase(self):
        return self._global_phase

    @global_phase.setter
    def global_phase(self, angle):
        self._global_phase = float(angle) % (2 * 3.141592653589793) if angle else 0.0

    def find_bit(self, bit):
        try:
            if isinstance(bit, Qubit):
                return self._qubit_indices[bit]
            if isinstance(bit, Clbit):
                return self._clbit_indices[bit]
        except KeyError:
            raise CircuitError("Could not locate provided bit: %r. Has it been added to the QuantumCircuit?" % (bit,))
        raise CircuitError("Could not locate bit of unknown type: %s" % type(bit))

    def __len__(self):
        return len(self._data)

    def __iter__(self):
        return iter(list(self._data))

    def __getitem__(self, i):
        return self._data[i]

    def size(self):
        return sum(1 for ci in self._data if ci.operation.name != "barrier")

    def count_ops(self):
        out = {}
        for ci in self._data:
            out[ci.operation.name] = out.get(ci.operation.name, 0) + 1
        return out

    # ---- argument conversion ------------------------------------------------------------
    @staticmethod
    def _convert(spec, bits, kind):
        if isinstance(spec, kind):
            return [spec]
        if isinstance(spec, Register):
            return list(spec)
        if isinstance(spec, (int,)) or (hasattr(spec, "__index__") and not isinstance(spec, (list, tuple, range))):
            try:
                return [bits[int(spec)]]
            except IndexError:
                raise CircuitError("Index %r out of range for size %d." % (spec, len(bits)))
        if isinstance(spec, slice):
            return bits[spec]
        if isinstance(spec, (list, tuple, range)):
            return [b for s in spec for b in QuantumCircuit._convert(s, bits, kind)]
        raise CircuitError("Invalid bit index: '%r' of type '%s'" % (spec, type(spec)))

    def qbit_argument_conversion(self, spec):
        return self._convert(spec, self._qubits, Qubit)

    def cbit_argument_conversion(self, spec):
        return self._convert(spec, self._clbits, Clbit)

    # ---- append ---------------------------------------------------------------------------
    def append(self, instruction, qargs=None, cargs=None):
        if isinstance(instruction, CircuitInstruction):
            instruction, qargs, cargs = instruction.operation, instruction.qubits, instruction.clbits
        if not isinstance(instruction, Operation):
            if hasattr(instruction, "to_instruction"):
                instruction = instruction.to_instruction()
            else:
                raise CircuitError("Object to append must be an Operation or have a to_instruction() method.")
        qs = [self.qbit_argument_conversion(a) for a in qargs or []]
        cs = [self.cbit_argument_conversion(a) for a in cargs or []]
        out = InstructionSet()
        for q, c in instruction.broadcast_arguments(qs, cs):
            self._check_dups(q)
            out.add(self._append(CircuitInstruction(instruction, q, c)))
        return out

    def _append(self, ci, qargs=None, cargs=None):
        if not isinstance(ci, CircuitInstruction):
            ci = CircuitInstruction(ci, qargs, cargs)
        self._data.append(ci)
        return ci

    @staticmethod
    def _check_dups(qubits):
        if len(set(qubits)) != len(qubits):
            raise CircuitError("duplicate qubit arguments")

    # ---- gates QCMRF.py and a {cx,id,rz,sx,x} circuit use; signatures as documented -----------
    def h(self, qubit):
        from .library.standard_gates import HGate
        return self.append(HGate(), [qubit], [])

    def x(self, qubit, label=None):
        from .library.standard_gates import XGate
        return self.append(XGate(label=label), [qubit], [])

    def id(self, qubit):
        from .library.standard_gates import IGate
        return self.append(IGate(), [qubit], [])

    def sx(self, qubit):
        from .library.standard_gates import SXGate
        return self.append(SXGate(), [qubit], [])

    def rz(self, phi, qubit):
        from .library.standard_gates import RZGate
        return self.append(RZGate(phi), [qubit], [])

    def p(self, theta, qubit):
        from .library.standard_gates import PhaseGate
        return self.append(PhaseGate(theta), [qubit], [])

    def u(self, theta, phi, lam, qubit):
        from .library.standard_gates import UGate
        return self.append(UGate(theta, phi, lam), [qubit], [])

    def t(self, qubit):
        from .library.standard_gates import TGate
        return self.append(TGate(), [qubit], [])

    def tdg(self, qubit):
        from .library.standard_gates import TdgGate
        return self.append(TdgGate(), [qubit], [])

    def cx(self, control_qubit, target_qubit, label=None, ctrl_state=None):
        from .library.standard_gates import CXGate
        return self.append(CXGate(label=label, ctrl_state=ctrl_state), [control_qubit, target_qubit], [])

    def cp(self, theta, control_qubit, target_qubit, label=None, ctrl_state=None):
        from .library.standard_gates import CPhaseGate
        return self.append(CPhaseGate(theta, label=label, ctrl_state=ctrl_state), [control_qubit, target_qubit], [])

    def ccx(self, control_qubit1, control_qubit2, target_qubit, ctrl_state=None):
        from .library.standard_gates import CCXGate
        return self.append(CCXGate(ctrl_state=ctrl_state), [control_qubit1, control_qubit2, target_qubit], [])

    def mcx(self, control_qubits, target_qubit, ancilla_qubits=None, mode="noancilla"):
        from .library.standard_gates import MCXGate
        if mode != "noancilla" or ancilla_qubits:
            raise NotImplementedError("strict double: only mcx(..., mode='noancilla') is modelled")
        ctrls = self.qbit_argument_conversion(control_qubits)
        tgt = self.qbit_argument_conversion(target_qubit)
        if len(tgt) != 1:
            raise CircuitError("mcx needs exactly one target")
        return self.append(MCXGate(len(ctrls)), ctrls + tgt, [])

    def measure(self, qubit, cbit):
        return self.append(Measure(), [qubit], [cbit])

    def barrier(self, *qargs, label=None):
        qs = [q for a in qargs for q in self.qbit_argument_conversion(a)] if qargs else list(self._qubits)
        return self.append(Barrier(len(qs), label=label), qs, [])

    # ---- composition ----------------------------------------------------------------------
    def copy_empty_like(self, name=None):
        c = QuantumCircuit(name=self.name if name is None else name, global_phase=self.global_phase)
        c.qregs, c.cregs = list(self.qregs), list(self.cregs)
        c._qubits, c._clbits = list(self._qubits), list(self._clbits)
        c._qubit_indices = {b: BitLocations(l.index, list(l.registers)) for b, l in self._qubit_indices.items()}
        c._clbit_indices = {b: BitLocations(l.index, list(l.registers)) for b, l in self._clbit_indices.items()}
        return c

    def copy(self, name=None):
        c = self.copy_empty_like(name)
        c._data = [ci.replace(operation=ci.operation.copy()) for ci in self._data]
        return c

    def to_instruction(self, parameter_map=None, label=None):
        return _to_operation(self, Instruction(self.name, self.num_qubits, self.num_clbits, [], label=label))

    def to_gate(self, parameter_map=None, label=None):
        if self.num_clbits:
            raise QiskitError("Circuit with classical bits cannot be converted to gate.")
        for ci in self._data:
            if not isinstance(ci.operation, Gate):
                raise QiskitError("One or more instructions cannot be converted to a gate. \"%s\" is not a gate instruction"
                                  % ci.operation.name)
        return _to_operation(self, Gate(self.name, self.num_qubits, [], label=label))

    def compose(self, other, qubits=None, clbits=None, front=False, inplace=False, wrap=False):
        if front or wrap:
            raise NotImplementedError("strict double: compose(front / wrap) is not modelled")
        dest = self if inplace else self.copy()
        if isinstance(other, Operation):
            if qubits is None:
                qubits = list(range(other.num_qubits))
            dest.append(other, qargs=qubits, cargs=clbits)
            return None if inplace else dest
        qs = dest._qubits[:other.num_qubits] if qubits is None else dest.qbit_argument_conversion(qubits)
        cs = dest._clbits[:other.num_clbits] if clbits is None else dest.cbit_argument_conversion(clbits)
        if len(qs) != other.num_qubits or len(cs) != other.num_clbits:
            raise CircuitError("Number of items in qubits / clbits parameter does not match the other circuit.")
        qm = dict(zip(other._qubits, qs))
        cm = dict(zip(other._clbits, cs))
        for ci in other._data:
            dest._append(CircuitInstruction(ci.operation.copy(), [qm[q] for q in ci.qubits], [cm[c] for c in ci.clbits]))
        dest.global_phase += other.global_phase
        return None if inplace else dest

    def inverse(self):
        """Qiskit's signature: NO arguments.  Same bits and registers, name + '_dg', negated global
        phase, every operation inverted on its own (a new object each) in reverse order."""
        inv = self.copy_empty_like(self.name + "_dg")
        inv.global_phase = -self.global_phase
        for ci in reversed(self._data):
            inv._append(ci.replace(operation=ci.operation.inverse()))
        return inv


def _to_operation(circuit, out):
    """circuit_to_instruction / circuit_to_gate: the definition is a COPY on a fresh register 'q'
    (+ 'c'), never the caller's circuit object"""
    regs = []
    if circuit.num_qubits:
        q = QuantumRegister(circuit.num_qubits, "q")
        regs.append(q)
    if circuit.num_clbits:
        c = ClassicalRegister(circuit.num_clbits, "c")
        regs.append(c)
    qm = {b: q[i] for i, b in enumerate(circuit._qubits)}
    cm = {b: c[i] for i, b in enumerate(circuit._clbits)}
    d = QuantumCircuit(*regs, name=out.name, global_phase=circuit.global_phase)
    for ci in circuit._data:
        if ci.operation.condition is not None:
            raise NotImplementedError("strict double: conditions inside to_instruction are not modelled")
        d._append(CircuitInstruction(ci.operation.copy(), [qm[b] for b in ci.qubits], [cm[b] for b in ci.clbits]))
    out.definition = d
    return out


from . import library  # noqa: E402,F401
