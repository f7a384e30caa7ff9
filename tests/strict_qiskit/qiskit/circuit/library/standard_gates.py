"""Standard gates of the strict double: names, parameter lists, ``inverse()`` results, ``ctrl_state``,
``to_matrix()`` and (where cheap) ``definition`` as in qiskit.circuit.library.standard_gates.  A
definition that is not modelled raises -- the engine must treat these names as primitives."""
import cmath
import math

import numpy

from .. import Gate, ControlledGate, QuantumCircuit, QuantumRegister

PI = math.pi


def _qc(n, name):
    return QuantumCircuit(QuantumRegister(n, "q"), name=name)


class HGate(Gate):
    def __init__(self, label=None, duration=None, unit="dt"):
        super().__init__("h", 1, [], label=label, duration=duration, unit=unit)

    def _define(self):
        qc = _qc(1, self.name)
        qc.append(U2Gate(0, PI), [0], [])
        self.definition = qc

    def inverse(self):
        return HGate()

    def __array__(self, dtype=None):
        return numpy.array([[1, 1], [1, -1]], dtype=dtype) / math.sqrt(2)


class XGate(Gate):
    def __init__(self, label=None, duration=None, unit="dt"):
        super().__init__("x", 1, [], label=label, duration=duration, unit=unit)

    def _define(self):
        qc = _qc(1, self.name)
        qc.append(U3Gate(PI, 0, PI), [0], [])
        self.definition = qc

    def inverse(self):
        return XGate()

    def __array__(self, dtype=None):
        return numpy.array([[0, 1], [1, 0]], dtype=dtype)


class IGate(Gate):
    def __init__(self, label=None, duration=None, unit="dt"):
        super().__init__("id", 1, [], label=label, duration=duration, unit=unit)

    def inverse(self):
        return IGate()

    def __array__(self, dtype=None):
        return numpy.eye(2, dtype=dtype)


class SXGate(Gate):
    def __init__(self, label=None, duration=None, unit="dt"):
        super().__init__("sx", 1, [], label=label, duration=duration, unit=unit)

    def inverse(self):
        return SXdgGate()

    def __array__(self, dtype=None):
        return numpy.array([[1 + 1j, 1 - 1j], [1 - 1j, 1 + 1j]], dtype=dtype) / 2


class SXdgGate(Gate):
    def __init__(self, label=None, duration=None, unit="dt"):
        super().__init__("sxdg", 1, [], label=label, duration=duration, unit=unit)

    def inverse(self):
        return SXGate()

    def __array__(self, dtype=None):
        return numpy.array([[1 - 1j, 1 + 1j], [1 + 1j, 1 - 1j]], dtype=dtype) / 2


class TGate(Gate):
    def __init__(self, label=None, duration=None, unit="dt"):
        super().__init__("t", 1, [], label=label, duration=duration, unit=unit)

    def inverse(self):
        return TdgGate()

    def __array__(self, dtype=None):
        return numpy.array([[1, 0], [0, cmath.exp(1j * PI / 4)]], dtype=dtype)


class TdgGate(Gate):
    def __init__(self, label=None, duration=None, unit="dt"):
        super().__init__("tdg", 1, [], label=label, duration=duration, unit=unit)

    def inverse(self):
        return TGate()

    def __array__(self, dtype=None):
        return numpy.array([[1, 0], [0, cmath.exp(-1j * PI / 4)]], dtype=dtype)


class RZGate(Gate):
    def __init__(self, phi, label=None, duration=None, unit="dt"):
        super().__init__("rz", 1, [phi], label=label, duration=duration, unit=unit)

    def inverse(self):
        return RZGate(-self.params[0])

    def __array__(self, dtype=None):
        lam = float(self.params[0])
        return numpy.array([[cmath.exp(-0.5j * lam), 0], [0, cmath.exp(0.5j * lam)]], dtype=dtype)


class PhaseGate(Gate):
    def __init__(self, theta, label=None, duration=None, unit="dt"):
        super().__init__("p", 1, [theta], label=label, duration=duration, unit=unit)

    def inverse(self):
        return PhaseGate(-self.params[0])

    def __array__(self, dtype=None):
        return numpy.array([[1, 0], [0, cmath.exp(1j * float(self.params[0]))]], dtype=dtype)


def _u(th, ph, lam, dtype):
    c, s = math.cos(th / 2), math.sin(th / 2)
    return numpy.array([[c, -cmath.exp(1j * lam) * s], [cmath.exp(1j * ph) * s, cmath.exp(1j * (ph + lam)) * c]], dtype=dtype)


class UGate(Gate):
    def __init__(self, theta, phi, lam, label=None, duration=None, unit="dt"):
        super().__init__("u", 1, [theta, phi, lam], label=label, duration=duration, unit=unit)

    def inverse(self):
        return UGate(-self.params[0], -self.params[2], -self.params[1])

    def __array__(self, dtype=None):
        return _u(*[float(p) for p in self.params], dtype)


class U3Gate(Gate):
    def __init__(self, theta, phi, lam, label=None, duration=None, unit="dt"):
        super().__init__("u3", 1, [theta, phi, lam], label=label, duration=duration, unit=unit)

    def inverse(self):
        return U3Gate(-self.params[0], -self.params[2], -self.params[1])

    def __array__(self, dtype=None):
        return _u(*[float(p) for p in self.params], dtype)


class U2Gate(Gate):
    def __init__(self, phi, lam, label=None, duration=None, unit="dt"):
        super().__init__("u2", 1, [phi, lam], label=label, duration=duration, unit=unit)

    def inverse(self):
        return U2Gate(-self.params[1] - PI, -self.params[0] + PI)

    def __array__(self, dtype=None):
        return _u(PI / 2, float(self.params[0]), float(self.params[1]), dtype)


class CXGate(ControlledGate):
    def __init__(self, label=None, ctrl_state=None, duration=None, unit="dt"):
        super().__init__("cx", 2, [], label=label, num_ctrl_qubits=1, ctrl_state=ctrl_state, base_gate=XGate(),
                         duration=duration, unit=unit)

    def inverse(self):
        return CXGate(ctrl_state=self.ctrl_state)


class CCXGate(ControlledGate):
    def __init__(self, label=None, ctrl_state=None, duration=None, unit="dt"):
        super().__init__("ccx", 3, [], label=label, num_ctrl_qubits=2, ctrl_state=ctrl_state, base_gate=XGate(),
                         duration=duration, unit=unit)

    def _define(self):
        qc = _qc(3, self.name)
        for g, qs in ((HGate(), [2]), (CXGate(), [1, 2]), (TdgGate(), [2]), (CXGate(), [0, 2]), (TGate(), [2]),
                      (CXGate(), [1, 2]), (TdgGate(), [2]), (CXGate(), [0, 2]), (TGate(), [1]), (TGate(), [2]),
                      (HGate(), [2]), (CXGate(), [0, 1]), (TGate(), [0]), (TdgGate(), [1]), (CXGate(), [0, 1])):
            qc.append(g, qs, [])
        self.definition = qc

    def inverse(self):
        return CCXGate(ctrl_state=self.ctrl_state)


class MCXGate(ControlledGate):
    """MCXGate(num_ctrl_qubits) is a FACTORY in Qiskit: 1 -> CXGate, 2 -> CCXGate, 3 -> C3XGate and
    4 -> C4XGate (both named 'mcx'), more -> MCXGrayCode ('mcx_gray')"""

    def __new__(cls, num_ctrl_qubits=None, label=None, ctrl_state=None, _name="mcx", duration=None, unit="dt"):
        explicit = {1: CXGate, 2: CCXGate}
        if cls is MCXGate and num_ctrl_qubits in explicit:
            gate = explicit[num_ctrl_qubits].__new__(explicit[num_ctrl_qubits])
            gate.__init__(label=label, ctrl_state=ctrl_state, duration=duration, unit=unit)
            return gate
        if cls is MCXGate and num_ctrl_qubits in (3, 4):
            sub = C3XGate if num_ctrl_qubits == 3 else C4XGate
            gate = object.__new__(sub)
            sub.__init__(gate, label=label, ctrl_state=ctrl_state, duration=duration, unit=unit)
            return gate
        if cls is MCXGate:
            gate = object.__new__(MCXGrayCode)
            MCXGrayCode.__init__(gate, num_ctrl_qubits, label=label, ctrl_state=ctrl_state, duration=duration, unit=unit)
            return gate
        return object.__new__(cls)

    def __init__(self, num_ctrl_qubits, label=None, ctrl_state=None, _name="mcx", duration=None, unit="dt"):
        super().__init__(_name, num_ctrl_qubits + 1, [], label=label, num_ctrl_qubits=num_ctrl_qubits,
                         ctrl_state=ctrl_state, base_gate=XGate(), duration=duration, unit=unit)

    def _define(self):
        raise NotImplementedError("strict double: the definition of %s is not modelled -- treat it as a primitive" % self._name)

    def inverse(self):
        return MCXGate(num_ctrl_qubits=self.num_ctrl_qubits, ctrl_state=self.ctrl_state)


class C3XGate(MCXGate):
    def __init__(self, label=None, ctrl_state=None, duration=None, unit="dt"):
        ControlledGate.__init__(self, "mcx", 4, [], label=label, num_ctrl_qubits=3, ctrl_state=ctrl_state,
                                base_gate=XGate(), duration=duration, unit=unit)

    def inverse(self):
        return C3XGate(ctrl_state=self.ctrl_state)


class C4XGate(MCXGate):
    def __init__(self, label=None, ctrl_state=None, duration=None, unit="dt"):
        ControlledGate.__init__(self, "mcx", 5, [], label=label, num_ctrl_qubits=4, ctrl_state=ctrl_state,
                                base_gate=XGate(), duration=duration, unit=unit)

    def inverse(self):
        return C4XGate(ctrl_state=self.ctrl_state)


class MCXGrayCode(MCXGate):
    def __init__(self, num_ctrl_qubits, label=None, ctrl_state=None, duration=None, unit="dt"):
        MCXGate.__init__(self, num_ctrl_qubits, label=label, ctrl_state=ctrl_state, _name="mcx_gray",
                         duration=duration, unit=unit)

    def inverse(self):
        return MCXGrayCode(num_ctrl_qubits=self.num_ctrl_qubits, ctrl_state=self.ctrl_state)


class CPhaseGate(ControlledGate):
    def __init__(self, theta, label=None, ctrl_state=None, duration=None, unit="dt"):
        super().__init__("cp", 2, [theta], label=label, num_ctrl_qubits=1, ctrl_state=ctrl_state,
                         base_gate=PhaseGate(theta), duration=duration, unit=unit)

    def _define(self):
        lam = self.params[0]
        qc = _qc(2, self.name)
        for g, qs in ((PhaseGate(lam / 2), [0]), (CXGate(), [0, 1]), (PhaseGate(-lam / 2), [1]), (CXGate(), [0, 1]),
                      (PhaseGate(lam / 2), [1])):
            qc.append(g, qs, [])
        self.definition = qc

    def inverse(self):
        return CPhaseGate(-self.params[0], ctrl_state=self.ctrl_state)
