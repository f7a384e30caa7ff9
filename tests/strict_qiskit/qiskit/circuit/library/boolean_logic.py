"""qiskit.circuit.library.AND(num_variable_qubits, flags=None, mcx_mode='noancilla') as Qiskit 0.45
builds it: registers 'variable' + 'result'; the circuit holds ONE gate named "and" (the inner circuit's
``to_gate()``), whose definition is  x(flipped) . mcx(controls -> result) . x(flipped)."""
from .. import QuantumCircuit, QuantumRegister


class AND(QuantumCircuit):
    def __init__(self, num_variable_qubits, flags=None, mcx_mode="noancilla"):
        if mcx_mode != "noancilla":
            raise NotImplementedError("strict double: only mcx_mode='noancilla' is modelled")
        self.num_variable_qubits = num_variable_qubits
        self.flags = flags
        qr_variable = QuantumRegister(num_variable_qubits, name="variable")
        qr_result = QuantumRegister(1, name="result")
        circuit = QuantumCircuit(qr_variable, qr_result, name="and")
        flags = flags or [1] * num_variable_qubits
        control_qubits = [q for q, flag in zip(qr_variable, flags) if flag != 0]
        flip_qubits = [q for q, flag in zip(qr_variable, flags) if flag < 0]
        if len(flip_qubits) > 0:
            circuit.x(flip_qubits)
        circuit.mcx(control_qubits, qr_result[:], None, mode=mcx_mode)
        if len(flip_qubits) > 0:
            circuit.x(flip_qubits)
        super().__init__(*circuit.qregs, name="and")
        self.compose(circuit.to_gate(), qubits=self.qubits, inplace=True)
