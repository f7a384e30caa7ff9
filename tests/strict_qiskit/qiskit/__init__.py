"""TEST INFRASTRUCTURE -- NOT Qiskit.  A strict, signature-faithful test double of the part of
qiskit-terra 0.45/0.46 that /root/reference/QCMRF.py and run_experiment.py touch and that the
engine's ingest reads (Qiskit itself is not installable in this image).

"Strict" means: every class and method here has the signature Qiskit documents (no private extras,
``QuantumCircuit.inverse()`` takes no arguments, ``append(instruction, qargs=None, cargs=None)``,
``CircuitInstruction(operation, qubits=(), clbits=())`` with tuple fields and no legacy tuple
unpacking, ``find_bit(bit).index``), objects are nested the way Qiskit nests them (``AND`` is a
circuit holding ONE gate "and" whose definition is X..X . MCX . X..X; ``append(circuit)`` goes
through ``to_instruction()``, which copies into a fresh register; ``inverse()`` inverts instruction
by instruction, names get "_dg", every instance is a distinct object), open controls rename the
gate (``ccx_o1``), and anything not modelled raises instead of guessing.

Put ``tests/strict_qiskit`` on ``sys.path`` BEFORE importing ``qcmrf_amd`` and the package takes its
"Qiskit is importable" branch (``HAVE_QISKIT``), which is the reference's real environment.
"""
from .circuit import QuantumCircuit, QuantumRegister, ClassicalRegister, AncillaRegister
from .compiler import transpile
from .exceptions import QiskitError

__version__ = "0.45.0+strict-double"
__all__ = ["QuantumCircuit", "QuantumRegister", "ClassicalRegister", "AncillaRegister", "transpile", "QiskitError"]
