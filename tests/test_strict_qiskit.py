"""The "Qiskit is importable" branch of the package (qcmrf_amd/qcmrf.py, qcmrf_amd/run_experiment.py) under a
strict, signature-faithful test double of Qiskit (tests/strict_qiskit: ``inverse()`` takes no arguments,
``append(instruction, qargs)``, ``CircuitInstruction(operation, qubits, clbits)`` without tuple unpacking,
``find_bit(bit).index``, a fresh ``AND`` per append, nested "and" gates, ``global_phase``, open-control names).
The reference always runs in that branch (QCMRF.py:6-9,225-234; run_experiment.py:10,52).  Child process: the
double must be importable as ``qiskit`` BEFORE qcmrf_amd is imported, and must not leak into the other tests."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(mode, timeout):
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "_strict_worker.py"), mode], capture_output=True, text=True,
                       timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    assert "strict-qiskit worker ok (%s)" % mode in r.stdout


def test_qcmrf_ingest_and_run_experiment_under_strict_qiskit_signatures():
    """QCMRF(...) for the seven reference graphs, ingest vs the oracle's gate stream, every fusion level on nested and
    transpiled circuits (numpy stand-in engine, amplitudes vs the closed form to 1e-12), run_experiment.main"""
    _run("cpu", 600)


def test_strict_double_rejects_what_qiskit_rejects():
    """the double is strict: private extras of the in-tree container do not exist on it"""
    code = ("import sys; sys.path.insert(0, %r); import qiskit\n"
            "from qiskit.circuit.library import AND, MCXGate\n"
            "c = qiskit.QuantumCircuit(3, name='c'); c.append(AND(2, [1, -1]), [0, 1, 2])\n"
            "for bad in (lambda: c.inverse(_shared={}), lambda: c._add, lambda: iter(c.data[0]), lambda: MCXGate(3).definition,\n"
            "            lambda: c.append(AND(2), [0, 1])):\n"
            "    try:\n"
            "        bad()\n"
            "    except (TypeError, AttributeError, NotImplementedError, qiskit.QiskitError):\n"
            "        continue\n"
            "    raise SystemExit('accepted')\n"
            "assert c.data[0].operation.definition is not c.data[0].operation.definition.data[0].operation.definition\n"
            "assert MCXGate(2, ctrl_state=1).name == 'ccx_o1' and c.inverse().data[0].operation.name == 'and_dg'\n"
            % os.path.join(HERE, "strict_qiskit"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_strict_qiskit_circuits_on_device():
    """the same worker through libqsv.so: QCMRF built on the strict double, nested and transpiled, all fusion
    levels against the closed form, and run_experiment.main (transpile branch) end to end on device 0"""
    _run("gpu", 900)
