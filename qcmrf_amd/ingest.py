"""Duck-typed ingest of a Qiskit-style circuit into the engine IR.

Replaces what Aer's assembler does with the circuits handed to ``simulator.run(T, shots=...)``
(/root/reference/run_experiment.py:56).  Works on any object with Qiskit's data model -- a real
``qiskit.QuantumCircuit`` (nested or transpiled), or ``qcmrf_amd.circuit.QuantumCircuit``:

    for ci in circuit.data:  ci.operation.name / .params / .definition ; ci.qubits ; ci.clbits
    circuit.find_bit(bit).index ; circuit.num_qubits ; circuit.num_clbits ; circuit.global_phase

Composite instructions (``cU_C<ii>``, ``cU_C<ii>_dg``, ``and``, ... -- QCMRF.py:218,225,232,234)
are expanded through ``operation.definition``; an operation that is neither primitive nor has a
definition raises ``ValueError`` naming it.  Measurements are deferred to the end of the circuit
(exact here: QCMRF.py never conditions on a classical bit and never touches a measured qubit
again, QCMRF.py:239); a circuit that does either is rejected.
"""
from __future__ import annotations

import re

import numpy as np

from . import ir

_OPEN_CTRL = re.compile(r"^(.*)_o(\d+)$")


class Ingested:
    def __init__(self, num_qubits, num_clbits):
        self.num_qubits = num_qubits
        self.num_clbits = num_clbits
        self.ops = []                 # ir.Op on logical qubits, program order
        self.measure = {}             # clbit index -> qubit index
        self.global_phase = 0.0
        self.creg_sizes = None        # [(name, size)] in declaration order, if the circuit has cregs
        self.n_source_ops = 0


def _index_of(circuit, bit, cache):
    k = id(bit)
    if k not in cache:
        cache[k] = circuit.find_bit(bit).index
    return cache[k]


def _unpack(ci):
    """CircuitInstruction (new style) or (op, qargs, cargs) tuple (legacy)"""
    if hasattr(ci, "operation"):
        return ci.operation, ci.qubits, ci.clbits
    op, qargs, cargs = ci
    return op, qargs, cargs


def _fparams(op):
    out = []
    for p in getattr(op, "params", ()) or ():
        try:
            out.append(float(p))
        except TypeError:
            raise ValueError("unbound or non-numeric parameter %r in gate %r" % (p, op.name))
    return out


def _ctrl_vals(op, n_ctrl):
    """control values from Qiskit's ``ctrl_state`` (bit i <-> control qubit i); default all ones"""
    state = getattr(op, "ctrl_state", None)
    if state is None:
        return [1] * n_ctrl
    return [(int(state) >> i) & 1 for i in range(n_ctrl)]


def _emit_primitive(out, name, op, q):
    """Append IR for primitive ``name`` on logical qubits ``q``; False if not primitive."""
    P = _fparams
    ops = out.ops
    if name in ("id", "i", "barrier", "delay"):
        return True
    if name == "x":
        ops.append(ir.op_x(q[0])); return True
    if name in ir.FIXED_1Q:
        ops.append(ir.op_u(q[0], ir.FIXED_1Q[name], label=name)); return True
    if name in ir.FIXED_PHASE:
        ops.append(ir.op_phase1(q[0], ir.FIXED_PHASE[name])); return True
    if name in ("p", "u1"):
        ops.append(ir.op_phase1(q[0], P(op)[0])); return True
    if name == "rz":
        lam = P(op)[0]
        ops.append(ir.op_diag([q[0]], [np.exp(-0.5j * lam), np.exp(0.5j * lam)])); return True
    if name == "rx":
        ops.append(ir.op_u(q[0], ir.rx(P(op)[0]), label="rx")); return True
    if name == "ry":
        ops.append(ir.op_u(q[0], ir.ry(P(op)[0]), label="ry")); return True
    if name in ("u", "u3"):
        ops.append(ir.op_u(q[0], ir.u3(*P(op)[:3]), label="u")); return True
    if name == "u2":
        ph, lam = P(op)[:2]
        ops.append(ir.op_u(q[0], ir.u3(np.pi / 2, ph, lam), label="u")); return True
    # controlled X family: all but the last qubit are controls (Qiskit argument order)
    if name in ("cx", "ccx", "mcx", "mcx_gray", "c3x", "c4x"):
        n_ctrl = len(q) - 1
        ops.append(ir.op_x(q[-1], q[:-1], _ctrl_vals(op, n_ctrl))); return True
    if name in ("cz", "ccz"):
        vals = _ctrl_vals(op, len(q) - 1) + [1]
        ops.append(ir.op_mcphase(q, np.pi, vals)); return True
    if name in ("cp", "cu1", "mcphase", "mcu1"):
        vals = _ctrl_vals(op, len(q) - 1) + [1]
        ops.append(ir.op_mcphase(q, P(op)[0], vals)); return True
    if name == "crz":
        lam = P(op)[0]
        v = _ctrl_vals(op, 1)[0]
        tab = np.ones(4, dtype=np.complex128)          # index = ctrl + 2*target
        tab[v] = np.exp(-0.5j * lam)
        tab[v + 2] = np.exp(0.5j * lam)
        ops.append(ir.op_diag([q[0], q[1]], tab)); return True
    if name in ("ch", "cy", "csx", "crx", "cry", "cu", "cu3"):
        base = name[1:]
        if base in ir.FIXED_1Q:
            m = ir.FIXED_1Q[base]
        elif base == "rx":
            m = ir.rx(P(op)[0])
        elif base == "ry":
            m = ir.ry(P(op)[0])
        else:
            pr = P(op)
            m = ir.u3(*pr[:3])
            if name == "cu" and len(pr) > 3:
                m = np.exp(1j * pr[3]) * m
        ops.append(ir.op_u(q[1], m, [q[0]], _ctrl_vals(op, 1), label=name)); return True
    if name == "swap":
        a, b = q
        ops.extend([ir.op_x(b, [a]), ir.op_x(a, [b]), ir.op_x(b, [a])]); return True
    return False


def _walk(circuit, qmap, cmap, out, depth):
    if depth > 32:
        raise ValueError("instruction definitions nest deeper than 32 levels")
    cache = {}
    out.global_phase += float(getattr(circuit, "global_phase", 0.0) or 0.0)
    for ci in circuit.data:
        op, qargs, cargs = _unpack(ci)
        q = [qmap[_index_of(circuit, b, cache)] for b in qargs]
        name = op.name
        if getattr(op, "condition", None) is not None:
            raise ValueError("classically conditioned operation %r is not supported" % name)
        if name == "measure":
            c = [cmap[_index_of(circuit, b, cache)] for b in cargs]
            out.measure[c[0]] = q[0]
            out._measured.add(q[0])
            out.n_source_ops += 1
            continue
        if name in ("barrier", "delay"):
            continue
        if name == "reset":
            raise ValueError("reset is not supported (deferred-measurement engine)")
        touched = out._measured.intersection(q)
        if touched:
            raise ValueError("gate %r acts on qubit %d after it was measured; mid-circuit measurement "
                             "with later use of the qubit is not supported" % (name, sorted(touched)[0]))
        m = _OPEN_CTRL.match(name)
        base = m.group(1) if m else name
        if _emit_primitive(out, base, op, q):
            out.n_source_ops += 1
            continue
        definition = getattr(op, "definition", None)
        if definition is not None:
            c = [cmap[_index_of(circuit, b, cache)] for b in cargs]
            _walk(definition, q, c, out, depth + 1)
            continue
        to_matrix = getattr(op, "to_matrix", None)
        if to_matrix is not None and len(q) <= 5:
            out.ops.append(ir.op_kq(q, np.asarray(to_matrix(), dtype=np.complex128)))
            out.n_source_ops += 1
            continue
        raise ValueError("unsupported operation %r on %d qubit(s): not a primitive of this engine and it "
                         "carries no definition" % (name, len(q)))


def ingest(circuit):
    nq = int(circuit.num_qubits)
    nc = int(getattr(circuit, "num_clbits", 0))
    out = Ingested(nq, nc)
    out._measured = set()
    _walk(circuit, list(range(nq)), list(range(nc)), out, 0)
    cregs = getattr(circuit, "cregs", None)
    if cregs:
        out.creg_sizes = [(getattr(r, "name", "c"), len(r)) for r in cregs]
    del out._measured
    return out
