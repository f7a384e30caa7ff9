"""``transpile(circuits, basis_gates=[...])`` of the test double: the basis translation is the
package's own stand-in (qcmrf_amd.transpile -- Qiskit's transpiler cannot be reproduced here), followed
by the level-1 clean-up shapes of tests/_qiskit_shapes.py, emitted as objects of THIS double, so the
backend is handed strict CircuitInstruction / Qubit objects and a non-zero global phase."""


def transpile(circuits, backend=None, basis_gates=None, coupling_map=None, initial_layout=None,
              optimization_level=None, seed_transpiler=None):
    if backend is not None or coupling_map is not None or initial_layout is not None:
        raise NotImplementedError("strict double: only transpile(circuits, basis_gates=...) is modelled")
    from qcmrf_amd.transpile import transpile as lower
    from .circuit import QuantumCircuit
    single = not isinstance(circuits, (list, tuple))
    outs = []
    for c in ([circuits] if single else circuits):
        t = lower(c, basis_gates=basis_gates, circuit_class=QuantumCircuit)
        if optimization_level is None or optimization_level >= 1:
            import _qiskit_shapes as shapes
            t = shapes.cancel_adjacent_cx(t)
            t = shapes.merge_1q_runs(t)
            t = shapes.cancel_adjacent_cx(t)
        outs.append(t)
    return outs[0] if single else outs
