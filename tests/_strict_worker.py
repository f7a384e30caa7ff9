"""TEST INFRASTRUCTURE: runs in a child process with tests/strict_qiskit FIRST on sys.path, so that
``import qiskit`` resolves to the strict, signature-faithful double and qcmrf_amd takes its
"Qiskit is importable" branch -- the reference's real environment (QCMRF.py:6-9, run_experiment.py:10-14).

    python tests/_strict_worker.py cpu     host path on the numpy stand-in engine (oracle/sharded_numpy.py)
    python tests/_strict_worker.py gpu     the same through libqsv.so on device 0
"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(HERE, "strict_qiskit")):
    if p in sys.path:
        sys.path.remove(p)
    sys.path.insert(0, p)

import numpy as np                                   # noqa: E402
import qiskit                                        # noqa: E402
assert "strict-double" in qiskit.__version__, qiskit.__file__
import qcmrf_amd                                     # noqa: E402
from qcmrf_amd import QCMRF, Aer, run_experiment, workloads   # noqa: E402
from qcmrf_amd.backend import QsvBackend             # noqa: E402
from qcmrf_amd.ingest import ingest                  # noqa: E402
from oracle import closed_form as cf, gate_stream as gs       # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "cpu"
assert qcmrf_amd.HAVE_QISKIT is True
assert qiskit.QuantumCircuit in QCMRF.__mro__, QCMRF.__mro__
from qcmrf_amd import qcmrf as _q                    # noqa: E402
assert _q.AND is qiskit.circuit.library.AND          # the real(-shaped) AND, not the in-tree one

models = json.load(open(os.path.join(HERE, "golden", "models_0.5.json")))
GRAPHS = models["GRAPHS"]
assert GRAPHS == workloads.REFERENCE_GRAPHS


def make_backend():
    be = QsvBackend()
    if mode == "cpu":
        from oracle.sharded_numpy import NumpyEngine
        be._engine_factory = lambda n, devices=(0,), rank=None, world_size=None: NumpyEngine(n, len(devices))
    return be


# 1. the constructor builds under Qiskit's signatures: inverse() without arguments, append(instruction, qargs),
#    a fresh AND per append, per-instruction inverses -- and the objects nest as Qiskit nests them
for j, C in enumerate(GRAPHS):
    th = models["THETAS"][str(j)][2]
    qc = QCMRF(C, th, with_measurements=True)
    n, m, W, dim = cf.model_shape(C)
    assert qc.num_qubits == W and qc.num_clbits == W and qc.dimension == dim
    names = [ci.operation.name for ci in qc.data]
    want = ["h"] * n
    for ii in range(m):
        want += ["h", "cU_C%d" % ii, "x", "cU_C%d_dg" % ii, "x", "h", "measure"]
    want += ["measure"] * n
    assert names == want, (names, want)
    ands = []
    for ci in qc.data:
        assert type(ci).__name__ == "CircuitInstruction" and isinstance(ci.qubits, tuple)
        try:
            iter(ci)
            raise SystemExit("CircuitInstruction must not unpack like a tuple")
        except TypeError:
            pass
        if ci.operation.name.startswith("cU_C"):
            sub = ci.operation.definition
            dg = ci.operation.name.endswith("_dg")
            assert [c.operation.name for c in sub.data] == (["and_dg", "cp", "and_dg"] if dg else ["and", "cp", "and"]) * (2 ** len(C[0]) if len(set(map(len, C))) == 1 else len(sub.data) // 3)
            for c in sub.data:
                if c.operation.name.startswith("and"):
                    ands.append(c.operation)
                    wrapped = c.operation.definition.data
                    assert len(wrapped) == 1 and wrapped[0].operation.name == c.operation.name     # AND holds ONE gate "and"
                    inner = [x.operation.name for x in wrapped[0].operation.definition.data]
                    assert inner[len(inner) // 2] in ("cx", "ccx", "mcx") and set(inner) <= {"x", "cx", "ccx", "mcx"}, inner
    assert len(set(map(id, ands))) == len(ands) == 4 * dim                       # no instruction object is shared

    # 2. ingest, gate by gate: the reference's stream, op for op (the oracle's independent restatement of QCMRF.py:199-243)
    ing = ingest(qc)
    ref = [o for o in gs.reference_stream(C, th) if o[0] != "measure"]
    assert len(ing.ops) == len(ref), (len(ing.ops), len(ref))
    for op, r in zip(ing.ops, ref):
        if r[0] == "h":
            assert op.kind == "u" and op.target == r[1]
        elif r[0] == "x":
            assert op.kind == "x" and op.target == r[1] and not op.ctrls
        elif r[0] == "mcx":
            assert op.kind == "x" and op.ctrls == tuple(r[1]) and op.target == r[2] and all(op.vals)
        elif r[0] == "cp":
            assert op.kind == "mcphase" and op.qubits == (r[2], r[3]) and op.angle == r[1]
    assert ing.measure == {**{n + 1 + i: n + 1 + i for i in range(m)}, **{q: q for q in range(n)}}

    # 3. every fusion level, nested and transpiled, amplitudes against the closed form
    be = make_backend()
    want_amp = cf.amplitudes(C, th)
    lowered = qiskit.transpile(qc, basis_gates=['cx', 'id', 'rz', 'sx', 'x'])
    assert isinstance(lowered, qiskit.QuantumCircuit) and set(lowered.count_ops()) <= {"cx", "id", "rz", "sx", "x", "measure"}
    for circ, levels in ((qc, (0, 1, 2, 3)), (lowered, (0, 3))):
        for fusion in levels:
            res = be.run(circ, shots=256, seed_simulator=7, fusion=fusion).result()
            assert sum(res.get_counts().values()) == 256
            err = float(np.abs(be.statevector() - want_amp).max())
            assert err < 1e-12, (j, fusion, err)
    # the fused program is the one the in-tree container gives: init + one diagonal factor per clique
    _, pl = be.compile(qc, 1)
    assert [o.kind for o in pl.ops] == ["init"] + ["diag"] * m, [o.kind for o in pl.ops]
    be.close()

# 4. run_experiment.main end to end: HAVE_QISKIT -> ``from qiskit import transpile`` -> lowered strict circuits -> counts
be = Aer.get_backend('qasm_simulator')
if mode == "cpu":
    from oracle.sharded_numpy import NumpyEngine
    be._engine_factory = lambda n, devices=(0,), rank=None, world_size=None: NumpyEngine(n, len(devices))
with tempfile.TemporaryDirectory() as tmp:
    counts = run_experiment.main(["--scale", "0.5", "--shots", "4000", "--reps", "2", "--outdir", tmp, "--seed-simulator", "11"])
    saved = json.load(open(os.path.join(tmp, "result_simulation_0.5.json")))
    mdl = json.load(open(os.path.join(tmp, "models_0.5.json")))
assert saved == counts and len(counts) == 14
for j, C in enumerate(GRAPHS):
    for i in range(2):
        if j == 0:                                                               # same RNG law, same first draws (graph-major order)
            assert mdl["THETAS"]["0"][i] == models["THETAS"]["0"][i]
        p = cf.probabilities(C, mdl["THETAS"][str(j)][i])
        c = counts[2 * j + i]
        assert sum(c.values()) == 4000
        obs = np.zeros_like(p)
        for k, v in c.items():
            assert p[int(k, 2)] > 0, k
            obs[int(k, 2)] = v
        big = p * 4000 >= 5
        chi2 = float(((obs[big] - 4000 * p[big]) ** 2 / (4000 * p[big])).sum())
        dof = int(big.sum())
        assert chi2 < dof + 6 * np.sqrt(2 * dof) + 10, (j, i, chi2, dof)
be.close()
print("strict-qiskit worker ok (%s)" % mode)
