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

import cmath
import math
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
        self.flat = None              # basis-gate circuits: what the flat walk already knows about each wire (see _walk_flat)


def _index_of(circuit, bit, cache):
    k = id(bit)
    if k not in cache:
        cache[k] = circuit.find_bit(bit).index
    return cache[k]


def _cbit(circuit, ci_map, bit, cache):
    i = ci_map.get(id(bit)) if ci_map is not None else None
    return _index_of(circuit, bit, cache) if i is None else i


def _bit_maps(circuit):
    """id(bit) -> index for the circuit's qubits and clbits (one pass; find_bit only as fallback)"""
    qs = getattr(circuit, "qubits", None)
    cs = getattr(circuit, "clbits", None)
    if qs is None:
        return None, None
    return {id(b): i for i, b in enumerate(qs)}, {id(b): i for i, b in enumerate(cs or ())}


def _unpack(ci):
    """CircuitInstruction (new style) or (op, qargs, cargs) tuple (legacy)"""
    try:
        return ci.operation, ci.qubits, ci.clbits
    except AttributeError:
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


def _h_x(ops, op, q): ops.append(ir.Op("x", target=q[0]))
def _h_fixed(name):
    m = ir.FIXED_1Q[name]
    return lambda ops, op, q: ops.append(ir.Op("u", target=q[0], mat=m, label=name))
def _h_phase(lam):
    tab = np.array([1.0, np.exp(1j * lam)], dtype=np.complex128)
    return lambda ops, op, q: ops.append(ir.Op("diag", qubits=(q[0],), table=tab))
def _h_p(ops, op, q): ops.append(ir.op_phase1(q[0], _fparams(op)[0]))
_RZ_TABLES = {}


def _h_rz(ops, op, q):
    lam = _fparams(op)[0]
    tab = _RZ_TABLES.get(lam)            # a lowered circuit repeats a handful of angles thousands of times
    if tab is None:
        if len(_RZ_TABLES) > 4096:
            _RZ_TABLES.clear()
        tab = np.array([np.exp(-0.5j * lam), np.exp(0.5j * lam)], dtype=np.complex128)
        tab.setflags(write=False)
        _RZ_TABLES[lam] = tab
    ops.append(ir.Op("diag", qubits=(q[0],), table=tab))
def _h_rx(ops, op, q): ops.append(ir.op_u(q[0], ir.rx(_fparams(op)[0]), label="rx"))
def _h_ry(ops, op, q): ops.append(ir.op_u(q[0], ir.ry(_fparams(op)[0]), label="ry"))
def _h_u(ops, op, q): ops.append(ir.op_u(q[0], ir.u3(*_fparams(op)[:3]), label="u"))
def _h_u2(ops, op, q):
    ph, lam = _fparams(op)[:2]
    ops.append(ir.op_u(q[0], ir.u3(np.pi / 2, ph, lam), label="u"))
def _h_mcx(ops, op, q):
    # controlled X family: all but the last qubit are controls (Qiskit argument order)
    ops.append(ir.Op("x", target=q[-1], ctrls=tuple(q[:-1]), vals=tuple(_ctrl_vals(op, len(q) - 1))))
def _h_cz(ops, op, q): ops.append(ir.op_mcphase(q, np.pi, _ctrl_vals(op, len(q) - 1) + [1]))
def _h_cp(ops, op, q):
    ops.append(ir.Op("mcphase", qubits=tuple(q), vals=tuple(_ctrl_vals(op, len(q) - 1)) + (1,),
                     angle=_fparams(op)[0]))
def _h_crz(ops, op, q):
    lam = _fparams(op)[0]
    v = _ctrl_vals(op, 1)[0]
    tab = np.ones(4, dtype=np.complex128)          # index = ctrl + 2*target
    tab[v] = np.exp(-0.5j * lam)
    tab[v + 2] = np.exp(0.5j * lam)
    ops.append(ir.op_diag([q[0], q[1]], tab))
def _h_c1q(name):
    base = name[1:]
    def h(ops, op, q):
        if base in ir.FIXED_1Q:
            m = ir.FIXED_1Q[base]
        elif base == "rx":
            m = ir.rx(_fparams(op)[0])
        elif base == "ry":
            m = ir.ry(_fparams(op)[0])
        else:
            pr = _fparams(op)
            m = ir.u3(*pr[:3])
            if name == "cu" and len(pr) > 3:
                m = np.exp(1j * pr[3]) * m
        ops.append(ir.op_u(q[1], m, [q[0]], _ctrl_vals(op, 1), label=name))
    return h
def _h_swap(ops, op, q):
    a, b = q
    ops.extend([ir.op_x(b, [a]), ir.op_x(a, [b]), ir.op_x(b, [a])])
def _h_nop(ops, op, q): pass


_PRIMITIVES = {"id": _h_nop, "i": _h_nop, "x": _h_x, "p": _h_p, "u1": _h_p, "rz": _h_rz, "rx": _h_rx,
               "ry": _h_ry, "u": _h_u, "u3": _h_u, "u2": _h_u2, "cz": _h_cz, "ccz": _h_cz, "cp": _h_cp,
               "cu1": _h_cp, "mcphase": _h_cp, "mcu1": _h_cp, "crz": _h_crz, "swap": _h_swap}
for _n in ir.FIXED_1Q:
    _PRIMITIVES[_n] = _h_fixed(_n)
for _n, _lam in ir.FIXED_PHASE.items():
    _PRIMITIVES[_n] = _h_phase(_lam)
for _n in ("cx", "ccx", "mcx", "mcx_gray", "c3x", "c4x"):
    _PRIMITIVES[_n] = _h_mcx
for _n in ("ch", "cy", "csx", "crx", "cry", "cu", "cu3"):
    _PRIMITIVES[_n] = _h_c1q(_n)


_MCX_NAMES = ("x", "cx", "ccx", "mcx", "mcx_gray", "c3x", "c4x")
_MCX_SET = frozenset(_MCX_NAMES)


def _flat(definition):
    """(operations, qubit argument tuples) of a definition's instructions"""
    data = definition.data
    try:
        return [ci.operation for ci in data], [ci.qubits for ci in data]
    except AttributeError:                                   # legacy (op, qargs, cargs) tuples
        return [ci[0] for ci in data], [ci[1] for ci in data]


def _unwrap(definition):
    """Qiskit nests its library circuits: ``AND(...)`` is a circuit holding ONE gate "and" on all of its qubits
    whose definition is the X..X . MCX . X..X, and ``append(circuit)`` wraps that once more (to_instruction).
    Follow such single-instruction wrappers (same qubits, same order, no condition, no phase) down to the
    definition that has the content; primitives are never opened (an ``x`` has a definition too: u3)."""
    for _ in range(8):
        data = definition.data
        if len(data) != 1:
            break
        ci = data[0]
        try:
            op, qargs = ci.operation, ci.qubits
        except AttributeError:
            op, qargs, _ = ci
        if op.name in _PRIMITIVES or getattr(op, "condition", None) is not None:
            break
        inner = getattr(op, "definition", None)
        # list == list compares element identity first: no Bit.__eq__ call when the bits are the circuit's own
        if inner is None or getattr(definition, "global_phase", 0) or list(qargs) != list(definition.qubits):
            break
        definition = inner
    return definition


def _and_key(definition):
    """Structural signature of a small flat definition -- gate names, qubit positions in instruction
    order, the control state of the middle gate -- or None if it cannot be X..X . MCX . X..X.
    Hot: a 34-qubit QCMRF circuit has 304 AND instances (two per clique state, QCMRF.py:225,227), every
    one a distinct object, but only 4 distinct signatures; everything beyond reading the signature off
    the object is memoised on it."""
    definition = _unwrap(definition)
    data = definition.data
    n = len(data)
    if not n & 1 or n > 33 or getattr(definition, "global_phase", 0):
        return None
    qs = getattr(definition, "qubits", None)
    if qs is None:
        return None
    try:
        rows = [(ci.operation, ci.qubits) for ci in data]
    except AttributeError:                                   # tuple-style instructions
        rows = [(ci[0], ci[1]) for ci in data]
    try:
        if any([o.condition for o, _ in rows]):
            return None
    except AttributeError:                                   # no .condition attribute at all (Qiskit >= 2)
        pass
    ix = dict(zip(map(id, qs), range(len(qs))))
    try:
        pos = tuple([ix[id(q)] for _, qa in rows for q in qa])
    except KeyError:                                         # bits that are equal to, but not, the definition's own
        pos = tuple([qs.index(q) for _, qa in rows for q in qa])
    return tuple([o.name for o, _ in rows]), pos, getattr(rows[n >> 1][0], "ctrl_state", None)


_SHAPES = {}


def _shape_of_key(key):
    """(ctrls, vals, target) for the signature of an X..X . MCX . X..X definition with the same X
    set on both sides, else None (memoised: a pure function of the signature)"""
    try:
        return _SHAPES[key]
    except KeyError:
        pass
    names, pos, state = key
    n = len(names)
    f = n >> 1
    res = None
    # sides: f one-qubit X gates each; the middle gate owns the positions between them
    if names[f] in _MCX_SET and all(nm == "x" for nm in names[:f] + names[f + 1:]) and len(pos) >= 2 * f + 1:
        head, mid, tail = list(pos[:f]), pos[f:len(pos) - f], list(pos[len(pos) - f:])
        ctrls, tgt = list(mid[:-1]), mid[-1]
        vals = [1] * len(ctrls) if state is None else [(int(state) >> i) & 1 for i in range(len(ctrls))]
        if (sorted(head) == sorted(tail) and len(set(head)) == f and len(set(mid)) == len(mid)
                and tgt not in head and all(x in ctrls for x in head)):
            res = (ctrls, [v ^ 1 if c in head else v for c, v in zip(ctrls, vals)], tgt)
    if len(_SHAPES) > 4096:
        _SHAPES.clear()
    _SHAPES[key] = res
    return res


def _conjugated_mcx_shape(definition):
    """(ctrls, vals, target) in the definition's own qubit numbering if it is X..X . MCX . X..X
    with the same X set on both sides, else None."""
    key = _and_key(definition)
    return None if key is None else _shape_of_key(key)


_PHASE_MASKS = {}


def _phase_mask(k, y):
    """float mask over the 2^(k+2) table entries (controls, scratch, other) on which the triple
    with clique state y puts its phase: other = 1 and scratch xor [controls == y] = 1"""
    m = _PHASE_MASKS.get((k, y))
    if m is None:
        j = np.arange(2 ** (k + 2))
        cbits = j & (2 ** k - 1)
        tb = (j >> k) & 1
        ob = (j >> (k + 1)) & 1
        m = (ob & (tb ^ (cbits == y))).astype(np.float64)
        m.setflags(write=False)
        _PHASE_MASKS[(k, y)] = m
    return m


_BLOCKS = {}


def _block_of_key(key):
    """(ctrls, scratch, other, mask matrix) for the signature of a definition made of nothing but
    ``AND . cp . AND`` triples on one (controls, scratch, other) set, else None (memoised)"""
    try:
        return _BLOCKS[key]
    except KeyError:
        pass
    names, lens, pos, and_keys = key
    n = len(names)
    res = None
    starts = [0]
    for ln in lens:
        starts.append(starts[-1] + ln)
    wires = None
    ys = []
    ok = True
    for k in range(0, n, 3):
        if names[k + 1] not in ("cp", "cu1") or lens[k + 1] != 2:
            ok = False
            break
        sa, sb = _shape_of_key(and_keys[k]), _shape_of_key(and_keys[k + 2])
        if sa is None or sb is None or sa[1] != sb[1]:
            ok = False
            break
        la, lb = pos[starts[k]:starts[k + 1]], pos[starts[k + 2]:starts[k + 3]]
        if len(la) <= max(sa[0] + [sa[2]]) or len(lb) <= max(sb[0] + [sb[2]]):
            ok = False
            break
        ctrls, tgt = [la[c] for c in sa[0]], la[sa[2]]
        if ctrls != [lb[c] for c in sb[0]] or tgt != lb[sb[2]]:
            ok = False
            break
        p0, p1 = pos[starts[k + 1]:starts[k + 2]]
        other = p0 if p1 == tgt else p1 if p0 == tgt else -1
        if other < 0 or other == tgt or other in ctrls:
            ok = False
            break
        if wires is None:
            wires = (ctrls, tgt, other)
        elif wires != (ctrls, tgt, other):
            ok = False
            break
        ys.append(sum(v << e for e, v in enumerate(sa[1])))
    if ok and wires is not None:
        kk = len(wires[0])
        n_src = n // 3 + sum(len(and_keys[k][0]) for k in range(0, n, 3)) + sum(len(and_keys[k][0]) for k in range(2, n, 3))
        res = (wires[0], wires[1], wires[2], np.stack([_phase_mask(kk, y) for y in ys], axis=1), n_src)
    if len(_BLOCKS) > 1024:
        _BLOCKS.clear()
    _BLOCKS[key] = res
    return res


def _emit_phase_block(definition, qmap, out):
    """A definition that is nothing but ``AND . cp . AND`` triples on one (controls, scratch, other)
    set -- the reference's ``cU_C`` and its inverse (QCMRF.py:218-228,234) -- is a diagonal:
    each triple puts e^{i lam} on { other = 1 and scratch xor [controls == y] = 1 }.  Emit that one
    table instead of 3 x 2^|C| gates (exact: MCX . D . MCX with D diagonal is diagonal).

    The whole block is recognised in ONE step: its structural signature (names, qubit positions,
    the signatures of its AND definitions) is read off the objects and looked up; only the phase
    angles are per-instance.  The signature is content, not object identity: nothing is cached on
    or about the caller's circuit."""
    data = definition.data
    n = len(data)
    if n < 3 or n % 3 or getattr(definition, "global_phase", 0):
        return False
    qs = getattr(definition, "qubits", None)
    if qs is None:
        return False
    ops, qargs = _flat(definition)
    for p in ops[1::3]:
        if getattr(p, "condition", None) is not None or getattr(p, "ctrl_state", None) not in (None, 1):
            return False
    # every AND is read afresh, compute and uncompute alike: QCMRF.py:225,227 builds a new object per append
    t = n // 3
    ands = ops[0::3] + ops[2::3]
    try:
        if any([o.condition for o in ands]):
            return False
    except AttributeError:                                  # no .condition attribute at all (Qiskit >= 2)
        pass
    defs = [getattr(o, "definition", None) for o in ands]
    if any([d is None for d in defs]):
        return False
    keys = [_and_key(d) for d in defs]
    if any([k is None for k in keys]):
        return False
    and_keys = [None] * n
    and_keys[0::3] = keys[:t]
    and_keys[2::3] = keys[t:]
    qi = {id(b): i for i, b in enumerate(qs)}
    try:
        pos = tuple([qi[id(q)] for qa in qargs for q in qa])
    except KeyError:                                        # bits equal to, but not, the definition's own objects
        pos = tuple([definition.find_bit(q).index for qa in qargs for q in qa])
    key = (tuple([o.name for o in ops]), tuple([len(qa) for qa in qargs]), pos, tuple(and_keys))
    blk = _block_of_key(key)
    if blk is None:
        return False
    ctrls, tgt, other, M, n_src = blk
    glob = [qmap[x] for x in ctrls]
    glob.append(qmap[tgt])
    glob.append(qmap[other])
    if out._measured and out._measured.intersection(glob):
        return False
    # the table is exp(i M lambda); the exponentials of ALL blocks of a circuit are taken in one call
    # when the walk is over (ingest -> _finish_phase_blocks)
    op = ir.Op("diag", qubits=tuple(glob), table=None)
    out.ops.append(op)
    out._phase_blocks.append((op, M, [_fparams(p)[0] for p in ops[1::3]]))
    out.n_source_ops += n_src
    return True


def _emit_conjugated_mcx(definition, qmap, out):
    """``X..X . MCX . X..X`` with the same X set on both sides (Qiskit's AND gate with negative
    flags, QCMRF.py:224-225) is one MCX with negated controls: emit that single op."""
    data = definition.data
    n = len(data)
    if n < 1 or n % 2 == 0 or n > 33 or getattr(definition, "global_phase", 0):
        return False
    f = n // 2
    mid, mq, _ = _unpack(data[f])
    if mid.name not in _MCX_NAMES or getattr(mid, "condition", None) is not None:
        return False
    qi, _ = _bit_maps(definition)
    if qi is None:
        return False
    head, tail = [], []
    for k in range(f):
        a, aq, _ = _unpack(data[k])
        b, bq, _ = _unpack(data[n - 1 - k])
        if a.name != "x" or b.name != "x" or len(aq) != 1 or len(bq) != 1:
            return False
        head.append(qi[id(aq[0])])
        tail.append(qi[id(bq[0])])
    if f and (sorted(head) != sorted(tail) or len(set(head)) != f):
        return False
    mq = [qi[id(b)] for b in mq]
    ctrls, tgt = mq[:-1], mq[-1]
    if tgt in head or any(x not in ctrls for x in head):
        return False
    vals = _ctrl_vals(mid, len(ctrls))
    if f:
        vals = [v ^ 1 if c in head else v for c, v in zip(ctrls, vals)]
    if out._measured and out._measured.intersection(qmap[i] for i in mq):
        return False                      # let the generic walk raise the precise error
    out.ops.append(ir.Op("x", target=qmap[tgt], ctrls=tuple(qmap[c] for c in ctrls), vals=tuple(vals)))
    out.n_source_ops += n
    return True


def _walk(circuit, qmap, cmap, out, depth):
    if depth > 32:
        raise ValueError("instruction definitions nest deeper than 32 levels")
    cache = {}
    out.global_phase += float(getattr(circuit, "global_phase", 0.0) or 0.0)
    qi, ci_map = _bit_maps(circuit)
    data = circuit.data
    try:
        triples = [(ci.operation, ci.qubits, ci.clbits) for ci in data]
    except AttributeError:                                   # legacy (op, qargs, cargs) tuples
        triples = [_unpack(ci) for ci in data]
    primitives = _PRIMITIVES
    for op, qargs, cargs in triples:
        try:
            q = [qmap[qi[id(b)]] for b in qargs]
        except (KeyError, TypeError):                        # no .qubits list, or bits that are equal to but not the circuit's own objects
            q = [qmap[_index_of(circuit, b, cache)] for b in qargs]
        name = op.name
        if getattr(op, "condition", None) is not None:
            raise ValueError("classically conditioned operation %r is not supported" % name)
        if name == "measure":
            c = [cmap[_cbit(circuit, ci_map, b, cache)] for b in cargs]
            out.measure[c[0]] = q[0]
            out._measured.add(q[0])
            out.n_source_ops += 1
            if out.keep_measures:                       # trajectory mode needs WHEN it happens
                out.ops.append(ir.Op("measure", target=q[0], mask=c[0]))
            continue
        if name in ("barrier", "delay"):
            continue
        if name == "reset":
            raise ValueError("reset is not supported (deferred-measurement engine)")
        touched = out._measured.intersection(q) if out._measured else None
        if touched:
            raise ValueError("gate %r acts on qubit %d after it was measured; mid-circuit measurement "
                             "with later use of the qubit is not supported" % (name, sorted(touched)[0]))
        handler = primitives.get(name)
        if handler is None and "_o" in name:              # Qiskit open-control naming: ccx_o1 ...
            m = _OPEN_CTRL.match(name)
            handler = primitives.get(m.group(1)) if m else None
        if handler is not None:
            handler(out.ops, op, q)
            out.n_source_ops += 1
            continue
        definition = getattr(op, "definition", None)
        if definition is not None:
            if out._part is not None and depth == 0 and out.peephole:
                # one process per GPU: the composite blocks of the top level are read by ONE rank each (round robin) and
                # exchanged afterwards (ingest: _exchange_blocks) -- reading them is most of the compile time, and every
                # rank reading all of them is host time that does not shrink with the rank count
                blk = out._nblk
                out._nblk += 1
                if blk % out._part[1] != out._part[0]:
                    ph = ir.Op("diag", qubits=(), table=None)
                    out.ops.append(ph)
                    out._remote[blk] = ph
                    continue
                n0, s0 = len(out.ops), out.n_source_ops
                if _emit_phase_block(definition, q, out) and len(out.ops) == n0 + 1:
                    out._mine.append((blk, out.ops[-1], out.n_source_ops - s0))
                    continue
                out._bail = True                          # not a block that comes back as one table: everyone reads everything
                return
            if out.peephole and (_emit_phase_block(definition, q, out) or _emit_conjugated_mcx(definition, q, out)):
                continue
            c = [cmap[_cbit(circuit, ci_map, b, cache)] for b in cargs]
            _walk(definition, q, c, out, depth + 1)
            continue
        to_matrix = getattr(op, "to_matrix", None)
        if to_matrix is not None and len(q) <= 5:
            out.ops.append(ir.op_kq(q, np.asarray(to_matrix(), dtype=np.complex128)))
            out.n_source_ops += 1
            continue
        raise ValueError("unsupported operation %r on %d qubit(s): not a primitive of this engine and it "
                         "carries no definition" % (name, len(q)))


_FLAT_NAMES = frozenset(("rz", "sx", "x", "cx", "id", "measure", "barrier", "delay"))
class _NotCompact(Exception):
    """the flat walk met something it has no compact record for: redo it with ir.Op objects"""


_PI = math.pi
_TWO_PI = 2.0 * math.pi
_HALF_PI = 0.5 * math.pi
_QUARTER_PI = 0.25 * math.pi


def _wrap(a):
    return a - _TWO_PI * round(a / _TWO_PI)


def _walk_flat(circuit, out, compact=False):
    """A circuit already lowered to the reference's basis {cx, id, rz, sx, x} (run_experiment.py:52) is
    thousands of one-qubit gates between CX gates.  Every maximal run of them on a wire is kept as what it IS
    while walking -- a global phase, an X flag and the phase angles between its Hadamards,

        e^{ig} X^f D(a_n) H D(a_{n-1}) H ... H D(a_0),      D(t) = diag(1, e^{it}),

    so that ``rz`` is one float addition, ``x`` a flag (or a pi on the angle behind the last H) and ``sx`` =
    e^{i pi/4} D(-pi/2) H D(-pi/2) appends an angle; H D(0) H = 1 and H D(pi) H = X fold on the spot.  The run is emitted
    where it BEGINS (right behind the previous gate on its wire) as ONE op that carries its angles (ir.classify_1q
    forms: 'D', 'A', 'h', 'G'; matrices are built only if somebody asks): 5.7-7.4 k instructions of the 34-qubit
    circuit become ~3 k ops without a single complex multiplication.  Exact; used at fusion >= 1 only (fusion 0
    promises the stream gate by gate).  Returns False (nothing emitted) if the circuit is not flat.
    compact: ``out.ops`` holds plain tuples (unlower's gate records) instead of ir.Op objects -- what the symbolic
    re-assembly reads, at a fraction of the cost of 5 k objects; ``unlower.rec_to_op`` turns one into the op it stands for."""
    data = circuit.data
    try:
        gates = [ci.operation for ci in data]
        names = [o.name for o in gates]
    except AttributeError:
        return False
    if not _FLAT_NAMES.issuperset(names):
        return False
    qi, ci_map = _bit_maps(circuit)
    if qi is None:
        return False
    qargs = [ci.qubits for ci in data]
    ops = out.ops
    pend, slot = {}, {}                    # wire -> [g, f, a_0, a_1, ...]; wire -> index of its placeholder in ops
    measured = out._measured
    # what the passes would otherwise find by walking the op list again (passes.split_leading / hoist_leading):
    touched = set()                        # wires something has been placed on
    leadslot = {}                          # wire -> index of the run that opens it (nothing on the wire before it)
    role = {}                              # wire -> 'c' / 't': how its first two-qubit gate uses it
    dense = set()                          # wires that are a dense target after their opening run (cx target, non-diagonal run)
    hrest = {}                             # wire -> Hadamard-like gates after its opening run (a general 2x2 counts as two)
    gph = 0.0
    cxs = {}

    def flush(q):
        nonlocal gph
        st = pend.pop(q, None)
        if st is None:
            return
        i = slot.pop(q)
        g, f = st[0], st[1]
        nh = len(st) - 3
        if nh == 0:
            a = _wrap(st[2])
            if f:
                if i != leadslot.get(q):
                    dense.add(q)
                if abs(a) < 1e-15:
                    gph += g
                    ops[i] = ("x", q) if compact else ir.Op("x", target=q)
                else:
                    ops[i] = ("a", q, _wrap(g), a) if compact else ir.Op("u", target=q, cls=("A", _wrap(g), a), label="run")
            elif abs(a) < 1e-15:
                gph += g                   # the run multiplied out to a number
            else:
                ops[i] = ("d", q, _wrap(g), a) if compact else ir.Op("diag", qubits=(q,), cls=("D", _wrap(g), a))
            return
        opening = leadslot.get(q) == i
        if not opening:
            dense.add(q)
        if nh == 1:
            a1, a0 = st[3], st[2]
            cls = ("h", _wrap(g + a1), _wrap(-a1), _wrap(a0 + _PI)) if f else ("h", _wrap(g), _wrap(a1), _wrap(a0))
        elif nh == 2:
            a2, a1, a0 = st[4], st[3], st[2]
            cls = ("G", _wrap(g + a2), _wrap(a1 + _PI), _wrap(a0), _wrap(-a2)) if f else ("G", _wrap(g), _wrap(a1), _wrap(a0), _wrap(a2))
        else:
            # three or more Hadamards left in one run (generic angles between them): multiply it out after all
            if compact:
                raise _NotCompact()
            m = np.array([[1.0, 0.0], [0.0, cmath.exp(1j * st[2])]], dtype=np.complex128)
            for a in st[3:]:
                m = np.array([[1.0, 0.0], [0.0, cmath.exp(1j * a)]]) @ ir.FIXED_1Q["h"] @ m
            if f:
                m = m[::-1, :]
            m = cmath.exp(1j * g) * m
            cls = ir.classify_1q(complex(m[0, 0]), complex(m[0, 1]), complex(m[1, 0]), complex(m[1, 1]))
            if cls is None:
                ops[i] = ir.Op("u", target=q, mat=np.ascontiguousarray(m), label="run")
                return
            if cls[0] in ("D", "A"):       # (cannot happen for three honest Hadamards, but stay exact)
                ops[i] = ir.Op("diag", qubits=(q,), cls=cls) if cls[0] == "D" else ir.Op("u", target=q, cls=cls, label="run")
                return
        if compact:
            ops[i] = ("h", q, cls[1], cls[2], cls[3]) if cls[0] == "h" else ("g", q, cls[1], cls[2], cls[3], cls[4])
        else:
            ops[i] = ir.Op("u", target=q, cls=cls, label="run")
        if not opening:
            hrest[q] = hrest.get(q, 0) + (1 if cls[0] == "h" else 2)

    out.global_phase += float(getattr(circuit, "global_phase", 0.0) or 0.0)
    n_src = 0
    try:
        conds = [o for o in gates if o.condition is not None]
    except AttributeError:                                   # no .condition attribute at all (Qiskit >= 2)
        conds = [o for o in gates if getattr(o, "condition", None) is not None]
    if conds:
        raise ValueError("classically conditioned operation %r is not supported" % conds[0].name)
    pget = pend.get
    for ci, name, gate, qa in zip(data, names, gates, qargs):
        if name == "rz" or name == "sx" or name == "x":
            q = qi[id(qa[0])]
            if measured and q in measured:
                raise ValueError("gate %r acts on qubit %d after it was measured; mid-circuit measurement "
                                 "with later use of the qubit is not supported" % (name, q))
            n_src += 1
            st = pget(q)
            if st is None:
                st = pend[q] = [0.0, 0, 0.0]
                if q not in touched:
                    touched.add(q)
                    leadslot[q] = len(ops)
                slot[q] = len(ops)
                ops.append(None)
            if name == "rz":
                lam = gate.params[0]
                if type(lam) is not float:
                    try:
                        lam = float(lam)
                    except TypeError:
                        raise ValueError("unbound or non-numeric parameter %r in gate 'rz'" % (lam,))
                # rz(lam) = e^{-i lam/2} D(lam);  D(lam) X = e^{i lam} X D(-lam)
                if st[1]:
                    st[-1] -= lam
                    st[0] += 0.5 * lam
                else:
                    st[-1] += lam
                    st[0] -= 0.5 * lam
            elif name == "x":
                st[1] ^= 1
            else:
                # sx = e^{i pi/4} D(-pi/2) H D(-pi/2);  with an X in front of the run:  H X = D(pi) H
                if st[1]:
                    st[1] = 0
                    st[-1] += _HALF_PI
                    st[0] -= _QUARTER_PI
                    new = _HALF_PI
                else:
                    st[-1] -= _HALF_PI
                    st[0] += _QUARTER_PI
                    new = -_HALF_PI
                if len(st) > 3:
                    mid = st[-1]
                    mid -= _TWO_PI * round(mid / _TWO_PI)
                    if -1e-13 < mid < 1e-13:                 # H D(0) H = 1
                        st.pop()
                        st[-1] += new
                        continue
                    if abs(abs(mid) - _PI) < 1e-13:          # H D(pi) H = X;  D(new) X = e^{i new} X D(-new)
                        st.pop()
                        st[-1] -= new
                        st[0] += new
                        st[1] = 1
                        continue
                st.append(new)
            continue
        if name == "cx":
            q, t = qi[id(qa[0])], qi[id(qa[1])]
            if measured and (q in measured or t in measured):
                raise ValueError("gate 'cx' acts on qubit %d after it was measured; mid-circuit measurement "
                                 "with later use of the qubit is not supported" % (q if q in measured else t))
            n_src += 1
            if q in pend:
                flush(q)
            if t in pend:
                flush(t)
            if q not in role:
                role[q] = "c"
                touched.add(q)
            if t not in role:
                role[t] = "t"
                touched.add(t)
            dense.add(t)
            o = cxs.get((q, t))                               # one op object per (control, target) of THIS walk: the passes read them, never write
            if o is None:
                o = cxs[(q, t)] = ("c", q, t) if compact else ir.Op("x", target=t, ctrls=(q,), vals=(1,))
            ops.append(o)
            continue
        if name == "measure":
            q = qi[id(qa[0])]
            n_src += 1
            flush(q)
            c = ci_map[id(ci.clbits[0])]
            out.measure[c] = q
            measured.add(q)
            if out.keep_measures:
                ops.append(ir.Op("measure", target=q, mask=c))
            continue
        if name == "id":
            n_src += 1
    for q in sorted(pend):
        flush(q)
    out.global_phase += gph
    lead = {q: ops[i] for q, i in leadslot.items() if ops[i] is not None}
    out.ops = [o for o in ops if o is not None]
    out.flat = {"lead": lead, "role": role, "dense": dense, "hrest": hrest, "compact": bool(compact)}
    out.n_source_ops += n_src
    return True


def ingest(circuit, peephole=False, keep_measures=False, comm=None, compact=False):
    """peephole=True additionally folds X..X . MCX . X..X definitions (Qiskit's AND with negative
    flags) into one MCX with negated controls while walking -- exact, and 5x fewer ops to fuse.
    comm (a process group of world > 1, every rank calling with the same circuit): the composite top-level blocks are
    read by one rank each and exchanged in ONE all-gather; the result is the same Ingested on every rank.
    compact (basis-gate circuits only): ``ops`` are unlower's gate records (tuples), ``flat["compact"]`` says so; only
    ``passes.optimise(..., flat=)`` at level 3 understands them."""
    nq = int(circuit.num_qubits)
    nc = int(getattr(circuit, "num_clbits", 0))
    out = Ingested(nq, nc)
    out.peephole = bool(peephole)
    out.keep_measures = bool(keep_measures)
    out._measured = set()
    out._phase_blocks = []
    out._part = (comm.rank, comm.world) if (comm is not None and comm.world > 1 and peephole and not keep_measures) else None
    out._nblk, out._remote, out._mine, out._bail, out._exchanged = 0, {}, [], False, False
    flat = False
    if peephole and not keep_measures:
        try:
            flat = _walk_flat(circuit, out, compact=compact and not keep_measures)
        except _NotCompact:
            out.ops, out.measure, out._measured, out.global_phase, out.n_source_ops, out.flat = [], {}, set(), 0.0, 0, None
            flat = _walk_flat(circuit, out, compact=False)
        except KeyError:                     # bits that are equal to, but not, the circuit's own objects: take the general walk
            out.ops, out.measure, out._measured, out.global_phase, out.n_source_ops, out.flat = [], {}, set(), 0.0, 0, None
    if not flat:
        try:
            _walk(circuit, list(range(nq)), list(range(nc)), out, 0)
        except Exception:
            if out._part is None or out._exchanged:
                raise
            out._bail = True                 # the others are waiting in the exchange: tell them, then fail here as a single process would
    _finish_phase_blocks(out)
    if out._part is not None and not _exchange_blocks(out, comm):
        return ingest(circuit, peephole=peephole, keep_measures=keep_measures)     # some block was no phase block: read it all here
    cregs = getattr(circuit, "cregs", None)
    if cregs:
        out.creg_sizes = [(getattr(r, "name", "c"), len(r)) for r in cregs]
    del out._measured
    return out


def _exchange_blocks(out, comm):
    """every rank contributes the blocks it read -- (index, qubits, table, source-gate count) -- and fills in the ones the
    others read; False if any rank met a block it could not turn into one table (then nobody uses the exchange)"""
    mine = [] if out._bail else [[blk, list(op.qubits), np.ascontiguousarray(op.table), int(n_src)] for blk, op, n_src in out._mine]
    out._exchanged = True
    parts = comm.allgather([bool(out._bail), mine])
    if any(p[0] for p in parts):
        return False
    got = {}
    for p in parts:
        for blk, qubits, table, n_src in p[1]:
            got[blk] = (qubits, table, n_src)
    mine_ids = {blk for blk, _, _ in out._mine}
    for blk, ph in out._remote.items():
        if blk not in got:
            return False
        qubits, table, n_src = got[blk]
        if out.measure and set(out.measure.values()).intersection(qubits):
            return False                                   # let the local walk raise the precise error
        ph.qubits, ph.table, ph._sup = tuple(int(x) for x in qubits), table, None
        out.n_source_ops += n_src
    return len(got) == len(mine_ids) + len(out._remote)


def _finish_phase_blocks(out):
    """tables of the recognised phase blocks: angle vectors of all blocks, ONE exp over them all"""
    blocks = out._phase_blocks
    del out._phase_blocks
    if not blocks:
        return
    angs = [M @ np.asarray(lam, dtype=np.float64) for _, M, lam in blocks]
    tab = np.exp(1j * np.concatenate(angs))
    off = 0
    for (op, _, _), a in zip(blocks, angs):
        op.table = tab[off:off + a.size]
        off += a.size
