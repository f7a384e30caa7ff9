"""Trajectory mode: take every mid-circuit measurement when it occurs and release the qubit.

The reference measures each clique's ancilla right after its block and never touches it again
(/root/reference/QCMRF.py:238-239).  Deferred to the end that costs one qubit per clique
(W = n + m + 1).  Taken when it occurs, the ancilla's slot can be recycled: the live state is only
the variables, the scratch qubit and ONE ancilla (n + 2 qubits), so MRFs with far more cliques
than any statevector of width W could hold become simulable -- at the price of following the
measurement outcomes: the engine walks the tree of outcomes depth first, splitting the shots at
every measurement by a binomial draw from the exact branch probability, and only visits branches
that still hold shots.  Everything on the device is the same hand-written HIP path (the segments
between measurements are compiled by the same exact fusion passes and run through ``qsv_exec``;
the branch probability is ``qsv_probabilities``, the collapse a one-qubit 0/1 diagonal that rides in
front of the child segment's program -- one pass, the state stays unnormalised -- the branch point
``qsv_copy_state``).  No closed-form knowledge of the circuit is used.
"""
from __future__ import annotations

import time

import numpy as np

from . import _lib, ingest as _ingest, ir, passes, planner, program


_ONE = np.array([0.0, 1.0])                               # the observable |1><1| of a measured slot


class _Segment:
    # prog[outcome]: this segment's program behind the projection of the PREVIOUS segment's measurement on ``outcome``
    # (and the X that hands a released slot back in |0>), in one record list: one pass over the state instead of three
    __slots__ = ("rec", "data", "n_ops", "measure_slot", "measure_clbit", "release", "prog")


def _live_plan(ops, n_qubits):
    """first/last use of every logical qubit; slot assignment with recycling"""
    first, last = {}, {}
    for k, op in enumerate(ops):
        qs = (op.target,) if op.kind == "measure" else op.support()
        for q in qs:
            first.setdefault(q, k)
            last[q] = k
    return first, last


def compile_trajectory(circuit, fusion=3):
    """-> (segments, width, final_measures [(slot, clbit)], num_clbits, creg_sizes, n_source_ops)"""
    ing = _ingest.ingest(circuit, peephole=fusion >= 1, keep_measures=True)
    ops = ing.ops
    first, last = _live_plan(ops, ing.num_qubits)
    # a measure is a release point iff it is the last thing that happens to its qubit
    slot_of, free, next_slot = {}, [], 0
    width = 0
    segments, cur = [], []
    pending_release = []                       # slots freed by the measure that closed the previous segment
    final_measures = []

    def slot(q):
        nonlocal next_slot, width
        if q not in slot_of:
            if free:
                slot_of[q] = free.pop(0)
            else:
                slot_of[q] = next_slot
                next_slot += 1
            width = max(width, next_slot)
        return slot_of[q]

    def remap(op):
        lay = {q: slot(q) for q in op.support()}
        o = ir.Op(op.kind, target=lay.get(op.target), ctrls=tuple(lay[c] for c in op.ctrls), vals=op.vals,
                  qubits=tuple(lay[q] for q in op.qubits), mat=op.mat, table=op.table, mats=op.mats,
                  angle=op.angle, label=op.label)
        return o

    staged = []                                # (kind, payload) in order: ("ops", [...]) / ("measure", slot, clbit, release)
    for k, op in enumerate(ops):
        if op.kind == "measure":
            s = slot(op.target)
            release = last[op.target] == k
            staged.append(("ops", cur))
            staged.append(("measure", s, op.mask, release))
            cur = []
            if release:
                del slot_of[op.target]
                free.append(s)
                free.sort()
        else:
            cur.append(remap(op))
    staged.append(("ops", cur))
    # trailing measures (nothing but measures after them) are sampled jointly from the final state
    while len(staged) >= 2 and staged[-1][0] == "ops" and not staged[-1][1] and staged[-2][0] == "measure":
        staged.pop()
        _, s, c, _ = staged.pop()
        final_measures.insert(0, (s, c))
    # fuse every segment with the ordinary exact passes (no init folding after the first one)
    segs = []
    first_seg = True
    i = 0
    while i < len(staged):
        kind = staged[i][0]
        assert kind == "ops"
        body = staged[i][1]
        if first_seg:
            fused = passes.optimise(body, level=fusion) if fusion > 0 else [ir.op_init(0)] + body
        else:
            fused = passes._fuse_body([ir.op_init(0)] + body, fusion, 10, 8)[1:] if fusion > 0 else list(body)
        first_seg = False
        sg = _Segment()
        sg.rec, sg.data = program.encode(fused)
        sg.n_ops = len(fused)
        sg.measure_slot = sg.measure_clbit = None
        sg.release = False
        sg.prog = None
        if segs:
            # the state stays UNNORMALISED along a branch (the projection is the 0/1 table, not 1/sqrt(p)): branch
            # probabilities are ratios and the sampler divides by the mass it finds
            prev = segs[-1]
            sg.prog = {}
            for outcome in (0, 1):
                pre = [ir.op_diag([prev.measure_slot], [1.0 - outcome, float(outcome)])]
                if outcome == 1 and prev.release:
                    pre.append(ir.op_x(prev.measure_slot))
                sg.prog[outcome] = program.encode(pre + list(fused))
        if i + 1 < len(staged):
            _, s, c, rel = staged[i + 1]
            sg.measure_slot, sg.measure_clbit, sg.release = s, c, rel
        segs.append(sg)
        i += 2
    return segs, max(width, 1), final_measures, ing.num_clbits, ing.creg_sizes, ing.n_source_ops


def run_trajectories(circuit, shots, seed, fusion=3, device=0, engine_factory=None, max_width=33):
    """returns (values: uint64 array of classical-register integers, metadata)"""
    t0 = time.perf_counter()
    segs, width, final_measures, num_clbits, creg_sizes, n_src = compile_trajectory(circuit, fusion)
    if width > max_width:
        raise MemoryError("trajectory mode still needs %d live qubits (limit %d)" % (width, max_width))
    make = engine_factory or _lib.Engine
    rng = np.random.RandomState(seed % (2 ** 32))
    pool = []
    created = []

    def get_engine():
        if pool:
            return pool.pop()
        e = make(width, devices=(device,))
        created.append(e)
        return e

    t1 = time.perf_counter()
    out_vals, out_cnts = [], []
    stats = {"nodes": 0, "copies": 0, "sweeps": 0}
    fm_slots = [s for s, _ in final_measures]

    def node(level, eng, k, bits, came_by):
        sg = segs[level]
        stats["nodes"] += 1
        stats["sweeps"] += sg.n_ops
        rec, data = (sg.rec, sg.data) if came_by is None else sg.prog[came_by]
        if len(rec):
            eng.exec(rec, data)
        if sg.measure_slot is None:                         # leaf: joint sample of what is left
            if fm_slots:
                smp = eng.sample(k, int(rng.randint(0, 2 ** 31 - 1)), fm_slots)
                vals = np.full(k, bits, dtype=np.uint64)
                for j, (_, c) in enumerate(final_measures):
                    vals |= ((smp >> np.uint64(j)) & np.uint64(1)) << np.uint64(c)
                uv, uc = np.unique(vals, return_counts=True)
                out_vals.extend(uv.tolist())
                out_cnts.extend(uc.tolist())
            else:
                out_vals.append(bits)
                out_cnts.append(k)
            return
        # mass on outcome 1 and total mass in one read pass with a FIXED summation order (qsv_expect_diag: per-workgroup
        # partial sums, no atomics): the same seed walks the same tree on every run
        p1, tot = eng.expect_diag([sg.measure_slot], _ONE)
        k1 = int(rng.binomial(k, min(max(p1 / tot, 0.0), 1.0))) if tot > 0 else 0
        k0 = k - k1
        other = None
        if k0 > 0 and k1 > 0:
            other = get_engine()
            other.copy_from(eng)
            stats["copies"] += 1
        # the projection on the outcome (and the X that hands a released slot back) ride in front of the child's program
        for outcome, kk, e in ((0, k0, eng), (1, k1, other if other is not None else eng)):
            if kk:
                node(level + 1, e, kk, bits | (outcome << sg.measure_clbit), outcome)
        if other is not None:
            pool.append(other)

    root = get_engine()
    node(0, root, int(shots), 0, None)
    t2 = time.perf_counter()
    for e in created:
        e.close()
    vals = np.asarray(out_vals, dtype=np.uint64)
    cnts = np.asarray(out_cnts, dtype=np.int64)
    meta = {"method": "trajectory", "live_qubits": width, "n_segments": len(segs), "n_source_ops": n_src,
            "branch_nodes": stats["nodes"], "state_copies": stats["copies"], "device_ops": stats["sweeps"],
            "engines": len(created), "time_compile": t1 - t0, "time_evolve": t2 - t1}
    return vals, cnts, num_clbits, creg_sizes, meta
