"""Logical -> physical qubit layout and shard-exchange planning.

The amplitude vector is split into P = 2^g shards by its g highest PHYSICAL index bits
("shard by high qubit index", BASELINE.json north_star).  A gate needs communication only when
it is non-diagonal on a shard bit; diagonal gates, controls and multiplexer selects on shard bits
are resolved per shard with no traffic (libqsv does that from the shard number).

``layout="reference"``  logical qubit q sits on physical bit q.  For a QCMRF circuit the
                        per-clique ancillas -- the only dense targets after fusion -- are the
                        highest qubits (QCMRF.py:231, index n+1+ii), i.e. exactly the shard bits:
                        the worst case, every late clique costs a half-shard exchange.
``layout="auto"``       qubits that are never a dense target (for a fused QCMRF circuit: the
                        variable qubits, which after the init write only ever select) become the
                        shard bits, preferring qubits in uniform superposition so every shard
                        carries equal weight.  Zero exchanges for fused QCMRF circuits.

Before a dense gate whose target currently sits on a shard bit, a ``swap`` (shard bit <-> local
bit) is inserted: the pairwise half-shard exchange.  The evicted local qubit is the one whose
next dense use is farthest away (never, if possible).
"""
from __future__ import annotations

import copy

from . import ir


class Plan:
    def __init__(self):
        self.ops = []            # ir.Op on PHYSICAL qubits, ``swap`` ops included
        self.layout = []         # final: layout[logical] = physical
        self.initial_layout = []
        self.n_exchanges = 0
        self.n_qubits = 0
        self.n_shards = 1


def _remap(op, lay):
    o = copy.copy(op)
    if op.kind in ("u", "x", "mux"):
        o.target = lay[op.target]
        o.ctrls = tuple(lay[c] for c in op.ctrls)
    elif op.kind in ("diag", "mcphase", "kq"):
        o.qubits = tuple(lay[q] for q in op.qubits)
    elif op.kind == "init":
        m = 0
        for q in range(len(lay)):
            if (op.mask >> q) & 1:
                m |= 1 << lay[q]
        o.mask = m
    return o


LANE_BITS = 6      # address bits 0..5 are the lane id of a wavefront load (1 KiB contiguous)
MULTI_R = 5        # register targets per k_multi pass (libqsv option multi_r)


def choose_layout(ops, n_qubits, n_shards, layout="auto", lane_targets=True):
    """layout[logical] = physical.

    auto, measured on MI355X (profiles/r01_multi_bits_W28.json): a k_multi pass streams at
    5.3-5.7 TB/s when its target bits sit just above the lane bits (6..~16) and drops to
    3.7-4.4 TB/s when they are adjacent bits around 17..24 (strides of 2-256 MiB camp on the same
    HBM channels).  So: never-dense qubits (pure selects/controls) take the lane bits and the top
    (shard) bits, dense targets are packed upward from bit 6 in order of first use."""
    g = n_shards.bit_length() - 1
    L = n_qubits - g
    if layout == "reference":
        return list(range(n_qubits))
    if layout != "auto":
        raise ValueError("layout must be 'auto' or 'reference', not %r" % (layout,))
    dense_first = {}
    dense_count = {q: 0 for q in range(n_qubits)}
    uniform = 0
    for k, op in enumerate(ops):
        if op.kind == "init":
            uniform |= op.mask
        for q in op.dense_targets():
            dense_first.setdefault(q, k)
            dense_count[q] += 1
    never = len(ops) + 1

    def badness(q):
        # never-dense + uniform first; then never-dense; then the latest / rarest dense use
        first = dense_first.get(q, never)
        return (0 if first == never else 1, 0 if (uniform >> q) & 1 else 1, -first, dense_count[q], -q)

    shard_q = sorted(sorted(range(n_qubits), key=badness)[:g])
    rest = [q for q in range(n_qubits) if q not in shard_q]
    dense = sorted((q for q in rest if q in dense_first), key=lambda q: (dense_first[q], q))
    # never-dense qubits: those in uniform superposition make good lane bits (every amplitude is
    # populated); those that stay |0> for the whole circuit (the AND scratch qubit, QCMRF.py:219)
    # go to the top so that the provably-zero half of the shard is one contiguous block
    quiet_u = [q for q in rest if q not in dense_first and (uniform >> q) & 1]
    quiet_z = [q for q in rest if q not in dense_first and not (uniform >> q) & 1]
    # Dense targets beyond what the register tile holds per pass ride on LANE bits: a gate whose
    # target is address bit < 6 is a wave shuffle inside the same k_multi pass, so a pass reaches
    # MULTI_R register targets plus up to 6 lane targets.  Targets are taken in first-use order:
    # per pass MULTI_R go to register positions (>= 6), that pass's share of the lane quota to lane
    # positions -- 15 cliques need 2 passes instead of 3, 19 cliques 3 instead of 4.
    lane_t, reg_t = [], list(dense)
    if lane_targets and len(dense) > MULTI_R and L >= 12:
        n_pass = max(1, -(-(len(dense) - LANE_BITS) // MULTI_R))
        n_lane_t = min(LANE_BITS, max(0, len(dense) - MULTI_R * n_pass))
        reg_t, k = [], 0
        for i in range(n_pass):
            reg_t += dense[k:k + MULTI_R]
            k += MULTI_R
            share = n_lane_t // n_pass + (1 if i < n_lane_t % n_pass else 0)
            lane_t += dense[k:k + share]
            k += share
        reg_t += dense[k:]
    n_quiet_lane = min(LANE_BITS - len(lane_t), max(0, L - len(dense)), len(quiet_u))
    order = quiet_u[:n_quiet_lane] + lane_t + reg_t + quiet_u[n_quiet_lane:] + quiet_z   # physical 0, 1, 2, ...
    lay = [0] * n_qubits
    for p, q in enumerate(order):
        lay[q] = p
    for p, q in enumerate(shard_q):
        lay[q] = L + p
    return lay


def plan(ops, n_qubits, n_shards=1, layout="auto", lane_targets=True):
    """ops on logical qubits (first op is ``init``) -> Plan with physical ops."""
    if n_shards < 1 or n_shards & (n_shards - 1):
        raise ValueError("number of shards must be a power of two")
    g = n_shards.bit_length() - 1
    L = n_qubits - g
    if L < 1:
        raise ValueError("%d qubits cannot be split into %d shards" % (n_qubits, n_shards))
    lay = choose_layout(ops, n_qubits, n_shards, layout, lane_targets)
    P = Plan()
    P.n_qubits, P.n_shards = n_qubits, n_shards
    P.initial_layout = list(lay)

    # next dense use per logical qubit, looked up lazily
    dense_at = {}
    for k, op in enumerate(ops):
        for q in op.dense_targets():
            dense_at.setdefault(q, []).append(k)

    def next_dense(q, k):
        for j in dense_at.get(q, ()):
            if j > k:
                return j
        return len(ops) + 1

    for k, op in enumerate(ops):
        if g:
            for t in op.dense_targets():
                if lay[t] < L:
                    continue
                busy = set(op.support())
                inv = {p: q for q, p in enumerate(lay)}
                cands = [inv[p] for p in range(L) if inv[p] not in busy]
                if not cands:
                    raise ValueError("gate %r leaves no local qubit free for an exchange" % (op,))
                victim = max(cands, key=lambda q: (next_dense(q, k), lay[q]))
                P.ops.append(ir.Op("swap", a=(lay[t],), b=(lay[victim],)))
                P.n_exchanges += 1
                lay[t], lay[victim] = lay[victim], lay[t]
        P.ops.append(_remap(op, lay))
    P.layout = list(lay)
    return P
