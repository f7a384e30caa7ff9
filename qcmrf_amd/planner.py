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

import os

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
    o._sup = None
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
DYN_LANES = 3      # lane bits 3..5 that libqsv lends to further targets of a pass (option dyn_lanes)
STATIC_LOW = 3     # lane bits 0..2: never lent (one 128-byte line per lane group)
GEN_TOP_MIN_L = 33 # k_init_prod keeps the top bits of a shard in registers from this many local qubits on


def choose_layout(ops, n_qubits, n_shards, layout="auto", lane_targets=True, dyn_lanes=DYN_LANES, want_heads=False):
    """layout[logical] = physical.

    auto, measured on MI355X (profiles/r01_multi_bits_W28.json): a k_multi pass streams at
    5.3-5.7 TB/s when its target bits sit just above the lane bits (6..~16) and drops to
    3.7-4.4 TB/s when they are adjacent bits around 17..24 (strides of 2-256 MiB camp on the same
    HBM channels).  So: never-dense qubits (pure selects/controls) take the lane bits and the top
    (shard) bits, dense targets are packed upward from bit 6 in order of first use."""
    g = n_shards.bit_length() - 1
    L = n_qubits - g
    if layout == "reference":
        return (list(range(n_qubits)), []) if want_heads else list(range(n_qubits))
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
    # target is carried by a lane bit is a wave shuffle inside the same k_multi pass.  Two ways:
    #   static   the qubit sits on address bit 0..5 for the whole circuit (6 such targets in all);
    #   borrowed libqsv lends lane bits 3..5 to up to DYN_LANES targets PER PASS (any bit 6..27; a wave
    #            load is then 8 x 128-byte lines instead of 1 KiB), leaving 3 static ones on bits 0..2.
    # Targets are dealt in first-use order: per pass the register (+ borrowed) quota goes to
    # positions >= 6, then that pass's share of the static quota to lane positions.  The borrowed
    # form is chosen when it saves a pass: 19 cliques (W = 34) 3 -> 2 passes.
    lane_t, reg_t, top_t = [], list(dense), []
    nd = len(dense)
    pass_heads = []          # first dense target of every pass after the first
    if lane_targets and nd > MULTI_R and L >= 12:
        n_pass_static = max(1, -(-(nd - LANE_BITS) // MULTI_R))
        n_pass_dyn = max(1, -(-(nd - STATIC_LOW) // (MULTI_R + dyn_lanes))) if dyn_lanes else n_pass_static
        if n_pass_dyn < n_pass_static:
            per_pass, n_pass, quota = MULTI_R + dyn_lanes, n_pass_dyn, STATIC_LOW
        else:
            per_pass, n_pass, quota = MULTI_R, n_pass_static, LANE_BITS
        # The FIRST pass of a circuit is the init-fused, write-only one, and its lane gates are nearly
        # free: every register bit there is a fresh |0> target, so the tile starts as one nonzero
        # amplitude per lane and libqsv applies the first round's lane gates to that scalar before the
        # tile exists (k_multi, INIT).  A read+write pass pays ~1 % per lane gate and 3-5 % for
        # borrowed lane bits.  So the first pass takes every lane-borne target it can -- its own
        # borrowed ones plus the whole static quota -- and the later passes run lean.
        first_big = min(nd, per_pass)
        first_lane = min(quota, nd - first_big)
        big = [list(dense[:first_big])]
        k = first_big
        lane_t += dense[k:k + first_lane]
        k += first_lane
        while k < nd:
            pass_heads.append(dense[k])
            big.append(list(dense[k:k + per_pass]))
            k += per_pass
        n_pass = len(big)
        # Positions: the read+write passes take the low bits (measured at W = 34,
        # scripts/placement_sweep.py: the same pass runs at 5.4 TB/s on bits 6..13 and 5.05 TB/s on
        # 14..21), the write-only first pass -- indifferent to where its targets sit -- the block above
        # ... and on a shard of >= 2^33 amplitudes the LAST pass keeps its register targets on the
        # top bits of the shard (5.78 TB/s against 5.42 with them on bits 6..10, same gates,
        # profiles/r01s2_placement_top34.log); its borrowed-lane targets stay low
        if L >= GEN_TOP_MIN_L and n_pass >= 2 and len(big[-1]) >= MULTI_R:
            top_t = big[-1][:MULTI_R]
            big[-1] = big[-1][MULTI_R:]
        if top_t:      # the last pass has its registers on top: the first (write-only) pass takes bits 6.. itself
            reg_t = big[0] + [q for b in big[1:] for q in b] + list(dense[k:])
        else:
            reg_t = [q for b in big[1:] for q in b] + big[0] + list(dense[k:])
    if not dense and L >= 14:
        # nothing left but the initial product state and its diagonal factors (passes.fold_fresh):
        # libqsv's generator multiplies a factor per AMPLITUDE if it touches one of its register
        # bits, per THREAD otherwise -- give those bits the qubits the fewest factors touch (a qubit
        # that stays |0> touches none; for a QCMRF circuit then the ancillas, one factor each).
        # Which bits those are follows the shard size exactly as in libqsv (flush_init_product):
        # the top bits of a shard of >= 2^33 amplitudes, bits 6..10 below that.
        cand = quiet_u + quiet_z
        uses = {q: 0 for q in cand}
        for op in ops:
            if op.kind in ("diag", "mcphase"):
                for q in op.qubits:
                    if q in uses:
                        uses[q] += 1
        regq = sorted(sorted(cand, key=lambda q: (uses[q], -q))[:MULTI_R], key=lambda q: (-uses[q], q))
        if len(cand) - len(regq) >= LANE_BITS:
            if L >= GEN_TOP_MIN_L:
                quiet_u = [q for q in quiet_u if q not in regq]
                quiet_z = [q for q in quiet_z if q not in regq] + regq      # highest local positions, fewest uses on top
            else:
                others = [q for q in quiet_u if q not in regq]
                quiet_z = [q for q in quiet_z if q not in regq]
                quiet_u = others[:LANE_BITS] + regq + others[LANE_BITS:]
    n_quiet_lane = min(LANE_BITS - len(lane_t), max(0, L - nd), len(quiet_u))
    # static lane targets take the lowest bits: bits 3..5 stay free to be lent out
    order = lane_t + quiet_u[:n_quiet_lane] + reg_t + quiet_u[n_quiet_lane:] + quiet_z + top_t   # physical 0, 1, 2, ...
    # A gate-by-gate program (controlled X / phase / 2x2 ops, e.g. the reference's own stream at
    # fusion 0) runs in GENERAL k_multi passes, where the economics differ (28 qubits, 60 ops a pass,
    # profiles/r02w_pass_budget.log): a controlled X costs 31 us with its target on a register bit but
    # 74 us on a lane bit (a wave shuffle of the whole tile), a 2x2 72 against 96 us, and an
    # UNCONTROLLED X nothing anywhere (X frame).  So the lane bits go to the qubits that are the
    # target of the fewest such ops, the most-targeted ones sit right above them.
    masked = [op for op in ops if op.kind == "mcphase" or (op.kind in ("x", "u") and len(op.ctrls) > 0)]
    free_x = sum(1 for op in ops if op.kind == "x" and len(op.ctrls) == 0)
    if lane_targets and L >= 12 and 2 * len(masked) > len(ops) - free_x:
        heat = {q: 0 for q in rest}
        touched = set()
        for op in ops:
            if op.kind in ("u", "mux", "kq") or (op.kind == "x" and len(op.ctrls) > 0):
                # (the one-qubit gate that opens a wire -- the H on every variable, QCMRF.py:204-205 -- is part of the init
                # write in libqsv, qsv_exec: it does not make its qubit a target)
                opening = op.kind == "u" and not op.ctrls and op.target not in touched
                for q in op.dense_targets():
                    if q in heat and not opening:
                        heat[q] += 1
            touched.update(op.support())
        zeros = set(quiet_z)
        # Qubits that are ONLY ever controls (the MRF variables in the reference's stream) go to the top of the shard: up
        # there a control selects whole workgroups, and libqsv resolves such controls per workgroup (the combo table of a
        # general pass: a CCX is not even looked at by the three quarters of the machine it cannot fire in).  On a lane
        # bit the same control only masks lanes -- every wave pays the op in full.  The lane bits go to the coldest qubits
        # that are targets at all (ancillas: two masked 2x2 each, the latest-used first), then to pure controls if any
        # lane is left.
        live = [q for q in rest if q not in zeros]
        pure_ctrl = [q for q in live if heat[q] == 0]
        tgt = [q for q in live if heat[q] > 0]
        by_cold = sorted(tgt, key=lambda q: (heat[q], -dense_first.get(q, never), q))
        if os.environ.get("QSV_PLANNER_CTRL_TOP", "1") == "0":       # experiment switch: the round-2 rule (coldest qubits on the lanes)
            by_cold = sorted(live, key=lambda q: (heat[q], 0 if (uniform >> q) & 1 else 1, q))
            pure_ctrl = []
            tgt = live
        lanes = by_cold[:LANE_BITS]
        lanes += pure_ctrl[:LANE_BITS - len(lanes)]
        above = sorted((q for q in tgt if q not in lanes), key=lambda q: (-heat[q], dense_first.get(q, never), q))
        order = lanes + above + [q for q in pure_ctrl if q not in lanes] + quiet_z
        pass_heads = []
    lay = [0] * n_qubits
    for p, q in enumerate(order):
        lay[q] = p
    for p, q in enumerate(shard_q):
        lay[q] = L + p
    if want_heads:
        return lay, pass_heads
    return lay


def plan(ops, n_qubits, n_shards=1, layout="auto", lane_targets=True, dyn_lanes=DYN_LANES, batch_swaps=True):
    """ops on logical qubits (first op is ``init``) -> Plan with physical ops.
    ``n_exchanges`` counts exchange STEPS: a batched swap of several shard bits is one."""
    if n_shards < 1 or n_shards & (n_shards - 1):
        raise ValueError("number of shards must be a power of two")
    g = n_shards.bit_length() - 1
    L = n_qubits - g
    if L < 1:
        raise ValueError("%d qubits cannot be split into %d shards" % (n_qubits, n_shards))
    lay, heads = choose_layout(ops, n_qubits, n_shards, layout, lane_targets, dyn_lanes, want_heads=True)
    heads = set(heads)
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
            need = [t for t in op.dense_targets() if lay[t] >= L]
            if need:
                # Only the op's dense targets have to be local: its controls / selects may well be the
                # ones evicted to a shard bit (libqsv resolves those per shard).  Victims by Belady's
                # rule: the local qubit whose next dense use is farthest away (never, if possible).
                busy = set(op.dense_targets())
                inv = {p: q for q, p in enumerate(lay)}
                victims = sorted((inv[p] for p in range(L) if inv[p] not in busy),
                                 key=lambda q: (next_dense(q, k), lay[q]), reverse=True)
                # ONE batched exchange: every other qubit sitting on a shard bit that will be a dense
                # target later rides along now (an all-to-all moves (2^k - 1) / 2^k of a shard over
                # 2^k - 1 links at once; k separate exchanges move k / 2 of it over one link each),
                # as long as the local qubit that makes room is needed later than it
                riders = sorted((q for q in range(n_qubits) if lay[q] >= L and q not in need
                                 and next_dense(q, k) <= len(ops)), key=lambda q: next_dense(q, k)) if batch_swaps else []
                pairs = []
                for t in need + riders:
                    if not victims:
                        if t in need:
                            raise ValueError("gate %r leaves no local qubit free for an exchange" % (op,))
                        break
                    if t not in need and next_dense(victims[0], k) <= next_dense(t, k):
                        break
                    pairs.append((t, victims.pop(0)))
                P.ops.append(ir.Op("swap", a=tuple(lay[t] for t, _ in pairs), b=tuple(lay[v] for _, v in pairs)))
                P.n_exchanges += 1
                for t, v in pairs:
                    lay[t], lay[v] = lay[v], lay[t]
        P.ops.append(_remap(op, lay))
        # pass boundary chosen with the layout: the first gate on the first target of a later pass
        hit = heads.intersection(op.dense_targets())
        if hit:
            P.ops[-1].new_pass = True
            heads -= hit
    P.layout = list(lay)
    return P
