"""GPU box: differential stress of qsv_exec's pass construction -- random programs of table ops
(multiplexed 2x2 with RX-like or general matrices, diagonals), controlled gates, X and phases on
random bits, with and without a leading init, random pass hints and engine options -- against the
numpy engine.  Not part of the test suite (minutes); run after touching qsv_multi.inc."""
import sys
sys.path.insert(0, ".")
import numpy as np
from qcmrf_amd import _lib, ir, program
from oracle.sharded_numpy import NumpyEngine



def run(n_cases=300, seed=7, verbose=True, only=None, override=None, widths=(14, 15, 16, 17), poison=False):
    """returns the number of mismatching programs.  widths: state sizes drawn from; at >= 21 qubits the
    workgroups of a pass no longer run all at once (in-place permutations that cross workgroups show)"""
    rs = np.random.RandomState(seed)
    bad = 0
    engines = {}
    for case in range(n_cases):
        W = int(rs.choice(list(widths)))
        P = int(rs.choice([1, 1, 2, 4]))
        key = (W, P)
        if key not in engines:
            engines[key] = _lib.Engine(W, devices=(0,) * P)
        eng = engines[key]
        style = rs.randint(0, 4)          # 0 rx-like simple, 1 general simple, 2 mixed general, 3 init + diagonals
        def mat():
            if style == 0:
                a = rs.rand() * 3
                return np.array([[np.cos(a), -1j * np.sin(a)], [-1j * np.sin(a), np.cos(a)]])
            q, _ = np.linalg.qr(rs.randn(2, 2) + 1j * rs.randn(2, 2))
            return q
        ops = []
        mask = 0
        for q in range(W):
            if rs.rand() < 0.7:
                mask |= 1 << q
        if rs.rand() < 0.7 or style == 3:
            ops.append(ir.op_init(mask))
        for _ in range(int(rs.randint(3, 26))):
            k = rs.randint(0, 10)
            qs = [int(x) for x in rs.permutation(W)[:4]]
            L = W - (P.bit_length() - 1)
            if qs[0] >= L:                                  # dense targets must be local to a shard
                qs[0] = int(rs.choice([q for q in range(L) if q not in qs[1:]]))
            if style == 3:
                kk = int(rs.randint(1, 5))
                ops.append(ir.op_diag(qs[:kk], np.exp(1j * rs.randn(2 ** kk)) * (0.5 + rs.rand(2 ** kk))))
            elif style in (0, 1) or k < 5:
                nsel = int(rs.randint(0, 4))
                sel = qs[1:1 + nsel]
                if style in (0, 1):
                    # simple passes need selects that are never targets: take them from the top bits
                    sel = [W - 1 - i for i in range(nsel) if W - 1 - i != qs[0]]
                ops.append(ir.op_mux(sel, qs[0], np.array([mat() for _ in range(2 ** len(sel))])) if sel else ir.op_u(qs[0], mat()))
                if style in (0, 1) and qs[0] >= W - 3:
                    ops.pop()
            elif k in (5, 6):
                nc = int(rs.randint(0, 3))
                ops.append(ir.op_x(qs[0], ctrls=qs[1:1 + nc], vals=[int(v) for v in rs.randint(0, 2, size=nc)]))
            elif k == 7:
                kk = int(rs.randint(1, 4))
                ops.append(ir.op_diag(qs[:kk], np.exp(1j * rs.randn(2 ** kk))))
            elif k == 8:
                kk = int(rs.randint(1, 4))
                ops.append(ir.op_mcphase(qs[:kk], float(rs.randn()), vals=[int(v) for v in rs.randint(0, 2, size=kk)]))
            elif rs.rand() < 0.4 and L >= 6:
                # a dense 2..5-qubit gate on local qubits (every form of the matrix-core kernels is among the random options)
                kk = int(rs.randint(2, min(5, L) + 1))
                tq = [int(x) for x in rs.permutation(L)[:kk]]
                u, _ = np.linalg.qr(rs.randn(2 ** kk, 2 ** kk) + 1j * rs.randn(2 ** kk, 2 ** kk))
                ops.append(ir.op_kq(tq, u))
            else:
                nc = int(rs.randint(0, 3))
                ops.append(ir.op_u(qs[0], mat(), ctrls=qs[1:1 + nc], vals=[int(v) for v in rs.randint(0, 2, size=nc)]))
            if ops and rs.rand() < 0.1 and ops[-1].kind != "init":
                ops[-1].new_pass = True
        ops = [o for o in ops if o is not None]
        if not ops or (len(ops) == 1 and ops[0].kind == "init"):
            ops.append(ir.op_diag([0], [1.0, 1.0j]))
        opts = {"multi_r": int(rs.choice([5, 5, 5, 4, 6, 3])), "dyn_lanes": int(rs.choice([3, 3, 0, 1, 2])),
                "lane_map": int(rs.choice([1, 1, 0])), "lane_targets": int(rs.choice([1, 1, 1, 0])),
                "zero_tracking": int(rs.choice([0, 0, 0, 1])), "init_prod": int(rs.choice([1, 1, 0])),
                "pass_hints": int(rs.choice([1, 1, 0])), "fused_sums": int(rs.choice([1, 0])),
                "xframe": int(rs.choice([1, 1, 0])), "multi_nt": int(rs.choice([-1, 1, 0])),
                "init_prod_nt": int(rs.choice([-1, 1])), "pass_budget": int(rs.choice([0, 0, 30, 100])),
                "general_r": int(rs.choice([4, 4, 5, 3, 2, 1])), "general_light_r": int(rs.choice([5, 5, 4, 3])),
                "pass_max_ops": int(rs.choice([64, 64, 8, 200])), "swizzle": int(rs.choice([1, 2, 2, 0])),
                "lane_map_min_l": int(rs.choice([26, 14, 14])),
                "kq_variant": int(rs.choice([-1, -1, 0, 3, 5, 6, 8])), "kq_order": int(rs.choice([1, 1, 0, 2, 4])),
                "kq3_tile": int(rs.choice([1, 1, 0, 2]))}
        if only is not None and (case != only if only >= 0 else case < -only):
            continue                                       # replay mode (CASE, or -CASE: from that case on): the generator state advances, nothing runs
        if override:
            opts.update(override)
        for kname, v in opts.items():
            eng.set_option(kname, v)
        rec, data = program.encode(ops)
        ref = NumpyEngine(W, P)
        if ops[0].kind != "init":
            ref.init_uniform((1 << W) - 1)
            eng.init_uniform((1 << W) - 1)
        ref.exec(rec, data)
        if poison:
            eng.poison_lds()                              # quiet NaNs in every compute unit's LDS: a read of an unstaged table shows
        eng.exec(rec, data)
        want = ref.amplitudes()
        got = eng.amplitudes()
        err = float(np.abs(got - want).max())
        tol = 1e-12 * max(1.0, float(np.abs(want).max()))
        nrm_ref = float((np.abs(want) ** 2).sum())
        nerr = abs(eng.norm() - nrm_ref)
        if err > tol or nerr > 1e-10 * max(1.0, nrm_ref):
            bad += 1
            print("MISMATCH case %d W=%d P=%d style=%d err=%.3e normerr=%.3e opts=%s" % (case, W, P, style, err, nerr, opts), flush=True)
            print("   ops:", [repr(o) + ("*" if o.new_pass else "") for o in ops], flush=True)
        if verbose and min(widths) >= 24:
            print("case %d W=%d P=%d ok so far (%d mismatches)" % (case, W, P, bad), flush=True)     # slow cases: keep the log alive
        if case % 50 == 49 and verbose:
            print("... %d cases, %d mismatches" % (case + 1, bad), flush=True)
    for e in engines.values():
        e.close()
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    only = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] not in ("big", "poison", "huge") else None
    if len(sys.argv) > 3 and sys.argv[3] == "poison":       # N SEED poison: LDS poisoned before every program
        bad = run(n, int(sys.argv[2]), poison=True)
        print("done: %d cases, %d mismatches" % (n, bad))
        sys.exit(1 if bad else 0)
    if len(sys.argv) > 3 and sys.argv[3] == "huge":         # N SEED huge: 26-27 qubits -- the size-gated forms (non-temporal, swizzle, lane map) with random gates
        bad = run(n, int(sys.argv[2]), widths=(26, 26, 27))
        print("done: %d cases, %d mismatches" % (n, bad))
        sys.exit(1 if bad else 0)
    if len(sys.argv) > 3 and sys.argv[3] == "big":          # N SEED big: wide states
        bad = run(n, int(sys.argv[2]), widths=(21, 22, 23))
        print("done: %d cases, %d mismatches" % (n, bad))
        sys.exit(1 if bad else 0)
    override = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in sys.argv[4:])      # replay one case: N SEED CASE opt=value ...
    bad = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 7, only=only, override=override)
    print("done: %d cases, %d mismatches" % (n, bad))
    sys.exit(1 if bad else 0)
