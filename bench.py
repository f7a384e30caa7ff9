#!/usr/bin/env python3
"""bench.py -- shots/sec of one QCMRF circuit run through the drop-in boundary on N MI355X.

    python bench.py [--gpus N --steps K --warmup W]                         (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
           --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...    (N > 1, one rank per GPU)

A "step" is one full ``backend.run(circuit, shots).result().get_counts()`` -- ingest, fusion
passes, state init, every gate sweep, the probability pass, sampling and the counts dict -- i.e.
exactly the call /root/reference/run_experiment.py:56-57 makes.  The circuit is built once
outside the timed region; the state vector lives in HBM throughout (nothing crosses PCIe except
the gate tables, a few KB, and the sampled outcomes).

Workload: ONE circuit for every N (strong scaling) -- BASELINE.json configs[4], the 34-qubit
2x7-grid MRF, 256 GiB of complex128 amplitudes: it fits the 288 GiB of a single MI355X, so
value(8) / value(1) is the 1 -> 8 GPU speed-up "at 34 qubits" of the north star (shards of
256 / 128 / 64 / 32 GiB at N = 1 / 2 / 4 / 8).  On a smaller card the widest grid MRF one device
holds is used instead (same rule on every rank).  --config i / --qubits W override.  At N = 1 the
line also carries ``other_configs``: configs[2] (28 qubits, the HBM-roofline config) and
configs[1] (20 qubits), same step definition, not part of ``value``.
theta = -halfnorm.rvs(scale=0.5), seed 1984; 4096 shots; seed_simulator 1984.

The JSON line carries ``roofline`` (dominant kernel, HIP-event timed on its launch stream over
the timed steps, algorithmic bytes of SURVEY.md 8(d)) and, at N=1, ``roofline_gate_sweeps_28q``
(BASELINE configs[2], the 28-qubit roofline config, with every fused gate SWEPT over the vector:
the read+write ``k_multi`` pass is the "gate-apply sweep" of the north star), ``gate_microbench_28q``
(one kernel per gate kind on a 28-qubit state: min / median / worst fraction of peak) and
``cpu_baseline`` (the plain-C oracle -- a port, Aer itself is not installable offline -- on a
bounded sample of the same workload).  At N>1 it also carries ``exchange_legs``: the same circuit in
the reference layout (ancillas on the shard bits), which forces shard-bit exchanges over RCCL and
over the peer-mapped transport -- so the RCCL communicator really spans the N ranks
(``rccl_ranks``).  ``--gates`` runs only the gate-apply micro-benchmark (one JSON object per gate kind).

Host processes rendezvous through qcmrf_amd.comm (standard library sockets; the launcher's RANK /
WORLD_SIZE / MASTER_* environment): bench.py imports no torch.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--qubits", type=int, default=0, help="override: grid MRF with this circuit width")
    ap.add_argument("--shots", type=int, default=4096)
    ap.add_argument("--fusion", type=int, default=3)
    ap.add_argument("--layout", default="auto")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra (untimed-for-value) legs")
    ap.add_argument("--variants", action="store_true", help="N > 1: run every extra leg (default there: the measured leg + the full-width-sweep leg)")
    ap.add_argument("--with-exchange", action="store_true", help="N > 1, with --variants: also the reference-layout leg, which needs shard-bit exchanges "
                                                            "(RCCL between GPUs, peer-mapped shards when ranks share one)")
    ap.add_argument("--virtual-shards", type=int, default=0,
                    help="1 GPU: split the vector into this many shards on device 0 (config-4 rehearsal: real "
                         "shard-bit resolution and exchange kernels, device copies instead of xGMI)")
    ap.add_argument("--config", type=int, default=0, help="force BASELINE configs[i] (1..4) regardless of --gpus")
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    ap.add_argument("--gates", action="store_true", help="gate-apply micro-benchmark (1 GPU)")
    ap.add_argument("--no-fold", action="store_true",
                    help="fold_fresh off: every fused gate is applied as a sweep of the full vector (k_multi passes)")
    ap.add_argument("--option", action="append", default=[], help="engine option name=int")
    ap.add_argument("--no-exchange-leg", action="store_true", help="N > 1: skip the reference-layout legs (RCCL / peer-mapped shard-bit exchanges)")
    ap.add_argument("--exchange-deadline", type=float, default=150.0, help="seconds one exchange leg may take before it is abandoned")
    ap.add_argument("--exchange-child", action="store_true", help=argparse.SUPPRESS)   # internal: run only the exchange legs, print their JSON
    return ap.parse_args()


CPU_SAMPLE_QUBITS = 28          # 4 GiB: the widest state the plain-C CPU leg is timed on (about 25 s)


def workload(args, hbm_total):
    """One circuit for every N (strong scaling): BASELINE configs[4], the 34-qubit grid MRF
    (256 GiB of amplitudes), whenever ONE device can hold it right now (free HBM of rank 0's device;
    an MI355X has 288 GiB) -- so that
    value(N) / value(1) is the 1 -> N speed-up "at 34 qubits" the north star asks for; otherwise the
    widest grid MRF one device holds.  --config / --qubits override."""
    from qcmrf_amd import workloads as wl
    if args.qubits:
        C = wl.for_width(args.qubits)
        name = "grid MRF, W=%d (n=%d, m=%d)" % (args.qubits, wl.width(C) - len(C) - 1, len(C))
    elif args.config:
        name, C = wl.baseline_config(args.config)
    else:
        W = 34
        while W > 20 and 16 * 2 ** W + (8 << 30) > hbm_total:
            W -= 1
        if W == 34:
            name, C = wl.baseline_config(4)
        else:
            C = wl.for_width(W)
            name = "grid MRF, W=%d (n=%d, m=%d): widest that fits one device" % (W, wl.width(C) - len(C) - 1, len(C))
    return name, C, wl.theta_halfnorm(wl.dimension(C))


# --------------------------------------------------------------------------------------------
def cpu_baseline(cliques, theta, shots, budget_s):
    """Plain-C oracle (OpenMP, all host threads) on the UNFUSED reference-order gate stream: the n
    initial H gates plus as many whole clique blocks as fit the time budget, extrapolated to all m
    blocks (every block has the same gate mix).  A circuit wider than CPU_SAMPLE_QUBITS does not
    fit a sensible host sample: the 28-qubit config is timed instead and scaled by the gate count
    and by 2^(W-28) (every gate is one memory-bound sweep of the 2^W vector).  Reported, not a target."""
    from oracle import cref, gate_stream as gs, closed_form as cf
    from qcmrf_amd import workloads as wl
    cref.build()
    cref.set_threads(cref.host_threads())          # affinity / cgroup share, not the box's 256 logical CPUs
    n, m, W, dim = cf.model_shape(cliques)
    scale, note = 1.0, ""
    if W > CPU_SAMPLE_QUBITS:
        n_gates_full = len(gs.reference_stream(cliques, theta, with_measurements=False))
        _, cliques = wl.baseline_config(2)
        theta = wl.theta_halfnorm(wl.dimension(cliques))
        n, m, W0, dim = cf.model_shape(cliques)
        n_gates_sample = len(gs.reference_stream(cliques, theta, with_measurements=False))
        scale = n_gates_full / n_gates_sample * 2.0 ** (W - W0)
        note = ("; the W=%d state (%.0f GiB) does not fit a host sample, so the 28-qubit config was timed and scaled by "
                "gates %d/%d x 2^%d = %.1fx" % (W, 16 * 2.0 ** W / 2 ** 30, n_gates_full, n_gates_sample, W - W0, scale))
        W = W0
    ops = gs.reference_stream(cliques, theta, with_measurements=False)
    # split into the H prologue and per-clique blocks (each starts with the H on its ancilla)
    starts = [i for i, op in enumerate(ops) if op[0] == "h" and op[1] > n]
    starts = starts[0::2] + [len(ops)]
    st = cref.RefState(W)
    t0 = time.perf_counter()
    st.run_stream(ops[:starts[0]])
    t_h = time.perf_counter() - t0
    t_blocks, done = 0.0, 0
    for b in range(m):
        t0 = time.perf_counter()
        st.run_stream(ops[starts[b]:starts[b + 1]])
        t_blocks += time.perf_counter() - t0
        done += 1
        if t_h + t_blocks > budget_s:
            break
    t0 = time.perf_counter()
    st.norm()
    t_prob = time.perf_counter() - t0
    est = (t_h + t_blocks / done * m + t_prob) * scale
    return {"value": shots / est, "unit": "shots/s", "cores": st.threads(), "kind": "port",
            "sample": "W=%d state (%.0f MiB) on host; %d H gates + %d of %d clique blocks (%d of %d gates) + norm pass "
                      "timed = %.1f s, extrapolated to %.1f s per circuit; unfused reference-order "
                      "stream; plain-C OpenMP restatement (Qiskit Aer not installable offline)%s"
                      % (W, 16 * 2 ** W / 2 ** 20, n, done, m, starts[done], len(ops), t_h + t_blocks + t_prob, est, note)}


# --------------------------------------------------------------------------------------------
def gate_microbench(args, W=None, quiet=False):
    """dense 1q at every target, X, CX, CCX(+-flags), CP, 3q diagonal, mux-RX, 5q dense on a
    W-qubit state: HIP-event time per launch -> GB/s against the algorithmic byte model."""
    from qcmrf_amd import _lib
    W = W or args.qubits or 28
    reps = max(args.steps, 5)
    rs = np.random.RandomState(0)

    def ru(k):
        q, _ = np.linalg.qr(rs.randn(2 ** k, 2 ** k) + 1j * rs.randn(2 ** k, 2 ** k))
        return q
    eng = _lib.Engine(W)
    for o in args.option:
        k, v = o.split("=")
        eng.set_option(k, int(v))
    eng.init_uniform((1 << W) - 1)
    eng.set_option("cache_sums", 0)                 # the norm pass below must really run every time
    A = float(2 ** W)
    cases = []
    for t in range(W):
        cases.append(("1q_t%02d" % t, 32 * A, lambda t=t, m=ru(1): eng.apply_1q(t, m)))
    cases += [
        ("x_t0", 32 * A, lambda: eng.apply_mcx([], 0)),
        ("x_t%d" % (W - 1), 32 * A, lambda: eng.apply_mcx([], W - 1)),
        ("cx_c3_t%d" % (W - 2), 16 * A, lambda: eng.apply_mcx([3], W - 2)),
        ("cx_c%d_t3" % (W - 2), 16 * A, lambda: eng.apply_mcx([W - 2], 3)),
        ("ccx_flags_c1c5_t12", 8 * A, lambda: eng.apply_mcx([1, 5], 12, [0, 1])),
        ("cp_c12_t%d" % (W - 1), 8 * A, lambda: eng.apply_mcphase([12, W - 1], 0.3)),
        ("diag3", 32 * A, lambda tab=np.exp(1j * rs.randn(8)): eng.apply_diag([2, 9, W - 1], tab)),
        ("mux_rx_c3_t%d" % (W - 1), 32 * A,
         lambda mats=np.array([ru(1) for _ in range(8)]): eng.apply_mux([4, 7, 12], W - 1, mats)),
        ("mux_rx_c3_t13", 32 * A,
         lambda mats=np.array([ru(1) for _ in range(8)]): eng.apply_mux([4, 7, 12], 13, mats)),
        ("kq5_low", 32 * A, lambda u=ru(5): eng.apply_kq([0, 1, 2, 3, 4], u)),
        ("kq5_mixed", 32 * A, lambda u=ru(5): eng.apply_kq([1, 6, 11, 17, W - 1], u)),
        ("kq3_high", 32 * A, lambda u=ru(3): eng.apply_kq([W - 3, W - 2, W - 1], u)),
        ("norm_pass", 16 * A, lambda: eng.norm()),
    ]
    out = []
    for name, nbytes, fn in cases:
        for _ in range(3):
            fn()
        eng.sync()
        eng.timer_begin()
        for _ in range(reps):
            fn()
        ms = eng.timer_end() / reps
        gbps = nbytes / ms / 1e6
        out.append({"gate": name, "ms": ms, "GBps": gbps, "frac_of_8TBps": gbps / HBM_PEAK_GBPS})
        if not quiet:
            print(json.dumps(out[-1]), flush=True)
    eng.close()
    return out


# --------------------------------------------------------------------------------------------
def timed_leg(backend, comm, qc, shots, steps, warmup, seed0=1984, options=(), **run_opts):
    """W warm-up runs, then exactly `steps` runs bracketed by barrier + device sync on both sides;
    elapsed = max over ranks.  Returns the per-kernel HIP-event aggregation as well."""
    for i in range(warmup):
        backend.run(qc, shots=shots, seed_simulator=seed0 + i, **run_opts).result()
    for o in options:
        k, v = o.split("=")
        backend.last_engine.set_option(k, int(v))
    comm.barrier()
    backend.last_engine.sync()
    t0 = time.perf_counter()
    agg = {}
    t = {"compile": 0.0, "evolve": 0.0, "sample": 0.0}
    for i in range(steps):
        res = backend.run(qc, shots=shots, seed_simulator=seed0 + warmup + i, profile=True, **run_opts).result()
        meta = res.metadata(0)
        for k in t:
            t[k] += meta["time_" + k]
        for k, v in meta["stats"]["kinds"].items():
            a = agg.setdefault(k, {"launches": 0, "bytes": 0.0, "ms": 0.0})
            for f in a:
                a[f] += v[f]
    backend.last_engine.sync()
    comm.barrier()
    elapsed = max(comm.allgather(time.perf_counter() - t0))
    counts = res.get_counts()                   # the merged dict on rank 0; empty on the other ranks (backend: gather_counts="root")
    assert sum(counts.values()) == (shots if comm.rank == 0 else 0), (comm.rank, sum(counts.values()))
    dom = max(agg, key=lambda k: agg[k]["ms"])
    d = agg[dom]
    return {"elapsed": elapsed, "meta": meta, "agg": agg, "dom": dom, "counts": counts,
            "breakdown_ms": {k: v / steps * 1e3 for k, v in t.items()},
            "kernels": {k: {"launches_per_step": a["launches"] / steps, "avg_ms": a["ms"] / a["launches"],
                            "GBps": a["bytes"] / a["ms"] / 1e6 if a["ms"] > 0 else None} for k, a in agg.items()},
            "roofline": {"bound": "hbm", "kernel": "k_" + dom, "achieved": d["bytes"] / d["ms"] / 1e6,
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": d["bytes"] / d["ms"] / 1e6 / HBM_PEAK_GBPS,
                         "traffic": None, "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                         "avg_launch_ms": d["ms"] / d["launches"], "rank": 0}}


def validate_last_step(backend, comm, qc, cliques, theta, counts, shots):
    """Sanity of the result the timed steps produced -- outside the timed region, product code only
    (qcmrf_amd.mrf is the host restatement of eval.py's Gibbs arithmetic, not the oracle): the state's norm
    summed over every rank's shard, the rate at which all real-part-extraction ancillas read 0 against
    Z / 2^n, and -- when enough shots pass the post-selection -- the conditional distribution of the
    variables against the Gibbs pmf (fidelity as run_experiment.py reports it)."""
    from qcmrf_amd import extract_probs, fidelity, mrf
    n = qc.num_vertices
    W = qc.num_qubits
    norm = float(np.sum(comm.allgather_f64(backend.last_engine.norm())))          # each rank holds its own shard's share
    P, rate = extract_probs(counts, n, W - n)
    p, lnZ = mrf.gibbs_pmf(cliques, theta)
    want = mrf.success_probability(cliques, theta)
    sigma = (want * (1.0 - want) / shots) ** 0.5
    # the conditional pmf is only worth comparing when enough shots passed the post-selection (an exact sampler scores
    # about 1 - (support - 1) / (4 N) against p with N of them); at the bench's 4096 shots it rarely is
    n_ok = int(round(rate * shots))
    support = int((p > 1e-12).sum())
    fid = float(fidelity(P, p)) if n_ok >= 4 * support else None
    # exact, whatever the shot count: one read pass over the resident state on every rank (qsv_expect_diag) gives
    # P(all ancillas 0) and <H> in the post-selected state; both against the Gibbs arithmetic to 1e-9
    hdiag = qc.hamiltonian_diagonal()                       # index bit q <-> qubit q, variable v on qubit n-1-v: gibbs_pmf's order
    s0, s1 = backend.expectation_diagonal(hdiag, list(range(n)), {q: 0 for q in range(n, W)})
    h_dev, h_want = s0 / s1, float(np.dot(p, hdiag))
    ok = (abs(norm - 1.0) < 1e-9 and abs(s1 - want) < 1e-9 and abs(h_dev - h_want) < 1e-9 * max(1.0, abs(h_want))
          and abs(rate - want) < 5.0 * sigma + 1e-3 and (fid is None or fid > 0.9))
    return {"ok": bool(ok), "norm_over_all_shards": norm, "p_all_ancillas_0_on_device": s1, "expected_Z_over_2n": want,
            "H_post_selected_on_device": h_dev, "H_gibbs": h_want,
            "sampled_success_rate": float(rate), "binomial_sigma": sigma, "post_selected_shots": n_ok, "fidelity_to_gibbs_pmf": fid,
            "note": "last timed step, computed after the timed region: norm and the two device expectations are exact checks "
                    "(1e-9) of the state in HBM across all ranks; the sampled rate is within 5 sigma; fidelity only when "
                    ">= 4 x support shots passed the post-selection"}


PMC_FILE = "profiles/pmc_traffic.json"


def pmc_traffic(dom, W):
    """HBM bytes per launch from the PMC counters -- NOT measured by this run: read from the committed
    summary of a separate rocprofv3 --pmc run of this same command (scripts/profile_round.sh), and
    labelled as such in the JSON line (``traffic`` itself stays null)."""
    try:
        v = json.load(open(os.path.join(ROOT, PMC_FILE))).get("k_" + dom, {}).get("hbm_bytes_per_launch_W%d" % W)
    except Exception:
        v = None
    return None if v is None else {"hbm_bytes_per_launch": v, "file": PMC_FILE,
                                   "note": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not this run"}


def run_exchange_legs(args, comm, qc, device, world):
    """reference layout + fold_fresh off on all ranks, once per transport.  Returns a dict (rank 0's
    view); a leg that raises is recorded; a leg that does not come back within the deadline makes
    the watchdog print what has been measured so far and end the process (every rank has one)."""
    import threading
    from qcmrf_amd.backend import QsvBackend
    out = {"rccl_ranks": 0, "layout": "reference (ancillas = dense targets on the shard bits)", "fold_fresh": False}
    for transport in ("rccl", "p2p"):
        done = threading.Event()
        leg = {}

        def work(transport=transport, leg=leg, done=done):
            be = None
            try:
                be = QsvBackend(fusion=args.fusion, layout="reference", comm=comm, device=device, fold_fresh=False,
                                exchange=transport)
                be.run(qc, shots=args.shots, seed_simulator=11).result()               # bootstrap + warm-up
                comm.barrier()
                t0 = time.perf_counter()
                n = 2
                for i in range(n):
                    r = be.run(qc, shots=args.shots, seed_simulator=12 + i, profile=True).result()
                be.last_engine.sync()
                comm.barrier()
                dt = max(comm.allgather(time.perf_counter() - t0))
                m = r.metadata(0)
                st = m["stats"]
                ex = st["kinds"].get("exchange", {})
                leg.update({"transport": getattr(be.last_engine, "transport", None), "ranks": world,
                            "ms_per_step": dt / n * 1e3, "exchanges_per_step": m["n_exchanges"],
                            "exchange_ms_each": ex.get("ms", 0.0) / max(1, ex.get("launches", 1)),
                            "exchange_bytes_sent_per_rank_per_step": st["exchange_bytes"],
                            "evolve_ms": m["time_evolve"] * 1e3})
            except Exception as e:                                                       # noqa: BLE001
                leg["error"] = repr(e)[:300]
            finally:
                try:
                    if be is not None:
                        be.close()
                except Exception:                                                        # noqa: BLE001
                    pass
                done.set()
        th = threading.Thread(target=work, daemon=True)
        th.start()
        if not done.wait(args.exchange_deadline):
            leg["error"] = "no result within %.0f s: abandoned" % args.exchange_deadline
            out[transport] = leg
            out["abandoned"] = transport
            return out            # the worker thread may still sit in a device call: the caller prints and exits
        try:
            oks = comm.allgather("error" not in leg)
        except Exception as e:                                                           # noqa: BLE001
            leg.setdefault("error", "a peer left during the leg: %r" % (e,))
            out[transport] = leg
            out["abandoned"] = transport
            return out
        if not all(oks):
            leg.setdefault("error", "failed on rank(s) %s" % [i for i, ok in enumerate(oks) if not ok])
        out[transport] = leg
        if transport == "rccl" and "error" not in leg and leg.get("transport") == "rccl":
            out["rccl_ranks"] = world
    return out


def exchange_legs_in_child_processes(args, rank):
    """Every rank runs the exchange legs in a CHILD process (same RANK / WORLD_SIZE, its own rendezvous
    endpoint): neither transport has ever seen two GPUs before the driver's scaling run, and a fault,
    abort or hang down there must not take the already measured line with it.  Rank 0 reads its
    child's JSON from a pipe; a child that outlives the deadline is killed."""
    import subprocess
    env = dict(os.environ)
    from qcmrf_amd import comm as qcomm
    env["QSV_COMM_ENDPOINT"] = qcomm.child_endpoint(env, "bench-legs")     # unix socket on one node, TCP across nodes: the parent's own rule
    cmd = [sys.executable, os.path.abspath(__file__), "--exchange-child", "--gpus", str(args.gpus), "--shots", str(args.shots),
           "--fusion", str(args.fusion), "--exchange-deadline", str(args.exchange_deadline)]
    if args.qubits:
        cmd += ["--qubits", str(args.qubits)]
    if args.config:
        cmd += ["--config", str(args.config)]
    limit = 2 * args.exchange_deadline + 180
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=sys.stderr, text=True)
        try:
            out, _ = p.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            p.kill()
            p.communicate()
            return {"rccl_ranks": 0, "error": "the exchange-leg process did not finish within %.0f s and was killed" % limit}
        lines = [ln for ln in (out or "").strip().splitlines() if ln.startswith("{")]
        if rank == 0:
            if lines:
                return json.loads(lines[-1])
            return {"rccl_ranks": 0, "error": "the exchange-leg process ended with code %s and no result" % p.returncode}
        return {"rccl_ranks": 0, "child_returncode": p.returncode}
    except Exception as e:                                                               # noqa: BLE001
        return {"rccl_ranks": 0, "error": repr(e)[:300]}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    # the HIP library first: it must be /opt/rocm's runtime that ends up in the process
    from qcmrf_amd import _lib, QCMRF
    from qcmrf_amd.backend import QsvBackend
    from qcmrf_amd import comm as qcomm
    from qcmrf_amd import workloads as wl
    _lib.load()
    n_dev = _lib.device_count()          # initialises /opt/rocm's HIP runtime before torch brings its own copy in

    if args.gates:
        gate_microbench(args)
        return

    comm = qcomm.from_environment(timeout_s=600)
    device = local_rank % max(1, n_dev)
    hbm_total = comm.bcast(_lib.device_memory(device)[0] if rank == 0 else None)    # FREE bytes on rank 0's device
    name, cliques, theta = workload(args, hbm_total)
    qc = QCMRF(cliques, theta)
    W = qc.num_qubits
    if args.exchange_child:
        legs = run_exchange_legs(args, comm, qc, device, world)
        if rank == 0:
            print(json.dumps(legs), flush=True)
        sys.stdout.flush()
        bad = bool(legs.get("abandoned")) or any(isinstance(v, dict) and "error" in v for v in legs.values())
        os._exit(1 if bad else 0)         # (_exit: a transport may still sit in a device call on some rank)
    backend = QsvBackend(fusion=args.fusion, layout=args.layout, comm=comm if world > 1 else None,
                         device=device, devices=(0,) * max(1, args.virtual_shards), fold_fresh=not args.no_fold)

    main_leg = timed_leg(backend, comm, qc, args.shots, args.steps, args.warmup, options=args.option)
    elapsed, meta, agg = main_leg["elapsed"], main_leg["meta"], main_leg["agg"]
    validation = validate_last_step(backend, comm, qc, cliques, theta, main_leg["counts"], args.shots)
    # host time per rank (what does not shrink with the rank count): compile + sample/merge/format of the timed steps
    host_by_rank = comm.allgather({k: round(v, 4) for k, v in main_leg["breakdown_ms"].items()}) if world > 1 else None

    # untimed-for-`value` extra legs (every rank takes part; reported under "variants")
    variants = {}

    def leg(name, n, circuit=None, **opts):
        backend.run(circuit or qc, shots=args.shots, seed_simulator=76, **opts)       # one untimed warm-up run
        comm.barrier()
        t0 = time.perf_counter()
        agg = {}
        for i in range(n):
            r = backend.run(circuit or qc, shots=args.shots, seed_simulator=77 + i, profile=True, **opts).result()
            for k, v in r.metadata(0)["stats"]["kinds"].items():
                a = agg.setdefault(k, {"launches": 0, "bytes": 0.0, "ms": 0.0})
                for f in a:
                    a[f] += v[f]
        backend.last_engine.sync()
        comm.barrier()
        dt = max(comm.allgather(time.perf_counter() - t0))
        m = r.metadata(0)
        variants[name] = {"shots_per_s": args.shots * n / dt, "ms_per_step": dt / n * 1e3,
                          "device_ops": m["n_device_ops"], "evolve_ms": m["time_evolve"] * 1e3,
                          "exchanges_per_step": m["n_exchanges"],
                          "kernels": {k: {"launches_per_step": a["launches"] / n, "avg_ms": a["ms"] / a["launches"],
                                          "GBps": a["bytes"] / a["ms"] / 1e6 if a["ms"] > 0 else None,
                                          "frac_of_8TBps": a["bytes"] / a["ms"] / 1e6 / HBM_PEAK_GBPS if a["ms"] > 0 else None}
                                      for k, a in agg.items()}}

    # N > 1: besides the measured leg only the full-width-sweep leg (how the sweep path scales),
    # unless --variants asks for more; a leg that fails is recorded, not fatal (the host collectives
    # time out after 10 minutes instead of leaving the other ranks in a barrier for good)
    if not args.no_variants:
        n = max(2, args.steps // 2)
        try:
            leg("full-width gate sweeps (fold_fresh off: init-fused pass + read/write k_multi passes)", n, fold_fresh=False)
        except Exception as e:                       # noqa: BLE001
            variants["full-width gate sweeps (fold_fresh off)"] = {"error": repr(e)[:300]}
    if not args.no_variants and (world == 1 or args.variants):
        leg("full-width gate sweeps + zero tracking (opt-in: skips the provably-zero part of the vector)", n,
            fold_fresh=False, engine_options={"zero_tracking": 1})
        if world > 1 and args.with_exchange:
            # ancillas (the dense targets) on the shard bits: every late clique costs a half-shard
            # exchange (RCCL between GPUs, peer-mapped when ranks share one)
            leg("reference layout (qubit q on bit q), fold_fresh off: shard-bit exchanges", 2, fold_fresh=False,
                layout="reference", engine_options={"zero_tracking": 0})
        if world == 1:
            leg("unfused reference-order gate stream (fusion=0)", 1, fusion=0, engine_options={"zero_tracking": 0})
        if world == 1:
            from qcmrf_amd.transpile import transpile
            leg("lowered to {cx,id,rz,sx,x} as run_experiment.py:52 (stand-in transpiler, not timed) -> fusion 3",
                2, circuit=transpile(qc), fusion=args.fusion, engine_options={"zero_tracking": 0})
        if world == 1:
            # run_experiment.py:56 hands the simulator a LIST of circuits: the backend compiles circuit
            # i + 1 on a helper thread while the device evolves and samples circuit i
            nb = 8
            t0 = time.perf_counter()
            rb = backend.run([qc] * nb, shots=args.shots, seed_simulator=5).result()
            dt = time.perf_counter() - t0
            assert len(rb.get_counts()) == nb
            variants["batch of %d circuits in one run() call (host compile overlapped with device work)" % nb] = {
                "shots_per_s": args.shots * nb / dt, "ms_per_step": dt / nb * 1e3}

    # the "next" rows of SURVEY.md 8(f), measured by this run (N = 1): the diagonal-observable read pass on the resident
    # 34-qubit state (f4: qsv_expect_diag) and trajectory mode beyond statevector reach (f3)
    expect_roof = None
    if world == 1 and not args.no_variants:
        eng = backend.last_engine
        if backend.last_plan is None or len(backend.last_plan.layout) != W:
            backend.run(qc, shots=0)
            eng = backend.last_engine
        n_var = qc.num_vertices
        hdiag = qc.hamiltonian_diagonal()
        fixed = {q: 0 for q in range(n_var, W)}
        backend.expectation_diagonal(hdiag, list(range(n_var)), fixed)
        eng.sync()
        reps = 5
        eng.timer_begin()
        for _ in range(reps):
            backend.expectation_diagonal(hdiag, list(range(n_var)), fixed)
        ms = eng.timer_end() / reps
        nbytes = 16.0 * 2.0 ** W
        expect_roof = {"bound": "hbm", "kernel": "k_expect_diag (<H> of QCMRF.Hamiltonian() post-selected on all ancillas 0: one read pass, %d-entry table)" % hdiag.size,
                       "achieved": nbytes / ms / 1e6, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": nbytes / ms / 1e6 / HBM_PEAK_GBPS,
                       "traffic": None, "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms, "launches": reps,
                       "note": "HIP events around qsv_expect_diag calls (kernel + the partial-sum read-back)"}
        try:
            from qcmrf_amd.backend import QsvBackend as _QB
            tC = wl.chain(22)                                   # 22 variables, 21 cliques: a 44-qubit circuit, 24 live qubits
            tq = QCMRF(tC, wl.theta_halfnorm(wl.dimension(tC), scale=0.25))
            backend.close()                                     # (the 256 GiB state leaves the device first)
            tb = _QB(method="trajectory", fusion=args.fusion)
            tb.run(tq, shots=256, seed_simulator=1)             # warm-up: engine creation, kernels
            t0 = time.perf_counter()
            tr = tb.run(tq, shots=args.shots, seed_simulator=2).result()
            dt = time.perf_counter() - t0
            tm = tr.metadata(0)
            assert sum(tr.get_counts().values()) == args.shots
            variants["trajectory mode (mid-circuit measurements taken as they occur, QCMRF.py:238-239): 22-variable chain MRF, "
                     "W = 44 circuit qubits"] = {
                "shots_per_s": args.shots / dt, "ms_per_step": dt * 1e3, "circuit_qubits": tq.num_qubits,
                "live_qubits": tm.get("live_qubits"), "segments": tm.get("n_segments"),
                "branch_nodes": tm.get("branch_nodes"), "state_copies": tm.get("state_copies"), "device_ops": tm.get("device_ops"),
                "shots": args.shots}
            tb.close()
        except Exception as e:                                   # noqa: BLE001
            variants["trajectory mode W = 44"] = {"error": repr(e)[:300]}

    # N = 1: the other single-GPU configs of BASELINE.json, same step definition (not part of `value`)
    other = {}
    if world == 1 and not args.no_variants and not (args.qubits or args.config or args.virtual_shards):
        for ci in (2, 1):
            oname, oc = wl.baseline_config(ci)
            oq = QCMRF(oc, wl.theta_halfnorm(wl.dimension(oc)))
            lg = timed_leg(backend, comm, oq, args.shots, max(args.steps, 10), 2)
            rf = lg["roofline"]
            rf["traffic_from_profiles"] = pmc_traffic(lg["dom"], oq.num_qubits)
            other[oname] = {"shots_per_s": args.shots * max(args.steps, 10) / lg["elapsed"],
                            "ms_per_step": lg["elapsed"] / max(args.steps, 10) * 1e3,
                            "breakdown_ms": lg["breakdown_ms"], "kernels": lg["kernels"], "roofline": rf}

    # N = 1: the north-star roofline, measured in THIS run: BASELINE configs[2] (28 qubits) with every
    # fused gate swept over the vector (fold_fresh off) -- the read+write k_multi pass is the
    # "gate-apply sweep"; and one kernel per gate kind on a 28-qubit state (--gates in brief)
    sweeps28, gates28 = None, None
    if world == 1 and not args.no_variants and not (args.qubits or args.config or args.virtual_shards):
        oname, oc = wl.baseline_config(2)
        oq = QCMRF(oc, wl.theta_halfnorm(wl.dimension(oc)))
        nst = max(args.steps, 20)
        lg = timed_leg(backend, comm, oq, args.shots, nst, 3, fold_fresh=False)
        km, ki = lg["agg"].get("multi"), lg["agg"].get("multi_init")
        if km and km["ms"] > 0:
            gb = km["bytes"] / km["ms"] / 1e6
            sweeps28 = {"bound": "hbm", "kernel": "k_multi<5,false,2> (read+write pass over the shard: the gates of one pass "
                                                  "are applied in registers between one load and one store of every amplitude)",
                        "achieved": gb, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gb / HBM_PEAK_GBPS,
                        "traffic": None, "traffic_from_profiles": pmc_traffic("multi", oq.num_qubits),
                        "algorithmic_bytes_per_launch": km["bytes"] / km["launches"], "avg_launch_ms": km["ms"] / km["launches"],
                        "launches_per_step": km["launches"] / nst, "steps": nst, "workload": oname,
                        "gates_per_step_on_device": lg["meta"]["n_device_ops"] - 1,
                        "ms_per_step": lg["elapsed"] / nst * 1e3, "shots_per_s": args.shots * nst / lg["elapsed"],
                        "breakdown_ms": lg["breakdown_ms"], "note": "fold_fresh=False; not part of `value`"}
            if ki and ki["ms"] > 0:
                sweeps28["init_fused_pass"] = {"kernel": "k_multi<5,true,2> (write-only first pass)", "avg_launch_ms": ki["ms"] / ki["launches"],
                                               "achieved": ki["bytes"] / ki["ms"] / 1e6, "frac": ki["bytes"] / ki["ms"] / 1e6 / HBM_PEAK_GBPS,
                                               "algorithmic_bytes_per_launch": ki["bytes"] / ki["launches"]}
        backend.close()
        g = gate_microbench(args, W=28, quiet=True)
        fr = sorted((x["frac_of_8TBps"], x["gate"]) for x in g)
        gates28 = {"qubits": 28, "n_cases": len(g), "min_frac": fr[0][0], "worst": fr[0][1], "median_frac": fr[len(fr) // 2][0],
                   "max_frac": fr[-1][0], "best": fr[-1][1], "n_below_0.60": sum(1 for f, _ in fr if f < 0.60),
                   "frac_by_gate": {x["gate"]: round(x["frac_of_8TBps"], 3) for x in g},
                   "note": "one dedicated kernel per gate, HIP-event time over %d launches each, algorithmic bytes of SURVEY.md 8(d); "
                           "controlled gates are charged the control-satisfied subspace only" % max(args.steps, 5)}

    # N > 1: the same circuit in the reference layout (qubit q on bit q: the ancillas, i.e. the dense
    # targets, sit on the shard bits -- QCMRF.py:231-236) with full-width sweeps, so the planner has
    # to exchange shard bits: once over RCCL (the communicator then really spans the N ranks) and
    # once over the peer-mapped transport.  Never part of `value`.  A watchdog keeps a transport
    # that hangs from taking the already measured line down with it.
    exchange_legs = None
    if world > 1 and not args.no_exchange_leg:
        backend.close()                       # its numbers are in; the legs allocate their own shards
        if n_dev >= world:
            exchange_legs = exchange_legs_in_child_processes(args, rank)
        else:
            # ranks sharing a GPU (rehearsal on a one-GPU box): a second process per rank would exceed
            # the box's bound on processes per GPU, and only the peer-mapped transport can run there
            # anyway (RCCL refuses two ranks on a device and says so): in-process, watchdog only
            exchange_legs = run_exchange_legs(args, comm, qc, device, world)
            if exchange_legs.get("abandoned"):
                if rank == 0:
                    print(json.dumps({"error": "exchange leg abandoned", "exchange_legs": exchange_legs}), flush=True)
                os._exit(1)

    if rank == 0:
        roof = main_leg["roofline"]
        roof["traffic_from_profiles"] = pmc_traffic(main_leg["dom"], W)
        line = {
            "metric": "shots/sec, n-qubit QCMRF circuit (fp64 statevector, ingest+evolve+sample)",
            "value": args.shots * args.steps / elapsed, "unit": "shots/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": name, "qubits": W, "shots": args.shots, "fusion": args.fusion, "fold_fresh": not args.no_fold,
                       "layout": args.layout, "state_GiB": 16.0 * 2 ** W / 2 ** 30,
                       "shard_GiB": 16.0 * 2 ** W / 2 ** 30 / max(args.virtual_shards, args.gpus),
                       "sweeps_per_step": sum(a["launches"] for k, a in agg.items() if k != "prob") // args.steps,
                       "gates_per_step_on_device": meta["n_device_ops"] - 1,
                       "source_gates": meta["n_source_ops"], "exchanges_per_step": meta["n_exchanges"],
                       "parallelism": "amplitude shards by high qubit x%d" % (args.virtual_shards or args.gpus)
                                      + (" (virtual shards on one device)" if args.virtual_shards else "")},
            "breakdown_ms": main_leg["breakdown_ms"],
            "breakdown_ms_by_rank": host_by_rank,
            "kernels": main_leg["kernels"],
            "roofline": roof,
            "validation": validation,
        }
        line["variants"] = variants
        # the same circuit with every fused gate swept over the full vector (fold_fresh off): the
        # read+write k_multi pass is the "gate-apply sweep" of BASELINE.json's metric
        for vname, v in variants.items():
            km = v.get("kernels", {}).get("multi") if vname.startswith("full-width gate sweeps (") else None
            if km and km.get("GBps"):
                line["roofline_gate_sweeps"] = {
                    "bound": "hbm", "kernel": "k_multi", "achieved": km["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": km["GBps"] / HBM_PEAK_GBPS, "traffic": None, "traffic_from_profiles": pmc_traffic("multi", W),
                    "avg_launch_ms": km["avg_ms"], "launches_per_step": km["launches_per_step"],
                    "shots_per_s": v["shots_per_s"], "ms_per_step": v["ms_per_step"],
                    "note": "variant, not part of `value`: fold_fresh=False"}
        if expect_roof:
            line["roofline_expect_diag"] = expect_roof
        if sweeps28:
            line["roofline_gate_sweeps_28q"] = sweeps28
        if gates28:
            line["gate_microbench_28q"] = gates28
        if exchange_legs is not None:
            line["exchange_legs"] = exchange_legs
            line["rccl_ranks"] = exchange_legs.get("rccl_ranks", 0)
        if other:
            line["other_configs"] = other
        if args.gpus == 1 and not args.no_cpu:
            backend.close()
            line["cpu_baseline"] = cpu_baseline(cliques, theta, args.shots, args.cpu_seconds)
            # the port executes the UNFUSED reference-order stream: the GPU number that does the same work is the
            # fusion-0 variant, not `value` (which runs the circuit fused into one write pass)
            like = variants.get("unfused reference-order gate stream (fusion=0)")
            if isinstance(like, dict) and "shots_per_s" in like and line["cpu_baseline"].get("value"):
                line["cpu_baseline"]["like_for_like"] = {
                    "gpu_variant": "unfused reference-order gate stream (fusion=0)", "gpu_shots_per_s": like["shots_per_s"],
                    "gpu_over_cpu": like["shots_per_s"] / line["cpu_baseline"]["value"]}
        print(json.dumps(line), flush=True)
    backend.close()
    if world > 1:
        comm.barrier()
        comm.close()
    if rank == 0 and not validation["ok"]:
        sys.exit("bench.py: the state the timed steps left in HBM failed validation (see \"validation\" in the line above)")


if __name__ == "__main__":
    main()
