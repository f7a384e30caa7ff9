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

Workload by N (BASELINE.json configs; --qubits overrides with a grid MRF of that width):
    N=1    configs[2]  28-qubit 2x6-grid MRF minus last edge        4 GiB state
    N=2,4  configs[3]  31-qubit random-graph MRF G(10,20)           32 GiB, 16 / 8 GiB shards
    N=8    configs[4]  34-qubit 2x7-grid MRF                        256 GiB, 32 GiB shards
theta = -halfnorm.rvs(scale=0.5), seed 1984; 4096 shots; seed_simulator 1984.

The JSON line carries ``roofline`` (dominant kernel, HIP-event timed on its launch stream over
the timed steps, algorithmic bytes of SURVEY.md 8(d)) and, at N=1, ``cpu_baseline`` (the plain-C
oracle -- a port, Aer itself is not installable offline -- on a bounded sample of the same
workload).  ``--gates`` runs the gate-apply micro-benchmark instead (one JSON object per gate kind).
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
    ap.add_argument("--with-exchange", action="store_true", help="N>1: also run the legs that need RCCL exchanges")
    ap.add_argument("--virtual-shards", type=int, default=0,
                    help="1 GPU: split the vector into this many shards on device 0 (config-4 rehearsal: real "
                         "shard-bit resolution and exchange kernels, device copies instead of xGMI)")
    ap.add_argument("--config", type=int, default=0, help="force BASELINE configs[i] (1..4) regardless of --gpus")
    ap.add_argument("--cpu-seconds", type=float, default=25.0)
    ap.add_argument("--gates", action="store_true", help="gate-apply micro-benchmark (1 GPU)")
    ap.add_argument("--option", action="append", default=[], help="engine option name=int")
    return ap.parse_args()


def workload(args):
    from qcmrf_amd import workloads as wl
    if args.qubits:
        C = wl.for_width(args.qubits)
        name = "grid MRF, W=%d (n=%d, m=%d)" % (args.qubits, wl.width(C) - len(C) - 1, len(C))
    else:
        name, C = wl.baseline_config(args.config or {1: 2, 2: 3, 4: 3, 8: 4}.get(args.gpus, 2))
    return name, C, wl.theta_halfnorm(wl.dimension(C))


# --------------------------------------------------------------------------------------------
def cpu_baseline(cliques, theta, shots, budget_s):
    """Plain-C oracle (OpenMP, all host threads) on the UNFUSED reference-order gate stream of the
    same circuit: the n initial H gates plus as many whole clique blocks as fit the time budget,
    extrapolated to all m blocks (every block has the same gate mix).  Reported, not a target."""
    from oracle import cref, gate_stream as gs, closed_form as cf
    cref.build()
    cref.set_threads(cref.host_threads())          # affinity / cgroup share, not the box's 256 logical CPUs
    n, m, W, dim = cf.model_shape(cliques)
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
    est = t_h + t_blocks / done * m + t_prob
    return {"value": shots / est, "unit": "shots/s", "cores": st.threads(), "kind": "port",
            "sample": "W=%d state (%.0f MiB) on host; %d H gates + %d of %d clique blocks (%d of %d gates) + norm pass "
                      "timed = %.1f s, extrapolated by block count to %.1f s per circuit; unfused reference-order "
                      "stream; plain-C OpenMP restatement (Qiskit Aer not installable offline)"
                      % (W, 16 * 2 ** W / 2 ** 20, n, done, m, starts[done], len(ops), t_h + t_blocks + t_prob, est)}


# --------------------------------------------------------------------------------------------
def gate_microbench(args):
    """dense 1q at every target, X, CX, CCX(+-flags), CP, 3q diagonal, mux-RX, 5q dense on a
    W-qubit state: HIP-event time per launch -> GB/s against the algorithmic byte model."""
    from qcmrf_amd import _lib
    W = args.qubits or 28
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
        print(json.dumps(out[-1]), flush=True)
    eng.close()
    return out


# --------------------------------------------------------------------------------------------
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
    from qcmrf_amd.comm import SingleProcess, TorchDistComm
    _lib.load()

    if args.gates:
        gate_microbench(args)
        return

    comm = TorchDistComm("gloo") if world > 1 else SingleProcess()
    name, cliques, theta = workload(args)
    qc = QCMRF(cliques, theta)
    W = qc.num_qubits
    backend = QsvBackend(fusion=args.fusion, layout=args.layout, comm=comm if world > 1 else None,
                         device=local_rank % max(1, _lib.device_count()),
                         devices=(0,) * max(1, args.virtual_shards))

    def step(i, profile=False):
        res = backend.run(qc, shots=args.shots, seed_simulator=1984 + i, profile=profile).result()
        return res

    for i in range(args.warmup):
        step(i)
    if backend.last_engine is not None:
        for o in args.option:
            k, v = o.split("=")
            backend.last_engine.set_option(k, int(v))
    comm.barrier()
    backend.last_engine.sync() if backend.last_engine else None
    t0 = time.perf_counter()
    agg = {}
    t_compile = t_evolve = t_sample = 0.0
    for i in range(args.steps):
        res = step(args.warmup + i, profile=True)
        meta = res.metadata(0)
        t_compile += meta["time_compile"]
        t_evolve += meta["time_evolve"]
        t_sample += meta["time_sample"]
        for k, v in meta["stats"]["kinds"].items():
            a = agg.setdefault(k, {"launches": 0, "bytes": 0.0, "ms": 0.0})
            for f in a:
                a[f] += v[f]
    backend.last_engine.sync()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max(comm.allgather(elapsed))
    counts = res.get_counts()
    assert sum(counts.values()) == args.shots

    # untimed-for-`value` extra legs (every rank takes part; reported under "variants")
    variants = {}

    def leg(name, n, circuit=None, **opts):
        comm.barrier()
        t0 = time.perf_counter()
        for i in range(n):
            r = backend.run(circuit or qc, shots=args.shots, seed_simulator=77 + i, **opts).result()
        backend.last_engine.sync()
        comm.barrier()
        dt = max(comm.allgather(time.perf_counter() - t0))
        m = r.metadata(0)
        variants[name] = {"shots_per_s": args.shots * n / dt, "ms_per_step": dt / n * 1e3,
                          "device_ops": m["n_device_ops"], "evolve_ms": m["time_evolve"] * 1e3}

    if not args.no_variants:
        n = max(2, args.steps // 2)
        leg("zero_tracking (opt-in: skips the provably-zero part of the vector)", n, engine_options={"zero_tracking": 1})
        if world == 1 or args.with_exchange:
            # N > 1: the unfused stream needs shard-bit exchanges (RCCL); opt-in there, because a
            # first-ever RCCL bring-up must not be able to take the main measurement down with it
            leg("unfused reference-order gate stream (fusion=0)", 1, fusion=0, engine_options={"zero_tracking": 0})
        if world == 1:
            from qcmrf_amd.transpile import transpile
            leg("lowered to {cx,id,rz,sx,x} as run_experiment.py:52 (stand-in transpiler, not timed) -> fusion 3",
                2, circuit=transpile(qc), fusion=args.fusion)
        backend.run(qc, shots=16, engine_options={"zero_tracking": 0})

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        dom = max(agg, key=lambda k: agg[k]["ms"])
        d = agg[dom]
        achieved = d["bytes"] / d["ms"] / 1e6                 # GB/s, per launch average
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get("k_" + dom, {}).get("hbm_bytes_per_launch_W%d" % W)
            except Exception:
                traffic = None
        line = {
            "metric": "shots/sec, n-qubit QCMRF circuit (fp64 statevector, ingest+evolve+sample)",
            "value": args.shots * args.steps / elapsed, "unit": "shots/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": name, "qubits": W, "shots": args.shots, "fusion": args.fusion,
                       "layout": args.layout, "state_GiB": 16.0 * 2 ** W / 2 ** 30,
                       "sweeps_per_step": sum(a["launches"] for k, a in agg.items() if k != "prob") // args.steps,
                       "gates_per_step_on_device": meta["n_device_ops"] - 1,
                       "source_gates": meta["n_source_ops"], "exchanges_per_step": meta["n_exchanges"],
                       "parallelism": "amplitude shards by high qubit x%d" % (args.virtual_shards or args.gpus)
                                      + (" (virtual shards on one device)" if args.virtual_shards else "")},
            "breakdown_ms": {"compile": t_compile / args.steps * 1e3, "evolve": t_evolve / args.steps * 1e3,
                             "sample": t_sample / args.steps * 1e3},
            "kernels": {k: {"launches_per_step": a["launches"] / args.steps,
                            "avg_ms": a["ms"] / a["launches"],
                            "GBps": a["bytes"] / a["ms"] / 1e6 if a["ms"] > 0 else None} for k, a in agg.items()},
            "roofline": {"bound": "hbm", "kernel": "k_" + dom, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                         "avg_launch_ms": d["ms"] / d["launches"], "rank": 0},
        }
        line["variants"] = variants
        if args.gpus == 1 and not args.no_cpu:
            backend.close()
            line["cpu_baseline"] = cpu_baseline(cliques, theta, args.shots, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    backend.close()
    if world > 1:
        comm.barrier()


if __name__ == "__main__":
    main()
