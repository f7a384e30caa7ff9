"""Drop-in for the simulator call of the reference:

    simulator = Aer.get_backend('qasm_simulator')          /root/reference/run_experiment.py:54
    result    = simulator.run(T, shots=SHOTS).result()     /root/reference/run_experiment.py:56
    counts    = result.get_counts()                        /root/reference/run_experiment.py:57

``run`` accepts one circuit or a list, nested (un-transpiled ``QCMRF`` objects) or lowered to
``{cx,id,rz,sx,x}``; counts come back as plain ``{bitstring: int}`` dicts (classical bit W-1
leftmost), a list of them when more than one circuit ran -- ``json.dumps``-able as
run_experiment.py:59-61 requires.

Execution = ingest -> exact fusion passes -> layout/shard plan -> ONE ``qsv_exec`` call into
libqsv.so (hand-written HIP, gfx950) -> probability pass + sampling on the device.  There is no
host-side simulation path: without the library or a GPU, ``run`` raises.
"""
from __future__ import annotations

import time

import numpy as np

from . import _lib, ingest as _ingest, passes, planner, program
from .comm import SingleProcess

_NAMES = ("qasm_simulator", "aer_simulator", "aer_simulator_statevector", "statevector_simulator",
          "qsv_simulator")


class Result:
    def __init__(self, experiments, backend_name):
        self._exps = experiments
        self.backend_name = backend_name
        self.success = True
        self.time_taken = sum(e["metadata"]["time_taken"] for e in experiments)

    @property
    def results(self):
        return self._exps

    def get_counts(self, experiment=None):
        if experiment is None:
            cs = [dict(e["counts"]) for e in self._exps]
            return cs[0] if len(cs) == 1 else cs
        if isinstance(experiment, int):
            return dict(self._exps[experiment]["counts"])
        for e in self._exps:
            if e["name"] == getattr(experiment, "name", experiment):
                return dict(e["counts"])
        raise KeyError("no experiment %r" % (experiment,))

    def metadata(self, i=0):
        return self._exps[i]["metadata"]

    def to_dict(self):
        return {"backend_name": self.backend_name, "success": True,
                "results": [{"name": e["name"], "shots": e["shots"], "counts": e["counts"],
                             "metadata": e["metadata"]} for e in self._exps]}


class Job:
    def __init__(self, result):
        self._result = result

    def result(self):
        return self._result

    def status(self):
        return "DONE"


def _format_keys(values, counts, num_clbits, creg_sizes):
    """integer outcomes (bit c = classical bit c) -> Qiskit count keys, vectorised:
    one (keys x width) matrix of '0'/'1' bytes, decoded once and sliced"""
    w = max(num_clbits, 1)
    values = np.ascontiguousarray(values, dtype=np.uint64)
    if w <= 64:
        # big-endian bytes -> bits -> UCS4 code points '0'/'1', VIEWED as fixed-width strings: no
        # per-key conversion at all (an S -> U astype costs more than everything else here together)
        bits = np.unpackbits(values.astype(">u8").view(np.uint8).reshape(-1, 8), axis=1)[:, 64 - w:]
        code = np.empty(bits.shape, dtype=np.uint32)
        np.add(bits, 48, out=code, casting="unsafe")
        keys = code.view("<U%d" % w).ravel().tolist()
    else:
        shifts = np.arange(w - 1, -1, -1, dtype=np.uint64)
        chars = (((values[:, None] >> np.minimum(shifts, np.uint64(63))) & np.uint64(1)) * (shifts < 64) + np.uint64(48)).astype(np.uint8)
        text = chars.tobytes().decode("ascii")
        keys = [text[i * w:(i + 1) * w] for i in range(len(values))]
    if creg_sizes and len(creg_sizes) > 1:           # one group per register, last register first
        cuts, hi = [], num_clbits
        for _, size in reversed(creg_sizes):
            cuts.append((num_clbits - hi, num_clbits - hi + size))
            hi -= size
        keys = [" ".join(k[a:b] for a, b in cuts) for k in keys]
    return dict(zip(keys, counts.tolist()))


class QsvBackend:
    """MI355X statevector backend.  Options (constructor or per ``run`` call):

    fusion      0 gate by gate | 1 + init/diagonal fusion | 2 + multiplexer fusion |
                3 + dense <=5-qubit windows with structure recovery (default)
    fold_fresh  (fusion 3) gates whose target nothing has touched yet become factors of the
                initial product state, written in the same pass as the state itself (default on)
    layout      'auto' (exchange-free where possible) | 'reference' (qubit q on index bit q)
    devices     HIP device id per shard owned by this process (repeat an id for virtual shards)
    method      'statevector' (default: all measurements deferred, one evolution, W qubits) |
                'trajectory' (mid-circuit measurements taken when they occur, measured qubits
                released: n+2 live qubits for a QCMRF circuit, see qcmrf_amd.trajectory)
    comm        process group for one-process-per-GPU launches (qcmrf_amd.comm)
    gather_counts  'root' (default: rank 0 returns the merged counts, the other ranks an empty dict) | 'all'
    spmd_ingest    (multi-rank) each rank reads 1/N of the circuit's composite blocks, one all-gather completes the program
                   (default off: in the only rehearsal available -- ranks sharing one GPU -- the extra collective cost more
                   than the reading it saved)
    device      HIP device of this rank when ``comm`` is given
    """

    def __init__(self, name="qasm_simulator", **options):
        self._name = name
        self.options = {"fusion": 3, "layout": "auto", "devices": (0,), "comm": None, "device": 0,
                        "profile": False, "engine_options": None, "method": "statevector", "fold_fresh": True,
                        "gather_counts": "root"}
        self.options.update(options)
        self._engine = None
        self._engine_key = None
        self.last_engine = None
        self.last_plan = None
        self._last_comm = None
        self._engine_factory = None      # test hook only; the default and only shipped engine is libqsv

    def name(self):
        return self._name

    def set_options(self, **options):
        self.options.update(options)

    # ---- engine cache: re-use the device allocation across circuits of one width ---------
    def _get_engine(self, n_qubits, opts):
        comm = opts["comm"] or SingleProcess()
        if comm.world > 1:
            key = (n_qubits, "rank", comm.rank, comm.world, opts["device"])
        else:
            key = (n_qubits, tuple(opts["devices"]))
        if self._engine is not None and self._engine_key == key:
            return self._engine
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        make = self._engine_factory or _lib.Engine
        if comm.world > 1:
            eng = make(n_qubits, devices=(opts["device"],), rank=comm.rank, world_size=comm.world)
            eng._comm_ready = False
        else:
            eng = make(n_qubits, devices=tuple(opts["devices"]))
            eng._comm_ready = True
        self._engine, self._engine_key = eng, key
        return eng

    def statevector(self):
        """amplitudes left by the last run, in LOGICAL qubit order (diagnostic; copies 2^W values
        to the host, so small circuits only).  Index bit q = logical qubit q."""
        eng, pl = self.last_engine, self.last_plan
        amp = eng.amplitudes()
        p = np.arange(amp.size, dtype=np.int64)
        l = np.zeros_like(p)
        for q, pos in enumerate(pl.layout):
            l |= ((p >> pos) & 1) << q
        out = np.empty_like(amp)
        out[l] = amp
        return out

    def expectation_diagonal(self, diag, qubits, fixed=None):
        """(sum |amp|^2 diag[j], sum |amp|^2) over the basis states in which every qubit of ``fixed``
        ({logical qubit: 0/1}) has the given value, on the state the LAST run left in HBM; ``diag`` is
        a real diagonal over the logical ``qubits`` (index bit b <-> qubits[b]).  The first divided by
        the second is the expectation in the post-selected state.  One read pass on the device
        (qsv_expect_diag); summed over the ranks of a multi-process run.  Replaces the opflow
        expectation of QCMRF.Hamiltonian() / sufficient_statistic() (QCMRF.py:159-193)."""
        eng, pl = self.last_engine, self.last_plan
        if eng is None or pl is None:
            raise RuntimeError("expectation_diagonal needs a state: run() a circuit on this backend first")
        phys = [pl.layout[q] for q in qubits]
        fm = fv = 0
        for q, v in (fixed or {}).items():
            fm |= 1 << pl.layout[q]
            fv |= int(bool(v)) << pl.layout[q]
        s0, s1 = eng.expect_diag(phys, np.ascontiguousarray(diag, dtype=np.float64), fm, fv)
        comm = self._last_comm
        if comm is not None and comm.world > 1:
            parts = comm.allgather((s0, s1))
            s0, s1 = sum(p[0] for p in parts), sum(p[1] for p in parts)
        return s0, s1

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    # ---- the call run_experiment.py:56 makes -------------------------------------------------
    def run(self, circuits, shots=1024, seed_simulator=None, **run_options):
        opts = dict(self.options)
        opts.update(run_options)
        single = not isinstance(circuits, (list, tuple))
        circs = [circuits] if single else list(circuits)
        if seed_simulator is None:
            seed_simulator = int(np.random.SeedSequence().entropy % (2 ** 63))
        exps = []
        if len(circs) > 1 and opts.get("method", "statevector") == "statevector":
            # a batch (run_experiment.py:52-56 hands over 70 circuits): compile circuit i+1 on a
            # helper thread while the device evolves and samples circuit i (ctypes releases the GIL
            # inside the blocking library calls)
            from concurrent.futures import ThreadPoolExecutor
            opts = dict(opts, _batch=True)          # (the helper thread must not talk on the process group's socket: no SPMD ingest here)
            with ThreadPoolExecutor(max_workers=1) as pool:
                nxt = pool.submit(self._prepare, circs[0], opts)
                for i in range(len(circs)):
                    prepared = nxt.result()
                    if i + 1 < len(circs):
                        nxt = pool.submit(self._prepare, circs[i + 1], opts)
                    exps.append(self._run_one(circs[i], int(shots), int(seed_simulator) + i, opts, prepared))
        else:
            for i, c in enumerate(circs):
                exps.append(self._run_one(c, int(shots), int(seed_simulator) + i, opts))
        return Job(Result(exps, self._name))

    def _prepare(self, circuit, opts):
        """host half of a run: ingest + passes + plan + encode (no device work)"""
        t0 = time.perf_counter()
        comm = opts["comm"] or SingleProcess()
        n_shards = comm.world if comm.world > 1 else len(opts["devices"])
        ing, pl = self.compile(circuit, n_shards, ingest_comm=comm if (comm.world > 1 and opts.get("spmd_ingest", False) and not opts.get("_batch")) else None,
                               **{k: opts[k] for k in ("fusion", "layout", "engine_options", "fold_fresh")})
        rec, data = program.encode(pl.ops)
        return ing, pl, rec, data, n_shards, time.perf_counter() - t0

    def compile(self, circuit, n_shards=1, **options):
        """ingest + passes + plan only (no GPU): returns (Ingested, Plan)"""
        opts = dict(self.options)
        opts.update(options)
        ing = _ingest.ingest(circuit, peephole=opts["fusion"] >= 1, comm=opts.get("ingest_comm"), compact=opts["fusion"] >= 3)
        # every qubit of a dense window has to be local to a shard at the same time
        g = max(1, int(n_shards)).bit_length() - 1
        ops = passes.optimise(ing.ops, level=opts["fusion"], fresh=bool(opts.get("fold_fresh", True)),
                              dense_kmax=max(1, min(5, ing.num_qubits - g)), flat=ing.flat)
        if ing.global_phase and opts.get("apply_global_phase", True):
            from . import ir
            ph = np.exp(1j * ing.global_phase)
            host = next((o for o in ops if o.kind == "diag"), None)
            if host is not None:                             # rides in a diagonal that is there anyway (one factor fewer on the device)
                host.table = host.table * ph
            else:
                ops.append(ir.op_diag([0], [ph, ph]))
        eo = opts.get("engine_options") or {}
        lane = bool(eo.get("lane_targets", 1)) and not eo.get("zero_tracking", 0)   # lane targets need a fully populated vector
        pl = planner.plan(ops, ing.num_qubits, n_shards, opts["layout"], lane_targets=lane,
                          dyn_lanes=int(eo.get("dyn_lanes", planner.DYN_LANES)))
        return ing, pl

    def _run_trajectory(self, circuit, shots, seed, opts):
        from . import trajectory
        t0 = time.perf_counter()
        vals, cnts, num_clbits, creg_sizes, meta = trajectory.run_trajectories(
            circuit, shots, seed, fusion=opts["fusion"], device=tuple(opts["devices"])[0],
            engine_factory=self._engine_factory)
        agg = {}
        for v, c in zip(vals.tolist(), cnts.tolist()):
            agg[v] = agg.get(v, 0) + c
        uv = np.fromiter(agg.keys(), dtype=np.uint64, count=len(agg))
        uc = np.fromiter(agg.values(), dtype=np.int64, count=len(agg))
        counts = _format_keys(uv, uc, num_clbits, creg_sizes) if len(agg) else {}
        meta.update({"n_qubits": int(circuit.num_qubits), "time_taken": time.perf_counter() - t0,
                     "time_sample": 0.0, "seed_simulator": seed, "fusion": opts["fusion"]})
        return {"name": getattr(circuit, "name", "circuit"), "shots": shots, "counts": counts, "metadata": meta}

    @staticmethod
    def _to_clbits(bits, clist, direct):
        """sampled words -> classical-register values (bit c = classical bit c)"""
        if direct:
            return bits
        vals = np.zeros(bits.shape, dtype=np.uint64)
        for j, c in enumerate(clist):
            vals |= ((bits >> np.uint64(j)) & np.uint64(1)) << np.uint64(c)
        return vals

    def _run_one(self, circuit, shots, seed, opts, prepared=None):
        if opts.get("method", "statevector") == "trajectory":
            return self._run_trajectory(circuit, shots, seed, opts)
        comm = opts["comm"] or SingleProcess()
        ing, pl, rec, data, n_shards, t_compile = prepared if prepared is not None else self._prepare(circuit, opts)
        t1 = time.perf_counter()
        t0 = t1 - t_compile

        eng = self._get_engine(ing.num_qubits, opts)
        # engine options are per RUN: whatever an earlier call set on the cached engine and this one
        # does not ask for goes back to the library default first (the plan above was compiled for
        # exactly this call's options)
        want = dict(opts["engine_options"] or {})
        applied = getattr(eng, "_applied_options", None)
        if applied is None:
            applied = eng._applied_options = {}
        for k in [k for k in applied if k not in want]:
            if k in _lib.OPTION_DEFAULTS:
                eng.set_option(k, _lib.OPTION_DEFAULTS[k])
            del applied[k]
        for k, v in want.items():
            eng.set_option(k, v)
            applied[k] = v
        if pl.n_exchanges and not eng._comm_ready:
            # collective: RCCL communicator over all ranks, or peer-mapped shards (ranks sharing a GPU)
            eng.comm_bootstrap(comm, device=opts["device"], transport=opts.get("exchange", "auto"))
            eng._comm_ready = True
        if opts["profile"]:
            eng.set_profiling(True)
        eng.reset_stats()
        eng.exec(rec, data)
        eng.sync()
        t2 = time.perf_counter()

        clist = sorted(ing.measure)
        # output bit c <- the qubit measured into classical bit c (-1: never written, stays 0), so
        # the sampled words come back already laid out as the classical register
        if clist and clist[-1] < 64:
            meas_phys = [-1] * (clist[-1] + 1)
            for c in clist:
                meas_phys[c] = pl.layout[ing.measure[c]]
            direct = True
        else:
            meas_phys = [pl.layout[ing.measure[c]] for c in clist]
            direct = False
        counts = {}
        if clist and shots > 0:
            if comm.world > 1:
                # One round trip and one one-way send per run: (1) all-gather of one double per rank, the shard's
                # probability mass -> the common multinomial split of the shots over the shards (same seed on every
                # rank); (2) each rank draws exactly split[rank] outcomes from its own shard and SENDS them to rank 0
                # as (distinct outcome, count) pairs -- a few hundred pairs -- without waiting for anything back.
                # Only rank 0 merges and formats: Result.get_counts() is the merged dict there and empty on the
                # other ranks (metadata "counts_on_rank": 0), whose step ends with the send -- formatting 4096 keys on
                # every rank was host time that does not shrink with the rank count (``gather_counts="all"``
                # restores the all-gather: every rank then returns the full dict).
                masses = np.asarray(comm.allgather_f64(eng.norm()) if hasattr(comm, "allgather_f64")
                                    else comm.allgather(eng.norm()), dtype=np.float64)
                split = np.random.RandomState(seed % (2 ** 32)).multinomial(shots, masses / masses.sum())
                k = int(split[comm.rank])
                mine = eng.sample(k, (seed * 1315423911 + comm.rank) % (2 ** 63), meas_phys) if k else np.zeros(0, dtype=np.uint64)
                uv, uc = np.unique(self._to_clbits(mine, clist, direct), return_counts=True)
                pairs = np.concatenate([uv, uc.astype(np.uint64)])
                everyone = opts.get("gather_counts", "root") == "all"
                if everyone or not hasattr(comm, "gather_bytes"):
                    raw = comm.allgather_bytes(pairs.tobytes()) if hasattr(comm, "allgather_bytes") else [np.asarray(b, dtype=np.uint64).tobytes() for b in comm.allgather(pairs)]
                else:
                    raw = comm.gather_bytes(pairs.tobytes())
                if raw is not None:
                    parts = [np.frombuffer(b, dtype=np.uint64) for b in raw]
                    av = np.concatenate([b[:b.size // 2] for b in parts])
                    ac = np.concatenate([b[b.size // 2:] for b in parts]).astype(np.int64)
                    # shards that differ only in an unmeasured qubit can produce the same outcome
                    uv, inv = np.unique(av, return_inverse=True)
                    uc = np.bincount(inv, weights=ac, minlength=uv.size).astype(np.int64) if uv.size != av.size else ac[np.argsort(av, kind="stable")]
                    counts = _format_keys(uv, uc, ing.num_clbits, ing.creg_sizes)
            else:
                vals = self._to_clbits(eng.sample(shots, seed, meas_phys), clist, direct)
                uv, uc = np.unique(vals, return_counts=True)
                counts = _format_keys(uv, uc, ing.num_clbits, ing.creg_sizes)
        elif shots > 0:
            counts = {}
        t3 = time.perf_counter()
        meta = {"n_qubits": ing.num_qubits, "n_source_ops": ing.n_source_ops, "n_device_ops": len(pl.ops),
                "n_exchanges": pl.n_exchanges, "n_shards": n_shards, "layout": list(pl.layout),
                "fusion": opts["fusion"], "time_compile": t1 - t0, "time_evolve": t2 - t1,
                "time_sample": t3 - t2, "time_taken": t3 - t0, "seed_simulator": seed}
        if comm.world > 1 and opts.get("gather_counts", "root") != "all":
            meta["counts_on_rank"] = 0
        if opts["profile"]:
            meta["stats"] = eng.stats()
            eng.set_profiling(False)
        self.last_engine = eng
        self.last_plan = pl
        self._last_comm = comm
        return {"name": getattr(circuit, "name", "circuit"), "shots": shots, "counts": counts, "metadata": meta}


class _Provider:
    """``from qcmrf_amd import Aer`` -> ``Aer.get_backend('qasm_simulator')`` (run_experiment.py:14,54)."""

    def __init__(self):
        self._cache = {}

    def get_backend(self, name="qasm_simulator", **options):
        if name not in _NAMES:
            raise ValueError("unknown backend %r; this provider serves %s" % (name, ", ".join(_NAMES)))
        if options or name not in self._cache:
            b = QsvBackend(name, **options)
            if options:
                return b
            self._cache[name] = b
        return self._cache[name]

    def backends(self):
        return list(_NAMES)


Aer = _Provider()


def get_backend(name="qasm_simulator", **options):
    return Aer.get_backend(name, **options)
