"""ctypes binding of libqsv.so (C ABI: include/qsv.h).

The library is hand-written HIP for gfx950 and is the only execution path: there is no CPU
fallback.  If the shared object is missing, or no HIP device is visible, the functions here
raise -- loudly -- instead of computing anything on the host.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QSV_LIBRARY") or os.path.join(_HERE, "csrc", "libqsv.so")   # QSV_LIBRARY: kernel experiments (scripts/)

MAX_CTRL = 16
MAX_KQ = 5
UNIQUE_ID_BYTES = 128
IPC_HANDLE_BYTES = 64

K_NAMES = ["init", "1q", "x", "diag", "mcphase", "mux", "kq", "prob", "swap", "exchange", "multi", "multi_init", "init_prod"]
K_COUNT = len(K_NAMES)

# defaults of the qsv_set_option knobs (qsv.hip: struct qsv_handle), so that a per-run override can be undone
OPTION_DEFAULTS = {"blocks_per_cu": 1 << 16, "unroll": 4, "lowt_shuffle": 1, "nontemporal": -1, "lane_targets": 1,
                   "cache_sums": 1, "fused_sums": 1, "pair_variant": 0, "kq_mfma": 1, "zero_tracking": 0, "lane_map": 1,
                   "init_prod_bit0": 0, "init_prod_r": 0, "init_prod": 1, "pass_hints": 1, "dyn_lanes": 3, "multi_r": 5,
                   "exchange_chunk_log2": 24, "xframe": 1, "pass_budget": 0, "trace_passes": 0, "single_shortcut": 1, "pass_max_ops": 56, "general_r": 4, "general_light_r": 5, "swizzle": 1, "lane_map_min_l": 26, "blocksum_variant": 6, "kq_chunked": 0, "fold_init_h": 1, "general_combos": 1, "lowctl_mask": 1, "kq_variant": -1, "kq3_tile": 1, "kq_blocks_per_cu": 0, "kq_debug": 0, "kq_order": 1, "multi_nt": -1, "init_prod_nt": -1}

OP_INIT_ZERO, OP_INIT_UNIFORM, OP_1Q, OP_MCX, OP_DIAG, OP_MCPHASE, OP_MUX, OP_KQ, OP_SWAP = range(9)
OPF_NEW_PASS = 1


class KindStats(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("algorithmic_bytes", C.c_double), ("device_ms", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("per_kind", KindStats * K_COUNT), ("exchanges", C.c_uint64), ("exchange_bytes", C.c_double),
                ("fused_gates", C.c_uint64)]


class QsvOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("target", C.c_int32), ("n", C.c_int32), ("flags", C.c_int32),
                ("qubits", C.c_int32 * MAX_CTRL), ("vals", C.c_int32 * MAX_CTRL),
                ("data_off", C.c_uint64), ("mask", C.c_uint64), ("angle", C.c_double)]


OP_DTYPE = np.dtype([("kind", "<i4"), ("target", "<i4"), ("n", "<i4"), ("flags", "<i4"),
                     ("qubits", "<i4", (MAX_CTRL,)), ("vals", "<i4", (MAX_CTRL,)),
                     ("data_off", "<u8"), ("mask", "<u8"), ("angle", "<f8")])
assert OP_DTYPE.itemsize == C.sizeof(QsvOp)

_P = C.POINTER
_i, _u64, _d, _vp = C.c_int, C.c_uint64, C.c_double, C.c_void_p
_ip, _dp, _u64p = _P(C.c_int), _P(C.c_double), _P(C.c_uint64)

# name -> (restype, argtypes); every symbol include/qsv.h declares
SIGNATURES = {
    "qsv_device_count": (_i, []),
    "qsv_create": (_i, [_i, _i, _ip, _P(_vp)]),
    "qsv_create_rank": (_i, [_i, _i, _i, _i, _P(_vp)]),
    "qsv_comm_unique_id": (_i, [_P(C.c_uint8)]),
    "qsv_comm_init": (_i, [_vp, _P(C.c_uint8)]),
    "qsv_destroy": (_i, [_vp]),
    "qsv_sync": (_i, [_vp]),
    "qsv_init_zero": (_i, [_vp]),
    "qsv_init_uniform": (_i, [_vp, _u64]),
    "qsv_apply_1q": (_i, [_vp, _i, _dp]),
    "qsv_apply_mc1q": (_i, [_vp, _i, _ip, _ip, _i, _dp]),
    "qsv_apply_mcx": (_i, [_vp, _i, _ip, _ip, _i]),
    "qsv_apply_diag": (_i, [_vp, _i, _ip, _dp]),
    "qsv_apply_mcphase": (_i, [_vp, _i, _ip, _ip, _d]),
    "qsv_apply_mux_1q": (_i, [_vp, _i, _ip, _i, _dp]),
    "qsv_apply_kq": (_i, [_vp, _i, _ip, _dp]),
    "qsv_swap_layout": (_i, [_vp, _i, _ip, _ip]),
    "qsv_probabilities": (_i, [_vp, _ip, _i, _dp]),
    "qsv_probabilities_cond": (_i, [_vp, _ip, _i, _u64, _u64, _dp]),
    "qsv_norm": (_i, [_vp, _dp]),
    "qsv_expect_diag": (_i, [_vp, _ip, _i, _dp, _u64, _u64, _dp]),
    "qsv_sample": (_i, [_vp, _u64, _u64, _ip, _i, _u64p]),
    "qsv_get_amplitudes": (_i, [_vp, _u64, _u64, _dp]),
    "qsv_set_amplitudes": (_i, [_vp, _u64, _u64, _dp]),
    "qsv_copy_state": (_i, [_vp, _vp]),
    "qsv_exec": (_i, [_vp, _vp, _i, _dp, _u64]),
    "qsv_set_profiling": (_i, [_vp, _i]),
    "qsv_reset_stats": (_i, [_vp]),
    "qsv_get_stats": (_i, [_vp, _P(Stats)]),
    "qsv_timer_begin": (_i, [_vp]),
    "qsv_timer_end": (_i, [_vp, _dp]),
    "qsv_set_option": (_i, [_vp, C.c_char_p, _i]),
    "qsv_last_error": (C.c_char_p, []),
    "qsv_version": (C.c_char_p, []),
    "qsv_device_memory": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "qsv_rccl_selftest": (C.c_int, [C.c_int, C.c_uint64]),
    "qsv_rccl_exchange_selftest": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "qsv_poison_lds": (C.c_int, [C.c_void_p]),
    "qsv_device_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "qsv_ipc_export": (C.c_int, [_vp, C.POINTER(C.c_uint8)]),
    "qsv_ipc_attach": (C.c_int, [_vp, C.POINTER(C.c_uint8), C.c_char_p, C.c_int]),
}

_lib = None


def load():
    """dlopen libqsv.so and bind every declared symbol.  Raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libqsv.so not found at %s -- build it first (python -c 'import __graft_entry__ as g; g.build()' "
            "or python -m qcmrf_amd.build).  There is no CPU fallback." % LIB_PATH)
    # the peer-mapped exchange transport shares shards between processes through dmabuf IPC: the
    # HSA runtime has to be told before it initialises (the pool's boxes export this already)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _raise(code):
    msg = load().qsv_last_error().decode("utf-8", "replace")
    if code == -1:
        raise ValueError("qsv: " + msg)
    if code == -2:
        raise MemoryError("qsv: " + msg)
    raise RuntimeError("qsv (%d): %s" % (code, msg))


def _chk(code):
    if code != 0:
        _raise(code)


def _ia(seq):
    a = np.ascontiguousarray(seq, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def _da(arr):
    a = np.ascontiguousarray(arr, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _cflat(arr):
    """complex array -> interleaved float64 view"""
    return np.ascontiguousarray(arr, dtype=np.complex128).view(np.float64).ravel()


def device_count():
    n = load().qsv_device_count()
    if n < 0:
        _raise(n)
    return n


def device_memory(device=0):
    """(free, total) bytes of HBM on one device"""
    f, t = C.c_uint64(0), C.c_uint64(0)
    _chk(load().qsv_device_memory(int(device), C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


def rccl_selftest(device=0, n_doubles=1 << 20):
    """1-rank RCCL bring-up + grouped send/recv to self on one device (diagnostic)"""
    _chk(load().qsv_rccl_selftest(int(device), int(n_doubles)))


def rccl_exchange_selftest(device=0, n_qubits=22, chunk_log2=14):
    """the RCCL pipeline of a batched exchange (double buffered, several peers) on a 1-rank communicator"""
    _chk(load().qsv_rccl_exchange_selftest(int(device), int(n_qubits), int(chunk_log2)))


def device_bus_id(device=0):
    buf = C.create_string_buffer(64)
    _chk(load().qsv_device_bus_id(int(device), buf, 64))
    return buf.value.decode()


def comm_unique_id():
    buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
    _chk(load().qsv_comm_unique_id(buf))
    return bytes(buf)


class Engine:
    """One statevector (all shards owned by this process) behind an opaque qsv handle.

    Qubit arguments are PHYSICAL positions; see qcmrf_amd.planner for the logical layout."""

    def __init__(self, n_qubits, devices=(0,), rank=None, world_size=None):
        self._lib = load()
        self._h = _vp()
        self.n_qubits = int(n_qubits)
        if rank is None:
            dev, dp = _ia(list(devices))
            self.n_shards = len(dev)
            self.world = self.n_shards
            self.rank = 0
            self.multiproc = False
            _chk(self._lib.qsv_create(self.n_qubits, len(dev), dp, C.byref(self._h)))
        else:
            self.n_shards = 1
            self.world = int(world_size)
            self.rank = int(rank)
            self.multiproc = self.world > 1
            _chk(self._lib.qsv_create_rank(self.n_qubits, self.world, self.rank, int(devices[0]), C.byref(self._h)))
        g = self.world.bit_length() - 1
        self.local_qubits = self.n_qubits - g

    # -- life cycle
    def close(self):
        if self._h:
            self._lib.qsv_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def comm_init(self, unique_id):
        buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        _chk(self._lib.qsv_comm_init(self._h, buf))

    def device_identity(self, device):
        """(host, PCI bus id) of a HIP device: two ranks with the same identity share one GPU"""
        import socket
        return socket.gethostname(), device_bus_id(device)

    def comm_bootstrap(self, comm, device=0, transport="auto"):
        """collective over ``comm``: set up the exchange transport between the ranks.

        'rccl'  rank 0 mints the RCCL unique id, everyone joins (RCCL over xGMI; the default
                whenever every rank has a GPU of its own)
        'p2p'   every rank maps the others' shards through HIP IPC and a pair swaps in place
        'auto'  p2p if two ranks share a device (RCCL refuses that: the one-GPU test box), else rccl"""
        if transport == "auto":
            ids = comm.allgather(self.device_identity(device))
            transport = "p2p" if len(set(ids)) < len(ids) else "rccl"
        if transport == "rccl":
            uid = comm.bcast(comm_unique_id() if comm.rank == 0 else None, src=0)
            self.comm_init(uid)
        elif transport == "p2p":
            buf = (C.c_uint8 * IPC_HANDLE_BYTES)()
            _chk(self._lib.qsv_ipc_export(self._h, buf))
            handles = b"".join(comm.allgather(bytes(buf)))
            name = comm.bcast("/qsv_%d_%d" % (os.getpid(), int.from_bytes(os.urandom(4), "little")) if comm.rank == 0 else None)
            hb = (C.c_uint8 * len(handles)).from_buffer_copy(handles)
            if comm.rank == 0:                      # create the segment before anyone else opens it
                _chk(self._lib.qsv_ipc_attach(self._h, hb, name.encode(), 1))
            comm.barrier()
            if comm.rank != 0:
                _chk(self._lib.qsv_ipc_attach(self._h, hb, name.encode(), 0))
            comm.barrier()
            if comm.rank == 0:                      # every rank holds its mapping: the name can go
                try:
                    os.unlink("/dev/shm" + name)
                except OSError:
                    pass
        else:
            raise ValueError("transport must be 'auto', 'rccl' or 'p2p', not %r" % (transport,))
        self.transport = transport
        return transport

    def sync(self):
        _chk(self._lib.qsv_sync(self._h))

    def poison_lds(self):
        """diagnostic: quiet NaNs into the LDS of every compute unit (qsv_poison_lds)"""
        _chk(self._lib.qsv_poison_lds(self._h))

    def set_option(self, name, value):
        _chk(self._lib.qsv_set_option(self._h, name.encode(), int(value)))

    # -- state preparation
    def init_zero(self):
        _chk(self._lib.qsv_init_zero(self._h))

    def init_uniform(self, mask):
        _chk(self._lib.qsv_init_uniform(self._h, int(mask)))

    # -- gates
    def apply_1q(self, t, m, ctrls=(), ctrl_vals=None):
        ma, mp = _da(_cflat(m))
        if len(ctrls) == 0:
            _chk(self._lib.qsv_apply_1q(self._h, int(t), mp))
        else:
            ca, cp = _ia(ctrls)
            va, vp = _ia([1] * len(ctrls) if ctrl_vals is None else ctrl_vals)
            _chk(self._lib.qsv_apply_mc1q(self._h, len(ca), cp, vp, int(t), mp))

    def apply_mcx(self, ctrls, t, ctrl_vals=None):
        ca, cp = _ia(ctrls)
        va, vp = _ia([1] * len(ca) if ctrl_vals is None else ctrl_vals)
        _chk(self._lib.qsv_apply_mcx(self._h, len(ca), cp, vp, int(t)))

    def apply_diag(self, qubits, table):
        qa, qp = _ia(qubits)
        ta, tp = _da(_cflat(table))
        if ta.size != 2 << len(qa):
            raise ValueError("diag table has %d entries for %d qubits" % (ta.size // 2, len(qa)))
        _chk(self._lib.qsv_apply_diag(self._h, len(qa), qp, tp))

    def apply_mcphase(self, qubits, angle, vals=None):
        qa, qp = _ia(qubits)
        va, vp = _ia([1] * len(qa) if vals is None else vals)
        _chk(self._lib.qsv_apply_mcphase(self._h, len(qa), qp, vp, float(angle)))

    def apply_mux(self, ctrls, t, mats):
        ca, cp = _ia(ctrls)
        ma, mp = _da(_cflat(mats))
        if ma.size != 8 << len(ca):
            raise ValueError("mux needs %d matrices" % (1 << len(ca)))
        _chk(self._lib.qsv_apply_mux_1q(self._h, len(ca), cp, int(t), mp))

    def apply_kq(self, qubits, u):
        qa, qp = _ia(qubits)
        ua, up = _da(_cflat(u))
        if ua.size != 2 << (2 * len(qa)):
            raise ValueError("kq matrix has wrong size")
        _chk(self._lib.qsv_apply_kq(self._h, len(qa), qp, up))

    def swap_layout(self, a, b):
        aa, ap = _ia(a)
        ba, bp = _ia(b)
        _chk(self._lib.qsv_swap_layout(self._h, len(aa), ap, bp))

    def exec(self, ops, data):
        """ops: numpy structured array of OP_DTYPE; data: float64 pool."""
        ops = np.ascontiguousarray(ops, dtype=OP_DTYPE)
        da, dp = _da(data if len(data) else np.zeros(1))
        _chk(self._lib.qsv_exec(self._h, ops.ctypes.data_as(_vp), len(ops), dp, len(data)))

    # -- measurement
    def probabilities(self, qubits, fix_mask=0, fix_val=0):
        qa, qp = _ia(qubits)
        out = np.zeros(1 << len(qa), dtype=np.float64)
        _chk(self._lib.qsv_probabilities_cond(self._h, qp, len(qa), int(fix_mask), int(fix_val),
                                              out.ctypes.data_as(_dp)))
        return out

    def expect_diag(self, qubits, table, fix_mask=0, fix_val=0):
        """(sum |amp|^2 table[j], sum |amp|^2) over the indices g with (g & fix_mask) == fix_val; table is a
        REAL diagonal over ``qubits`` (index bit b <-> qubits[b]); this process's shards only"""
        qa, qp = _ia(qubits)
        ta, tp = _da(table)
        if ta.size != 1 << len(qa):
            raise ValueError("diagonal observable has %d entries for %d qubits" % (ta.size, len(qa)))
        out = np.zeros(2, dtype=np.float64)
        _chk(self._lib.qsv_expect_diag(self._h, qp, len(qa), tp, int(fix_mask), int(fix_val), out.ctypes.data_as(_dp)))
        return float(out[0]), float(out[1])

    def norm(self):
        v = C.c_double()
        _chk(self._lib.qsv_norm(self._h, C.byref(v)))
        return v.value

    def sample(self, shots, seed, meas_qubits=None):
        out = np.zeros(int(shots), dtype=np.uint64)
        if meas_qubits is None:
            _chk(self._lib.qsv_sample(self._h, int(shots), int(seed), None, 0, out.ctypes.data_as(_u64p)))
        else:
            qa, qp = _ia(meas_qubits)
            _chk(self._lib.qsv_sample(self._h, int(shots), int(seed), qp, len(qa), out.ctypes.data_as(_u64p)))
        return out

    def amplitudes(self, start=0, count=None):
        if count is None:
            count = (1 << self.n_qubits) - start
        out = np.empty(int(count), dtype=np.complex128)
        _chk(self._lib.qsv_get_amplitudes(self._h, int(start), int(count), out.view(np.float64).ctypes.data_as(_dp)))
        return out

    def copy_from(self, other):
        """device-to-device copy of another engine's state (same shape)"""
        _chk(self._lib.qsv_copy_state(self._h, other._h))

    def set_amplitudes(self, start, values):
        v = np.ascontiguousarray(values, dtype=np.complex128)
        _chk(self._lib.qsv_set_amplitudes(self._h, int(start), len(v), v.view(np.float64).ctypes.data_as(_dp)))

    # -- instrumentation
    def set_profiling(self, on):
        _chk(self._lib.qsv_set_profiling(self._h, int(bool(on))))

    def reset_stats(self):
        _chk(self._lib.qsv_reset_stats(self._h))

    def stats(self):
        st = Stats()
        _chk(self._lib.qsv_get_stats(self._h, C.byref(st)))
        out = {"exchanges": int(st.exchanges), "exchange_bytes": float(st.exchange_bytes),
               "fused_gates": int(st.fused_gates), "kinds": {}}
        for i, name in enumerate(K_NAMES):
            k = st.per_kind[i]
            if k.launches:
                out["kinds"][name] = {"launches": int(k.launches), "bytes": float(k.algorithmic_bytes),
                                      "ms": float(k.device_ms)}
        return out

    def timer_begin(self):
        _chk(self._lib.qsv_timer_begin(self._h))

    def timer_end(self):
        v = C.c_double()
        _chk(self._lib.qsv_timer_end(self._h, C.byref(v)))
        return v.value
