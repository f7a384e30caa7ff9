// qsv.hip -- host side of libqsv.so: handles, shard resolution, launches, sampling, exchange.
// C ABI declared in include/qsv.h (which cites the reference call each entry point replaces).
// gfx950 only; no CPU fallback of any kind: without a HIP device every entry point fails.
#include "qsv_kernels.h"
#include "qsv_kmulti_inst.h"
QSV_KMULTI_FOR_GENERAL(QSV_KMULTI_DECLARE)
QSV_KMULTI_FOR_MODE(QSV_KMULTI_DECLARE, 1)
QSV_KMULTI_FOR_MODE(QSV_KMULTI_DECLARE, 2)
#include "../../include/qsv.h"

#include <rccl/rccl.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(e_ == hipErrorOutOfMemory ? QSV_E_NOMEM : QSV_E_HIP, "%s: %s (%s:%d)",   \
                  #expr, hipGetErrorString(e_), __FILE__, __LINE__);                       \
  } while (0)
#define CHK(expr)                 \
  do {                            \
    int r_ = (expr);              \
    if (r_ != QSV_OK) return r_;  \
  } while (0)

// ------------------------------------------------------------------------------------------
// RCCL, bound lazily (dlopen) so single-GPU use never touches it
// ------------------------------------------------------------------------------------------
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static int rccl_load() {
  if (g_rccl.lib) return QSV_OK;
  const char* names[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
  void* lib = nullptr;
  for (const char* n : names) {
    lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (lib) break;
  }
  if (!lib) return fail(QSV_E_RCCL, "cannot dlopen librccl: %s", dlerror());
#define BIND(field, sym)                                                        \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, sym));     \
  if (!g_rccl.field) return fail(QSV_E_RCCL, "librccl lacks %s", sym)
  BIND(GetUniqueId, "ncclGetUniqueId");
  BIND(CommInitRank, "ncclCommInitRank");
  BIND(CommDestroy, "ncclCommDestroy");
  BIND(Send, "ncclSend");
  BIND(Recv, "ncclRecv");
  BIND(GroupStart, "ncclGroupStart");
  BIND(GroupEnd, "ncclGroupEnd");
  BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
  g_rccl.lib = lib;
  return QSV_OK;
}
// RCCL announces itself on stdout when a communicator comes up ("RCCL version : ..."); stdout
// belongs to the caller (bench.py prints exactly one JSON line there): keep the banner on stderr
struct StdoutToStderr {
  int saved = -1;
  StdoutToStderr() { fflush(stdout); saved = dup(1); if (saved >= 0) dup2(2, 1); }
  ~StdoutToStderr() { if (saved >= 0) { fflush(stdout); dup2(saved, 1); close(saved); } }
};
#define NCCLCHK(expr)                                                                     \
  do {                                                                                    \
    ncclResult_t r_ = (expr);                                                             \
    if (r_ != ncclSuccess)                                                                \
      return fail(QSV_E_RCCL, "%s: %s", #expr, g_rccl.GetErrorString(r_));                \
  } while (0)

// ------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------
struct Pending { int kind; hipEvent_t e0, e1; };

struct Shard {
  int device = 0;
  int index = 0;                 // global shard number (= rank in multi-process mode)
  hipStream_t stream = nullptr;
  cplx* amp = nullptr;
  // small-table arena: pinned host staging + device mirror, bump allocated
  char* h_arena = nullptr;
  char* d_arena = nullptr;
  size_t arena_bytes = 0, arena_top = 0;
  // sampling workspace
  double* d_sums = nullptr;      // one per QSV_SBLOCK amplitudes
  // exchange staging (allocated on first use)
  cplx* xbuf[2] = {nullptr, nullptr};
  size_t xbuf_amps = 0;
  // RCCL exchange pipeline: pack / unpack run on a second stream, overlapping the send/recv of the
  // neighbouring chunk on the main one; two staging slots per direction, one segment per peer
  hipStream_t xstream = nullptr;
  cplx* xsend[2] = {nullptr, nullptr};
  cplx* xrecv[2] = {nullptr, nullptr};
  size_t xstage_amps = 0;
  hipEvent_t ev_packed[2] = {nullptr, nullptr}, ev_sent[2] = {nullptr, nullptr}, ev_unpacked[2] = {nullptr, nullptr};
  hipEvent_t ev_start = nullptr;
  int n_cu = 256;
  uint64_t zmask = 0;            // zero tracking: local bits known |0>; memory with such a bit set is unwritten
  std::vector<double> h_sums;    // host copy of the block sums, valid until the state changes
  bool sums_valid = false;
  // sums left behind by the last k_multi pass of a program, one per workgroup tile (no read pass)
  double* d_tsums = nullptr;
  size_t tsums_cap = 0;
  double* d_super = nullptr;     // one sum per QSV_SUPER tiles (device) ...
  double* h_tsums = nullptr;     // ... and their pinned host copy, valid until the state changes
  size_t h_tsums_cap = 0;
  bool h_tsums_valid = false;
  bool tile_valid = false, tile_fresh = false;
  int tile_R = 0;
  RegPos tile_rp;
  LanePos tile_lp;
  BitIns tile_ins;
  uint64_t tile_nblocks = 0;
  uint64_t tile_xor = 0;         // X frame of that pass: tile addresses are XOR-ed with it
  uint64_t* d_sblk = nullptr;    // sampling scratch (block index, residual, result per shot)
  double* d_sres = nullptr;
  uint64_t* d_sout = nullptr;
  size_t sample_cap = 0;
  hipEvent_t ev_ready = nullptr;  // qsv_copy_state: everything asked of this shard so far has run (the destination's stream waits for it)
  hipEvent_t ev_copied = nullptr; // ... and this shard has been read completely (its own stream waits for that)
  double* d_red = nullptr;       // scratch of the reductions (marginals, expectation partial sums): grown on demand, kept
  size_t red_cap = 0;            // ... in doubles
  std::vector<Pending> pending;
  std::vector<hipEvent_t> free_events;
};

struct qsv_handle {
  int W = 0, L = 0, gbits = 0;   // global qubits, local qubits per shard, shard bits
  int P = 1;                     // total shards
  bool multiproc = false;
  int rank = 0;
  ncclComm_t comm = nullptr;
  // peer-mapped transport: the other ranks' shards (IPC) and the arrival counters shared with them
  std::vector<cplx*> peer_amp;
  std::atomic<uint64_t>* ipc_arrive = nullptr;   // one counter per rank, QSV_IPC_SLOT apart
  size_t ipc_map_bytes = 0;
  uint64_t ipc_seq = 0;
  std::vector<Shard> shards;     // shards owned by this process
  qsv_stats stats;
  bool profiling = false;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  // options
  int opt_blocks_per_cu = 1 << 16;  // measured on MI355X: one tile per workgroup (no grid-stride) streams fastest
  int opt_unroll = 4;
  int opt_lowt_shuffle = 1;
  int opt_nt = -1;                    // one-gate sweeps non-temporal: -1 by shard size and bit positions (single_nt), 0 never, 1 always
  int opt_multi_r = 5;                // max distinct targets per k_multi pass (0: never group)
  int opt_kq_variant = -1;            // dense K = 4, 5 gates: 0 four real products per complex one, 1 three (Gauss), 2 three + next batch prefetched, 3 / 4 = 2 / 1 with the A fragments in LDS, 5 two batches per step 32 groups wide (lane swaps), 6 / 7 / 8 the batch staged through LDS with 1 / 2 / 3 batches of loads in flight; -1 (default): by K and placement (launch_kq_mfma)
  int opt_kq_order = 1;               // which target is index bit 0, 1, .. for the matrix-core kernels: 0 as given, 1 a target on address bit 11 first, 2 ascending, 3 descending, 4 nearest bit 11 first
  int opt_kq_debug = 0;               // measurement only (needs QSV_MEASUREMENT_KNOBS): the LDS-staged dense-gate kernel without products (1) / without memory traffic (2) / as a plain copy (3)
  int opt_lowctl_mask = 1;            // one-gate sweeps: controls on address bits 0..2 as a mask over whole lines (k_pair_m)
  int opt_general_combos = 1;         // general k_multi passes: controls on workgroup-uniform bits resolved per workgroup (combo table)
  int opt_fold_init_h = 1;            // qsv_exec: H on a still-|0> qubit right after an init is part of the init write
  int exec_ops_left = 1 << 30;            // qsv_exec: ops of the running program not yet consumed (group_fits lets the tail ride along)
  int opt_kq_blocks_per_cu = 0;       // k_kq_mfma grid: workgroups per CU (0: 64)
  int opt_kq3_tile = 1;               // dense 3-qubit gates on the vector units (k_kq_tile) instead of I (x) U on a 16 x 16 matrix-core tile
  int opt_kq_chunked = 0;             // k_kq_mfma: every wave walks a contiguous run of batches instead of a grid-stride loop (experiment)
  int opt_blocksum_variant = 6;       // access pattern of the read-only passes (qsv_measure.inc): 2 wave-contiguous, 4 no grid-stride loop, 1 index swizzle
  int opt_swz = 1;                    // one-gate kernels: lane bit 5 of a wave access carries address bit 11 (swz_5_11)
  int opt_general_light_r = 5;        // tile a general pass with little arithmetic is padded to (its register TARGETS stay <= general_r)
  int opt_pass_max_ops = 56;          // ops per k_multi pass at most (34-qubit unfused stream: 56 -> 8 passes at 0.70 of peak, 827 ms; 64 -> 7 at 0.65, 783 ms; 96 -> 5 at 0.51, 712 ms)
  int opt_single_shortcut = 1;        // a pass of ONE op runs as its dedicated kernel (0: as a k_multi pass like any other -- an experiment)
  int opt_general_r = 4;              // ... of a GENERAL pass (masked ops / register selects); also the tile it is padded to
  int opt_pair_variant = 0;           // experiments: see run_single
  int opt_lane_targets = 1;           // gates on address bits < 6 ride in k_multi passes as wave shuffles
  int opt_init_prod_bit0 = 0;         // first register bit of the generator: 0 by shard size (default), < 0 the top bits, 6 bits 6.. + lane map
  int opt_init_prod_r = 0;            // amplitudes per thread of the generator, log2: 0 by shard size (default), else 3..6
  int opt_init_prod = 1;              // init followed by diagonals only -> k_init_prod (write-only generator)
  int opt_pass_hints = 1;             // honour QSV_OPF_NEW_PASS (planner-chosen pass boundaries)
  int opt_lane_map_min_l = 26;        // local qubits from which the lane map is applied to tiles other than bits 6..10
  int opt_lane_map = 1;               // access pattern of passes without borrowed lanes: 0 plain, 1 auto, else explicit 5-bit fields
  int opt_dyn_lanes = 3;              // lane bits 3..5 lent per pass to targets anywhere below bit 28 (0..3)
  int opt_cache_sums = 1;             // 0: every norm / sample recomputes the block sums (benchmarks)
  int opt_fused_sums = 1;             // last k_multi pass of a program also leaves the per-tile |amp|^2 sums
  int opt_kq_mfma = 1;                // dense k >= 3 gates on the f64 matrix cores
  int opt_zero_tracking = 0;          // opt-in: skip the part of the shard that is provably still zero
  int opt_multi_nt = -1;              // k_multi with non-temporal loads + stores: -1 shards of >= 2^26 amplitudes, 0 never, 1 always
  int opt_init_prod_nt = -1;          // generator with non-temporal stores: -1 by shard size, 0 never, 1 always
  int opt_xframe = 1;                 // uncontrolled X gates inside a pass become an XOR on its store addresses
  int opt_trace_passes = 0;           // 1: one stderr line per k_multi pass (R, mode, ops by update shape) -- a diagnostic
  int opt_pass_budget = 0;            // opt-in cap on the arithmetic of a general pass, percent of one read+write of the shard (0: none)
  uint64_t opt_xchunk = 1ull << 24;   // amplitudes per exchange chunk (256 MiB)
};

static inline uint64_t amps_local(const qsv_handle* h) { return 1ull << h->L; }

static int shard_set(const Shard& s) {
  HIPCHK(hipSetDevice(s.device));
  return QSV_OK;
}

static unsigned grid_for(const qsv_handle* h, const Shard& s, uint64_t work, uint64_t per_block) {
  uint64_t need = (work + per_block - 1) / per_block;
  uint64_t cap = (uint64_t)s.n_cu * (uint64_t)h->opt_blocks_per_cu;
  if (need < 1) need = 1;
  // HIP rejects a launch whose gridDim.x * blockDim.x reaches 2^32 (hit by a 256 GiB shard:
  // 2^34 amplitudes / 4 per thread); every kernel sized through here has a grid-stride loop
  return (unsigned)std::min<uint64_t>(std::min(need, cap), (1ull << 24) - 1ull);
}

// device copy of a small host table, asynchronous on the shard stream
static int arena_put(Shard& s, const void* src, size_t bytes, void** dptr) {
  const size_t slot = (bytes + 255) & ~size_t(255);
  if (slot > s.arena_bytes) return fail(QSV_E_BADARG, "table of %zu bytes exceeds arena", bytes);
  if (s.arena_top + slot > s.arena_bytes) {
    HIPCHK(hipStreamSynchronize(s.stream));   // every earlier table has been consumed
    s.arena_top = 0;
  }
  memcpy(s.h_arena + s.arena_top, src, bytes);
  HIPCHK(hipMemcpyAsync(s.d_arena + s.arena_top, s.h_arena + s.arena_top, bytes,
                        hipMemcpyHostToDevice, s.stream));
  *dptr = s.d_arena + s.arena_top;
  s.arena_top += slot;
  return QSV_OK;
}

static int get_event(Shard& s, hipEvent_t* e) {
  if (!s.free_events.empty()) { *e = s.free_events.back(); s.free_events.pop_back(); return QSV_OK; }
  HIPCHK(hipEventCreate(e));
  return QSV_OK;
}

static int drain_pending(qsv_handle* h) {
  for (Shard& s : h->shards) {
    if (s.pending.empty()) continue;
    CHK(shard_set(s));
    HIPCHK(hipStreamSynchronize(s.stream));
    for (Pending& p : s.pending) {
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, p.e0, p.e1));
      h->stats.per_kind[p.kind].device_ms += ms;
      s.free_events.push_back(p.e0);
      s.free_events.push_back(p.e1);
    }
    s.pending.clear();
  }
  return QSV_OK;
}

// bracket one kernel launch: stats (+ events when profiling)
template <class F>
static int launch(qsv_handle* h, Shard& s, int kind, double bytes, F&& f) {
  Pending p{kind, nullptr, nullptr};
  if (h->profiling) {
    CHK(get_event(s, &p.e0));
    CHK(get_event(s, &p.e1));
    HIPCHK(hipEventRecord(p.e0, s.stream));
  }
  f();
  HIPCHK(hipGetLastError());
  if (kind != QSV_K_PROB) { s.sums_valid = false; s.tile_valid = false; s.h_tsums_valid = false; }   // any state change drops cached sums
  if (h->profiling) {
    HIPCHK(hipEventRecord(p.e1, s.stream));
    s.pending.push_back(p);
    if (s.pending.size() > 4096) CHK(drain_pending(h));
  }
  h->stats.per_kind[kind].launches += 1;
  h->stats.per_kind[kind].algorithmic_bytes += bytes;
  return QSV_OK;
}

// ------------------------------------------------------------------------------------------
// life cycle
// ------------------------------------------------------------------------------------------
static int shard_alloc(qsv_handle* h, Shard& s) {
  CHK(shard_set(s));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, s.device));
  s.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  HIPCHK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
  const uint64_t n = amps_local(h);
  HIPCHK(hipMalloc(&s.amp, n * sizeof(cplx)));
  s.arena_bytes = 8u << 20;
  HIPCHK(hipHostMalloc(&s.h_arena, s.arena_bytes, hipHostMallocDefault));
  HIPCHK(hipMalloc(&s.d_arena, s.arena_bytes));
  const uint64_t nblk = (n + QSV_SBLOCK - 1) / QSV_SBLOCK;
  HIPCHK(hipMalloc(&s.d_sums, nblk * sizeof(double)));
  return QSV_OK;
}

static int ilog2_exact(int x) {
  if (x <= 0 || (x & (x - 1))) return -1;
  int l = 0;
  while ((1 << l) < x) ++l;
  return l;
}

extern "C" int qsv_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(QSV_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

extern "C" int qsv_device_memory(int device_id, uint64_t* free_bytes, uint64_t* total_bytes) {
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return fail(QSV_E_BADARG, "device id %d not in [0,%d)", device_id, ndev);
  HIPCHK(hipSetDevice(device_id));
  size_t f = 0, t = 0;
  HIPCHK(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return QSV_OK;
}

extern "C" int qsv_device_bus_id(int device_id, char* out, int len) {
  if (!out || len < 16) return fail(QSV_E_BADARG, "bus id buffer too small");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return fail(QSV_E_BADARG, "device id %d not in [0,%d)", device_id, ndev);
  HIPCHK(hipDeviceGetPCIBusId(out, len, device_id));
  return QSV_OK;
}

static int create_common(int n_qubits, int P, qsv_handle** out, qsv_handle** hh) {
  if (!out) return fail(QSV_E_BADARG, "out is NULL");
  const int g = ilog2_exact(P);
  if (g < 0) return fail(QSV_E_BADARG, "number of shards %d is not a power of two", P);
  if (n_qubits < 1 || n_qubits > 40) return fail(QSV_E_BADARG, "n_qubits %d out of range", n_qubits);
  if (n_qubits - g < 1) return fail(QSV_E_BADARG, "%d qubits cannot be split into %d shards", n_qubits, P);
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail(QSV_E_HIP, "no HIP device visible");
  qsv_handle* h = new qsv_handle();
  memset(&h->stats, 0, sizeof h->stats);
  h->W = n_qubits;
  h->gbits = g;
  h->L = n_qubits - g;
  h->P = P;
  *hh = h;
  return QSV_OK;
}

extern "C" int qsv_destroy(qsv_handle* h) {
  if (!h) return QSV_OK;
  for (Shard& s : h->shards) {
    hipSetDevice(s.device);
    if (s.stream) hipStreamSynchronize(s.stream);
    for (Pending& p : s.pending) { hipEventDestroy(p.e0); hipEventDestroy(p.e1); }
    for (hipEvent_t e : s.free_events) hipEventDestroy(e);
    if (s.amp) hipFree(s.amp);
    if (s.d_arena) hipFree(s.d_arena);
    if (s.h_arena) hipHostFree(s.h_arena);
    if (s.d_sums) hipFree(s.d_sums);
    for (int b = 0; b < 2; ++b) if (s.xbuf[b]) hipFree(s.xbuf[b]);
    for (int b = 0; b < 2; ++b) {
      if (s.xsend[b]) hipFree(s.xsend[b]);
      if (s.xrecv[b]) hipFree(s.xrecv[b]);
      if (s.ev_packed[b]) hipEventDestroy(s.ev_packed[b]);
      if (s.ev_sent[b]) hipEventDestroy(s.ev_sent[b]);
      if (s.ev_unpacked[b]) hipEventDestroy(s.ev_unpacked[b]);
    }
    if (s.ev_start) hipEventDestroy(s.ev_start);
    if (s.xstream) { hipStreamSynchronize(s.xstream); hipStreamDestroy(s.xstream); }
    if (s.d_sblk) { hipFree(s.d_sblk); hipFree(s.d_sres); hipFree(s.d_sout); }
    if (s.d_tsums) hipFree(s.d_tsums);
    if (s.d_red) hipFree(s.d_red);
    if (s.ev_copied) hipEventDestroy(s.ev_copied);
    if (s.ev_ready) hipEventDestroy(s.ev_ready);
    if (s.h_tsums) hipHostFree(s.h_tsums);
    if (s.d_super) hipFree(s.d_super);
    if (s.stream) hipStreamDestroy(s.stream);
  }
  if (h->t0) hipEventDestroy(h->t0);
  if (h->t1) hipEventDestroy(h->t1);
  if (h->comm && g_rccl.lib) g_rccl.CommDestroy(h->comm);
  for (size_t r = 0; r < h->peer_amp.size(); ++r)
    if (h->peer_amp[r] && (int)r != h->rank) hipIpcCloseMemHandle(h->peer_amp[r]);
  if (h->ipc_arrive) munmap((void*)h->ipc_arrive, h->ipc_map_bytes);
  delete h;
  return QSV_OK;
}

extern "C" int qsv_create(int n_qubits, int n_devices, const int* device_ids, qsv_handle** out) {
  qsv_handle* h = nullptr;
  CHK(create_common(n_qubits, n_devices, out, &h));
  int ndev = 0;
  hipGetDeviceCount(&ndev);
  h->shards.resize(n_devices);
  for (int i = 0; i < n_devices; ++i) {
    Shard& s = h->shards[i];
    s.device = device_ids ? device_ids[i] : 0;
    s.index = i;
    if (s.device < 0 || s.device >= ndev) {
      qsv_destroy(h);
      return fail(QSV_E_BADARG, "device id %d not in [0,%d)", s.device, ndev);
    }
    int r = shard_alloc(h, s);
    if (r != QSV_OK) { qsv_destroy(h); return r; }
  }
  // peer access between distinct devices of one process (exchange by peer copy)
  for (Shard& a : h->shards)
    for (Shard& b : h->shards)
      if (a.device != b.device) {
        int can = 0;
        hipDeviceCanAccessPeer(&can, a.device, b.device);
        if (can) { hipSetDevice(a.device); hipDeviceEnablePeerAccess(b.device, 0); (void)hipGetLastError(); }
      }
  *out = h;
  return QSV_OK;
}

extern "C" int qsv_create_rank(int n_qubits, int world_size, int rank, int device_id, qsv_handle** out) {
  qsv_handle* h = nullptr;
  CHK(create_common(n_qubits, world_size, out, &h));
  if (rank < 0 || rank >= world_size) { delete h; return fail(QSV_E_BADARG, "rank %d not in [0,%d)", rank, world_size); }
  int ndev = 0;
  hipGetDeviceCount(&ndev);
  if (device_id < 0 || device_id >= ndev) { delete h; return fail(QSV_E_BADARG, "device id %d not in [0,%d)", device_id, ndev); }
  h->multiproc = world_size > 1;
  h->rank = rank;
  h->shards.resize(1);
  h->shards[0].device = device_id;
  h->shards[0].index = rank;
  int r = shard_alloc(h, h->shards[0]);
  if (r != QSV_OK) { qsv_destroy(h); return r; }
  *out = h;
  return QSV_OK;
}

extern "C" int qsv_comm_unique_id(uint8_t id[QSV_UNIQUE_ID_BYTES]) {
  CHK(rccl_load());
  ncclUniqueId uid;
  NCCLCHK(g_rccl.GetUniqueId(&uid));
  static_assert(sizeof(uid) == QSV_UNIQUE_ID_BYTES, "ncclUniqueId size");
  memcpy(id, &uid, QSV_UNIQUE_ID_BYTES);
  return QSV_OK;
}

extern "C" int qsv_comm_init(qsv_handle* h, const uint8_t id[QSV_UNIQUE_ID_BYTES]) {
  if (!h || !id) return fail(QSV_E_BADARG, "NULL argument");
  if (!h->multiproc) return QSV_OK;
  if (h->comm) return QSV_OK;
  CHK(rccl_load());
  CHK(shard_set(h->shards[0]));
  ncclUniqueId uid;
  memcpy(&uid, id, QSV_UNIQUE_ID_BYTES);
  StdoutToStderr quiet;
  NCCLCHK(g_rccl.CommInitRank(&h->comm, h->P, uid, h->rank));
  return QSV_OK;
}

// ------------------------------------------------------------------------------------------
// peer-mapped (IPC) transport
// ------------------------------------------------------------------------------------------
#define QSV_IPC_SLOT 8            // uint64 counters per rank (one 64-byte line each)

extern "C" int qsv_ipc_export(qsv_handle* h, uint8_t out[QSV_IPC_HANDLE_BYTES]) {
  if (!h || !out) return fail(QSV_E_BADARG, "NULL argument");
  static_assert(sizeof(hipIpcMemHandle_t) <= QSV_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
  CHK(shard_set(h->shards[0]));
  hipIpcMemHandle_t hd;
  HIPCHK(hipIpcGetMemHandle(&hd, h->shards[0].amp));
  memset(out, 0, QSV_IPC_HANDLE_BYTES);
  memcpy(out, &hd, sizeof hd);
  return QSV_OK;
}

extern "C" int qsv_ipc_attach(qsv_handle* h, const uint8_t* handles, const char* shm_name, int create) {
  if (!h || !handles || !shm_name) return fail(QSV_E_BADARG, "NULL argument");
  if (!h->multiproc) return QSV_OK;
  if (!h->peer_amp.empty()) return QSV_OK;
  CHK(shard_set(h->shards[0]));
  const size_t bytes = (size_t)h->P * QSV_IPC_SLOT * sizeof(uint64_t);
  const int fd = shm_open(shm_name, create ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
  if (fd < 0) return fail(QSV_E_HIP, "shm_open(%s) failed", shm_name);
  if (create && ftruncate(fd, (off_t)bytes) != 0) { close(fd); return fail(QSV_E_HIP, "ftruncate(%s) failed", shm_name); }
  void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return fail(QSV_E_HIP, "mmap(%s) failed", shm_name);
  h->ipc_arrive = reinterpret_cast<std::atomic<uint64_t>*>(m);     // a fresh segment reads as zeros
  h->ipc_map_bytes = bytes;
  h->peer_amp.assign(h->P, nullptr);
  h->peer_amp[h->rank] = h->shards[0].amp;
  for (int r = 0; r < h->P; ++r) {
    if (r == h->rank) continue;
    hipIpcMemHandle_t hd;
    memcpy(&hd, handles + (size_t)r * QSV_IPC_HANDLE_BYTES, sizeof hd);
    void* p = nullptr;
    HIPCHK(hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess));
    h->peer_amp[r] = reinterpret_cast<cplx*>(p);
  }
  return QSV_OK;
}

// every rank of a group posts once and waits for all its peers: all ranks run the same program, so
// their counters advance in lockstep (each exchange = one post before, one after, on every rank)
static int ipc_group_barrier(qsv_handle* h, const std::vector<int>& peers) {
  const uint64_t seq = ++h->ipc_seq;
  h->ipc_arrive[(size_t)h->rank * QSV_IPC_SLOT].store(seq, std::memory_order_release);
  const auto t0 = std::chrono::steady_clock::now();
  for (int peer : peers)
    while (h->ipc_arrive[(size_t)peer * QSV_IPC_SLOT].load(std::memory_order_acquire) < seq) {
      sched_yield();
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
        return fail(QSV_E_HIP, "rank %d waited 120 s for rank %d at exchange step %llu", h->rank, peer, (unsigned long long)seq);
    }
  return QSV_OK;
}

extern "C" int qsv_rccl_selftest(int device_id, uint64_t n_doubles) {
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device_id < 0 || device_id >= ndev) return fail(QSV_E_BADARG, "device id %d not in [0,%d)", device_id, ndev);
  if (n_doubles < 1 || n_doubles > (1ull << 28)) return fail(QSV_E_BADARG, "n_doubles out of range");
  CHK(rccl_load());
  HIPCHK(hipSetDevice(device_id));
  ncclUniqueId uid;
  NCCLCHK(g_rccl.GetUniqueId(&uid));
  ncclComm_t comm = nullptr;
  StdoutToStderr quiet;
  NCCLCHK(g_rccl.CommInitRank(&comm, 1, uid, 0));
  hipStream_t st = nullptr;
  double *a = nullptr, *b = nullptr;
  int rc = QSV_OK;
  std::vector<double> hin(n_doubles), hout(n_doubles, 0.0);
  for (uint64_t i = 0; i < n_doubles; ++i) hin[i] = 0.5 * (double)i - 3.0;
  auto body = [&]() -> int {
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIPCHK(hipMalloc(&a, n_doubles * sizeof(double)));
    HIPCHK(hipMalloc(&b, n_doubles * sizeof(double)));
    HIPCHK(hipMemcpyAsync(a, hin.data(), n_doubles * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(b, 0, n_doubles * sizeof(double), st));
    NCCLCHK(g_rccl.GroupStart());
    NCCLCHK(g_rccl.Send(a, n_doubles, ncclDouble, 0, comm, st));
    NCCLCHK(g_rccl.Recv(b, n_doubles, ncclDouble, 0, comm, st));
    NCCLCHK(g_rccl.GroupEnd());
    HIPCHK(hipMemcpyAsync(hout.data(), b, n_doubles * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (uint64_t i = 0; i < n_doubles; ++i)
      if (hout[i] != hin[i]) return fail(QSV_E_RCCL, "rccl self-test: element %llu came back as %g, sent %g", (unsigned long long)i, hout[i], hin[i]);
    return QSV_OK;
  };
  rc = body();
  if (a) hipFree(a);
  if (b) hipFree(b);
  if (st) hipStreamDestroy(st);
  g_rccl.CommDestroy(comm);
  return rc;
}

extern "C" int qsv_poison_lds(qsv_handle* h) {
  if (!h) return fail(QSV_E_BADARG, "NULL handle");
  for (Shard& s : h->shards) {
    CHK(shard_set(s));
    unsigned long long* sink = nullptr;
    HIPCHK(hipMalloc(&sink, sizeof *sink));
    // 52 KiB per workgroup: exactly three fit a compute unit's 160 KiB, and 3 x CUs workgroups that all stay resident
    // (the spin) land three on every unit
    hipLaunchKernelGGL(k_poison_lds, dim3((unsigned)s.n_cu * 3), dim3(QSV_TPB), 52 * 1024, s.stream, sink, 52 * 1024 / 8, 4000);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s.stream));
    HIPCHK(hipFree(sink));
  }
  return QSV_OK;
}

extern "C" int qsv_sync(qsv_handle* h) {
  if (!h) return fail(QSV_E_BADARG, "NULL handle");
  for (Shard& s : h->shards) {
    CHK(shard_set(s));
    HIPCHK(hipStreamSynchronize(s.stream));
  }
  return QSV_OK;
}

// ------------------------------------------------------------------------------------------
// shard resolution helpers
// ------------------------------------------------------------------------------------------
static int check_qubit(const qsv_handle* h, int q, const char* what) {
  if (q < 0 || q >= h->W) return fail(QSV_E_BADARG, "%s qubit %d not in [0,%d)", what, q, h->W);
  return QSV_OK;
}
static inline int shard_bit(const qsv_handle* h, const Shard& s, int q) {   // q >= L
  return (s.index >> (q - h->L)) & 1;
}

// controls -> (skip shard?, local controls)
struct LocalCtrl { bool skip = false; std::vector<int> q, v; };
static LocalCtrl resolve_ctrl(const qsv_handle* h, const Shard& s, int n, const int* ctrls, const int* vals) {
  LocalCtrl lc;
  for (int i = 0; i < n; ++i) {
    const int v = vals ? (vals[i] ? 1 : 0) : 1;
    if (ctrls[i] >= h->L) { if (shard_bit(h, s, ctrls[i]) != v) lc.skip = true; }
    else { lc.q.push_back(ctrls[i]); lc.v.push_back(v); }
  }
  return lc;
}

static int check_distinct(const qsv_handle* h, int n, const int* q, int extra) {
  uint64_t seen = 0;
  if (extra >= 0) seen |= 1ull << extra;
  for (int i = 0; i < n; ++i) {
    CHK(check_qubit(h, q[i], "gate"));
    if (seen & (1ull << q[i])) return fail(QSV_E_BADARG, "duplicate qubit %d in gate", q[i]);
    seen |= 1ull << q[i];
  }
  return QSV_OK;
}

static BitIns make_ins(std::vector<int> pos) {
  std::sort(pos.begin(), pos.end());
  BitIns b;
  b.n = (int)pos.size();
  for (int i = 0; i < b.n; ++i) b.pos[i] = pos[i];
  return b;
}

// ------------------------------------------------------------------------------------------
// state preparation
// ------------------------------------------------------------------------------------------
extern "C" int qsv_init_uniform(qsv_handle* h, uint64_t qubit_mask) {
  if (!h) return fail(QSV_E_BADARG, "NULL handle");
  if (h->W < 64 && (qubit_mask >> h->W)) return fail(QSV_E_BADARG, "mask has bits beyond qubit %d", h->W - 1);
  const int pc = __builtin_popcountll(qubit_mask);
  const double val = std::pow(2.0, -0.5 * pc);
  const uint64_t n = amps_local(h);
  const uint64_t lmask = n - 1;
  for (Shard& s : h->shards) {
    CHK(shard_set(s));
    // shard bits outside the mask must be 0 for the shard to hold any weight
    const uint64_t hi = (uint64_t)s.index << h->L;
    const double v = (hi & ~qubit_mask) ? 0.0 : val;
    const uint64_t nonmask = ~qubit_mask & lmask;
    s.zmask = 0;
    CHK(launch(h, s, QSV_K_INIT, 16.0 * (double)n, [&] {
      hipLaunchKernelGGL(k_init, dim3(grid_for(h, s, n, QSV_TPB * 4)), dim3(QSV_TPB), 0, s.stream,
                         s.amp, n, nonmask, v);
    }));
  }
  return QSV_OK;
}
extern "C" int qsv_init_zero(qsv_handle* h) { return qsv_init_uniform(h, 0ull); }

// The rest of the host side, in dependency order (one translation unit):
#include "qsv_gates.inc"     // LocalOp, validation, per-shard resolution, single-gate launchers, qsv_apply_*
#include "qsv_multi.inc"     // PendingGroup, k_multi pass construction (round schedule, modes), zero tracking
#include "qsv_layout.inc"    // qsv_apply_kq (VALU + MFMA), qsv_swap_layout: local swap and shard-bit exchange
#include "qsv_measure.inc"   // qsv_norm / qsv_sample / qsv_probabilities / amplitude copies
#include "qsv_exec.inc"      // qsv_exec, stats, timers, options
