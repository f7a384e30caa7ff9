// qsv.hip -- host side of libqsv.so: handles, shard resolution, launches, sampling, exchange.
// C ABI declared in include/qsv.h (which cites the reference call each entry point replaces).
// gfx950 only; no CPU fallback of any kind: without a HIP device every entry point fails.
#include "qsv_kernels.h"
#include "../../include/qsv.h"

#include <rccl/rccl.h>
#include <dlfcn.h>

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
  int n_cu = 256;
  uint64_t zmask = 0;            // zero tracking: local bits known |0>; memory with such a bit set is unwritten
  std::vector<double> h_sums;    // host copy of the block sums, valid until the state changes
  bool sums_valid = false;
  // sums left behind by the last k_multi pass of a program, one per workgroup tile (no read pass)
  double* d_tsums = nullptr;
  size_t tsums_cap = 0;
  bool tile_valid = false, tile_fresh = false;
  int tile_R = 0;
  RegPos tile_rp;
  BitIns tile_ins;
  uint64_t tile_nblocks = 0;
  uint64_t* d_sblk = nullptr;    // sampling scratch (block index, residual, result per shot)
  double* d_sres = nullptr;
  uint64_t* d_sout = nullptr;
  size_t sample_cap = 0;
  std::vector<Pending> pending;
  std::vector<hipEvent_t> free_events;
};

struct qsv_handle {
  int W = 0, L = 0, gbits = 0;   // global qubits, local qubits per shard, shard bits
  int P = 1;                     // total shards
  bool multiproc = false;
  int rank = 0;
  ncclComm_t comm = nullptr;
  std::vector<Shard> shards;     // shards owned by this process
  qsv_stats stats;
  bool profiling = false;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  // options
  int opt_blocks_per_cu = 1 << 16;  // measured on MI355X: one tile per workgroup (no grid-stride) streams fastest
  int opt_unroll = 4;
  int opt_lowt_shuffle = 1;
  int opt_nt = 0;
  int opt_multi_r = 5;                // max distinct targets per k_multi pass (0: never group)
  int opt_pair_variant = 0;           // experiments: see run_single
  int opt_lane_targets = 1;           // gates on address bits < 6 ride in k_multi passes as wave shuffles
  int opt_fused_sums = 1;             // last k_multi pass of a program also leaves the per-tile |amp|^2 sums
  int opt_kq_mfma = 1;                // dense k >= 3 gates on the f64 matrix cores
  int opt_zero_tracking = 0;          // opt-in: skip the part of the shard that is provably still zero
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
  return (unsigned)std::min<uint64_t>(std::min(need, cap), 0x7fffffffull);
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
  if (kind != QSV_K_PROB) { s.sums_valid = false; s.tile_valid = false; }   // any state change drops cached sums
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
    if (s.d_sblk) { hipFree(s.d_sblk); hipFree(s.d_sres); hipFree(s.d_sout); }
    if (s.d_tsums) hipFree(s.d_tsums);
    if (s.stream) hipStreamDestroy(s.stream);
  }
  if (h->t0) hipEventDestroy(h->t0);
  if (h->t1) hipEventDestroy(h->t1);
  if (h->comm && g_rccl.lib) g_rccl.CommDestroy(h->comm);
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
  NCCLCHK(g_rccl.CommInitRank(&h->comm, h->P, uid, h->rank));
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

// ------------------------------------------------------------------------------------------
// gates: every gate is first RESOLVED per shard (shard-bit controls evaluated, tables sliced)
// into a LocalOp on local address bits, then either launched alone (dedicated kernel) or
// grouped with its neighbours into one register-tiled k_multi pass (qsv_exec).
// ------------------------------------------------------------------------------------------
struct LocalOp {
  int type = 0;               // 0 mux table | 1 diag table | 2 controlled 2x2 | 3 controlled phase
  bool is_x = false;          // type 2 that is a plain X: swap kernel when launched alone
  int target = -1;            // local address bit (types 0, 2)
  std::vector<int> list;      // types 0,1: gather list; table index bit e <- address bit list[e]
  std::vector<double> table;  // type 0: 8 doubles per entry; type 1: 2 doubles per entry
  std::vector<int> cq, cv;    // types 2,3: local controls and their values
  double m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// slice a 2^k table (entry = `ent` doubles) over `qubits` down to the local qubits of shard s
static void slice_table(const qsv_handle* h, const Shard& s, int k, const int* qubits, const double* table,
                        int ent, std::vector<int>& lq, std::vector<double>& out) {
  std::vector<int> lpos;
  uint32_t gfix = 0;
  lq.clear();
  for (int b = 0; b < k; ++b) {
    if (qubits[b] >= h->L) { if (shard_bit(h, s, qubits[b])) gfix |= 1u << b; }
    else { lpos.push_back(b); lq.push_back(qubits[b]); }
  }
  const int kl = (int)lpos.size();
  out.resize((size_t)ent << kl);
  for (uint32_t jl = 0; jl < (1u << kl); ++jl) {
    uint32_t j = gfix;
    for (int b = 0; b < kl; ++b) if ((jl >> b) & 1u) j |= 1u << lpos[b];
    memcpy(&out[(size_t)jl * ent], &table[(size_t)j * ent], sizeof(double) * ent);
  }
}

// argument validation shared by the one-gate entry points and qsv_exec
static int validate_gate(const qsv_handle* h, int kind, int n, const int* q, int target, const void* data) {
  switch (kind) {
    case QSV_OP_1Q: case QSV_OP_MCX:
      if (n < 0 || n > QSV_MAX_CTRL || (n && !q)) return fail(QSV_E_BADARG, "n_ctrl %d out of range", n);
      CHK(check_qubit(h, target, "target"));
      CHK(check_distinct(h, n, q, target));
      if (kind == QSV_OP_1Q && !data) return fail(QSV_E_BADARG, "matrix is NULL");
      break;
    case QSV_OP_MUX:
      if (n < 0 || n > 10 || (n && !q) || !data) return fail(QSV_E_BADARG, "mux needs 0..10 controls and matrices");
      CHK(check_qubit(h, target, "target"));
      CHK(check_distinct(h, n, q, target));
      break;
    case QSV_OP_DIAG:
      if (n < 1 || n > QSV_MAX_CTRL || !q || !data) return fail(QSV_E_BADARG, "diag needs 1..%d qubits and a table", QSV_MAX_CTRL);
      CHK(check_distinct(h, n, q, -1));
      return QSV_OK;
    case QSV_OP_MCPHASE:
      if (n < 1 || n > QSV_MAX_CTRL || !q) return fail(QSV_E_BADARG, "mcphase needs 1..%d qubits", QSV_MAX_CTRL);
      CHK(check_distinct(h, n, q, -1));
      return QSV_OK;
    default:
      return fail(QSV_E_BADARG, "not a gate kind: %d", kind);
  }
  if (target >= h->L)
    return fail(QSV_E_UNSUPPORTED, "target qubit %d is a shard bit (local qubits: %d); qsv_swap_layout it first", target, h->L);
  return QSV_OK;
}

// false: this shard is untouched by the gate (a shard-bit control does not match)
static bool resolve_gate(const qsv_handle* h, const Shard& s, int kind, int n, const int* q, const int* vals,
                         int target, const double* data, double angle, LocalOp& lo) {
  lo = LocalOp();
  if (kind == QSV_OP_1Q || kind == QSV_OP_MCX || kind == QSV_OP_MCPHASE) {
    LocalCtrl lc = resolve_ctrl(h, s, n, q, vals);
    if (lc.skip) return false;
    lo.cq = lc.q;
    lo.cv = lc.v;
    if (kind == QSV_OP_MCPHASE) {
      lo.type = 3;
      lo.m[0] = std::cos(angle);
      lo.m[1] = std::sin(angle);
    } else {
      lo.type = 2;
      lo.target = target;
      lo.is_x = kind == QSV_OP_MCX;
      if (lo.is_x) { lo.m[2] = 1.0; lo.m[4] = 1.0; }
      else memcpy(lo.m, data, sizeof lo.m);
    }
    return true;
  }
  if (kind == QSV_OP_MUX) {
    lo.type = 0;
    lo.target = target;
    slice_table(h, s, n, q, data, 8, lo.list, lo.table);
  } else {
    lo.type = 1;
    slice_table(h, s, n, q, data, 2, lo.list, lo.table);
  }
  return true;
}

template <int KIND, bool NT>
static void launch_pair(const qsv_handle* h, const Shard& s, uint64_t npairs, const BitIns& ins,
                        uint64_t fixed, uint64_t tbit, const Mat2& m) {
  const int U = h->opt_unroll;
  if (U >= 4 && npairs % (QSV_TPB * 4) == 0)
    hipLaunchKernelGGL((k_pair<KIND, 4, false, NT>), dim3(grid_for(h, s, npairs, QSV_TPB * 4)), dim3(QSV_TPB), 0,
                       s.stream, s.amp, npairs, ins, fixed, tbit, m);
  else if (U >= 2 && npairs % (QSV_TPB * 2) == 0)
    hipLaunchKernelGGL((k_pair<KIND, 2, false, NT>), dim3(grid_for(h, s, npairs, QSV_TPB * 2)), dim3(QSV_TPB), 0,
                       s.stream, s.amp, npairs, ins, fixed, tbit, m);
  else
    hipLaunchKernelGGL((k_pair<KIND, 1, true, NT>), dim3(grid_for(h, s, npairs, QSV_TPB)), dim3(QSV_TPB), 0,
                       s.stream, s.amp, npairs, ins, fixed, tbit, m);
}

// one resolved gate, one dedicated kernel
static int run_single(qsv_handle* h, Shard& s, const LocalOp& lo) {
  CHK(shard_set(s));
  const uint64_t n = amps_local(h);
  if (lo.type == 2) {
    const int nc = (int)lo.cq.size(), t = lo.target;
    const uint64_t npairs = n >> (1 + nc);
    const double bytes = 32.0 * (double)(n >> nc);
    const int kind = lo.is_x ? QSV_K_X : QSV_K_1Q;
    Mat2 mm;
    memcpy(mm.v, lo.m, sizeof mm.v);
    if (!lo.is_x && nc == 0 && t < 6 && h->opt_lowt_shuffle && n % (QSV_TPB * 4) == 0) {
      return launch(h, s, kind, bytes, [&] {
        if (h->opt_nt)
          hipLaunchKernelGGL((k_lowt<4, true>), dim3(grid_for(h, s, n, QSV_TPB * 4)), dim3(QSV_TPB), 0, s.stream, s.amp, n, t, mm);
        else
          hipLaunchKernelGGL((k_lowt<4, false>), dim3(grid_for(h, s, n, QSV_TPB * 4)), dim3(QSV_TPB), 0, s.stream, s.amp, n, t, mm);
      });
    }
    std::vector<int> pos = lo.cq;
    pos.push_back(t);
    const BitIns ins = make_ins(pos);
    uint64_t fixed = 0;
    for (int i = 0; i < nc; ++i) if (lo.cv[i]) fixed |= 1ull << lo.cq[i];
    const uint64_t tbit = 1ull << t;
    if (h->opt_pair_variant && !lo.is_x && npairs % (QSV_TPB * 8 * 8) == 0) {
      const int v = h->opt_pair_variant;
      return launch(h, s, kind, bytes, [&] {
#define PX(U, NL, NS, RM) hipLaunchKernelGGL((k_pair_x<U, NL, NS, RM>), dim3((unsigned)(npairs / (QSV_TPB * U))), dim3(QSV_TPB), 0, s.stream, s.amp, npairs, ins, fixed, tbit, mm)
        switch (v) {
          case 1: PX(4, false, false, false); break;   // baseline shape without the grid-stride loop
          case 2: PX(4, false, true, false); break;    // nt stores only
          case 3: PX(4, true, false, false); break;    // nt loads only
          case 4: PX(4, false, false, true); break;    // XCD remap
          case 5: PX(8, false, false, false); break;   // deeper unroll
          case 6: PX(8, false, false, true); break;
          case 7: PX(2, false, false, false); break;
          case 8: PX(4, false, true, true); break;
          default: PX(8, false, true, false); break;
        }
#undef PX
      });
    }
    return launch(h, s, kind, bytes, [&] {
      if (lo.is_x) { if (h->opt_nt) launch_pair<1, true>(h, s, npairs, ins, fixed, tbit, mm); else launch_pair<1, false>(h, s, npairs, ins, fixed, tbit, mm); }
      else         { if (h->opt_nt) launch_pair<0, true>(h, s, npairs, ins, fixed, tbit, mm); else launch_pair<0, false>(h, s, npairs, ins, fixed, tbit, mm); }
    });
  }
  if (lo.type == 3) {
    const int nc = (int)lo.cq.size();
    const uint64_t nsub = n >> nc;
    const BitIns ins = make_ins(lo.cq);
    uint64_t fixed = 0;
    for (int i = 0; i < nc; ++i) if (lo.cv[i]) fixed |= 1ull << lo.cq[i];
    const cplx ph = make_double2(lo.m[0], lo.m[1]);
    return launch(h, s, QSV_K_MCPHASE, 32.0 * (double)nsub, [&] {
      if (nsub % (QSV_TPB * 4) == 0)
        hipLaunchKernelGGL((k_mcphase<4, false>), dim3(grid_for(h, s, nsub, QSV_TPB * 4)), dim3(QSV_TPB), 0, s.stream, s.amp, nsub, ins, fixed, ph);
      else
        hipLaunchKernelGGL((k_mcphase<1, true>), dim3(grid_for(h, s, nsub, QSV_TPB)), dim3(QSV_TPB), 0, s.stream, s.amp, nsub, ins, fixed, ph);
    });
  }
  const int kl = (int)lo.list.size();
  void* dtab = nullptr;
  CHK(arena_put(s, lo.table.data(), lo.table.size() * sizeof(double), &dtab));
  BitList bl;
  bl.n = kl;
  for (int b = 0; b < kl; ++b) bl.pos[b] = lo.list[b];
  if (lo.type == 1) {
    const int ntab = 1 << kl;
    const bool lds = kl <= 11;
    const size_t shm = lds ? (size_t)ntab * sizeof(cplx) : 0;
    return launch(h, s, QSV_K_DIAG, 32.0 * (double)n, [&] {
      const cplx* tp = reinterpret_cast<const cplx*>(dtab);
      if (n % (QSV_TPB * 4) == 0) {
        const dim3 g(grid_for(h, s, n, QSV_TPB * 4));
        if (lds) {
          if (h->opt_nt) hipLaunchKernelGGL((k_diag<4, false, true, true>), g, dim3(QSV_TPB), shm, s.stream, s.amp, n, bl, tp, ntab);
          else           hipLaunchKernelGGL((k_diag<4, false, true, false>), g, dim3(QSV_TPB), shm, s.stream, s.amp, n, bl, tp, ntab);
        } else           hipLaunchKernelGGL((k_diag<4, false, false, false>), g, dim3(QSV_TPB), 0, s.stream, s.amp, n, bl, tp, ntab);
      } else {
        const dim3 g(grid_for(h, s, n, QSV_TPB));
        if (lds) hipLaunchKernelGGL((k_diag<1, true, true, false>), g, dim3(QSV_TPB), shm, s.stream, s.amp, n, bl, tp, ntab);
        else     hipLaunchKernelGGL((k_diag<1, true, false, false>), g, dim3(QSV_TPB), 0, s.stream, s.amp, n, bl, tp, ntab);
      }
    });
  }
  // type 0: uniformly controlled 2x2
  const int nmat = 1 << kl, t = lo.target;
  const uint64_t npairs = n >> 1;
  const size_t shm = (size_t)nmat * 64;
  return launch(h, s, QSV_K_MUX, 32.0 * (double)n, [&] {
    const double* mp = reinterpret_cast<const double*>(dtab);
    if (npairs % (QSV_TPB * 4) == 0 && h->opt_unroll >= 4) {
      const dim3 g(grid_for(h, s, npairs, QSV_TPB * 4));
      if (h->opt_nt) hipLaunchKernelGGL((k_mux<4, false, true>), g, dim3(QSV_TPB), shm, s.stream, s.amp, npairs, t, bl, mp, nmat);
      else           hipLaunchKernelGGL((k_mux<4, false, false>), g, dim3(QSV_TPB), shm, s.stream, s.amp, npairs, t, bl, mp, nmat);
    } else if (npairs % (QSV_TPB * 2) == 0) {
      hipLaunchKernelGGL((k_mux<2, false, false>), dim3(grid_for(h, s, npairs, QSV_TPB * 2)), dim3(QSV_TPB), shm, s.stream, s.amp, npairs, t, bl, mp, nmat);
    } else {
      hipLaunchKernelGGL((k_mux<1, true, false>), dim3(grid_for(h, s, npairs, QSV_TPB)), dim3(QSV_TPB), shm, s.stream, s.amp, npairs, t, bl, mp, nmat);
    }
  });
}

static int apply_gate(qsv_handle* h, int kind, int n, const int* q, const int* vals, int target,
                      const double* data, double angle) {
  if (!h) return fail(QSV_E_BADARG, "NULL handle");
  CHK(validate_gate(h, kind, n, q, target, kind == QSV_OP_MCX || kind == QSV_OP_MCPHASE ? (const void*)h : (const void*)data));
  LocalOp lo;
  for (Shard& s : h->shards)
    if (resolve_gate(h, s, kind, n, q, vals, target, data, angle, lo)) CHK(run_single(h, s, lo));
  return QSV_OK;
}

extern "C" int qsv_apply_1q(qsv_handle* h, int t, const double m[8]) { return apply_gate(h, QSV_OP_1Q, 0, nullptr, nullptr, t, m, 0); }
extern "C" int qsv_apply_mc1q(qsv_handle* h, int n_ctrl, const int* ctrls, const int* ctrl_vals, int t, const double m[8]) {
  return apply_gate(h, QSV_OP_1Q, n_ctrl, ctrls, ctrl_vals, t, m, 0);
}
extern "C" int qsv_apply_mcx(qsv_handle* h, int n_ctrl, const int* ctrls, const int* ctrl_vals, int t) {
  return apply_gate(h, QSV_OP_MCX, n_ctrl, ctrls, ctrl_vals, t, nullptr, 0);
}
extern "C" int qsv_apply_mcphase(qsv_handle* h, int n_ctrl, const int* ctrls, const int* vals, double angle) {
  return apply_gate(h, QSV_OP_MCPHASE, n_ctrl, ctrls, vals, -1, nullptr, angle);
}
extern "C" int qsv_apply_diag(qsv_handle* h, int k, const int* qubits, const double* table) {
  return apply_gate(h, QSV_OP_DIAG, k, qubits, nullptr, -1, table, 0);
}
extern "C" int qsv_apply_mux_1q(qsv_handle* h, int k, const int* ctrls, int t, const double* mats) {
  return apply_gate(h, QSV_OP_MUX, k, ctrls, nullptr, t, mats, 0);
}

// ------------------------------------------------------------------------------------------
// k_multi pass: a run of resolved gates on <= opt_multi_r distinct targets, one HBM sweep
// ------------------------------------------------------------------------------------------
struct PendingGroup {
  std::vector<LocalOp> ops;
  std::vector<int> targets;       // distinct target bits, in first-use order
  std::vector<int> selects;       // every table-select bit used by a table op of the group
  std::vector<int> lane_targets;  // targets on address bits < 6, handled by wave shuffles (no register bit)
  bool lane_mode = false;         // decided when the group opens: lane bits are really the lane id
  bool opened = false;
  bool simple = true;             // only table ops, and no select bit is a target of the group
  size_t table_cplx = 0;
  bool init = false;              // an init write is waiting to be merged into the pass
  uint64_t nonmask = 0;
  double initval = 0;
};

static bool groupable(const qsv_handle* h, const LocalOp& lo) {
  if (h->opt_multi_r < 1) return false;
  if ((lo.type == 0 || lo.type == 1) && (int)lo.list.size() > QSV_MULTI_MAXLIST) return false;
  return true;
}
static size_t table_cplx_of(const LocalOp& lo) {
  if (lo.type == 0) return (size_t)4 << lo.list.size();
  if (lo.type == 1) return (size_t)1 << lo.list.size();
  return 0;
}
static bool has_bit(const std::vector<int>& v, int q) { return std::find(v.begin(), v.end(), q) != v.end(); }

// does this op's target ride on a lane bit (shuffle) in group g?
static bool is_lane_target(const PendingGroup& g, const LocalOp& lo) {
  return g.lane_mode && lo.target >= 0 && lo.target < 6 && !has_bit(g.targets, lo.target);
}

// would the group still be "simple" (table ops only, selects disjoint from register targets)?
static bool stays_simple(const PendingGroup& g, const LocalOp& lo) {
  if (!g.simple || lo.type > 1) return false;
  for (int q : lo.list) if (has_bit(g.targets, q) || q == lo.target) return false;
  if (lo.target >= 0 && !is_lane_target(g, lo) && has_bit(g.selects, lo.target)) return false;
  return true;
}
static bool group_fits(const qsv_handle* h, const PendingGroup& g, const LocalOp& lo) {
  size_t nt = g.targets.size();
  if (is_lane_target(g, lo)) {
    if (!has_bit(g.lane_targets, lo.target) && g.lane_targets.size() >= 6) return false;
  } else if (lo.target >= 0 && !has_bit(g.targets, lo.target)) {
    if (has_bit(g.lane_targets, lo.target)) return false;     // cannot be both in one pass
    ++nt;
  }
  // the general kernel (controls / selects on register bits, masked 2x2) is built for R <= 4
  const int rmax = stays_simple(g, lo) ? h->opt_multi_r : std::min(h->opt_multi_r, 4);
  if ((int)nt > rmax) return false;
  if (g.table_cplx + table_cplx_of(lo) > 2560 - 4) return false;  // 40 KiB of LDS tables
  return g.ops.size() < 64;
}
static void group_add(const qsv_handle* h, const Shard& s, PendingGroup& g, LocalOp&& lo) {
  if (!g.opened) {
    // lane bits are the lane id only if nothing is inserted below bit 6: no known-zero bit there
    // (zero tracking) and a shard wide enough for 64-lane rows
    g.opened = true;
    g.lane_mode = h->opt_lane_targets && h->L >= 12 && s.zmask == 0;   // full wavefronts, lane id = address bits 0..5
  }
  g.simple = stays_simple(g, lo);
  if (is_lane_target(g, lo)) { if (!has_bit(g.lane_targets, lo.target)) g.lane_targets.push_back(lo.target); }
  else if (lo.target >= 0 && !has_bit(g.targets, lo.target)) g.targets.push_back(lo.target);
  if (lo.type <= 1) for (int q : lo.list) if (!has_bit(g.selects, q)) g.selects.push_back(q);
  g.table_cplx += table_cplx_of(lo);
  g.ops.push_back(std::move(lo));
}

template <int R>
static void launch_multi(const qsv_handle* h, const Shard& s, bool init, int mode, uint64_t nthreads, const BitIns& ins,
                         const RegPos& rp, const MultiOp* dops, const MultiSlot* dslots, int nrounds,
                         const cplx* dtab, int ntab, uint64_t nonmask, double initval, unsigned zreg, double* tsums) {
  const dim3 g((unsigned)((nthreads + QSV_TPB - 1) / QSV_TPB));
  const size_t shm = (size_t)std::max(ntab, 1) * sizeof(cplx);
#define QSV_LM(I, S) hipLaunchKernelGGL((k_multi<R, I, S>), g, dim3(QSV_TPB), shm, s.stream, s.amp, nthreads, ins, rp, dops, dslots, nrounds, dtab, ntab, nonmask, initval, zreg, tsums)
  if (init) { if (mode == 2) QSV_LM(true, 2); else if (mode == 1) QSV_LM(true, 1); else QSV_LM(true, 0); }
  else      { if (mode == 2) QSV_LM(false, 2); else if (mode == 1) QSV_LM(false, 1); else QSV_LM(false, 0); }
#undef QSV_LM
}

// zero tracking: write the zeros that were only implied so far
static int materialize(qsv_handle* h, Shard& s) {
  if (!s.zmask) return QSV_OK;
  const uint64_t n = amps_local(h);
  const uint64_t zm = s.zmask;
  s.zmask = 0;
  CHK(shard_set(s));
  const int nz = __builtin_popcountll(zm);
  return launch(h, s, QSV_K_INIT, 16.0 * ((double)n - (double)(n >> nz)), [&] {
    hipLaunchKernelGGL(k_fill_zero, dim3(grid_for(h, s, n, QSV_TPB * 4)), dim3(QSV_TPB), 0, s.stream, s.amp, n, zm);
  });
}

static int flush_group(qsv_handle* h, Shard& s, PendingGroup& g, bool final_pass = false) {
  const uint64_t n = amps_local(h);
  CHK(shard_set(s));
  if (g.ops.empty()) {
    if (g.init) {
      s.zmask = 0;
      CHK(launch(h, s, QSV_K_INIT, 16.0 * (double)n, [&] {
        hipLaunchKernelGGL(k_init, dim3(grid_for(h, s, n, QSV_TPB * 4)), dim3(QSV_TPB), 0, s.stream, s.amp, n, g.nonmask, g.initval);
      }));
    }
    g = PendingGroup();
    return QSV_OK;
  }
  if (g.ops.size() == 1 && !g.init) {
    CHK(materialize(h, s));
    const int r = run_single(h, s, g.ops[0]);
    g = PendingGroup();
    return r;
  }
  // register bits: the targets, padded with free bits >= 6 so every lane keeps >= 8 loads in flight
  std::vector<int> reg = g.targets;
  const int want = std::min(h->L, std::max((int)reg.size(), std::min(3, h->opt_multi_r)));
  for (int pass = 0; pass < 2; ++pass) {             // first bits >= 6 that are not table selects, then any
    for (int b = std::min(6, h->L - 1); (int)reg.size() < want && b < h->L; ++b)
      if (!has_bit(reg, b) && (pass == 1 || (!has_bit(g.selects, b) && !((s.zmask >> b) & 1ull)))) reg.push_back(b);
  }
  if (g.lane_targets.empty())
    for (int b = 0; (int)reg.size() < want && b < h->L; ++b)
      if (!has_bit(reg, b)) reg.push_back(b);
  const int R = (int)reg.size();
  RegPos rp;
  memset(&rp, 0, sizeof rp);
  for (int c = 0; c < R; ++c) rp.pos[c] = reg[c];
  // zero tracking: register bits still known |0> are not read; known-zero bits outside the tile
  // stay zero, so only the subspace where they are 0 is enumerated at all
  uint64_t regmask = 0;
  for (int q : reg) regmask |= 1ull << q;
  const uint64_t zin = s.zmask, zout = zin & ~regmask;
  unsigned zreg = 0;
  for (int c = 0; c < R; ++c) if ((zin >> reg[c]) & 1ull) zreg |= 1u << c;
  std::vector<int> inspos = reg;
  for (int b = 0; b < h->L; ++b) if ((zout >> b) & 1ull) inspos.push_back(b);
  const int nzout = __builtin_popcountll(zout);
  s.zmask = zout;
  const BitIns ins = make_ins(inspos);
  auto reg_index = [&](int q) -> int {
    for (int c = 0; c < R; ++c) if (reg[c] == q) return c;
    return -1;
  };
  std::vector<MultiOp> mops(g.ops.size());
  std::vector<double> tables;
  for (size_t i = 0; i < g.ops.size(); ++i) {
    const LocalOp& lo = g.ops[i];
    MultiOp& mo = mops[i];
    memset(&mo, 0, sizeof mo);
    mo.type = lo.type;
    mo.bit = lo.target >= 0 ? reg_index(lo.target) : 0;
    if (lo.target >= 0 && mo.bit < 0) {                  // a lane target: wave-shuffle form
      mo.type = lo.type == 0 ? 4 : 5;
      mo.bit = lo.target;
    }
    mo.uniform = 1;
    if (lo.type == 0 || lo.type == 1) {
      mo.nlist = (int)lo.list.size();
      mo.tab = (int)(tables.size() / 2);
      for (int e = 0; e < mo.nlist; ++e) {
        const int c = reg_index(lo.list[e]);
        if (c >= 0) { mo.pos[e] = -1; mo.regw[c] = (lo.type == 0 ? 1 : 1) << e; mo.uniform = 0; }
        else mo.pos[e] = lo.list[e];
      }
      tables.insert(tables.end(), lo.table.begin(), lo.table.end());
    } else {
      for (size_t k = 0; k < lo.cq.size(); ++k) {
        const int c = reg_index(lo.cq[k]);
        if (c >= 0) { mo.rmask |= 1u << c; if (lo.cv[k]) mo.rval |= 1u << c; }
        else { mo.tmask |= 1ull << lo.cq[k]; if (lo.cv[k]) mo.tval |= 1ull << lo.cq[k]; }
      }
      memcpy(mo.m, lo.m, sizeof mo.m);
      if (lo.type == 2 && lo.is_x) mo.nlist = 1;           // register swap instead of arithmetic
    }
  }
  bool simple = true;
  for (const MultiOp& mo : mops) if ((mo.type != 0 && mo.type != 1 && mo.type != 4) || !mo.uniform) simple = false;
  // rounds of 1 + R slots: slot 0 = list of ops without a register target (diag / phase / lane-bit
  // gates), slot 1 + b = optional 2x2 gate on register bit b; a round runs its list first, then the
  // bits in order.  An op joins the current round if program order allows it or if it commutes
  // with what it would overtake (neither op's target lies in the other's support).
  const int NS = R + 1;
  std::vector<MultiSlot> slots;
  std::vector<MultiOp> sorted;
  sorted.reserve(mops.size() + NS);
  {
    auto support_mask = [&](size_t i) -> uint64_t {          // every address bit op i reads or writes
      const LocalOp& lo = g.ops[i];
      uint64_t m = 0;
      if (lo.target >= 0) m |= 1ull << lo.target;
      for (int q : lo.list) m |= 1ull << q;
      for (int q : lo.cq) m |= 1ull << q;
      return m;
    };
    auto target_mask = [&](size_t i) -> uint64_t { return g.ops[i].target >= 0 ? 1ull << g.ops[i].target : 0ull; };
    auto commute = [&](size_t i, size_t j) -> bool {
      return !(target_mask(i) & support_mask(j)) && !(target_mask(j) & support_mask(i));
    };
    struct Round { std::vector<int> list; std::vector<int> gate; std::vector<int> members; };
    std::vector<Round> rounds(1);
    rounds[0].gate.assign(std::max(R, 1), -1);
    int bp = 0;                                            // gates of the current round occupy bits < bp
    for (size_t i = 0; i < mops.size(); ++i) {
      const MultiOp& mo = mops[i];
      Round* cur = &rounds.back();
      const bool is_list = mo.type == 1 || mo.type >= 3;
      bool fits;
      if (is_list) {
        // the list runs before the round's gates: fine if there are none yet, or if op i commutes
        // with every gate already placed in this round
        fits = true;
        for (int gi : cur->gate) if (gi >= 0 && !commute(i, (size_t)gi)) { fits = false; break; }
      } else {
        fits = cur->gate[mo.bit] < 0;
        if (fits && mo.bit < bp)                           // would run before later-placed higher bits? no: lower bits run first
          for (int gi : cur->gate) if (gi >= 0 && mops[gi].bit > mo.bit && !commute(i, (size_t)gi)) { fits = false; break; }
      }
      if (!fits) {
        rounds.emplace_back();
        cur = &rounds.back();
        cur->gate.assign(std::max(R, 1), -1);
        bp = 0;
      }
      if (is_list) cur->list.push_back((int)i);
      else { cur->gate[mo.bit] = (int)i; bp = std::max(bp, mo.bit + 1); }
    }
    MultiOp ident;                                         // simple passes run a gate in every slot
    memset(&ident, 0, sizeof ident);
    ident.uniform = 1;
    ident.tab = (int)(tables.size() / 2);
    if (simple && R > 0) { const double id4[8] = {1, 0, 0, 0, 0, 0, 1, 0}; tables.insert(tables.end(), id4, id4 + 8); }
    for (const Round& rd : rounds) {
      MultiSlot ls;
      ls.first = (int)sorted.size();
      ls.ndiag = (int)rd.list.size();
      ls.has = 0;
      ls.pad = 0;
      for (int idx : rd.list) sorted.push_back(mops[idx]);
      slots.push_back(ls);
      for (int b = 0; b < R; ++b) {
        MultiSlot gs;
        gs.first = (int)sorted.size();
        gs.ndiag = 0;
        gs.has = rd.gate[b] >= 0;
        gs.pad = 0;
        if (rd.gate[b] >= 0) sorted.push_back(mops[rd.gate[b]]);
        else if (simple) { ident.bit = b; sorted.push_back(ident); }
        slots.push_back(gs);
      }
    }
  }
  const int nrounds = (int)slots.size() / NS;
  void* dops = nullptr;
  void* dslots = nullptr;
  void* dtab = nullptr;
  CHK(arena_put(s, sorted.data(), sorted.size() * sizeof(MultiOp), &dops));
  CHK(arena_put(s, slots.data(), slots.size() * sizeof(MultiSlot), &dslots));
  if (tables.empty()) tables.assign(2, 0.0);
  CHK(arena_put(s, tables.data(), tables.size() * sizeof(double), &dtab));
  const int ntab = (int)(tables.size() / 2);
  const uint64_t nthreads = n >> (R + nzout);
  const double n_written = (double)(n >> nzout);
  const double n_read = g.init ? 0.0 : (double)(n >> (nzout + __builtin_popcount(zreg)));
  const double bytes = 16.0 * (n_written + n_read);
  h->stats.fused_gates += g.ops.size();
  const bool init = g.init;
  const uint64_t nonmask = g.nonmask;
  const double initval = g.initval;
  // the program's last pass also reduces |amp|^2 per workgroup tile: measurement then needs no read pass
  double* tsums = nullptr;
  if (final_pass && h->opt_fused_sums && nthreads % QSV_TPB == 0 && nthreads >= QSV_TPB) {
    const uint64_t nb = nthreads / QSV_TPB;
    if (s.tsums_cap < nb) {
      if (s.d_tsums) HIPCHK(hipFree(s.d_tsums));
      HIPCHK(hipMalloc(&s.d_tsums, nb * sizeof(double)));
      s.tsums_cap = nb;
    }
    tsums = s.d_tsums;
    s.tile_R = R;
    s.tile_rp = rp;
    s.tile_ins = ins;
    s.tile_nblocks = nb;
  }
  // MODE 2: simple pass whose every 2x2 table entry is RX-like (real diagonal, imaginary off-diagonal)
  int mode = simple ? 1 : 0;
  if (simple) {
    bool rx = true;
    for (const MultiOp& mo : sorted) {
      if (mo.type == 1) { rx = false; break; }           // a complex diagonal does not fit the RX-like form
      if (mo.type != 0 && mo.type != 4) continue;
      const size_t nent = (size_t)1 << mo.nlist;
      for (size_t e = 0; e < nent && rx; ++e) {
        const double* m = &tables[2 * ((size_t)mo.tab + 4 * e)];
        if (m[1] != 0.0 || m[7] != 0.0 || m[2] != 0.0 || m[4] != 0.0) rx = false;
      }
      if (!rx) break;
    }
    if (rx) mode = 2;
  }
  // the init-fused pass (write only) is a different kernel instantiation: accounted on its own
  const int r = launch(h, s, init ? QSV_K_MULTI_INIT : QSV_K_MULTI, bytes, [&] {
    const MultiOp* o = reinterpret_cast<const MultiOp*>(dops);
    const MultiSlot* sl = reinterpret_cast<const MultiSlot*>(dslots);
    const cplx* tp = reinterpret_cast<const cplx*>(dtab);
    switch (R) {
      case 0: launch_multi<0>(h, s, init, mode, nthreads, ins, rp, o, sl, nrounds, tp, ntab, nonmask, initval, zreg, tsums); break;
      case 1: launch_multi<1>(h, s, init, mode, nthreads, ins, rp, o, sl, nrounds, tp, ntab, nonmask, initval, zreg, tsums); break;
      case 2: launch_multi<2>(h, s, init, mode, nthreads, ins, rp, o, sl, nrounds, tp, ntab, nonmask, initval, zreg, tsums); break;
      case 3: launch_multi<3>(h, s, init, mode, nthreads, ins, rp, o, sl, nrounds, tp, ntab, nonmask, initval, zreg, tsums); break;
      case 4: launch_multi<4>(h, s, init, mode, nthreads, ins, rp, o, sl, nrounds, tp, ntab, nonmask, initval, zreg, tsums); break;
      case 5: launch_multi<5>(h, s, init, mode, nthreads, ins, rp, o, sl, nrounds, tp, ntab, nonmask, initval, zreg, tsums); break;
      default: launch_multi<6>(h, s, init, mode, nthreads, ins, rp, o, sl, nrounds, tp, ntab, nonmask, initval, zreg, tsums); break;
    }
  });
  s.tile_fresh = tsums != nullptr;
  g = PendingGroup();
  return r;
}

template <int K>
static void launch_kq(const qsv_handle* h, const Shard& s, uint64_t ngroups, const BitIns& ins,
                      const KqOffs& offs, const cplx* u) {
  const size_t shm = sizeof(cplx) << (2 * K);
  hipLaunchKernelGGL((k_kq<K>), dim3(grid_for(h, s, ngroups, QSV_TPB)), dim3(QSV_TPB), shm, s.stream,
                     s.amp, ngroups, ins, offs, u);
}

template <int K>
static void launch_kq_mfma(const qsv_handle* h, const Shard& s, uint64_t nbatch, const BitIns& ins,
                           const KqOffs& offs, const double* ur, const double* ui) {
  // every wave keeps U in registers; give each at least ~8 batches to amortise loading it
  const uint64_t blocks = std::max<uint64_t>(1, std::min<uint64_t>((nbatch + 31) / 32, (uint64_t)s.n_cu * 64));
  hipLaunchKernelGGL((k_kq_mfma<K>), dim3((unsigned)blocks), dim3(QSV_TPB), 0, s.stream, s.amp, nbatch, ins, offs, ur, ui);
}

extern "C" int qsv_apply_kq(qsv_handle* h, int k, const int* qubits, const double* u) {
  if (!h) return fail(QSV_E_BADARG, "NULL handle");
  if (k < 1 || k > QSV_MAX_KQ || !qubits || !u) return fail(QSV_E_BADARG, "kq needs 1..%d qubits and a matrix", QSV_MAX_KQ);
  CHK(check_distinct(h, k, qubits, -1));
  for (int b = 0; b < k; ++b)
    if (qubits[b] >= h->L)
      return fail(QSV_E_UNSUPPORTED, "qubit %d of a dense gate is a shard bit (local qubits: %d); qsv_swap_layout it first", qubits[b], h->L);
  if (h->L < k) return fail(QSV_E_BADARG, "dense %d-qubit gate on %d local qubits", k, h->L);
  const uint64_t n = amps_local(h);

  // matrix-core path: K = 4, 5 natively; K = 3 embedded as I (x) U on one extra (free) qubit
  const int km = k >= 4 ? k : 4;
  if (h->opt_kq_mfma && k >= 3 && h->L - km >= 3) {
    std::vector<int> q(qubits, qubits + k);
    if (k == 3)
      for (int b = 0; b < h->L; ++b)
        if (std::find(q.begin(), q.end(), b) == q.end()) { q.push_back(b); break; }
    const int D = 1 << km, d = 1 << k;
    std::vector<double> ri(2 * (size_t)D * D, 0.0);       // Ur then Ui, row-major D x D
    for (int r = 0; r < D; ++r)
      for (int c = 0; c < D; ++c) {
        if ((r >> k) != (c >> k)) continue;                // block diagonal in the padding bit
        const double* e = u + 2 * ((size_t)(r & (d - 1)) * d + (c & (d - 1)));
        ri[(size_t)r * D + c] = e[0];
        ri[(size_t)D * D + (size_t)r * D + c] = e[1];
      }
    KqOffs offs;
    memset(&offs, 0, sizeof offs);
    for (int j = 0; j < D; ++j)
      for (int b = 0; b < km; ++b)
        if ((j >> b) & 1) offs.off[j] |= 1ull << q[b];
    const BitIns ins = make_ins(q);
    const uint64_t nbatch = (n >> km) / 8;
    for (Shard& s : h->shards) {
      CHK(shard_set(s));
      void* dtab = nullptr;
      CHK(arena_put(s, ri.data(), ri.size() * sizeof(double), &dtab));
      const double* ur = reinterpret_cast<const double*>(dtab);
      const double* ui = ur + (size_t)D * D;
      CHK(launch(h, s, QSV_K_KQ, 32.0 * (double)n, [&] {
        if (km == 4) launch_kq_mfma<4>(h, s, nbatch, ins, offs, ur, ui);
        else         launch_kq_mfma<5>(h, s, nbatch, ins, offs, ur, ui);
      }));
    }
    return QSV_OK;
  }

  const uint64_t ngroups = n >> k;
  KqOffs offs;
  memset(&offs, 0, sizeof offs);
  for (int j = 0; j < (1 << k); ++j)
    for (int b = 0; b < k; ++b)
      if ((j >> b) & 1) offs.off[j] |= 1ull << qubits[b];
  const BitIns ins = make_ins(std::vector<int>(qubits, qubits + k));
  for (Shard& s : h->shards) {
    CHK(shard_set(s));
    void* dtab = nullptr;
    CHK(arena_put(s, u, sizeof(double) * 2 << (2 * k), &dtab));
    const cplx* up = reinterpret_cast<const cplx*>(dtab);
    CHK(launch(h, s, QSV_K_KQ, 32.0 * (double)n, [&] {
      switch (k) {
        case 1: launch_kq<1>(h, s, ngroups, ins, offs, up); break;
        case 2: launch_kq<2>(h, s, ngroups, ins, offs, up); break;
        case 3: launch_kq<3>(h, s, ngroups, ins, offs, up); break;
        case 4: launch_kq<4>(h, s, ngroups, ins, offs, up); break;
        default: launch_kq<5>(h, s, ngroups, ins, offs, up); break;
      }
    }));
  }
  return QSV_OK;
}

// ------------------------------------------------------------------------------------------
// layout swaps and the shard-bit exchange
// ------------------------------------------------------------------------------------------
static int ensure_xbuf(qsv_handle* h, Shard& s, uint64_t amps) {
  if (s.xbuf_amps >= amps) return QSV_OK;
  CHK(shard_set(s));
  for (int b = 0; b < 2; ++b) {
    if (s.xbuf[b]) { HIPCHK(hipFree(s.xbuf[b])); s.xbuf[b] = nullptr; }
    HIPCHK(hipMalloc(&s.xbuf[b], amps * sizeof(cplx)));
  }
  s.xbuf_amps = amps;
  return QSV_OK;
}

static int swap_local(qsv_handle* h, int a, int b) {
  if (a > b) std::swap(a, b);
  const uint64_t n = amps_local(h);
  const uint64_t nq = n >> 2;
  const BitIns ins = make_ins({a, b});
  for (Shard& s : h->shards) {
    CHK(shard_set(s));
    CHK(launch(h, s, QSV_K_SWAP, 32.0 * (double)(n >> 1), [&] {
      hipLaunchKernelGGL(k_swap_bits, dim3(grid_for(h, s, nq, QSV_TPB * 2)), dim3(QSV_TPB), 0, s.stream,
                         s.amp, nq, ins, 1ull << a, 1ull << b);
    }));
  }
  return QSV_OK;
}

// swap shard bit G (>= L) with local bit j
static int exchange(qsv_handle* h, int G, int j) {
  for (Shard& s : h->shards) s.sums_valid = false;
  const int gb = G - h->L;
  const uint64_t n = amps_local(h);
  const uint64_t nhalf = n >> 1;
  h->stats.exchanges += 1;
  if (!h->multiproc) {
    // every pair (A: bit gb = 0, B = A | 1<<gb) owned by this process
    for (Shard& A : h->shards) {
      if ((A.index >> gb) & 1) continue;
      Shard& B = h->shards[A.index | (1 << gb)];
      if (A.device == B.device) {
        CHK(shard_set(A));
        HIPCHK(hipStreamSynchronize(B.stream));
        CHK(launch(h, A, QSV_K_EXCHANGE, 32.0 * (double)n, [&] {
          hipLaunchKernelGGL(k_swap_shards, dim3(grid_for(h, A, nhalf, QSV_TPB * 2)), dim3(QSV_TPB), 0, A.stream,
                             A.amp, B.amp, nhalf, j);
        }));
        HIPCHK(hipStreamSynchronize(A.stream));
      } else {
        const uint64_t chunk = std::min<uint64_t>(nhalf, h->opt_xchunk);
        CHK(ensure_xbuf(h, A, chunk));
        CHK(ensure_xbuf(h, B, chunk));
        for (uint64_t p0 = 0; p0 < nhalf; p0 += chunk) {
          CHK(shard_set(A));
          hipLaunchKernelGGL(k_pack, dim3(grid_for(h, A, chunk, QSV_TPB * 2)), dim3(QSV_TPB), 0, A.stream, A.amp, A.xbuf[0], p0, chunk, j, 1);
          HIPCHK(hipGetLastError());
          CHK(shard_set(B));
          hipLaunchKernelGGL(k_pack, dim3(grid_for(h, B, chunk, QSV_TPB * 2)), dim3(QSV_TPB), 0, B.stream, B.amp, B.xbuf[0], p0, chunk, j, 0);
          HIPCHK(hipGetLastError());
          HIPCHK(hipStreamSynchronize(B.stream));
          CHK(shard_set(A));
          HIPCHK(hipStreamSynchronize(A.stream));
          HIPCHK(hipMemcpyPeer(B.xbuf[1], B.device, A.xbuf[0], A.device, chunk * sizeof(cplx)));
          HIPCHK(hipMemcpyPeer(A.xbuf[1], A.device, B.xbuf[0], B.device, chunk * sizeof(cplx)));
          hipLaunchKernelGGL(k_unpack, dim3(grid_for(h, A, chunk, QSV_TPB * 2)), dim3(QSV_TPB), 0, A.stream, A.amp, A.xbuf[1], p0, chunk, j, 1);
          HIPCHK(hipGetLastError());
          CHK(shard_set(B));
          hipLaunchKernelGGL(k_unpack, dim3(grid_for(h, B, chunk, QSV_TPB * 2)), dim3(QSV_TPB), 0, B.stream, B.amp, B.xbuf[1], p0, chunk, j, 0);
          HIPCHK(hipGetLastError());
          h->stats.exchange_bytes += 2.0 * (double)chunk * sizeof(cplx);
        }
        HIPCHK(hipStreamSynchronize(B.stream));
        CHK(shard_set(A));
        HIPCHK(hipStreamSynchronize(A.stream));
        h->stats.per_kind[QSV_K_EXCHANGE].launches += 1;
        h->stats.per_kind[QSV_K_EXCHANGE].algorithmic_bytes += 32.0 * (double)n;
      }
    }
    return QSV_OK;
  }
  // one shard per process: RCCL send/recv with the partner rank, chunked through staging buffers
  if (!h->comm) return fail(QSV_E_RCCL, "exchange needs qsv_comm_init on every rank first");
  Shard& s = h->shards[0];
  CHK(shard_set(s));
  const int u = (s.index >> gb) & 1;
  const int peer = s.index ^ (1 << gb);
  const int v = 1 - u;                 // my entries with bit j == 1-u travel
  const uint64_t chunk = std::min<uint64_t>(nhalf, h->opt_xchunk);
  CHK(ensure_xbuf(h, s, chunk));
  Pending p{QSV_K_EXCHANGE, nullptr, nullptr};
  if (h->profiling) { CHK(get_event(s, &p.e0)); CHK(get_event(s, &p.e1)); HIPCHK(hipEventRecord(p.e0, s.stream)); }
  for (uint64_t p0 = 0; p0 < nhalf; p0 += chunk) {
    hipLaunchKernelGGL(k_pack, dim3(grid_for(h, s, chunk, QSV_TPB * 2)), dim3(QSV_TPB), 0, s.stream, s.amp, s.xbuf[0], p0, chunk, j, v);
    HIPCHK(hipGetLastError());
    NCCLCHK(g_rccl.GroupStart());
    NCCLCHK(g_rccl.Send(s.xbuf[0], chunk * 2, ncclDouble, peer, h->comm, s.stream));
    NCCLCHK(g_rccl.Recv(s.xbuf[1], chunk * 2, ncclDouble, peer, h->comm, s.stream));
    NCCLCHK(g_rccl.GroupEnd());
    hipLaunchKernelGGL(k_unpack, dim3(grid_for(h, s, chunk, QSV_TPB * 2)), dim3(QSV_TPB), 0, s.stream, s.amp, s.xbuf[1], p0, chunk, j, v);
    HIPCHK(hipGetLastError());
    h->stats.exchange_bytes += (double)chunk * sizeof(cplx);
  }
  if (h->profiling) { HIPCHK(hipEventRecord(p.e1, s.stream)); s.pending.push_back(p); }
  h->stats.per_kind[QSV_K_EXCHANGE].launches += 1;
  h->stats.per_kind[QSV_K_EXCHANGE].algorithmic_bytes += 32.0 * (double)nhalf;
  return QSV_OK;
}

extern "C" int qsv_swap_layout(qsv_handle* h, int npairs, const int* a, const int* b) {
  if (!h) return fail(QSV_E_BADARG, "NULL handle");
  if (npairs < 0 || (npairs && (!a || !b))) return fail(QSV_E_BADARG, "bad swap list");
  for (int i = 0; i < npairs; ++i) {
    CHK(check_qubit(h, a[i], "swap"));
    CHK(check_qubit(h, b[i], "swap"));
    if (a[i] == b[i]) continue;
    const int lo = std::min(a[i], b[i]), hi = std::max(a[i], b[i]);
    if (hi < h->L) CHK(swap_local(h, lo, hi));
    else if (lo < h->L) CHK(exchange(h, hi, lo));
    else return fail(QSV_E_UNSUPPORTED, "swap of two shard bits (%d,%d) is not implemented; route through a local bit", lo, hi);
  }
  return QSV_OK;
}

// ------------------------------------------------------------------------------------------
// measurement
// ------------------------------------------------------------------------------------------
static int block_sums(qsv_handle* h, Shard& s, std::vector<double>& sums) {
  if (s.tile_valid) {                        // left behind by the last k_multi pass: no read pass
    CHK(shard_set(s));
    sums.resize(s.tile_nblocks);
    HIPCHK(hipMemcpyAsync(sums.data(), s.d_tsums, s.tile_nblocks * sizeof(double), hipMemcpyDeviceToHost, s.stream));
    HIPCHK(hipStreamSynchronize(s.stream));
    return QSV_OK;
  }
  if (s.sums_valid) { sums = s.h_sums; return QSV_OK; }
  const uint64_t n = amps_local(h);
  const uint64_t nblk = (n + QSV_SBLOCK - 1) / QSV_SBLOCK;
  CHK(shard_set(s));
  CHK(launch(h, s, QSV_K_PROB, 16.0 * (double)n, [&] {
    hipLaunchKernelGGL(k_blocksum, dim3((unsigned)std::min<uint64_t>(nblk, (uint64_t)s.n_cu * 16)), dim3(QSV_TPB), 0,
                       s.stream, s.amp, n, s.d_sums, nblk);
  }));
  s.h_sums.resize(nblk);
  HIPCHK(hipMemcpyAsync(s.h_sums.data(), s.d_sums, nblk * sizeof(double), hipMemcpyDeviceToHost, s.stream));
  HIPCHK(hipStreamSynchronize(s.stream));
  s.sums_valid = true;
  sums = s.h_sums;
  return QSV_OK;
}

static double pairwise_sum(const double* x, size_t n) {
  if (n <= 64) { double s = 0; for (size_t i = 0; i < n; ++i) s += x[i]; return s; }
  const size_t m = n / 2;
  return pairwise_sum(x, m) + pairwise_sum(x + m, n - m);
}

extern "C" int qsv_norm(qsv_handle* h, double* out) {
  if (!h || !out) return fail(QSV_E_BADARG, "NULL argument");
  double tot = 0;
  std::vector<double> sums;
  for (Shard& s : h->shards) {
    CHK(block_sums(h, s, sums));
    tot += pairwise_sum(sums.data(), sums.size());
  }
  *out = tot;
  return QSV_OK;
}

extern "C" int qsv_sample(qsv_handle* h, uint64_t shots, uint64_t seed, const int* meas_qubits, int n_meas,
                          uint64_t* out_bits) {
  if (!h || (shots && !out_bits)) return fail(QSV_E_BADARG, "NULL argument");
  if (meas_qubits && (n_meas < 0 || n_meas > 64)) return fail(QSV_E_BADARG, "n_meas %d out of range", n_meas);
  if (meas_qubits) for (int i = 0; i < n_meas; ++i) CHK(check_qubit(h, meas_qubits[i], "measured"));
  if (shots == 0) return QSV_OK;
  const size_t ns = h->shards.size();
  std::vector<std::vector<double>> sums(ns);
  std::vector<double> mass(ns);
  double total = 0;
  for (size_t i = 0; i < ns; ++i) {
    CHK(block_sums(h, h->shards[i], sums[i]));
    mass[i] = pairwise_sum(sums[i].data(), sums[i].size());
    total += mass[i];
  }
  if (!(total > 0)) return fail(QSV_E_BADARG, "state has zero norm on this process; nothing to sample");
  // sorted uniforms in [0,total)
  std::mt19937_64 rng(seed);
  std::vector<double> r(shots);
  for (uint64_t s = 0; s < shots; ++s) r[s] = (double)(rng() >> 11) * (1.0 / 9007199254740992.0) * total;
  std::sort(r.begin(), r.end());
  std::vector<uint64_t> idx(shots);
  int last_shard = -1;
  for (size_t i = 0; i < ns; ++i) if (mass[i] > 0) last_shard = (int)i;
  uint64_t s0 = 0;
  double base = 0;
  for (size_t i = 0; i < ns && s0 < shots; ++i) {
    if (!(mass[i] > 0)) continue;
    Shard& sh = h->shards[i];
    const double top = ((int)i == last_shard) ? INFINITY : base + mass[i];
    uint64_t s1 = s0;
    while (s1 < shots && r[s1] < top) ++s1;
    const uint64_t cnt = s1 - s0;
    if (cnt) {
      // walk the blocks of this shard; rounding slack is clamped to the last populated block
      std::vector<uint64_t> blk(cnt);
      std::vector<double> res(cnt);
      const std::vector<double>& bs = sums[i];
      size_t last_nz = 0;
      for (size_t bb = 0; bb < bs.size(); ++bb) if (bs[bb] > 0) last_nz = bb;
      size_t b = 0;
      double pre = base;                 // mass before block b
      for (uint64_t q = 0; q < cnt; ++q) {
        const double x = r[s0 + q];
        while (b < last_nz && x >= pre + bs[b]) { pre += bs[b]; ++b; }
        blk[q] = b;
        res[q] = std::max(0.0, x - pre);
      }
      CHK(shard_set(sh));
      if (sh.sample_cap < cnt) {
        if (sh.d_sblk) { HIPCHK(hipFree(sh.d_sblk)); HIPCHK(hipFree(sh.d_sres)); HIPCHK(hipFree(sh.d_sout)); }
        sh.sample_cap = std::max<size_t>(cnt, 8192);
        HIPCHK(hipMalloc(&sh.d_sblk, sh.sample_cap * sizeof(uint64_t)));
        HIPCHK(hipMalloc(&sh.d_sres, sh.sample_cap * sizeof(double)));
        HIPCHK(hipMalloc(&sh.d_sout, sh.sample_cap * sizeof(uint64_t)));
      }
      HIPCHK(hipMemcpyAsync(sh.d_sblk, blk.data(), cnt * sizeof(uint64_t), hipMemcpyHostToDevice, sh.stream));
      HIPCHK(hipMemcpyAsync(sh.d_sres, res.data(), cnt * sizeof(double), hipMemcpyHostToDevice, sh.stream));
      if (sh.tile_valid) {
        const dim3 g((unsigned)std::min<uint64_t>(cnt, 65535));
#define LT(RR) hipLaunchKernelGGL((k_locate_tile<RR>), g, dim3(QSV_TPB), 0, sh.stream, sh.amp, sh.tile_ins, sh.tile_rp, sh.d_sblk, sh.d_sres, sh.d_sout, cnt)
        switch (sh.tile_R) {
          case 0: LT(0); break; case 1: LT(1); break; case 2: LT(2); break; case 3: LT(3); break;
          case 4: LT(4); break; case 5: LT(5); break; default: LT(6); break;
        }
#undef LT
      } else {
        hipLaunchKernelGGL(k_locate, dim3((unsigned)std::min<uint64_t>(cnt, 65535)), dim3(64), 0, sh.stream,
                           sh.amp, amps_local(h), sh.d_sblk, sh.d_sres, sh.d_sout, cnt);
      }
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(&idx[s0], sh.d_sout, cnt * sizeof(uint64_t), hipMemcpyDeviceToHost, sh.stream));
      HIPCHK(hipStreamSynchronize(sh.stream));
      const uint64_t hi = (uint64_t)sh.index << h->L;
      for (uint64_t q = 0; q < cnt; ++q) idx[s0 + q] |= hi;
    }
    base += mass[i];
    s0 = s1;
  }
  // shots come out sorted by index; decorrelate the order with the same generator
  for (uint64_t s = shots - 1; s > 0; --s) {
    const uint64_t k = rng() % (s + 1);
    std::swap(idx[s], idx[k]);
  }
  for (uint64_t s = 0; s < shots; ++s) {
    if (!meas_qubits) { out_bits[s] = idx[s]; continue; }
    uint64_t bits = 0;
    for (int j = 0; j < n_meas; ++j) bits |= ((idx[s] >> meas_qubits[j]) & 1ull) << j;
    out_bits[s] = bits;
  }
  return QSV_OK;
}

extern "C" int qsv_probabilities_cond(qsv_handle* h, const int* qubits, int k, uint64_t fix_mask, uint64_t fix_val,
                                      double* out) {
  if (!h || !out || (k && !qubits)) return fail(QSV_E_BADARG, "NULL argument");
  if (k < 0 || k > 26) return fail(QSV_E_BADARG, "marginal over %d qubits unsupported (max 26)", k);
  for (int i = 0; i < k; ++i) CHK(check_qubit(h, qubits[i], "marginal"));
  const int ntab = 1 << k;
  BitList bl;
  bl.n = k;
  for (int b = 0; b < k; ++b) bl.pos[b] = qubits[b];
  std::vector<double> part(ntab);
  for (int i = 0; i < ntab; ++i) out[i] = 0.0;
  const uint64_t n = amps_local(h);
  for (Shard& s : h->shards) {
    CHK(shard_set(s));
    double* d_out = nullptr;
    HIPCHK(hipMalloc(&d_out, ntab * sizeof(double)));
    HIPCHK(hipMemsetAsync(d_out, 0, ntab * sizeof(double), s.stream));
    const uint64_t hi = (uint64_t)s.index << h->L;
    const bool lds = k <= 12;
    CHK(launch(h, s, QSV_K_PROB, 16.0 * (double)n, [&] {
      const dim3 g(grid_for(h, s, n, QSV_TPB * 8));
      if (lds) hipLaunchKernelGGL((k_marginal<true>), g, dim3(QSV_TPB), ntab * sizeof(double), s.stream, s.amp, n, hi, bl, fix_mask, fix_val, d_out, ntab);
      else     hipLaunchKernelGGL((k_marginal<false>), g, dim3(QSV_TPB), 0, s.stream, s.amp, n, hi, bl, fix_mask, fix_val, d_out, ntab);
    }));
    HIPCHK(hipMemcpyAsync(part.data(), d_out, ntab * sizeof(double), hipMemcpyDeviceToHost, s.stream));
    HIPCHK(hipStreamSynchronize(s.stream));
    HIPCHK(hipFree(d_out));
    for (int i = 0; i < ntab; ++i) out[i] += part[i];
  }
  return QSV_OK;
}
extern "C" int qsv_probabilities(qsv_handle* h, const int* qubits, int k, double* out) {
  return qsv_probabilities_cond(h, qubits, k, 0ull, 0ull, out);
}

static int amp_copy(qsv_handle* h, uint64_t start, uint64_t count, double* out, const double* in) {
  if (!h || (count && !out && !in)) return fail(QSV_E_BADARG, "NULL argument");
  const uint64_t n = amps_local(h);
  uint64_t done = 0;
  while (done < count) {
    const uint64_t g = start + done;
    const int si = (int)(g >> h->L);
    Shard* sh = nullptr;
    for (Shard& s : h->shards) if (s.index == si) sh = &s;
    if (!sh) return fail(QSV_E_BADARG, "amplitude %llu lives on shard %d, which this process does not own", (unsigned long long)g, si);
    const uint64_t off = g & (n - 1);
    const uint64_t m = std::min(count - done, n - off);
    CHK(shard_set(*sh));
    HIPCHK(hipStreamSynchronize(sh->stream));
    if (out) HIPCHK(hipMemcpy(out + 2 * done, sh->amp + off, m * sizeof(cplx), hipMemcpyDeviceToHost));
    else   { HIPCHK(hipMemcpy(sh->amp + off, in + 2 * done, m * sizeof(cplx), hipMemcpyHostToDevice)); sh->sums_valid = false; }
    done += m;
  }
  return QSV_OK;
}
extern "C" int qsv_copy_state(qsv_handle* dst, qsv_handle* src) {
  if (!dst || !src) return fail(QSV_E_BADARG, "NULL handle");
  if (dst->W != src->W || dst->P != src->P || dst->shards.size() != src->shards.size())
    return fail(QSV_E_BADARG, "qsv_copy_state: handles differ in shape (%d/%d qubits, %d/%d shards)", dst->W, src->W, dst->P, src->P);
  const size_t bytes = amps_local(src) * sizeof(cplx);
  for (size_t i = 0; i < src->shards.size(); ++i) {
    Shard& a = src->shards[i];
    Shard& b = dst->shards[i];
    if (a.index != b.index) return fail(QSV_E_BADARG, "qsv_copy_state: shard order differs");
    CHK(shard_set(a));
    HIPCHK(hipStreamSynchronize(a.stream));
    CHK(shard_set(b));
    HIPCHK(hipMemcpyAsync(b.amp, a.amp, bytes, hipMemcpyDeviceToDevice, b.stream));
    b.zmask = a.zmask;
    b.sums_valid = false;
    b.tile_valid = false;
    dst->stats.per_kind[QSV_K_SWAP].launches += 1;
    dst->stats.per_kind[QSV_K_SWAP].algorithmic_bytes += 2.0 * (double)bytes;
  }
  return QSV_OK;
}
extern "C" int qsv_get_amplitudes(qsv_handle* h, uint64_t start, uint64_t count, double* out) { return amp_copy(h, start, count, out, nullptr); }
extern "C" int qsv_set_amplitudes(qsv_handle* h, uint64_t start, uint64_t count, const double* in) { return amp_copy(h, start, count, nullptr, in); }

// ------------------------------------------------------------------------------------------
// batched execution: resolve per shard, block consecutive gates into k_multi passes
// ------------------------------------------------------------------------------------------
extern "C" int qsv_exec(qsv_handle* h, const qsv_op* ops, int n_ops, const double* data, uint64_t n_data) {
  if (!h || (n_ops && !ops)) return fail(QSV_E_BADARG, "NULL argument");
  const size_t ns = h->shards.size();
  std::vector<PendingGroup> pend(ns);
  auto flush_all = [&](bool final_pass = false) -> int {
    for (size_t i = 0; i < ns; ++i) CHK(flush_group(h, h->shards[i], pend[i], final_pass));
    return QSV_OK;
  };
  for (Shard& s : h->shards) s.tile_fresh = false;
  for (int i = 0; i < n_ops; ++i) {
    const qsv_op& o = ops[i];
    if (o.n < 0 || o.n > QSV_MAX_CTRL) return fail(QSV_E_BADARG, "op %d: n=%d out of range", i, o.n);
    const double* d = data ? data + o.data_off : nullptr;
    auto need = [&](uint64_t cnt) -> int {
      if (!data || o.data_off + cnt > n_data) return fail(QSV_E_BADARG, "op %d: data range [%llu,+%llu) outside pool of %llu", i,
                                                          (unsigned long long)o.data_off, (unsigned long long)cnt, (unsigned long long)n_data);
      return QSV_OK;
    };
    switch (o.kind) {
      case QSV_OP_INIT_ZERO:
      case QSV_OP_INIT_UNIFORM: {
        CHK(flush_all());
        const uint64_t mask = o.kind == QSV_OP_INIT_ZERO ? 0ull : o.mask;
        if (h->W < 64 && (mask >> h->W)) return fail(QSV_E_BADARG, "mask has bits beyond qubit %d", h->W - 1);
        if (h->opt_multi_r < 1) { CHK(qsv_init_uniform(h, mask)); break; }
        const double val = std::pow(2.0, -0.5 * __builtin_popcountll(mask));
        for (size_t k = 0; k < ns; ++k) {
          const Shard& s = h->shards[k];
          const uint64_t hi = (uint64_t)s.index << h->L;
          pend[k].init = true;
          pend[k].initval = (hi & ~mask) ? 0.0 : val;
          pend[k].nonmask = ~mask & (amps_local(h) - 1);
          h->shards[k].zmask = h->opt_zero_tracking ? pend[k].nonmask : 0ull;
        }
        break;
      }
      case QSV_OP_1Q: case QSV_OP_MCX: case QSV_OP_DIAG: case QSV_OP_MCPHASE: case QSV_OP_MUX: {
        if (o.kind == QSV_OP_1Q) CHK(need(8));
        if (o.kind == QSV_OP_DIAG) CHK(need(2ull << o.n));
        if (o.kind == QSV_OP_MUX) CHK(need(8ull << o.n));
        CHK(validate_gate(h, o.kind, o.n, o.qubits, o.target, o.kind == QSV_OP_MCX || o.kind == QSV_OP_MCPHASE ? (const void*)h : (const void*)d));
        LocalOp lo;
        for (size_t k = 0; k < ns; ++k) {
          Shard& s = h->shards[k];
          if (!resolve_gate(h, s, o.kind, o.n, o.qubits, o.vals, o.target, d, o.angle, lo)) continue;
          if (!groupable(h, lo)) {
            CHK(flush_group(h, s, pend[k]));
            CHK(materialize(h, s));
            CHK(run_single(h, s, lo));
            continue;
          }
          if (!group_fits(h, pend[k], lo)) CHK(flush_group(h, s, pend[k]));
          group_add(h, s, pend[k], std::move(lo));
        }
        break;
      }
      case QSV_OP_KQ:
        CHK(flush_all());
        for (Shard& s : h->shards) CHK(materialize(h, s));
        CHK(need(2ull << (2 * o.n)));
        CHK(qsv_apply_kq(h, o.n, o.qubits, d));
        break;
      case QSV_OP_SWAP:
        CHK(flush_all());
        for (Shard& s : h->shards) CHK(materialize(h, s));
        CHK(qsv_swap_layout(h, o.n, o.qubits, o.vals));
        break;
      default:
        return fail(QSV_E_BADARG, "op %d: unknown kind %d", i, o.kind);
    }
  }
  CHK(flush_all(true));
  for (Shard& s : h->shards) {
    CHK(materialize(h, s));                  // writes zeros only: the tile sums stay right
    s.tile_valid = s.tile_fresh;
    s.tile_fresh = false;
  }
  return QSV_OK;
}

// ------------------------------------------------------------------------------------------
// instrumentation
// ------------------------------------------------------------------------------------------
extern "C" int qsv_set_profiling(qsv_handle* h, int on) {
  if (!h) return fail(QSV_E_BADARG, "NULL handle");
  if (!on) CHK(drain_pending(h));
  h->profiling = on != 0;
  return QSV_OK;
}
extern "C" int qsv_reset_stats(qsv_handle* h) {
  if (!h) return fail(QSV_E_BADARG, "NULL handle");
  CHK(drain_pending(h));
  memset(&h->stats, 0, sizeof h->stats);
  return QSV_OK;
}
extern "C" int qsv_get_stats(qsv_handle* h, qsv_stats* out) {
  if (!h || !out) return fail(QSV_E_BADARG, "NULL argument");
  CHK(qsv_sync(h));
  CHK(drain_pending(h));
  *out = h->stats;
  return QSV_OK;
}
extern "C" int qsv_timer_begin(qsv_handle* h) {
  if (!h) return fail(QSV_E_BADARG, "NULL handle");
  Shard& s = h->shards[0];
  CHK(shard_set(s));
  if (!h->t0) { HIPCHK(hipEventCreate(&h->t0)); HIPCHK(hipEventCreate(&h->t1)); }
  HIPCHK(hipEventRecord(h->t0, s.stream));
  return QSV_OK;
}
extern "C" int qsv_timer_end(qsv_handle* h, double* ms) {
  if (!h || !ms || !h->t0) return fail(QSV_E_BADARG, "timer not started");
  Shard& s = h->shards[0];
  CHK(shard_set(s));
  HIPCHK(hipEventRecord(h->t1, s.stream));
  HIPCHK(hipEventSynchronize(h->t1));
  float f = 0.f;
  HIPCHK(hipEventElapsedTime(&f, h->t0, h->t1));
  *ms = f;
  return QSV_OK;
}
extern "C" int qsv_set_option(qsv_handle* h, const char* name, int value) {
  if (!h || !name) return fail(QSV_E_BADARG, "NULL argument");
  if (!strcmp(name, "blocks_per_cu")) { if (value < 1) return fail(QSV_E_BADARG, "blocks_per_cu < 1"); h->opt_blocks_per_cu = value; }
  else if (!strcmp(name, "unroll")) h->opt_unroll = value;
  else if (!strcmp(name, "lowt_shuffle")) h->opt_lowt_shuffle = value;
  else if (!strcmp(name, "nontemporal")) h->opt_nt = value;
  else if (!strcmp(name, "lane_targets")) h->opt_lane_targets = value != 0;
  else if (!strcmp(name, "fused_sums")) h->opt_fused_sums = value != 0;
  else if (!strcmp(name, "pair_variant")) h->opt_pair_variant = value;
  else if (!strcmp(name, "kq_mfma")) h->opt_kq_mfma = value != 0;
  else if (!strcmp(name, "zero_tracking")) h->opt_zero_tracking = value != 0;
  else if (!strcmp(name, "multi_r")) { if (value < 0 || value > QSV_MULTI_MAXR) return fail(QSV_E_BADARG, "multi_r out of range"); h->opt_multi_r = value; }
  else if (!strcmp(name, "exchange_chunk_log2")) { if (value < 4 || value > 32) return fail(QSV_E_BADARG, "exchange_chunk_log2 out of range"); h->opt_xchunk = 1ull << value; }
  else return fail(QSV_E_BADARG, "unknown option %s", name);
  return QSV_OK;
}
extern "C" const char* qsv_last_error(void) { return g_err.c_str(); }
extern "C" const char* qsv_version(void) { return "qsv 0.1 (gfx950)"; }
