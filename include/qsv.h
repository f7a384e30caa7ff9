/*
 * qsv.h -- C ABI of libqsv.so, the MI355X (gfx950) fp64 statevector engine that replaces
 * the Qiskit-Aer call of the reference's hot path:
 *
 *     simulator = Aer.get_backend('qasm_simulator')            /root/reference/run_experiment.py:54
 *     result    = simulator.run(T, shots=SHOTS).result()       /root/reference/run_experiment.py:56
 *     counts    = result.get_counts()                          /root/reference/run_experiment.py:57
 *
 * The reference's "FFI" for this path is the Python -> Aer C++ extension call hidden inside
 * run(); this header is what a ctypes (or cgo / JNI) binding of that call binds instead.
 * Every entry point names the reference gate / step it executes (file:line into
 * /root/reference).  See INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *  - amplitude vector: 2^n_qubits complex128, interleaved (re, im); qubit q <-> bit q of the
 *    basis index (Qiskit little-endian).  Qubit numbers here are PHYSICAL positions; the
 *    Python host keeps the logical->physical layout map.
 *  - sharding: P = 2^g shards by the g highest physical qubits; shard s holds indices
 *    [s << L, (s+1) << L), L = n_qubits - g.  Qubits >= L are "shard bits".
 *  - every pointer argument is a caller-owned host buffer, read or filled synchronously;
 *    nothing is retained after return.  No callbacks.
 *  - return 0 on success; <0 on error (QSV_E_*); text via qsv_last_error() (thread local).
 *  - a handle is not thread-safe; distinct handles may be used from distinct threads.
 *  - gate calls are asynchronous on the device stream(s); results are synchronised by
 *    qsv_sync / qsv_probabilities / qsv_sample / qsv_get_amplitudes / qsv_get_stats.
 */
#ifndef QSV_H
#define QSV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qsv_handle qsv_handle;

#define QSV_OK             0
#define QSV_E_BADARG      -1
#define QSV_E_NOMEM       -2
#define QSV_E_HIP         -3
#define QSV_E_RCCL        -4
#define QSV_E_UNSUPPORTED -5

#define QSV_MAX_CTRL      16   /* controls of one gate / qubits of one diagonal        */
#define QSV_MAX_KQ         5   /* dense k-qubit unitary                                */
#define QSV_UNIQUE_ID_BYTES 128

/* kernel kinds, index into qsv_stats.per_kind[] */
enum {
  QSV_K_INIT = 0, QSV_K_1Q, QSV_K_X, QSV_K_DIAG, QSV_K_MCPHASE, QSV_K_MUX, QSV_K_KQ,
  QSV_K_PROB, QSV_K_SWAP, QSV_K_EXCHANGE, QSV_K_MULTI, QSV_K_MULTI_INIT, QSV_K_INIT_PROD, QSV_K_COUNT
};

typedef struct {
  uint64_t launches;          /* kernel launches of this kind                              */
  double   algorithmic_bytes; /* SURVEY.md 8(d) byte model, summed                         */
  double   device_ms;         /* HIP-event time on the launch stream (profiling on only)   */
} qsv_kind_stats;

typedef struct {
  qsv_kind_stats per_kind[QSV_K_COUNT];
  uint64_t exchanges;         /* shard-bit exchanges performed                              */
  double   exchange_bytes;    /* bytes sent over the fabric by this process                 */
  uint64_t fused_gates;       /* gates executed inside multi-gate (QSV_K_MULTI) passes      */
} qsv_stats;

/* ---- life cycle -------------------------------------------------------------------- */

/* number of visible HIP devices (<0 on error) */
int qsv_device_count(void);

/* HBM of one device in bytes (free now / total): lets the caller size a state vector to the
 * card (2^34 complex128 = 256 GiB fits the 288 GiB of one MI355X). */
int qsv_device_memory(int device_id, uint64_t* free_bytes, uint64_t* total_bytes);

/* PCI bus id of a device ("0000:05:00.0"): two ranks that report the same id on the same host
 * share one GPU (RCCL refuses that; the peer-mapped transport below does not mind) */
int qsv_device_bus_id(int device_id, char* out, int len);

/* One process drives n_devices shards (n_devices a power of two).  device_ids[i] is the HIP
 * device of shard i; repeating an id places several ("virtual") shards on one GPU.
 * Replaces: the state allocation inside Aer's run() (run_experiment.py:56). */
int qsv_create(int n_qubits, int n_devices, const int* device_ids, qsv_handle** out);

/* One process per GPU (torchrun-style launch): this process owns shard `rank` of
 * `world_size` (a power of two) on HIP device device_id.  Exchanges go over RCCL once
 * qsv_comm_init has been called on every rank. */
int qsv_create_rank(int n_qubits, int world_size, int rank, int device_id, qsv_handle** out);

/* RCCL bootstrap for qsv_create_rank handles: rank 0 calls qsv_comm_unique_id, the host
 * side broadcasts the 128 bytes, every rank calls qsv_comm_init (collective). */
int qsv_comm_unique_id(uint8_t id[QSV_UNIQUE_ID_BYTES]);
int qsv_comm_init(qsv_handle* h, const uint8_t id[QSV_UNIQUE_ID_BYTES]);

/* Peer-mapped exchange (alternative transport for qsv_create_rank handles on ONE node): every
 * rank exports an IPC handle of its shard, the host side all-gathers the QSV_IPC_HANDLE_BYTES
 * each and every rank attaches the others; `shm_name` names a small POSIX shared-memory segment
 * (created by the rank that passes create != 0, before the others attach) through which a pair
 * of ranks synchronises around an exchange.  With it a shard-bit exchange is ONE kernel per rank
 * that swaps its half of the pairs in place, reading and writing the partner's shard directly
 * (over xGMI between GPUs; plain HBM when two ranks share a device, which RCCL refuses) -- no
 * staging buffers, each amplitude crosses the link once.  RCCL stays the default transport
 * whenever every rank has a device of its own. */
#define QSV_IPC_HANDLE_BYTES 64
int qsv_ipc_export(qsv_handle* h, uint8_t out[QSV_IPC_HANDLE_BYTES]);
int qsv_ipc_attach(qsv_handle* h, const uint8_t* handles /* world x QSV_IPC_HANDLE_BYTES */,
                   const char* shm_name, int create);

/* Diagnostic: bring RCCL up on one device as a 1-rank communicator and push `n_doubles` through
 * the same grouped ncclSend/ncclRecv + stream sequence the shard exchange uses (rank 0 to itself),
 * then compare.  Exercises the dlopen binding, communicator life cycle and call order on a box
 * with a single GPU, where RCCL refuses two ranks on one device.  0 on success. */
int qsv_rccl_selftest(int device_id, uint64_t n_doubles);

/* Diagnostic: the RCCL side of a BATCHED exchange (several shard bits at once = an all-to-all inside a
 * group of shards) on a 1-rank communicator with rank 0 as its own peers: staging buffers, the
 * two-stream double-buffered pack / grouped send+recv / unpack pipeline in chunks of 2^chunk_log2
 * amplitudes over a 2^n_qubits shard, verified element by element.  0 on success. */
int qsv_rccl_exchange_selftest(int device_id, int n_qubits, int chunk_log2);

/* Diagnostic: leave quiet NaNs in the LDS of every compute unit of the handle's devices (LDS is not cleared
 * between kernels).  A kernel that reads a table it never staged then produces NaN deterministically instead
 * of "usually fine" -- used by the GPU tests before the init passes, whose idle workgroups skip the staging. */
int qsv_poison_lds(qsv_handle* h);

int qsv_destroy(qsv_handle* h);
int qsv_sync(qsv_handle* h);

/* ---- state preparation -------------------------------------------------------------- */

/* |0...0>  -- implicit initial state of a QuantumCircuit (QCMRF.py:78). */
int qsv_init_zero(qsv_handle* h);

/* H on every qubit of qubit_mask applied to |0...0>, written directly:
 * amp = 2^{-popcount/2} where (index & ~mask) == 0, else 0.   (QCMRF.py:204-205) */
int qsv_init_uniform(qsv_handle* h, uint64_t qubit_mask);

/* ---- gates -------------------------------------------------------------------------- */

/* dense 2x2 on qubit t; m = row-major {re,im} x 4.   h / sx (QCMRF.py:231,236; basis 'sx'
 * run_experiment.py:52). */
int qsv_apply_1q(qsv_handle* h, int t, const double m[8]);

/* multi-controlled 2x2: fires where bit ctrls[i] == ctrl_vals[i] (NULL = all ones). */
int qsv_apply_mc1q(qsv_handle* h, int n_ctrl, const int* ctrls, const int* ctrl_vals,
                   int t, const double m[8]);

/* X / CX / CCX / MCX(k) with +-control flags: the AND gate of QCMRF.py:224-225,227, the
 * x of QCMRF.py:233,235, basis 'cx','x'. */
int qsv_apply_mcx(qsv_handle* h, int n_ctrl, const int* ctrls, const int* ctrl_vals, int t);

/* k-qubit diagonal: amp[i] *= table[j], j = sum_b bit(i, qubits[b]) << b; table = 2^k x {re,im}.
 * rz / p / merged phase blocks (cU_C of QCMRF.py:218-228 is one such table). */
int qsv_apply_diag(qsv_handle* h, int k, const int* qubits, const double* table);

/* e^{i angle} on the subspace where bit ctrls[i] == ctrl_vals[i] for all i (n_ctrl >= 1).
 * cp(2 gamma, n, anc) of QCMRF.py:226 is n_ctrl = 2. */
int qsv_apply_mcphase(qsv_handle* h, int n_ctrl, const int* ctrls, const int* ctrl_vals,
                      double angle);

/* uniformly controlled 2x2 on t: mats[j] (8 doubles each), j = control bits, ctrls[0] = LSB.
 * The whole real-part-extraction sandwich H cU X cU^dg X H of QCMRF.py:231-236 is one of
 * these. */
int qsv_apply_mux_1q(qsv_handle* h, int k, const int* ctrls, int t, const double* mats);

/* dense 2^k x 2^k unitary (k <= QSV_MAX_KQ), row-major {re,im}; index bit b <-> qubits[b].
 * Fused blocks of a transpiled circuit (run_experiment.py:52). */
int qsv_apply_kq(qsv_handle* h, int k, const int* qubits, const double* u);

/* physically swap the amplitude-index positions a[i] <-> b[i].  Both local: a permutation
 * sweep.  One of them a shard bit: the pairwise half-shard exchange (in-place swap kernel for
 * virtual shards and peer-mapped ranks, peer copy between devices of one process, RCCL
 * send/recv between ranks).  Several such pairs on distinct qubits in ONE call are executed as
 * one batched exchange: an all-to-all inside every group of 2^k shards (each shard sends 1/2^k
 * of itself to each of its 2^k - 1 partners, all links busy at once) instead of k sequential
 * half-shard exchanges.  The RCCL path is chunked and double buffered (pack / unpack on a second
 * stream while the neighbouring chunk is on the wire). */
int qsv_swap_layout(qsv_handle* h, int npairs, const int* a, const int* b);

/* ---- measurement (QCMRF.py:239,243; shots of run_experiment.py:56) -------------------- */

/* marginal distribution over `qubits` (k <= 26): out[j] += sum |amp|^2, j as in qsv_apply_diag.
 * Only amplitudes with (index & fix_mask) == fix_val contribute (fix_mask = 0: all).
 * Multi-process handles return this rank's partial sums. */
int qsv_probabilities(qsv_handle* h, const int* qubits, int k, double* out);
int qsv_probabilities_cond(qsv_handle* h, const int* qubits, int k,
                           uint64_t fix_mask, uint64_t fix_val, double* out);

/* expectation of a real DIAGONAL observable on the resident state: table has 2^k doubles, indexed
 * like qsv_apply_diag's (k <= 26; qubits may include shard bits).  Only amplitudes whose global
 * index g has (g & fix_mask) == fix_val contribute:
 *     out[0] = sum |amp[g]|^2 table[j(g)]      out[1] = sum |amp[g]|^2      (both over those g)
 * so out[0] / out[1] is the expectation in the post-selected state (fix_mask = 0: plain <O>,
 * out[1] = norm).  One read pass over the shard(s), 16 B per amplitude.  Multi-process handles
 * return this rank's partial sums.
 * Replaces: the opflow expectation of QCMRF.Hamiltonian() / sufficient_statistic()
 * (QCMRF.py:159-193): H = -sum theta_{C,y} Phi_{C,y} is diagonal in the computational basis. */
int qsv_expect_diag(qsv_handle* h, const int* qubits, int k, const double* table,
                    uint64_t fix_mask, uint64_t fix_val, double out[2]);

/* sum |amp|^2 over this process's shards */
int qsv_norm(qsv_handle* h, double* out);

/* draw `shots` basis states from |amp|^2 over this process's shards (normalised by their
 * mass); out_bits[s] bit j = value of qubit meas_qubits[j] (n_meas <= 64; an entry of -1 leaves
 * bit j at 0: a classical bit no measurement writes).
 * meas_qubits == NULL: out_bits[s] = the full basis index. */
int qsv_sample(qsv_handle* h, uint64_t shots, uint64_t seed, const int* meas_qubits,
               int n_meas, uint64_t* out_bits);

/* copy amplitudes [start, start+count) of the GLOBAL index space into out (2*count doubles);
 * the range must lie inside shards owned by this process. */
int qsv_get_amplitudes(qsv_handle* h, uint64_t start, uint64_t count, double* out);
int qsv_set_amplitudes(qsv_handle* h, uint64_t start, uint64_t count, const double* in);

/* dst <- src (same n_qubits and shard structure), device to device.  Branch points of the
 * trajectory mode: the mid-circuit measure of QCMRF.py:238-239 taken when it occurs. */
int qsv_copy_state(qsv_handle* dst, qsv_handle* src);

/* ---- batched execution -------------------------------------------------------------- */

enum {
  QSV_OP_INIT_ZERO = 0, QSV_OP_INIT_UNIFORM, QSV_OP_1Q, QSV_OP_MCX, QSV_OP_DIAG,
  QSV_OP_MCPHASE, QSV_OP_MUX, QSV_OP_KQ, QSV_OP_SWAP
};

/* scheduling hint from the planner: close the current multi-gate pass before this gate (the
 * result does not depend on it; it only decides which gates share one sweep of the shard) */
#define QSV_OPF_NEW_PASS 1

typedef struct {
  int32_t  kind;                   /* QSV_OP_*                                             */
  int32_t  target;                 /* target qubit (1Q, MCX, MUX)                          */
  int32_t  n;                      /* number of controls / qubits in qubits[]              */
  int32_t  flags;                  /* QSV_OPF_*                                            */
  int32_t  qubits[QSV_MAX_CTRL];   /* controls, or qubit list (DIAG, KQ), or swap a-list   */
  int32_t  vals[QSV_MAX_CTRL];     /* control values, or swap b-list                       */
  uint64_t data_off;               /* offset in doubles into `data` (matrix / table)       */
  uint64_t mask;                   /* INIT_UNIFORM qubit mask                              */
  double   angle;                  /* MCPHASE                                              */
} qsv_op;

/* run a whole program in one call (one ctypes crossing per circuit) */
int qsv_exec(qsv_handle* h, const qsv_op* ops, int n_ops, const double* data, uint64_t n_data);

/* ---- instrumentation ---------------------------------------------------------------- */

/* on: bracket every kernel launch with HIP events on its own stream (costs a little) */
int qsv_set_profiling(qsv_handle* h, int on);
int qsv_reset_stats(qsv_handle* h);
int qsv_get_stats(qsv_handle* h, qsv_stats* out);

/* HIP-event stopwatch on shard 0's stream: begin records, end records+synchronises */
int qsv_timer_begin(qsv_handle* h);
int qsv_timer_end(qsv_handle* h, double* ms);

/* Tuning knobs (defaults in brackets; none of them changes a result, only how it is computed).  Unknown
 * name or value out of range: QSV_E_BADARG.
 *   passes      multi_r [5]        register targets of a k_multi pass at most (0: one kernel per gate)
 *               general_r [4]      ... of a GENERAL pass (masked X / 2x2 / phase, register selects); its tile width
 *               general_light_r [5] tile width a general pass with little arithmetic is padded to
 *               pass_max_ops [64]  ops per pass at most        pass_budget [0]   arithmetic cap of a general pass, % of one sweep
 *               lane_targets [1]   targets < 6 ride on lane bits (wave shuffles)   dyn_lanes [3]  lane bits 3..5 lent per pass
 *               lane_map [1]       lane bit 5 on address bit 11 when the tile allows (2: only a tile on bits 6..10; 0: never)    pass_hints [1]  honour QSV_OPF_NEW_PASS
 *               xframe [1]         uncontrolled X = XOR on store addresses / pending across passes (never a data move)
 *               single_shortcut [1] a one-op pass runs as its dedicated kernel     trace_passes [0]  one stderr line per pass
 *   generator   init_prod [1]      init x diagonal factors in one write-only pass   init_prod_r [0 = by shard size], init_prod_bit0 [0 = by size]
 *   memory      nontemporal [-1], multi_nt [-1], init_prod_nt [-1]   non-temporal loads/stores: -1 by shard size, 0 never, 1 always
 *   measurement cache_sums [1], fused_sums [1]   keep / produce per-tile |amp|^2 sums in the last pass of a program
 *   kernels     unroll [4], lowt_shuffle [1], pair_variant [0], kq_mfma [1], blocks_per_cu [65536]
 *               swizzle [1]        one-gate kernels: lane bit 5 of a wave access carries address bit 11 (two 512-byte runs 32 KiB apart);
 *                                  1 from 2^26 amplitudes per shard, 2 from 2^14, 0 never        lane_map_min_l [26]  same for k_multi tiles other than bits 6..10
 *   other       zero_tracking [0]  skip amplitudes known to be zero (opt-in)       exchange_chunk_log2 [24]  amplitudes per exchange chunk */
int qsv_set_option(qsv_handle* h, const char* name, int value);

const char* qsv_last_error(void);
const char* qsv_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QSV_H */
