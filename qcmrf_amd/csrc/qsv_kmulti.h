// qsv_kmulti.h -- the register-tiled multi-gate pass (k_multi) and its tile geometry.  Its own header so that
// the instantiations compile in translation units of their own (qsv_kmulti_inst.h).
#pragma once
#include <cstddef>
#include "qsv_common.h"

// ---------------------------------------------------------------------------------------
// k_multi<R>: register-tiled multi-gate sweep -- several gates per HBM pass.
//
// Each lane owns the 2^R amplitudes that differ only in R "register bits" (the target qubits of
// the gates of this pass).  With every register bit >= 6 the lane id supplies address bits 0..5,
// so each of the 2^R loads of a wavefront is one contiguous 1 KiB global_load_dwordx4; the whole
// 2^R-dimensional subspace a gate needs is then in the lane's own VGPRs (R = 6: 64 complex128 =
// 256 of the 512 VGPRs a CDNA4 lane may hold) -- no LDS traffic, no shuffles, no barriers for
// the amplitudes.  LDS holds only the gate tables.  Gates are applied in program order; controls
// and diagonal selects may sit on any bit (register, lane or block bits).
// HBM traffic: one read + one write of the shard (32 B / amplitude) for up to ~a dozen gates.
// ---------------------------------------------------------------------------------------
#define QSV_MULTI_MAXR 6
#define QSV_MULTI_MAXLIST 10

// Field order matters to the general pass: its interpreter reads an op's control words with scalar loads, and the scalar data
// cache is small (16 KiB shared by a few CUs).  Everything a controlled X or a controlled phase needs -- the bulk of the reference's
// unfused stream -- sits in the FIRST 64-byte line of the (64-byte aligned) descriptor: a 56-op pass then keeps 3.5 KiB of
// descriptors hot instead of striding through 10 KiB with every op's words straddling two lines.
struct alignas(64) MultiOp {
  int shape;                      // general passes: which of the eleven update shapes (GS_*) this op is
  int bit;                        // register bit of the target (types 0, 2); lane bit (types 4, 5)
  int tab;                        // table offset in LDS, in complex128 units
  int type;                       // 0 mux table | 1 diag table | 2 controlled 2x2 | 3 controlled phase | 4, 5: types 0, 2 on a lane bit
  unsigned long long tmask, tval; // types 2,3: condition on the lane/block part of the address
  unsigned long long rfire;       // types 2,3,5: bit j set <=> (j & rmask) == rval, j = register index of an amplitude
  double m[8];                    // type 2: 2x2 row-major {re,im}; type 3: m[0..1] = phase (m[0..1] still in the first line)
  int uniform;                    // 1: table index (types 0, 1, 4) / control condition (2, 3, 5: rmask == 0) independent of the register bits
  int nlist;                      // entries of pos[] (types 0, 1); types 2, 5: != 0 marks a plain X
  unsigned int rmask, rval;       // types 2,3: condition on the register index
  int regw[QSV_MULTI_MAXR];       // types 0,1: table-index weight of register bit c
  int pos[QSV_MULTI_MAXLIST];     // types 0,1: address bit of list entry e; -1 if it is a register bit
};
static_assert(sizeof(MultiOp) % 64 == 0 && offsetof(MultiOp, m) + 16 <= 64, "the hot words of a MultiOp share its first cache line");
// update shapes of a general pass.  Gates on a register bit: table with one matrix per thread / per
// pair, controlled X, one controlled matrix; then the ops without a register target: diagonal table
// with one entry per thread / per amplitude, controlled phase, lane-bit gate as a table (per thread /
// per amplitude), controlled X on a lane bit, one controlled matrix on a lane bit
enum { GS_TAB_T = 0, GS_TAB_P = 1, GS_X = 2, GS_MAT = 3, GS_DIAG_T = 4, GS_DIAG_A = 5, GS_PHASE = 6, GS_LTAB_T = 7,
       GS_LTAB_A = 8, GS_LX = 9, GS_LMAT = 10 };
struct RegPos { int pos[QSV_MULTI_MAXR]; };
// Address bit carried by lane bits 3, 4, 5 of a wavefront.  Lane bits 0..2 are always address
// bits 0..2 (8 lanes x 16 B = one 128-byte line per lane group); the upper three default to
// {3,4,5} (a wave load = 1 KiB contiguous) or are lent, per pass, to target qubits anywhere below
// bit 28: the gate on such a bit is then a wave shuffle like any other lane-bit gate, and a wave
// load becomes 8 separate 128-byte lines.  `ins` of the pass holds these positions too.
struct LanePos { int pos[3]; };

// pos[0] < 0: plain mapping -- thread index bits fill the non-inserted address bits in order.
// thread part / block part of a tile base address; thread index t = wave:2 | lane:6
__host__ __device__ __forceinline__ uint32_t tile_base_thr(uint32_t t, const BitIns& ins, const LanePos& lp) {
  if (lp.pos[0] < 0) return (uint32_t)ins_bits((uint64_t)t, ins);
  const uint32_t u = ((t >> 6) << 3) | (t & 7u);
  return (uint32_t)ins_bits((uint64_t)u, ins) | (((t >> 3) & 1u) << lp.pos[0]) | (((t >> 4) & 1u) << lp.pos[1]) |
         (((t >> 5) & 1u) << lp.pos[2]);
}
__device__ __forceinline__ uint64_t tile_base_blk(uint64_t block, const BitIns& ins, const LanePos& lp) {
  if (lp.pos[0] < 0) return ins_bits(block * QSV_TPB, ins);
  return ins_bits(block << 5, ins);          // QSV_TPB / 64 waves x 8 low lanes per workgroup
}

// thread part of a table index: bit e <- address bit pos[e]
__device__ __forceinline__ uint32_t multi_jt(const MultiOp& op, uint64_t base) {
  uint32_t jt = 0;
  for (int e = 0; e < op.nlist; ++e)
    if (op.pos[e] >= 0) jt |= (uint32_t)((base >> op.pos[e]) & 1ull) << e;
  return jt;
}

// table-op passes (MODE 1, 2): a 2x2 gate on register bit B, one matrix per thread from the LDS table
template <int R, int B, int MODE>
__device__ __forceinline__ void multi_2x2_bit(cplx (&a)[1 << R], const MultiOp& op, uint64_t base,
                                              const cplx* __restrict__ lt) {
  static_assert(MODE >= 1, "general passes go through gen_op");
  constexpr int NP = (R > 0) ? (1 << (R - 1)) : 0;
  const cplx* mp = lt + op.tab + 4 * multi_jt(op, base);
  const cplx m00 = mp[0], m01 = mp[1], m10 = mp[2], m11 = mp[3];
  if constexpr (MODE == 2) {
    // every matrix of the pass has a real diagonal and an imaginary off-diagonal (RX-like: the
    // real-part-extraction blocks, [[c, -is], [-is, c]]): half the flops of a general 2x2
    const double c0 = m00.x, s0 = m01.y, s1 = m10.y, c1 = m11.x;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int j0 = ((p >> B) << (B + 1)) | (p & ((1 << B) - 1)), j1 = j0 | (1 << B);
      const cplx x = a[j0], y = a[j1];
      a[j0] = make_double2(fma(c0, x.x, -s0 * y.y), fma(c0, x.y, s0 * y.x));
      a[j1] = make_double2(fma(c1, y.x, -s1 * x.y), fma(c1, y.y, s1 * x.x));
    }
  } else {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int j0 = ((p >> B) << (B + 1)) | (p & ((1 << B) - 1)), j1 = j0 | (1 << B);
      const cplx x = a[j0], y = a[j1];
      a[j0] = cmad(m01, y, cmul(m00, x));
      a[j1] = cmad(m11, y, cmul(m10, x));
    }
  }
}

// A 2x2 gate whose target is a LANE bit (address bit < 6): the partner amplitude of every register
// sits in lane ^ (1 << bit) of the same wavefront, so the gate is a wave shuffle plus one complex
// multiply-add per amplitude -- no extra HBM pass and no register bit spent.  With the six lane
// bits a pass reaches R + 6 distinct targets.
template <int R, int MODE>
__device__ __forceinline__ void multi_diag(cplx (&a)[1 << R], const MultiOp& op, uint64_t base,
                                           const cplx* __restrict__ lt) {
  static_assert(MODE >= 1, "general passes go through gen_op");
  // ONE update path for every list op of a table-op pass: new = dg * own + of * partner, where a
  // diagonal is the special case of = 0.  Branching between two whole-tile updates (diagonal vs
  // lane gate) made hipcc keep both results alive: +70 VGPRs at R = 5, i.e. spills that showed up
  // as +19 % HBM traffic in the PMC counters.
  const uint32_t jt = multi_jt(op, base);
  const bool lane_op = op.type == 4;
  const int lm = lane_op ? (1 << op.bit) : 1;
  const bool up = lane_op && ((threadIdx.x >> op.bit) & 1);   // this lane holds the |1> half of the pair
  const cplx* mp = lt + op.tab + (lane_op ? 4 * jt : jt);
  const cplx dg = lane_op ? (up ? mp[3] : mp[0]) : mp[0];
  const cplx of = lane_op ? (up ? mp[2] : mp[1]) : make_double2(0.0, 0.0);
  if constexpr (MODE == 2) {
    const double c = dg.x, sn = of.y;                   // RX-like tables only (host guarantees: no diagonals)
#pragma unroll
    for (int j = 0; j < (1 << R); ++j) {
      const double ox = __shfl_xor(a[j].x, lm, 64), oy = __shfl_xor(a[j].y, lm, 64);
      a[j] = make_double2(fma(c, a[j].x, -sn * oy), fma(c, a[j].y, sn * ox));
    }
  } else {
#pragma unroll
    for (int j = 0; j < (1 << R); ++j) {
      cplx o;
      o.x = __shfl_xor(a[j].x, lm, 64);
      o.y = __shfl_xor(a[j].y, lm, 64);
      a[j] = cmad(of, o, cmul(dg, a[j]));
    }
  }
}

// ---- GENERAL passes (MODE 0): every update in place ------------------------------------------
// A general pass interprets a list of ops of eleven shapes.  Written as plain expressions, each
// shape leaves its results in fresh registers and hipcc reconciles the shapes at every merge point
// with a COPY OF THE WHOLE TILE (2^R x 2 v_mov_b64 per slot and per list op, taken or not: at
// R = 4 six tile copies per round + one per list op -- more than half of the VALU work of the
// reference's unfused stream, whose rounds hold ONE controlled X each).  Three rules keep the
// tile where it is (found on reduced kernels; hipcc then emits no copy at all):
//   1. an update is ONE inline-assembly statement, arithmetic included, whose outputs are TIED to
//      the registers the amplitudes already live in (arithmetic left outside gets sunk below the
//      statement and keeps the old value alive across it);
//   2. no if / else with an update on both sides: the shapes are tested one after the other
//      against a selector the compiler cannot see through (QSV_OPQ), so every update sits in a
//      one-sided branch;
//   3. controls on lane / block bits are a real exec-masked branch (a wave none of whose lanes
//      match skips the op), register controls a scalar test per amplitude -- an op that does not
//      fire costs its tests only.
__device__ __forceinline__ void swap_inplace(cplx& x, cplx& y) {
  double t;
  asm("v_mov_b64 %4, %0\n\tv_mov_b64 %0, %2\n\tv_mov_b64 %2, %4\n\t"
      "v_mov_b64 %4, %1\n\tv_mov_b64 %1, %3\n\tv_mov_b64 %3, %4"
      : "+v"(x.x), "+v"(x.y), "+v"(y.x), "+v"(y.y), "=&v"(t));
}
__device__ __forceinline__ void set_inplace(cplx& a, double vx, double vy) {    // a <- (vx, vy)
  asm("v_mov_b64 %0, %2\n\tv_mov_b64 %1, %3" : "+v"(a.x), "+v"(a.y) : "v"(vx), "v"(vy));
}
__device__ __forceinline__ void cmul_inplace(cplx& a, cplx d) {                 // a <- a * d
  double t;
  asm("v_mul_f64 %2, %1, %4\n\tv_mul_f64 %1, %1, %3\n\tv_fma_f64 %1, %0, %4, %1\n\tv_fma_f64 %0, %0, %3, -%2"
      : "+v"(a.x), "+v"(a.y), "=&v"(t) : "v"(d.x), "v"(d.y));
}
__device__ __forceinline__ void cmul_inplace_s(cplx& a, cplx d) {               // same, d in SGPRs (one scalar operand per instruction)
  double t;
  asm("v_mul_f64 %2, %1, %4\n\tv_mul_f64 %1, %1, %3\n\tv_fma_f64 %1, %0, %4, %1\n\tv_fma_f64 %0, %0, %3, -%2"
      : "+v"(a.x), "+v"(a.y), "=&v"(t) : "s"(d.x), "s"(d.y));
}
// (x, y) <- M (x, y): 16 multiply-adds + 3 moves.  MC = "v": matrix in VGPRs (a table entry per thread);
// MC = "s": the op's own matrix straight from SGPRs (every instruction below names exactly ONE matrix
// element, which is the one scalar operand a gfx9 VOP3 instruction may have)
#define QSV_MAT2_INPLACE(NAME, MC)                                                                                      \
__device__ __forceinline__ void NAME(cplx& x, cplx& y, cplx m00, cplx m01, cplx m10, cplx m11) {                       \
  double t0, t1, t2;                                                                                                    \
  asm("v_mul_f64 %4, %7, %0\n\tv_fma_f64 %4, -%8, %1, %4\n\tv_fma_f64 %4, %9, %2, %4\n\tv_fma_f64 %4, -%10, %3, %4\n\t"     \
      "v_mul_f64 %5, %7, %1\n\tv_fma_f64 %5, %8, %0, %5\n\tv_fma_f64 %5, %9, %3, %5\n\tv_fma_f64 %5, %10, %2, %5\n\t"       \
      "v_mul_f64 %6, %11, %0\n\tv_fma_f64 %6, -%12, %1, %6\n\tv_fma_f64 %6, %13, %2, %6\n\tv_fma_f64 %6, -%14, %3, %6\n\t"  \
      "v_mul_f64 %3, %13, %3\n\tv_fma_f64 %3, %14, %2, %3\n\tv_fma_f64 %3, %11, %1, %3\n\tv_fma_f64 %3, %12, %0, %3\n\t"    \
      "v_mov_b64 %2, %6\n\tv_mov_b64 %0, %4\n\tv_mov_b64 %1, %5"                                                      \
      : "+v"(x.x), "+v"(x.y), "+v"(y.x), "+v"(y.y), "=&v"(t0), "=&v"(t1), "=&v"(t2)                                     \
      : MC(m00.x), MC(m00.y), MC(m01.x), MC(m01.y), MC(m10.x), MC(m10.y), MC(m11.x), MC(m11.y));                        \
}
QSV_MAT2_INPLACE(mat2_inplace, "v")     // rows: t0 = x'.re, t1 = x'.im, t2 = y'.re, then y.im in place
QSV_MAT2_INPLACE(mat2_inplace_s, "s")
#undef QSV_MAT2_INPLACE
// a <- dg * a + of * (ox, oy)   (the partner amplitude arrives by wave shuffle)
__device__ __forceinline__ void lane_mix_inplace(cplx& a, double ox, double oy, cplx dg, cplx of) {
  double t;
  asm("v_mul_f64 %2, %1, %6\n\tv_mul_f64 %1, %1, %5\n\tv_fma_f64 %1, %0, %6, %1\n\tv_fma_f64 %1, %7, %4, %1\n\t"
      "v_fma_f64 %1, %8, %3, %1\n\tv_fma_f64 %0, %0, %5, -%2\n\tv_fma_f64 %0, %7, %3, %0\n\tv_fma_f64 %0, -%8, %4, %0"
      : "+v"(a.x), "+v"(a.y), "=&v"(t) : "v"(ox), "v"(oy), "v"(dg.x), "v"(dg.y), "v"(of.x), "v"(of.y));
}
__device__ __forceinline__ double shfl_at(double v, int byte_addr) {            // v of lane byte_addr / 4
  const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int R>
__device__ __forceinline__ int multi_jr(const MultiOp& op, int j) {
  int jr = 0;
#pragma unroll
  for (int c = 0; c < R; ++c) if ((j >> c) & 1) jr += op.regw[c];
  return jr;
}

#define QSV_OPQ(x) ({ int _t = (x); asm volatile("" : "+s"(_t)); _t; })
// the control part of an op, read in ONE scalar load batch at the top of gen_op (it used to be three dependent
// ones: shape / bit, then the masks inside the shape's branch, then the fire mask behind the exec test)
struct GenCtl { uint64_t tmask, tval; uint32_t flo, fhi; int bit; };   // re-hide a scalar before every test (rule 2)
// does amplitude j fire?  32-bit halves of the host-made mask: one s_bitcmp1_b32 + branch per test, firing path in line
#define QSV_FIRES(flo, fhi, j) __builtin_expect(((((j) < 32 ? (flo) : (fhi)) >> ((j) & 31)) & 1u) != 0u, 1)
#define QSV_PAIR(p, B) const int j0 = (((p) >> (B)) << ((B) + 1)) | ((p) & ((1 << (B)) - 1)), j1 = j0 | (1 << (B))

// gate of a general pass on register bit B
template <int R, int B>
__device__ __forceinline__ void gen_gate(cplx (&a)[1 << R], const MultiOp& op, int shape, const GenCtl& ctl, uint64_t base,
                                         const cplx* __restrict__ lt) {
  constexpr int NP = 1 << (R - 1);
  if (QSV_OPQ(shape) < GS_X) {
    if (QSV_OPQ(shape) == GS_TAB_T) {
      const cplx* mp = lt + op.tab + 4 * multi_jt(op, base);
      const cplx m00 = mp[0], m01 = mp[1], m10 = mp[2], m11 = mp[3];
#pragma unroll
      for (int p = 0; p < NP; ++p) { QSV_PAIR(p, B); mat2_inplace(a[j0], a[j1], m00, m01, m10, m11); }
    }
    if (QSV_OPQ(shape) == GS_TAB_P) {
      const uint32_t jt = multi_jt(op, base);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        QSV_PAIR(p, B);
        const cplx* mp = lt + op.tab + 4 * (jt + multi_jr<R>(op, j0));
        mat2_inplace(a[j0], a[j1], mp[0], mp[1], mp[2], mp[3]);
      }
    }
  }
  if (QSV_OPQ(shape) >= GS_X) {
    if ((base & ctl.tmask) == ctl.tval) {                // exec-masked: a wave without a matching lane skips the op
      const uint32_t flo = ctl.flo, fhi = ctl.fhi;
      if (QSV_OPQ(shape) == GS_X) {
#pragma unroll
        for (int p = 0; p < NP; ++p) { QSV_PAIR(p, B); if (QSV_FIRES(flo, fhi, j0)) swap_inplace(a[j0], a[j1]); }
      }
      if (QSV_OPQ(shape) == GS_MAT) {
        const cplx m00 = make_double2(op.m[0], op.m[1]), m01 = make_double2(op.m[2], op.m[3]);
        const cplx m10 = make_double2(op.m[4], op.m[5]), m11 = make_double2(op.m[6], op.m[7]);
#pragma unroll
        for (int p = 0; p < NP; ++p) { QSV_PAIR(p, B); if (QSV_FIRES(flo, fhi, j0)) mat2_inplace_s(a[j0], a[j1], m00, m01, m10, m11); }
      }
    }
  }
}

// op of a general pass without a register target
template <int R>
__device__ __forceinline__ void gen_list(cplx (&a)[1 << R], const MultiOp& op, int shape, const GenCtl& ctl, uint64_t base,
                                         const cplx* __restrict__ lt) {
  if (QSV_OPQ(shape) < GS_LTAB_T) {
    if (QSV_OPQ(shape) == GS_DIAG_T) {
      const cplx d = lt[op.tab + multi_jt(op, base)];
#pragma unroll
      for (int j = 0; j < (1 << R); ++j) cmul_inplace(a[j], d);
    }
    if (QSV_OPQ(shape) == GS_DIAG_A) {
      const uint32_t jt = multi_jt(op, base);
#pragma unroll
      for (int j = 0; j < (1 << R); ++j) cmul_inplace(a[j], lt[op.tab + jt + multi_jr<R>(op, j)]);
    }
    if (QSV_OPQ(shape) == GS_PHASE) {
      if ((base & ctl.tmask) == ctl.tval) {
        const cplx ph = make_double2(op.m[0], op.m[1]);
        const uint32_t flo = ctl.flo, fhi = ctl.fhi;
#pragma unroll
        for (int j = 0; j < (1 << R); ++j)
          if (QSV_FIRES(flo, fhi, j)) cmul_inplace_s(a[j], ph);
      }
    }
  }
  if (QSV_OPQ(shape) >= GS_LTAB_T) {
    // the partner amplitude sits in lane ^ (1 << bit): ds_bpermute with the byte address made ONCE per op
    // (__shfl_xor recomputes its six-instruction address for every value it moves)
    const int lane = threadIdx.x & 63;
    const bool up = (lane >> op.bit) & 1;                // this lane holds the |1> half of the pair
    const int partner = (lane ^ (1 << op.bit)) << 2;
    if (QSV_OPQ(shape) == GS_LTAB_T) {
      const cplx* mp = lt + op.tab + 4 * multi_jt(op, base);
      const cplx dg = up ? mp[3] : mp[0], of = up ? mp[2] : mp[1];
#pragma unroll
      for (int j = 0; j < (1 << R); ++j) lane_mix_inplace(a[j], shfl_at(a[j].x, partner), shfl_at(a[j].y, partner), dg, of);
    }
    if (QSV_OPQ(shape) == GS_LTAB_A) {
      const uint32_t jt = multi_jt(op, base);
#pragma unroll
      for (int j = 0; j < (1 << R); ++j) {
        const cplx* mp = lt + op.tab + 4 * (jt + multi_jr<R>(op, j));
        const cplx dg = up ? mp[3] : mp[0], of = up ? mp[2] : mp[1];
        lane_mix_inplace(a[j], shfl_at(a[j].x, partner), shfl_at(a[j].y, partner), dg, of);
      }
    }
    if (QSV_OPQ(shape) >= GS_LX) {
      // tmask never contains the target bit, so a lane and its partner fire together.  No exec-masked
      // branch here (the shuffles need every lane on): a lane whose controls do not match reads ITSELF
      // (X) or mixes with the identity (matrix); a wave without any matching lane skips the op.
      const bool ct = (base & ctl.tmask) == ctl.tval;
      if (QSV_OPQ(__builtin_amdgcn_readfirstlane((int)(__builtin_amdgcn_ballot_w64(ct) != 0)))) {
        const uint32_t flo = ctl.flo, fhi = ctl.fhi;
        // no control on a register bit (the usual case): one block without tests, so the 2^R x 4
        // shuffles are in flight together instead of one amplitude's at a time
        const int all = __builtin_amdgcn_readfirstlane(op.uniform);
        const int from = ct ? partner : (lane << 2);
        const cplx dg = !ct ? make_double2(1.0, 0.0) : up ? make_double2(op.m[6], op.m[7]) : make_double2(op.m[0], op.m[1]);
        const cplx of = !ct ? make_double2(0.0, 0.0) : up ? make_double2(op.m[4], op.m[5]) : make_double2(op.m[2], op.m[3]);
        if (QSV_OPQ(all)) {
          if (QSV_OPQ(shape) == GS_LX) {
#pragma unroll
            for (int j = 0; j < (1 << R); ++j) set_inplace(a[j], shfl_at(a[j].x, from), shfl_at(a[j].y, from));
          }
          if (QSV_OPQ(shape) == GS_LMAT) {
#pragma unroll
            for (int j = 0; j < (1 << R); ++j) lane_mix_inplace(a[j], shfl_at(a[j].x, partner), shfl_at(a[j].y, partner), dg, of);
          }
        }
        if (!QSV_OPQ(all)) {
          if (QSV_OPQ(shape) == GS_LX) {
#pragma unroll
            for (int j = 0; j < (1 << R); ++j)
              if (QSV_FIRES(flo, fhi, j)) set_inplace(a[j], shfl_at(a[j].x, from), shfl_at(a[j].y, from));
          }
          if (QSV_OPQ(shape) == GS_LMAT) {
#pragma unroll
            for (int j = 0; j < (1 << R); ++j)
              if (QSV_FIRES(flo, fhi, j)) lane_mix_inplace(a[j], shfl_at(a[j].x, partner), shfl_at(a[j].y, partner), dg, of);
          }
        }
      }
    }
  }
}

// a wave-uniform 64-bit value, in scalar registers (readfirstlane returns a SIGNED int: widen through uint32_t, or a low
// half with bit 31 set smears ones over the high half)
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((unsigned long long)hi << 32) | (unsigned long long)lo;
}

// One op of a general pass.  The ops come as a flat list in program order: no rounds, no empty slots.
template <int R>
__device__ __forceinline__ void gen_op(cplx (&a)[1 << R], const MultiOp& op, uint64_t base, const cplx* __restrict__ lt) {
  const int shape = __builtin_amdgcn_readfirstlane(op.shape);
  GenCtl ctl;
  {
    uint32_t tml = (uint32_t)op.tmask, tmh = (uint32_t)(op.tmask >> 32), tvl = (uint32_t)op.tval, tvh = (uint32_t)(op.tval >> 32);
    uint32_t fl = (uint32_t)op.rfire, fh = (uint32_t)(op.rfire >> 32);
    int bit0 = op.bit;
    asm volatile("" : "+s"(tml), "+s"(tmh), "+s"(tvl), "+s"(tvh), "+s"(fl), "+s"(fh), "+s"(bit0));   // here, not where first used
    ctl.bit = bit0;
    ctl.tmask = ((uint64_t)tmh << 32) | tml; ctl.tval = ((uint64_t)tvh << 32) | tvl; ctl.flo = fl; ctl.fhi = fh;
  }
  if (QSV_OPQ(shape) < GS_DIAG_T) {
    if constexpr (R > 0) {
      const int b = ctl.bit;
      if (QSV_OPQ(b) < 2) {
        if (QSV_OPQ(b) == 0) gen_gate<R, 0>(a, op, shape, ctl, base, lt);
        if constexpr (R > 1) if (QSV_OPQ(b) == 1) gen_gate<R, 1>(a, op, shape, ctl, base, lt);
      }
      if constexpr (R > 2) {
        if (QSV_OPQ(b) >= 2) {
          if (QSV_OPQ(b) == 2) gen_gate<R, 2>(a, op, shape, ctl, base, lt);
          if constexpr (R > 3) if (QSV_OPQ(b) == 3) gen_gate<R, 3>(a, op, shape, ctl, base, lt);
          if constexpr (R > 4) if (QSV_OPQ(b) == 4) gen_gate<R, 4>(a, op, shape, ctl, base, lt);
          if constexpr (R > 5) if (QSV_OPQ(b) == 5) gen_gate<R, 5>(a, op, shape, ctl, base, lt);
        }
      }
    }
  }
  if (QSV_OPQ(shape) >= GS_DIAG_T) gen_list<R>(a, op, shape, ctl, base, lt);
}

// Schedule of a TABLE-OP pass (MODE 1, 2): the host lays the gates out in ROUNDS of 1 + R slots.
// Slot 0 of a round is a LIST of ops without a register target (diagonals, lane-bit gates), slot
// 1 + b holds one 2x2 table gate on register bit b (an identity table where the circuit has none);
// a round runs its list, then bits 0..R-1.  The kernel body is straight-line over b inside one
// runtime loop over rounds -- no branch on the target bit or around an update, so plain expressions
// already update every amplitude in place.
struct MultiSlot { int first; int ndiag; int has; int pad; };   // list slot: ops[first .. first+ndiag); gate slot: ops[first] if has
// combo table of a general pass, as ints in the `slots` buffer: [0] number of combo bits nb (0: none, walk the flat list),
// [1..8] their address positions, then (HDR ints in, 8-byte aligned) 2^nb entries of QSV_COMBO_WORDS 64-bit masks over the
// flat op list: bit i set <=> op i can fire in the workgroups whose combo bits spell this entry's index
#define QSV_COMBO_HDR 12
#define QSV_COMBO_MAXBITS 8
#define QSV_COMBO_WORDS 2

template <int R, int B, int MODE>
__device__ __forceinline__ void multi_slot(cplx (&a)[1 << R], const MultiOp* __restrict__ ops,
                                           const MultiSlot* __restrict__ rs, uint64_t base,
                                           const cplx* __restrict__ lt) {
  if constexpr (R > 0) {
    const MultiSlot sl = rs[1 + B];
    // SIMPLE passes apply a gate in EVERY slot (the host fills gaps with an identity table):
    // with no branch around the update, hipcc updates the tile in place instead of keeping an
    // old and a new copy alive across the merge.
    multi_2x2_bit<R, B, MODE>(a, ops[sl.first], base, lt);
  }
}

// INIT: do not read the shard; start from the uniform-superposition product state instead
// (amp = val where (index & nonmask) == 0): the init write and the first gate pass become one.
// SIMPLE: every op of the pass is a table op whose select bits are all lane/block bits (the
// shape of a fused QCMRF circuit): the general paths are compiled out.
// NT: non-temporal loads AND stores of the amplitudes (measured on MI355X, profiles/r02_nt_variants.log:
// either alone gains 1-2 %, both together 9 % on a 4 GiB shard -- 5.93 -> 6.50 TB/s -- and 3 % on a
// 256 GiB one; a shard that fits the caches keeps the plain form)
template <int R, bool INIT, int MODE, bool NT>
__global__ __launch_bounds__(QSV_TPB, (R == 5 && MODE ? 2 : 1)) void k_multi(cplx* __restrict__ amp, uint64_t nthreads,
                                                   BitIns ins, RegPos rp, LanePos lp,
                                                   const MultiOp* __restrict__ ops,
                                                   const MultiSlot* __restrict__ slots, int nrounds,
                                                   const cplx* __restrict__ tables, int ntab,
                                                   uint64_t nonmask, double initval,
                                                   unsigned int zreg, double* __restrict__ tile_sums, uint64_t xmask) {
  // xmask (X frame): uncontrolled X gates of the pass are not executed as data movement at all --
  // the host conjugates every later op of the pass by them and the pass STORES each amplitude at
  // (its address XOR xmask): a wave store stays one contiguous run, the permutation is free.
  // The host passes only bits for which that is race-free: register and LANE bits of a read+write
  // pass (the wave that stores an address is the wave that loaded it, after its own loads), any bit
  // of a write-only INIT pass.  An X on a bit that selects another wave / workgroup stays pending on
  // the host (it conjugates the ops of the following passes too) until a pass can take it.
  // zreg (zero tracking): register bits whose qubit is still known to be |0> on entry -- every
  // amplitude with such a bit set is zero by construction and is not read (memory there may be
  // unwritten).  `ins` then also holds the known-zero NON-register bits, so only the populated
  // subspace is enumerated at all.
  extern __shared__ double4 lds_raw[];
  cplx* lt = reinterpret_cast<cplx*>(lds_raw);
  // address = (uniform 64-bit pointer: shard + block part + register offset) + 32-bit lane part
  const uint64_t base_blk = tile_base_blk(blockIdx.x, ins, lp);      // wave-uniform
  const uint32_t base_thr = tile_base_thr(threadIdx.x, ins, lp);
  const uint64_t base = base_blk | base_thr;
  // The gate tables go to LDS -- unless this is an init pass and the whole workgroup starts from zeros (see
  // `live` below: all but one workgroup in 2^(ancillas outside the tile) of a QCMRF circuit): it then runs
  // no op, needs no table, and is a plain fill.
  const bool need_tables = !INIT || __syncthreads_or((int)((base & nonmask) == 0));
  if (need_tables)
    for (int i = threadIdx.x; i < ntab; i += QSV_TPB) lt[i] = tables[i];
  __syncthreads();
  const uint64_t gb = (uint64_t)blockIdx.x * QSV_TPB;
  if (gb + threadIdx.x >= nthreads) return;
  cplx* __restrict__ pblk = amp + base_blk;
  // register-bit offsets once, in SGPRs (otherwise every one of the 2^R loads re-reads its
  // positions from the kernel arguments and waits for them)
  uint64_t ob[R > 0 ? R : 1];
#pragma unroll
  for (int c = 0; c < R; ++c) {
    ob[c] = 1ull << rp.pos[c];
    if constexpr (R <= 4) asm volatile("" : "+s"(ob[c]));   // pinning costs registers R = 5, 6 do not have
  }
  cplx a[1 << R];
  int list0_done = 0;
  bool live = true;               // INIT: does this wavefront hold a nonzero amplitude at all?
  if (INIT) {
    // Every register bit of a fused QCMRF pass is a fresh |0> target: the tile starts as ONE
    // nonzero amplitude per lane (j = 0).  The first round's list ops (lane gates, diagonals) map
    // zeros to zeros, so they are applied to that scalar BEFORE the tile exists -- 4 shuffles per
    // lane gate instead of 128, and nothing for a write-only pass to hide (three lane gates cost
    // 12 ms of a 55 ms pass at 34 qubits otherwise).  No tile register is live across this branch.
    cplx f = make_double2(((base & nonmask) == 0) ? initval : 0.0, 0.0);
    if constexpr (MODE == 2) {
      bool all_dead = R > 0;
#pragma unroll
      for (int c = 0; c < R; ++c) all_dead = all_dead && (ob[c] & nonmask);
      if (all_dead && nrounds > 0 && need_tables) {           // (a workgroup that staged no tables holds zeros only: nothing to apply)
        const MultiSlot sl = slots[0];
        for (int d = 0; d < sl.ndiag; ++d) {
          const MultiOp& op = ops[sl.first + d];
          const uint32_t jt = multi_jt(op, base);
          const bool up = (threadIdx.x >> op.bit) & 1;
          const cplx* mp = lt + op.tab + 4 * jt;
          const double c = up ? mp[3].x : mp[0].x, sn = up ? mp[2].y : mp[1].y;
          const double ox = __shfl_xor(f.x, 1 << op.bit, 64), oy = __shfl_xor(f.y, 1 << op.bit, 64);
          f = make_double2(fma(c, f.x, -sn * oy), fma(c, f.y, sn * ox));
        }
        list0_done = 1;
      }
    }
#pragma unroll
    for (int j = 0; j < (1 << R); ++j) {
      uint64_t off = 0;
#pragma unroll
      for (int c = 0; c < R; ++c) if ((j >> c) & 1) off |= ob[c];
      a[j] = ((off & nonmask) == 0) ? f : make_double2(0.0, 0.0);
    }
    // Most wavefronts of an init pass start from NOTHING: an amplitude is nonzero only where every qubit
    // outside the uniform-superposition mask reads 0 (all ancillas: one address in 2^m), and gates map
    // zero tiles to zero tiles.  Such a wave skips the ops and just writes its zeros, so a write-only pass
    // has (almost) no arithmetic left to hide.
    live = __builtin_amdgcn_ballot_w64(f.x != 0.0 || f.y != 0.0) != 0;
  } else if (zreg == 0) {                       // the common case: no branch between the loads
#pragma unroll
    for (int j = 0; j < (1 << R); ++j) {
      uint64_t off = 0;
#pragma unroll
      for (int c = 0; c < R; ++c) if ((j >> c) & 1) off |= ob[c];
      a[j] = NT ? ld_nt((pblk + off) + base_thr) : (pblk + off)[base_thr];
    }
  } else {
#pragma unroll
    for (int j = 0; j < (1 << R); ++j) {
      uint64_t off = 0;
#pragma unroll
      for (int c = 0; c < R; ++c) if ((j >> c) & 1) off |= ob[c];
      if ((unsigned)j & zreg) a[j] = make_double2(0.0, 0.0);
      else a[j] = NT ? ld_nt((pblk + off) + base_thr) : (pblk + off)[base_thr];
    }
  }
  if constexpr (MODE == 0) {
    // general pass: `ops` is a flat list in program order, `nrounds` its length.  `slots` carries the host's
    // COMBO TABLE (GenCombos): the controls that sit on workgroup-uniform address bits are resolved per workgroup --
    // for every value of (up to 8 of) those bits the host lists the ops that can fire at all, in program order, and a
    // workgroup walks only its own list.  A CCX of the reference's stream whose two controls select workgroups is then
    // neither loaded nor tested by the three quarters of the machine it does not concern.
    const int* ch = reinterpret_cast<const int*>(slots);
    const int nb = __builtin_amdgcn_readfirstlane(ch[0]);
    uint32_t ci = 0;
    for (int e = 0; e < nb; ++e) ci |= (uint32_t)((base_blk >> ch[1 + e]) & 1ull) << e;
    ci = __builtin_amdgcn_readfirstlane(ci);
    // this workgroup's ops as a bit mask over the flat list (QSV_COMBO_WORDS x 64 ops; nb = 0: one entry, every op): the
    // next op's index comes out of a register, so its descriptor load does not wait for an index load first.  ONE loop:
    // the interpreter body (gen_op) exists once in the kernel.
    const unsigned long long* mk = reinterpret_cast<const unsigned long long*>(ch + QSV_COMBO_HDR) + (size_t)ci * QSV_COMBO_WORDS;
    static_assert(QSV_COMBO_WORDS == 2, "two mask words are spelled out below");
    const unsigned long long m0 = uniform_u64(mk[0]), m1 = uniform_u64(mk[1]);
    // the plain loop over the flat list (the shape the in-place interpreter was tuned in: anything cleverer -- iterating the
    // set bits, a second copy of the loop -- cost more than it saved); an op this workgroup cannot fire costs one scalar
    // bit test, its descriptor is never loaded
    const int nops = (!INIT || live) ? nrounds : 0;
    for (int i = 0; i < nops; ++i) {
      const unsigned long long mw = i < 64 ? m0 : m1;
      if (!((mw >> (i & 63)) & 1ull)) continue;
      gen_op<R>(a, ops[i], base, lt);
    }
  } else {
    constexpr int NS = R + 1;
    const int nr = (!INIT || live) ? nrounds : 0;
    for (int r = 0; r < nr; ++r) {
      const MultiSlot* rs = slots + r * NS;
      {
        const MultiSlot sl = rs[0];
        const int nd = (r == 0 && list0_done) ? 0 : sl.ndiag;     // round 0's list already went into the scalar
        for (int d = 0; d < nd; ++d) multi_diag<R, MODE>(a, ops[sl.first + d], base, lt);
      }
      multi_slot<R, 0, MODE>(a, ops, rs, base, lt);
      if constexpr (R > 1) multi_slot<R, 1, MODE>(a, ops, rs, base, lt);
      if constexpr (R > 2) multi_slot<R, 2, MODE>(a, ops, rs, base, lt);
      if constexpr (R > 3) multi_slot<R, 3, MODE>(a, ops, rs, base, lt);
      if constexpr (R > 4) multi_slot<R, 4, MODE>(a, ops, rs, base, lt);
      if constexpr (R > 5) multi_slot<R, 5, MODE>(a, ops, rs, base, lt);
    }
  }
  // store side of the X frame: register, lane and block part of the mask
  uint64_t regbits = 0;
#pragma unroll
  for (int c = 0; c < R; ++c) regbits |= ob[c];
  const uint64_t xreg = xmask & regbits, xrest = xmask & ~regbits;
  const uint32_t thrbits = tile_base_thr(QSV_TPB - 1, ins, lp);
  cplx* __restrict__ pst = amp + (base_blk ^ (xrest & ~(uint64_t)thrbits));
  const uint32_t thr_st = base_thr ^ (uint32_t)(xrest & thrbits);
  double psum = 0.0;
#pragma unroll
  for (int j = 0; j < (1 << R); ++j) {
    uint64_t off = 0;
#pragma unroll
    for (int c = 0; c < R; ++c) if ((j >> c) & 1) off |= ob[c];
    if (NT) st_nt((pst + (off ^ xreg)) + thr_st, a[j]);
    else (pst + (off ^ xreg))[thr_st] = a[j];
    psum = fma(a[j].x, a[j].x, fma(a[j].y, a[j].y, psum));
  }
  // last pass of a program: leave sum |amp|^2 of this workgroup's tile behind, so that measurement
  // needs no separate read pass over the shard (workgroups are full: nthreads % 256 == 0 is
  // checked on the host before tile_sums is passed)
  if (tile_sums) {
    __shared__ double wpart[QSV_TPB / 64];
    psum = wave_sum(psum);
    if ((threadIdx.x & 63) == 0) wpart[threadIdx.x >> 6] = psum;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = (wpart[0] + wpart[1]) + (wpart[2] + wpart[3]);
  }
}

