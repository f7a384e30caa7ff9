// qsv_common.h -- types and device helpers shared by every kernel of the fp64 statevector engine (gfx950, wave64).
//
// Every gate is a sweep over a shard of 2^L complex128 amplitudes resident in HBM; the
// kernels are bandwidth kernels (0.44 flop/B for a 2x2) and are written for coalesced
// 16-byte-per-lane (global_load_dwordx4) streams: one wave instruction = 1 KiB.
//
// Roofline per kernel (algorithmic bytes):
//   k_pair / k_lowt / k_mux / k_diag / k_kq : 32 B per amplitude touched  (HBM bound)
//   k_mcphase                               : 32 B per amplitude of the controlled subspace
//   k_init                                  : 16 B per amplitude (write only)
//   k_blocksum                              : 16 B per amplitude (read only)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define QSV_TPB 256
#define QSV_MAXB 28          // >= QSV_MAX_CTRL + 1 insert positions; marginals gather up to 26 bits

typedef double2 cplx;

struct BitIns {              // sorted ascending bit positions at which a zero bit is inserted
  int n;
  int pos[QSV_MAXB];
};
struct BitList {             // gather list: result bit b <- index bit pos[b]
  int n;
  int pos[QSV_MAXB];
};
struct Mat2 { double v[8]; };  // row-major {re,im}: m00 m01 m10 m11

__host__ __device__ __forceinline__ uint64_t ins_bits(uint64_t x, const BitIns& b) {
  for (int j = 0; j < b.n; ++j) {
    const int p = b.pos[j];
    const uint64_t lo = x & ((1ull << p) - 1ull);
    x = ((x >> p) << (p + 1)) | lo;
  }
  return x;
}
// Which amplitudes one wave instruction touches is a free choice in a sweep kernel (any bijection between
// work items and amplitudes will do), and it matters: a wavefront access made of two 512-byte runs
// 32 KiB apart -- lane bit 5 carrying address bit 11 instead of bit 5 -- streams 8-13 % faster on MI355X
// than one contiguous 1 KiB (profiles/r02_lane_map_high_target.log; no other bit does it).  swz_5_11
// exchanges the two bits of an amplitude index; a kernel applies it when neither is a bit the gate
// singles out (target, control, select), so a pair stays a pair.
__host__ __device__ __forceinline__ uint64_t swz_5_11(uint64_t i) {
  const uint64_t d = ((i >> 5) ^ (i >> 11)) & 1ull;
  return i ^ (d << 5) ^ (d << 11);
}
// ... and with the partner of bit 11 chosen by the host: the address bit lane bit 5 of an enumeration lands on (bit 5
// itself unless the gate's controls / target take it: then the next free one)
__host__ __device__ __forceinline__ uint64_t swz_a_11(uint64_t i, int a) {
  const uint64_t d = ((i >> a) ^ (i >> 11)) & 1ull;
  return i ^ (d << a) ^ (d << 11);
}
__device__ __forceinline__ uint32_t gather_bits(uint64_t x, const BitList& b) {
  uint32_t j = 0;
  for (int k = 0; k < b.n; ++k) j |= (uint32_t)((x >> b.pos[k]) & 1ull) << k;
  return j;
}
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cplx cmad(cplx a, cplx b, cplx c) {   // a*b + c
  return make_double2(fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y)));
}
__device__ __forceinline__ cplx ld(const cplx* p) { return *p; }
__device__ __forceinline__ void st(cplx* p, cplx v) { *p = v; }
__device__ __forceinline__ cplx ld_nt(const cplx* p) {
  return make_double2(__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y));
}
__device__ __forceinline__ void st_nt(cplx* p, cplx v) {
  __builtin_nontemporal_store(v.x, &p->x);
  __builtin_nontemporal_store(v.y, &p->y);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
