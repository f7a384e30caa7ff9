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
