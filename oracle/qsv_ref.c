/*
 * qsv_ref.c -- CPU restatement of the statevector arithmetic behind
 * /root/reference/run_experiment.py:54-57 (Qiskit Aer's CPU statevector: third party, not
 * vendored, version unpinned ~0.13.x).  TEST INFRASTRUCTURE ONLY: used by tests/ as a second
 * checker at sizes numpy is too slow for, and by bench.py's cpu_baseline leg ("port").
 * Never linked into, loaded by, or called from the shipped package.
 *
 * Published algorithm restated: 2^n complex128 amplitudes, qubit q = bit q of the index; a
 * gate is one OpenMP-parallel sweep over the amplitude pairs / subspace it touches
 * (Aer: QubitVector::apply_matrix / apply_mcx / apply_mcphase / apply_diagonal_matrix).
 * Gate semantics: Qiskit circuit library (h, x, mcx, cp ...) as used by QCMRF.py:205-236.
 *
 * Parity pin: see oracle/__init__.py ("parity unpinned" at 1e-10 by the reference itself;
 * pinned statistically by its committed Aer counts and exactly by algebra).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double re, im; } c128;

static inline uint64_t insert_zero(uint64_t x, int p) {
  const uint64_t lo = x & ((1ull << p) - 1ull);
  return ((x >> p) << (p + 1)) | lo;
}

static void sort_int(int* a, int n) {
  for (int i = 1; i < n; ++i) {
    int v = a[i], j = i - 1;
    while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; --j; }
    a[j + 1] = v;
  }
}

void ref_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int ref_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void ref_init_zero(c128* s, int nq) {
  const uint64_t n = 1ull << nq;
#pragma omp parallel for schedule(static)
  for (uint64_t i = 0; i < n; ++i) { s[i].re = 0.0; s[i].im = 0.0; }
  s[0].re = 1.0;
}

/* (multi-)controlled 2x2 on qubit t; m = row-major {re,im} x 4; is_x: pure swap */
void ref_apply_1q(c128* s, int nq, int t, const double* m, int n_ctrl, const int* ctrls,
                  const int* vals, int is_x) {
  int pos[64];
  uint64_t fixed = 0;
  for (int i = 0; i < n_ctrl; ++i) {
    pos[i] = ctrls[i];
    if (!vals || vals[i]) fixed |= 1ull << ctrls[i];
  }
  pos[n_ctrl] = t;
  sort_int(pos, n_ctrl + 1);
  const uint64_t npairs = 1ull << (nq - 1 - n_ctrl);
  const uint64_t tbit = 1ull << t;
  const c128 m00 = {m ? m[0] : 0, m ? m[1] : 0}, m01 = {m ? m[2] : 1, m ? m[3] : 0};
  const c128 m10 = {m ? m[4] : 1, m ? m[5] : 0}, m11 = {m ? m[6] : 0, m ? m[7] : 0};
#pragma omp parallel for schedule(static)
  for (uint64_t p = 0; p < npairs; ++p) {
    uint64_t i0 = p;
    for (int j = 0; j <= n_ctrl; ++j) i0 = insert_zero(i0, pos[j]);
    i0 |= fixed;
    const uint64_t i1 = i0 | tbit;
    const c128 a = s[i0], b = s[i1];
    if (is_x) { s[i0] = b; s[i1] = a; }
    else {
      s[i0].re = m00.re * a.re - m00.im * a.im + m01.re * b.re - m01.im * b.im;
      s[i0].im = m00.re * a.im + m00.im * a.re + m01.re * b.im + m01.im * b.re;
      s[i1].re = m10.re * a.re - m10.im * a.im + m11.re * b.re - m11.im * b.im;
      s[i1].im = m10.re * a.im + m10.im * a.re + m11.re * b.im + m11.im * b.re;
    }
  }
}

/* e^{i angle} where every listed qubit matches its value */
void ref_apply_mcphase(c128* s, int nq, int n, const int* qubits, const int* vals, double angle) {
  int pos[64];
  uint64_t fixed = 0;
  for (int i = 0; i < n; ++i) {
    pos[i] = qubits[i];
    if (!vals || vals[i]) fixed |= 1ull << qubits[i];
  }
  sort_int(pos, n);
  const uint64_t nsub = 1ull << (nq - n);
  const double c = cos(angle), sn = sin(angle);
#pragma omp parallel for schedule(static)
  for (uint64_t p = 0; p < nsub; ++p) {
    uint64_t i = p;
    for (int j = 0; j < n; ++j) i = insert_zero(i, pos[j]);
    i |= fixed;
    const c128 a = s[i];
    s[i].re = a.re * c - a.im * sn;
    s[i].im = a.re * sn + a.im * c;
  }
}

/* table[j], j = sum_b bit(qubits[b]) << b */
void ref_apply_diag(c128* s, int nq, int k, const int* qubits, const double* table) {
  const uint64_t n = 1ull << nq;
#pragma omp parallel for schedule(static)
  for (uint64_t i = 0; i < n; ++i) {
    uint32_t j = 0;
    for (int b = 0; b < k; ++b) j |= (uint32_t)((i >> qubits[b]) & 1ull) << b;
    const double dr = table[2 * j], di = table[2 * j + 1];
    const c128 a = s[i];
    s[i].re = a.re * dr - a.im * di;
    s[i].im = a.re * di + a.im * dr;
  }
}

/* uniformly controlled 2x2 on t; mats[j] (8 doubles), j from ctrls, ctrls[0] = LSB */
void ref_apply_mux(c128* s, int nq, int k, const int* ctrls, int t, const double* mats) {
  const uint64_t npairs = 1ull << (nq - 1);
  const uint64_t tbit = 1ull << t;
#pragma omp parallel for schedule(static)
  for (uint64_t p = 0; p < npairs; ++p) {
    const uint64_t i0 = insert_zero(p, t), i1 = i0 | tbit;
    uint32_t j = 0;
    for (int b = 0; b < k; ++b) j |= (uint32_t)((i0 >> ctrls[b]) & 1ull) << b;
    const double* m = mats + 8 * (size_t)j;
    const c128 a = s[i0], b2 = s[i1];
    s[i0].re = m[0] * a.re - m[1] * a.im + m[2] * b2.re - m[3] * b2.im;
    s[i0].im = m[0] * a.im + m[1] * a.re + m[2] * b2.im + m[3] * b2.re;
    s[i1].re = m[4] * a.re - m[5] * a.im + m[6] * b2.re - m[7] * b2.im;
    s[i1].im = m[4] * a.im + m[5] * a.re + m[6] * b2.im + m[7] * b2.re;
  }
}

double ref_norm(const c128* s, int nq) {
  const uint64_t n = 1ull << nq;
  double acc = 0.0;
#pragma omp parallel for reduction(+ : acc) schedule(static)
  for (uint64_t i = 0; i < n; ++i) acc += s[i].re * s[i].re + s[i].im * s[i].im;
  return acc;
}

/* out[j] += |amp|^2 over indices with (i & fmask) == fval; j gathered from `qubits` */
void ref_marginal(const c128* s, int nq, int k, const int* qubits, uint64_t fmask, uint64_t fval,
                  double* out) {
  const uint64_t n = 1ull << nq;
  for (uint64_t i = 0; i < n; ++i) {
    if ((i & fmask) != fval) continue;
    uint32_t j = 0;
    for (int b = 0; b < k; ++b) j |= (uint32_t)((i >> qubits[b]) & 1ull) << b;
    out[j] += s[i].re * s[i].re + s[i].im * s[i].im;
  }
}

/* cumulative-sum sampler (Aer: sample_measure): shots sorted uniforms walked against |amp|^2 */
void ref_sample(const c128* s, int nq, uint64_t shots, const double* sorted_u, uint64_t* out) {
  const uint64_t n = 1ull << nq;
  double run = 0.0;
  uint64_t k = 0, last_nz = 0;
  for (uint64_t i = 0; i < n && k < shots; ++i) {
    const double p = s[i].re * s[i].re + s[i].im * s[i].im;
    if (p > 0) last_nz = i;
    run += p;
    while (k < shots && sorted_u[k] < run) out[k++] = i;
  }
  while (k < shots) out[k++] = last_nz;
}
