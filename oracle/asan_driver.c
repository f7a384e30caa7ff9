/* TEST INFRASTRUCTURE: drives every entry point of oracle/qsv_ref.c on small random states under
 * AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; `make -C oracle asan`).  Out-of-bounds
 * indexing, misaligned or overflowing bit arithmetic in the oracle's sweeps would be reported here;
 * the GPU pool has no sanitizer, so this is the one the oracle gets. */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef struct { double re, im; } c128;
void ref_set_threads(int n);
int ref_num_threads(void);
void ref_init_zero(c128* s, int nq);
void ref_apply_1q(c128* s, int nq, int t, const double* m, int n_ctrl, const int* ctrls, const int* vals, int is_x);
void ref_apply_mcphase(c128* s, int nq, int n, const int* qubits, const int* vals, double angle);
void ref_apply_diag(c128* s, int nq, int k, const int* qubits, const double* table);
void ref_apply_mux(c128* s, int nq, int k, const int* ctrls, int t, const double* mats);
double ref_norm(const c128* s, int nq);
void ref_marginal(const c128* s, int nq, int k, const int* qubits, uint64_t fmask, uint64_t fval, double* out);
void ref_sample(const c128* s, int nq, uint64_t shots, const double* sorted_u, uint64_t* out);

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static double unif(void) { return (double)(rnd() >> 11) * (1.0 / 9007199254740992.0); }

/* k distinct qubits out of nq */
static void pick(int nq, int k, int* out) {
  int used[64] = {0};
  for (int i = 0; i < k; ++i) {
    int q;
    do q = (int)(rnd() % (uint64_t)nq); while (used[q]);
    used[q] = 1;
    out[i] = q;
  }
}
static void rot(double a, double* m) { /* RX-like unitary, row-major {re,im} x 4 */
  m[0] = cos(a); m[1] = 0; m[2] = 0; m[3] = -sin(a); m[4] = 0; m[5] = -sin(a); m[6] = cos(a); m[7] = 0;
}

int main(void) {
  ref_set_threads(2);
  for (int nq = 1; nq <= 12; ++nq) {
    const uint64_t n = 1ull << nq;
    c128* s = malloc(n * sizeof *s);          /* exactly 2^nq: any overrun is ASan's */
    ref_init_zero(s, nq);
    double h[8] = {M_SQRT1_2, 0, M_SQRT1_2, 0, M_SQRT1_2, 0, -M_SQRT1_2, 0};
    for (int q = 0; q < nq; ++q) ref_apply_1q(s, nq, q, h, 0, NULL, NULL, 0);
    for (int trial = 0; trial < 40; ++trial) {
      int q[16], v[16];
      const int nc = (int)(rnd() % (uint64_t)(nq < 5 ? nq : 5));       /* 0 .. min(nq-1, 4) controls */
      pick(nq, nc + 1, q);
      for (int i = 0; i < nc; ++i) v[i] = (int)(rnd() & 1);
      double m[8];
      rot(unif() * 3, m);
      ref_apply_1q(s, nq, q[nc], m, nc, q, trial & 1 ? v : NULL, 0);
      pick(nq, nc + 1, q);
      ref_apply_1q(s, nq, q[nc], NULL, nc, q, v, 1);                   /* MCX with +- flags: m == NULL */
      const int k = 1 + (int)(rnd() % (uint64_t)(nq < 6 ? nq : 6));
      pick(nq, k, q);
      for (int i = 0; i < k; ++i) v[i] = (int)(rnd() & 1);
      ref_apply_mcphase(s, nq, k, q, trial & 2 ? v : NULL, unif() * 6 - 3);
      double* tab = malloc((2u << k) * sizeof *tab);
      for (int j = 0; j < (1 << k); ++j) { const double a = unif() * 6; tab[2 * j] = cos(a); tab[2 * j + 1] = sin(a); }
      ref_apply_diag(s, nq, k, q, tab);
      free(tab);
      const int kc = (int)(rnd() % (uint64_t)(nq < 5 ? nq : 5));
      pick(nq, kc + 1, q);
      double* mats = malloc((8u << kc) * sizeof *mats);
      for (int j = 0; j < (1 << kc); ++j) rot(unif() * 3, mats + 8 * j);
      ref_apply_mux(s, nq, kc, q, q[kc], mats);
      free(mats);
    }
    const double nrm = ref_norm(s, nq);
    if (fabs(nrm - 1.0) > 1e-10) { fprintf(stderr, "norm drifted to %.15g at nq=%d\n", nrm, nq); return 1; }
    int q[16];
    const int k = nq < 4 ? nq : 4;
    pick(nq, k, q);
    double* out = calloc(1u << k, sizeof *out);
    ref_marginal(s, nq, k, q, 0, 0, out);
    double tot = 0;
    for (int j = 0; j < (1 << k); ++j) tot += out[j];
    if (fabs(tot - nrm) > 1e-10) { fprintf(stderr, "marginal sums to %.15g\n", tot); return 1; }
    ref_marginal(s, nq, k, q, 1ull << (nq - 1), 1ull << (nq - 1), out);
    free(out);
    enum { SHOTS = 257 };
    double u[SHOTS];
    uint64_t idx[SHOTS];
    for (int i = 0; i < SHOTS; ++i) u[i] = nrm * (i + 0.5) / SHOTS;        /* sorted, last ones near the total */
    u[SHOTS - 1] = nrm * 1.0000001;                                          /* rounding slack past the end */
    ref_sample(s, nq, SHOTS, u, idx);
    for (int i = 0; i < SHOTS; ++i) if (idx[i] >= n) { fprintf(stderr, "sample out of range\n"); return 1; }
    free(s);
  }
  printf("asan driver ok (%d threads)\n", ref_num_threads());
  return 0;
}
