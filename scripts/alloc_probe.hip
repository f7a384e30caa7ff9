// GPU box: does the bandwidth of a linear fill depend on the SIZE / ALIGNMENT of the allocation it
// lives in?  (k_init streams at 7.2 TB/s in a 128+ GiB shard but 6.0-6.4 TB/s in 4-64 GiB ones.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void fill(double2* p, uint64_t n, double v) {
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = make_double2(v, 0.0);
}
static double run(double2* p, uint64_t n) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const unsigned grid = (unsigned)((n / 1024 < (1u << 24) - 1) ? n / 1024 : (1u << 24) - 1);
  for (int i = 0; i < 2; ++i) fill<<<grid, 256>>>(p, n, 1.0);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) fill<<<grid, 256>>>(p, n, 2.0);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return 16.0 * (double)n * 5 / (ms * 1e-3) / 1e9;
}
int main() {
  for (int w = 28; w <= 33; ++w) {
    const uint64_t n = 1ull << w;
    for (int mode = 0; mode < 3; ++mode) {
      // 0: exact allocation; 1: allocation padded to 130 GiB, state at its start; 2: exact + 1 GiB, state aligned to 1 GiB
      uint64_t bytes = n * 16;
      if (mode == 1) bytes = (bytes > (130ull << 30)) ? bytes : (130ull << 30);
      if (mode == 2) bytes += 1ull << 30;
      char* base = nullptr;
      if (hipMalloc(&base, bytes) != hipSuccess) { printf("w=%d mode %d: alloc failed\n", w, mode); continue; }
      char* p = base;
      if (mode == 2) p = (char*)(((uintptr_t)base + (1ull << 30) - 1) & ~((1ull << 30) - 1));
      printf("w=%d (%5.1f GiB) mode %d: %7.0f GB/s  (base %% 2MiB = %llu)\n", w, n * 16 / 1073741824.0, mode, run((double2*)p, n),
             (unsigned long long)((uintptr_t)base & ((1ull << 21) - 1)));
      fflush(stdout);
      hipFree(base);
    }
  }
  return 0;
}
