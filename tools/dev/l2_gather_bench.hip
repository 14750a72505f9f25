// Micro-benchmark: throughput of 64-lane gathers of 16-byte (two adjacent doubles) or 8-byte entries at random positions of
// a table that lives in L2 (default 800 KB: the exact log-sum-exp table of the overlap fill; 1.15 MB: its pair-emission table),
// with the load issued plainly, non-temporal, or at agent scope (sc1: no L1 allocation).  Four independent gathers in flight
// per wave, 12 waves per CU: what is measured is the L1-miss path's line rate, not latency.
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_gather_bench l2_gather_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct __attribute__((packed, aligned(8))) D2 { double v[2]; };

enum Mode { PLAIN16, NT16, AGENT8x2, PLAIN8, NT8, AGENT8, PLAIN16_ALIGNED };

template <int MODE>
__device__ __forceinline__ double fetch(const double* __restrict__ tab, uint32_t n) {
  if (MODE == PLAIN16) { const D2 f = *(const D2*)(tab + n); return f.v[0] + f.v[1]; }
  if (MODE == PLAIN16_ALIGNED) { const double2 f = *(const double2*)(tab + (n & ~1u)); return f.x + f.y; }
  if (MODE == NT16) {
    const double a = __builtin_nontemporal_load(tab + n), b = __builtin_nontemporal_load(tab + n + 1);
    return a + b;
  }
  if (MODE == AGENT8x2) {
    const double a = __hip_atomic_load(tab + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double b = __hip_atomic_load(tab + n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return a + b;
  }
  if (MODE == PLAIN8) return tab[n];
  if (MODE == NT8) return __builtin_nontemporal_load(tab + n);
  return __hip_atomic_load(tab + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const double* __restrict__ tab, uint32_t entries, double* out, int iters) {
  double acc = 0;
  uint32_t n = 2654435761u * (blockIdx.x * 256 + threadIdx.x + 1);
  for (int it = 0; it < iters; it += 4) {
    const uint32_t n1 = n * 1664525u + 1013904223u, n2 = n1 * 1664525u + 1013904223u, n3 = n2 * 1664525u + 1013904223u;
    const double a = fetch<MODE>(tab, (n >> 8) % entries), b = fetch<MODE>(tab, (n1 >> 8) % entries);
    const double c = fetch<MODE>(tab, (n2 >> 8) % entries), d = fetch<MODE>(tab, (n3 >> 8) % entries);
    acc += (a + b) + (c + d);
    n = n3 * 1664525u + 1013904223u;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char* what, uint32_t entries) {
  const int blocks = 256 * 3 * 4, iters = 2048;   // 3 workgroups of 4 waves per CU, four rounds
  double *d_tab, *d_out;
  hipMalloc(&d_tab, (size_t)(entries + 2) * 8); hipMalloc(&d_out, (size_t)blocks * 256 * 8);
  std::vector<double> h(entries + 2);
  for (uint32_t k = 0; k < entries + 2; ++k) h[k] = 1.0 / (k + 1);
  hipMemcpy(d_tab, h.data(), (size_t)(entries + 2) * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, d_tab, entries, d_out, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, d_tab, entries, d_out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_cu = (double)blocks / 256 * 4 * iters;   // wave-gathers per CU
  printf("%-34s table %7.0f KB  %8.3f ms  %7.1f ns per wave-gather per CU = %5.2f clk per lane at 2.1 GHz\n", what, entries * 8 / 1024.0, ms,
         ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.1 / 64);
  hipFree(d_tab); hipFree(d_out);
}

int main() {
  for (uint32_t entries : {100001u, 147456u, 4000u}) {
    run<PLAIN16>("16 B (8-aligned), plain", entries);
    run<PLAIN16_ALIGNED>("16 B (16-aligned), plain", entries);
    run<NT16>("2 x 8 B, non-temporal", entries);
    run<AGENT8x2>("2 x 8 B, agent scope", entries);
    run<PLAIN8>("8 B, plain", entries);
    run<NT8>("8 B, non-temporal", entries);
    run<AGENT8>("8 B, agent scope", entries);
  }
  return 0;
}
