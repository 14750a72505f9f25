// Micro-benchmark: cost of a 64-lane LDS gather of 16-byte (or 8-byte) entries at random rows, with the table laid out
// plainly or as R lane-interleaved copies (lane k reads copy k % R; copy r's row n sits at ((n * R) + r) * 16 bytes).
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_gather_bench lds_gather_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int R, int ROWS, int WAVES>
__global__ __launch_bounds__(256) void k_gather(const uint32_t* __restrict__ idx, double* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  for (int k = threadIdx.x; k < ROWS * R * 2; k += 256) lds[k] = (double)k;
  __syncthreads();
  if ((threadIdx.x >> 6) >= WAVES) return;
  const int lane = threadIdx.x & 63, r = lane % R;
  double acc = 0;
  uint32_t n = idx[threadIdx.x];
  for (int it = 0; it < iters; ++it) {
    const double2 v = *(const double2*)(lds + ((size_t)(n % ROWS) * R + r) * 2);
    acc += v.x + v.y;
    n = n * 1664525u + 1013904223u + (uint32_t)(v.x);   // next row depends on the data (serialises the gathers like a chain)
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int R, int ROWS, int WAVES>
__global__ __launch_bounds__(256) void k_gather_ilp(const uint32_t* __restrict__ idx, double* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  for (int k = threadIdx.x; k < ROWS * R * 2; k += 256) lds[k] = (double)k;
  __syncthreads();
  if ((threadIdx.x >> 6) >= WAVES) return;
  const int lane = threadIdx.x & 63, r = lane % R;
  double acc = 0;
  uint32_t n = idx[threadIdx.x];
  for (int it = 0; it < iters; it += 4) {   // four independent gathers in flight: throughput, not latency
    uint32_t n1 = n * 1664525u + 1013904223u, n2 = n1 * 1664525u + 1013904223u, n3 = n2 * 1664525u + 1013904223u;
    const double2 a = *(const double2*)(lds + ((size_t)(n % ROWS) * R + r) * 2);
    const double2 b = *(const double2*)(lds + ((size_t)(n1 % ROWS) * R + r) * 2);
    const double2 c = *(const double2*)(lds + ((size_t)(n2 % ROWS) * R + r) * 2);
    const double2 d = *(const double2*)(lds + ((size_t)(n3 % ROWS) * R + r) * 2);
    acc += a.x + b.y + c.x + d.y;
    n = n3 * 1664525u + 1013904223u;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int R, int ROWS, int WAVES, bool ILP>
static void run(const char* what) {
  const int blocks = 256 * 4, iters = 4096;
  uint32_t* d_idx; double* d_out;
  std::vector<uint32_t> h(256);
  for (int k = 0; k < 256; ++k) h[k] = 2654435761u * (k + 1);
  hipMalloc(&d_idx, 1024); hipMalloc(&d_out, blocks * 256 * 8);
  hipMemcpy(d_idx, h.data(), 1024, hipMemcpyHostToDevice);
  const size_t lds = (size_t)ROWS * R * 16;
  auto fn = ILP ? k_gather_ilp<R, ROWS, WAVES> : k_gather<R, ROWS, WAVES>;
  hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), lds, 0, d_idx, d_out, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), lds, 0, d_idx, d_out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // wave-gathers per CU: blocks / 256 CUs * WAVES * iters
  const double per_cu = (double)blocks / 256 * WAVES * iters;
  printf("%-44s R=%d rows=%5d lds=%6zu B waves/WG=%d  %8.3f ms  %6.1f ns per wave-gather per CU (%.1f clk at 2.1 GHz)\n", what, R, ROWS, lds, WAVES, ms,
         ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.1);
  hipFree(d_idx); hipFree(d_out);
}

int main() {
  run<1, 1281, 4, false>("plain table, dependent gathers");
  run<1, 1281, 4, true>("plain table, 4 gathers in flight");
  run<4, 641, 4, true>("4 interleaved copies, 4 in flight");
  run<8, 321, 4, true>("8 interleaved copies, 4 in flight");
  run<16, 161, 4, true>("16 interleaved copies, 4 in flight");
  run<2, 1281, 4, true>("2 interleaved copies, 4 in flight");
  run<1, 1281, 1, true>("plain table, 1 wave per WG, 4 in flight");
  run<4, 641, 1, true>("4 copies, 1 wave per WG, 4 in flight");
  return 0;
}
