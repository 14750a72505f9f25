// Developer microbenchmark: can the compare masks of a fill step leave the wavefront by SCALAR stores (s_store_dwordx4, no vector
// instruction at all) at the rate k_viterbi_fill produces them -- 20 masks of 8 bytes per wavefront-step, ~160 fp64 vector
// instructions between them?  Compares (a) v_cmp + v_addc into a lane word + one 16-byte vector store per 4 steps (today's
// traceback tile) with (b) v_cmp + s_store_dwordx4 of the masks.   hipcc --offload-arch=gfx950 -O3 -o build/sstore_bench tools/dev/sstore_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, uint4* out2, const double* in, int steps) {
  const int lane = threadIdx.x & 63;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  double v[10];
  for (int i = 0; i < 10; ++i) v[i] = in[(wave * 64 + lane) * 10 + i];
  unsigned long long* wp = out + (size_t)wave * steps * 20;
  const unsigned long long wbase = (unsigned long long)wp;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)wbase), hi = __builtin_amdgcn_readfirstlane((unsigned)(wbase >> 32));
  unsigned long long sp = ((unsigned long long)hi << 32) | lo;
  uint4 tile = {0, 0, 0, 0};
  for (int t = 0; t < steps; ++t) {
    unsigned acc = 0;
    unsigned long long m[20];
#pragma unroll
    for (int c = 0; c < 20; ++c) {
      // ~8 fp64 vector instructions of filler per mask (the recurrence's adds and maxes)
      double x = v[c % 10], y = v[(c + 3) % 10];
#pragma unroll
      for (int f = 0; f < 3; ++f) { x = x + y; y = fmax(y, x * 0.999); }
      v[c % 10] = x * 0.5; v[(c + 3) % 10] = y * 0.5;
      m[c] = __builtin_amdgcn_fcmp(x, y, 2);
      if (MODE == 0) {
        unsigned long long co;
        asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(acc), "=s"(co) : "v"(acc), "s"(m[c]));
      }
    }
    if (MODE == 0) {
      ((unsigned*)&tile)[t & 3] = acc;
      if ((t & 3) == 3) out2[((size_t)wave * (steps / 4) + (t >> 2)) * 64 + lane] = tile;
    } else {
#pragma unroll
      for (int c = 0; c < 20; c += 2)
        asm volatile("s_store_dwordx4 %0, %1, %2" :: "s"(__uint128_t(m[c]) | (__uint128_t(m[c + 1]) << 64)), "s"(sp), "n"(c * 8) : "memory");
      asm volatile("" ::: "memory");
      sp += 160;
    }
  }
  if (MODE == 1) asm volatile("s_dcache_wb" ::: "memory");
  if (v[0] == 12345.678) out[0] = 1;
}
int main() {
  const int waves = 25000, steps = 1000, blocks = waves / 4;
  unsigned long long* out; uint4* out2; double* in;
  hipMalloc(&out, (size_t)waves * steps * 20 * 8 + 4096);
  hipMalloc(&out2, (size_t)waves * (steps / 4) * 64 * 16);
  hipMalloc(&in, (size_t)waves * 64 * 10 * 8);
  std::vector<double> h((size_t)waves * 64 * 10);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0 + (i % 977) * 1e-3;
  hipMemcpy(in, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, out2, in, steps);
      else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, out2, in, steps);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("mode %d (%s) rep %d: %.3f ms  %s\n", mode, mode ? "s_store masks" : "v_addc + tile store", rep, ms, hipGetErrorString(hipGetLastError()));
    }
  return 0;
}
