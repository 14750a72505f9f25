// Micro-benchmark: issue cost of the vector instructions the fills are made of, in clocks per wavefront instruction per SIMD
// (eight independent chains per wave, four waves per SIMD: throughput, not latency).  Measured on MI355X (relative numbers: the
// clock under load is below the 2.4 GHz the print-out assumes): f64 add / fma / mul / max / compare 5.5-6, f64 conversions, fract and
// ldexp 4.5-5, 32-bit integer / fp32 add 3.2-3.5, v_readlane_b32 5.5, v_mov_b32_dpp 4.9, v_exp_f32 8.7.  Back-to-back
// v_cndmask_b32 reading VCC shows 23 against 5 for the form that reads an SGPR pair, but replacing the VCC-form selects of
// k_backward_fill's inner loop by the SGPR form changed nothing (16.06 vs 16.09 ms): an artefact of the back-to-back stream.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate_bench valu_rate_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAINS 8
#define BODY(ASM, CONS)                                                                                   \
  for (int it = 0; it < iters; ++it) {                                                                    \
    _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) asm volatile(ASM : CONS);                          \
  }

enum Op { ADD_F64, FMA_F64, MUL_F64, MAX_F64, CMP_F64, CVT_F64_F32, CVT_F32_F64, CVT_I32_F64, CVT_F64_I32, FRACT_F64, LDEXP_F64,
          CNDMASK, CNDMASK_K, CNDMASK_E64, MAX_CMP_PAIR, ADD_U32, READLANE, MOV_DPP, ADD_F32, EXP_F32, LSHL_ADD_U64, MAD_U64_U32, N_OPS };
static const char* kNames[N_OPS] = {"v_add_f64", "v_fma_f64", "v_mul_f64", "v_max_f64", "v_cmp_gt_f64", "v_cvt_f64_f32", "v_cvt_f32_f64",
                                    "v_cvt_i32_f64", "v_cvt_f64_i32", "v_fract_f64", "v_ldexp_f64", "v_cndmask_b32", "v_cndmask (const src)", "v_cndmask_e64 sgpr", "v_cmp+2cndmask /3", "v_add_u32",
                                    "v_readlane_b32", "v_mov_b32_dpp", "v_add_f32", "v_exp_f32", "v_lshl_add_u64", "v_mad_u64_u32"};

template <int OP>
__global__ __launch_bounds__(256) void k_rate(double* out, int iters) {
  double d[CHAINS];
  float f[CHAINS];
  int i[CHAINS], j2[CHAINS];
  unsigned long long u[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) { d[c] = 1.0 + threadIdx.x * 1e-3 + c; f[c] = 1.0f + c; i[c] = threadIdx.x + c; j2[c] = c; u[c] = threadIdx.x + c; }
  const double k = 1.0000001;
  const int kk = threadIdx.x * 3;
  const unsigned long long mask = 0x5555555555555555ull;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(k));
      if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[c]) : "v"(k));
      if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[c]) : "v"(k));
      if (OP == MAX_F64) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[c]) : "v"(k));
      if (OP == CMP_F64) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d[c]), "v"(k) : "vcc");
      if (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[c]) : "v"(f[c]));
      if (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[c]) : "v"(d[c]));
      if (OP == CVT_I32_F64) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i[c]) : "v"(d[c]));
      if (OP == CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[c]) : "v"(i[c]));
      if (OP == FRACT_F64) asm volatile("v_fract_f64 %0, %0" : "+v"(d[c]));
      if (OP == LDEXP_F64) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(d[c]));
      if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i[c]) : "v"(i[(c + 1) % CHAINS]) : );
      if (OP == CNDMASK_K) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i[c]) : "v"(kk));
      if (OP == CNDMASK_E64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(i[c]) : "v"(kk), "s"(mask));
      if (OP == MAX_CMP_PAIR) asm volatile("v_cmp_gt_f64 vcc, %2, %3\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(i[c]), "+v"(j2[c]) : "v"(d[c]), "v"(k), "v"(kk) : "vcc");
      if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i[c]) : "v"(i[(c + 1) % CHAINS]));
      if (OP == READLANE) { int s; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(i[c])); asm volatile("" : : "s"(s)); }
      if (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(i[c]) : "v"(i[(c + 1) % CHAINS]));
      if (OP == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[c]) : "v"(f[(c + 1) % CHAINS]));
      if (OP == EXP_F32) asm volatile("v_exp_f32 %0, %0" : "+v"(f[c]));
      if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(u[c]) : "v"(u[(c + 1) % CHAINS]));
      if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(u[c]) : "v"(i[c]), "v"(i[(c + 1) % CHAINS]) : "vcc");
    }
  }
  double acc = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) acc += d[c] + f[c] + i[c] + j2[c] + (double)u[c];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
static void run(double* d_out) {
  const int blocks = 256 * 4, iters = 20000;   // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 100);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = 4.0 * iters * CHAINS * (OP == MAX_CMP_PAIR ? 3 : 1);   // 4 waves per SIMD
  printf("%-16s %8.3f ms  %6.2f clk per wavefront instruction per SIMD at 2.4 GHz\n", kNames[OP], ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
}

int main() {
  double* d_out; (void)hipMalloc(&d_out, 256 * 4 * 256 * 8);
  run<ADD_F64>(d_out); run<FMA_F64>(d_out); run<MUL_F64>(d_out); run<MAX_F64>(d_out); run<CMP_F64>(d_out);
  run<CVT_F64_F32>(d_out); run<CVT_F32_F64>(d_out); run<CVT_I32_F64>(d_out); run<CVT_F64_I32>(d_out); run<FRACT_F64>(d_out);
  run<LDEXP_F64>(d_out); run<CNDMASK>(d_out); run<CNDMASK_K>(d_out); run<CNDMASK_E64>(d_out); run<MAX_CMP_PAIR>(d_out); run<ADD_U32>(d_out); run<READLANE>(d_out); run<MOV_DPP>(d_out); run<ADD_F32>(d_out);
  run<EXP_F32>(d_out); run<LSHL_ADD_U64>(d_out); run<MAD_U64_U32>(d_out);
  return 0;
}
