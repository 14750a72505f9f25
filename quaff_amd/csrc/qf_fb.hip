// qf_fb.hip — Forward and Backward fills with E-step counts (the `quaff train` / `count` hot path):
// QuaffForwardMatrix ctor src/qmodel.cpp:1343-1391, QuaffBackwardMatrix ctor + transCount :1393-1510,
// QuaffCountingTask::run :2238-2271.  Same skewed G x B wavefront as the Viterbi fill (qf_kernels.hip); max
// is replaced by the reference's table log-sum-exp (src/logsumexp.cpp:34-103: 1e-4-step table, linear
// interpolation, cut-off at 10), whose table is built on the host and shared.  The Forward matrix IS
// materialised (24 B/cell, step-major so every store/load is a coalesced segment); Backward re-reads it and
// never stores its own matrix.  Results are compared with 1e-4 relative tolerance (north_star), so sums may
// be re-associated: per-column count partials are combined in an LDS ring and flushed with fp64 atomics.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "qf_dpp.hpp"
#include "qf_kernels.hpp"

namespace qf {

struct __attribute__((packed, aligned(4))) U32x4f { uint32_t v[4]; };   // 16-byte load from a 4-byte aligned address

#define QF_NEG_INF (-__builtin_huge_val())
#ifndef QF_FWD32_WAVES
#define QF_FWD32_WAVES 4    // (32, 3): Forward's three diagonals per lane fit 128 registers
#endif
#ifndef QF_BWD32_WAVES
#define QF_BWD32_WAVES 2    // Backward's do not (208 bytes of scratch at four wavefronts per SIMD, 64 at three)
#endif
#ifndef QF_BWD_WAVES
#define QF_BWD_WAVES 2
#endif

// Forward's rows are written once, 18 GB per E-step of the bench, and read once by Backward milliseconds later: stored non-temporal
// they do not churn the L2 on their way out (Forward 11.9 -> 10.6 ms; Backward reading them non-temporal: 13.6 -> 14.0 ms, not kept)
#ifndef QF_FW_NT
#define QF_FW_NT 1   // bit 0: Forward's row stores non-temporal; bit 1: Backward's row loads (A/B)
#endif
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldrow(const float4* p) {
  if (QF_FW_NT & 2) { const f4v v = __builtin_nontemporal_load((const f4v*)p); return make_float4(v.x, v.y, v.z, v.w); }
  return *p;
}

// log_sum_exp, src/logsumexp.cpp:34-50 + log_sum_exp_unary :84-103 (x >= 10, NaN, inf -> 0).
// n = (int)(x / 1e-4) is evaluated as x * 1e4: at worst the neighbouring interval of the same piecewise-
// linear function is used (the interpolant is continuous), far inside the 1e-4 tolerance.
__device__ __forceinline__ double lse2(const double* __restrict__ tab, double a, double b) {
  const double mx = a > b ? a : b, mn = a > b ? b : a;
  const double diff = mx - mn;  // -inf - -inf = NaN -> treated as 0 below, like the reference's a == b case
  if (!(diff < 10.0)) return a == b ? mx + tab[0] : mx;
  const int n = (int)(diff * 10000.0);
  const double dx = diff - n * .0001;
  const double f0 = tab[n], f1 = tab[n + 1];
  return mx + (f0 + (f1 - f0) * (dx * 10000.0));
}

// Expected counts are compared at 1e-4 relative: their exponential goes through the single-precision hardware exp2
// (relative error ~1e-7 for the terms that matter, |x| of a few units; <= 1e-5 for the negligible ones near underflow;
// below ~-87 the count flushes to 0).  The double-precision software exp was ~250 of Backward's ~300 VALU per cell.
__device__ __forceinline__ float count_expf(double x) { return __expf((float)x); }
__device__ __forceinline__ double count_exp(double x) { return (double)__expf((float)x); }

// Expected counts from different bands meet in shared accumulators (the emission tables in global memory, the context-dependent
// transition counts of a wavefront in LDS).  Added as floating point with atomics, their totals depend on the order the
// hardware serves them in: run to run, and with every way of cutting a batch into pieces, contexts or devices.  They are added
// as FIXED POINT instead, which is exact integer arithmetic: the total is the same whatever the order (the reference adds
// per-read counts in read order, src/qmodel.cpp:2416-2422; any fixed total is as good as its).  An accumulator is two 64-bit
// words that never exchange a carry while terms arrive, so both adds are fire-and-forget atomics (an add that has to return the
// old value for its carry parks the wavefront for an LDS / L2 round trip per term: measured +35 % on Backward):
//   word 1 += floor(v 2^32)                  the term down to 2^-32; 2^32 of headroom above a count total of 4e9
//   word 0 += floor(frac(v 2^32) 2^32)       the next 32 bits, below 2^32 per term: headroom for 2^32 terms
// value = word 1 / 2^32 + word 0 / 2^64, normalised into 64.64 fixed point once, after the last add (k_sum_count_replicas).
// A term is a non-negative finite double; what it has below 2^-64 is truncated (of the term itself, not an accumulated error).
__device__ __forceinline__ void fx_add_words(unsigned long long* acc, unsigned long long w0, unsigned long long w1) {
  if (w0) atomicAdd(acc, w0);
  if (w1) atomicAdd(acc + 1, w1);
}
__device__ __forceinline__ void fx_add(unsigned long long* acc, double v) {
  if (!(v > 0.0)) return;   // zero, and never a NaN into an integer conversion
  // floor(v 2^32) is read off the mantissa of floor(v 2^32) + 2^52, which needs v 2^32 < 2^52.  Every term this kernel forms is
  // a sum of posterior weights over at most one band's columns (v <= 2^20 for the longest read the library takes): a term of
  // 2^19 or more goes in as two exact halves.
  const bool big = v >= 524288.0;
  if (big) v *= 0.5;
  const double t = v * 4294967296.0, ft = floor(t);
  const unsigned long long w0 = (unsigned long long)__double2uint_rz((t - ft) * 4294967296.0);
  const unsigned long long w1 = (unsigned long long)__double_as_longlong(ft + 4503599627370496.0) & 0xFFFFFFFFFFFFFull;
  fx_add_words(acc, w0, w1);
  if (big) fx_add_words(acc, w0, w1);
}
// A band's running totals (the context-free transition counts: sums over all the band's cells; -global full DP against a
// reference of millions of bases pushes the expected delete count past fx_add's 2^20): integer part and fraction separately,
// exact for any finite term below 2^31.
__device__ __forceinline__ void fx_add_total(unsigned long long* acc, double v) {
  if (!(v > 0.0)) return;
  if (v < 262144.0) { fx_add(acc, v); return; }
  const double ip = floor(v);
  fx_add(acc, v - ip);                                              // (exact: v < 2^53)
  fx_add_words(acc, 0ull, (unsigned long long)ip << 32);
}

// (context k-mer, quality) of a match-emission row number as the context words carry it (FbArgs::em_qmajor)
__device__ __forceinline__ void em_row_decode(const FbArgs& a, uint32_t er, uint32_t& mk, uint32_t& q) {
  if (a.em_qmajor) { mk = er & (a.Km - 1u); q = (er >> a.em_kshift) + a.em_qmin; }
  else { mk = er / (kNQualDev + 1); q = er % (kNQualDev + 1); }
}

// Forward / Backward are compared at 1e-4 relative, so their log(1 + exp(-x)) need not be the reference's table
// interpolant bit for bit.  The 800 KB table is a 64-way L2 gather per call; the same function as kLsePieces quadratic
// pieces on a 1/128 grid (20 KB; host: ensure_lse, qf_api.hip) sits in LDS: within 3.3e-10 of log1p(exp(-x)), the
// accuracy of the reference's own 1e-4-step linear interpolant.  What these kernels spend their time on is exactly this
// lookup -- a 64-lane LDS gather with bank conflicts (~38 cycles of the CU's LDS per 16-byte-per-lane read, measured) --
// so a piece is 16 bytes (fp64 constant term, fp32 linear and quadratic terms): one ds_read_b128 per call.  The x >= 10
// cut-off is the reference's (src/logsumexp.cpp:84-90): the last piece is all zero and x is clamped onto it; -inf - -inf =
// NaN and +-inf land there too (v_min_f64 returns its non-NaN operand), which gives max + 0 like the reference's a == b /
// isinf rules.
constexpr int kLseDoubles = kLsePieces * 2;   // LDS footprint in doubles
__device__ __forceinline__ void lseh_load(double* s_h, const double* __restrict__ g_h, int tid, int nthreads) {
  for (int k = tid; k < kLseDoubles; k += nthreads) s_h[k] = g_h[k];
}
// v_max_f64 / v_min_f64 as they are: the compiler's fmax / fmin first canonicalise both operands (one more v_max_f64 each)
// to quiet signalling NaNs, which these values never are; the instructions themselves return the non-NaN operand.
__device__ __forceinline__ double vmax_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double vmin_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double lseh(const double* hs, double a, double b) {
  const double mx = vmax_f64(a, b);
  const double u = vmin_f64(fabs(a - b) * 128.0, (double)(kLsePieces - 1));
  const int n = (int)u;
  const double t = __builtin_amdgcn_fract(u);
  const LsePiece q = ((const LsePiece*)hs)[n];
  return mx + fma(t, fma(t, (double)q.c2, (double)q.c1), q.c0);
}

// ------------------------------------------------------------------------------------------------
// Forward fill: the skewed G x B wavefront of k_viterbi_fill (qf_kernels.hip) with log-sum-exp for max.
//  * dynamic LDS: the lse pieces, the context-dependent transition scores, and (EMLDS) the emission tables;
//  * the context word, reference-token window and emission scores of step t+1 are fetched at step t;
//  * no per-cell validity masks: cells above the band's last diagonal or outside the reference read a -inf match emission,
//    which keeps them at -inf (see emis below); the start term and the end terms sit in wave-uniform branches that are
//    taken only while some lane is on its first / last column;
//  * neighbour lanes exchange by DPP shifts.
// Storage: packed fp32 rows (qf_device.hpp: fw_row_floats) -- the store path, not arithmetic, bounds this kernel
// (measured: 15 eight-byte stores per step cost 11 of its 19 ms), so a step writes 4 sixteen-byte stores per lane.
// ------------------------------------------------------------------------------------------------
#ifndef QF_FWD_WAVES
#define QF_FWD_WAVES 3
#endif
template <int G, int B, bool GAPCTX, bool EMLDS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(G == 32 ? QF_FWD32_WAVES : B <= 5 ? QF_FWD_WAVES : B <= 8 ? 2 : 1))) void k_forward_fill(FbArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds_fb[];
  const uint32_t Kg = a.dp.Kg;
  const uint32_t n_em = a.dp.ematch_ninf_off / 8 + 4;
  double* s_trans = lds_fb + kLseDoubles;
  double* s_em = s_trans + ((4 * Kg + 1) & ~1u);
  lseh_load(lds_fb, a.lse_h, threadIdx.x, 256);
  for (uint32_t k = threadIdx.x; k < 4 * Kg; k += 256) s_trans[k] = a.dp.trans[k];
  if (EMLDS) {
    for (uint32_t k = threadIdx.x; k < n_em; k += 256) s_em[k] = a.dp.ematch[k];
    for (uint32_t k = threadIdx.x; k < kInsRows; k += 256) s_em[n_em + k] = a.dp.eins[k];
  }
  __syncthreads();
  const double* hs = lds_fb;
  constexpr int UPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int grp = lane / G, l = lane % G;
  const uint32_t uidx = wave * UPW + grp;
  const bool active = uidx < a.n_cls_units;
  uint32_t uid = 0;
  int dlo = 0, dhi = -1, xLen = 0, yLen = 0;
  uint64_t xb = 0, yb = 0, xw = 0, fw_off = 0;
  if (active) {
    uid = a.cls_list[uidx];
    const Unit u = a.units[uid];
    const uint32_t r = u.pair / a.n_refs, x = u.pair % a.n_refs;
    xb = a.ref_off[x]; xLen = (int)(a.ref_off[x + 1] - xb); xw = a.ref_woff[x];
    yb = a.read_off[r]; yLen = (int)(a.read_off[r + 1] - yb);
    dlo = u.dlo; dhi = u.dhi; fw_off = u.tb_off;
  }
  int T = active ? yLen + G - 1 : 0;
  for (int o = 32; o; o >>= 1) T = max(T, __shfl_xor(T, o));
  const int d0 = dlo + l * B;
  const int bmax = active ? dhi - d0 : -1;  // slots b > bmax are outside the band
  const double i2m = a.dp.i2m, d2m = a.dp.d2m, i2i = a.dp.i2i, d2d = a.dp.d2d;
  const double* __restrict__ ematch = EMLDS ? s_em : a.dp.ematch;
  const double* __restrict__ eins = EMLDS ? s_em + n_em : a.dp.eins;
  const double* __restrict__ trans = s_trans;
  const bool local = a.dp.local != 0;
  const double c_m2m = trans[0], c_m2i = trans[Kg], c_m2d = trans[2 * Kg];
  constexpr int NF = fw_row_floats(B);
  float4* __restrict__ fwrow = (float4*)(a.fw + fw_off);
  double* __restrict__ fwend = a.fw + fw_off + (uint64_t)(yLen + G - 1) * G * NF / 2;

  double M[B], I[B], D[B];
#pragma unroll
  for (int b = 0; b < B; ++b) M[b] = I[b] = D[b] = QF_NEG_INF;
  double pubM = QF_NEG_INF, pubD = QF_NEG_INF;
  double endTerm[B];  // mat(i,yLen) + m2e for this lane's rows of the last column
#pragma unroll
  for (int b = 0; b < B; ++b) endTerm[b] = QF_NEG_INF;

  // reference tokens: a window of the lane's B rows (slot b at bits 2b), one new token per step taken from the 2-bit packed
  // reference (a 64-bit pair of words per 16 steps, the next word fetched a chunk ahead) -- as k_viterbi_fill
  const uint32_t* __restrict__ xp = a.ref_packed + xw;
  const int nxw = (xLen + 15) / 16 + 2;
  const int rtop0 = d0 - l + B - 1;
  const int q0 = rtop0 >> 4, sh0 = 2 * (rtop0 & 15);
  auto xword = [&](int q) -> uint32_t { return xp[min(max(q, 0), nxw - 1)]; };
  uint32_t xlo, xhi = xword(q0), xnx = xword(q0 + 1);
  uint32_t win = 0;
  {
    const uint8_t* xt = a.ref_tok + xb;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const int row = d0 - l - 1 + b;
      const uint32_t t = (row >= 0 && row < xLen) ? xt[row] : 0u;
      win |= t << (2 * b);
    }
  }
  const uint32_t* __restrict__ ctx = a.ctx + yb;
  // context words: the word of step t (column j = t - l + 1, index j - 1 = t - l) is loaded two steps ahead, because the
  // emission scores of step t+1 are fetched at step t
  auto ctxword = [&](int t) -> uint32_t { return ctx[min(t - l, yLen + 4)]; };
  uint32_t gkPrev = 0;
  const uint32_t ninf_off = a.dp.ematch_ninf_off;
  // match emission of slot b at reference row i: -inf above the band's last diagonal and outside the reference.  That alone
  // keeps every cell outside the band / matrix at -inf (match by its emission, insert and delete by induction from -inf
  // neighbours), except the delete state just outside, which a cell of the band never reads.
  auto emis = [&](uint32_t w, uint32_t window, int b, int i) -> double {
    uint32_t off = ((w & 0x7FFFu) << 5) | (((window >> (2 * b)) & 3u) << 3);
    if (b > bmax || (uint32_t)(i - 1) >= (uint32_t)xLen) off = ninf_off;
    return *(const double*)((const char*)ematch + off);
  };
  uint32_t wN = ctxword(0), wNN = ctxword(1);
  const uint32_t tok0 = (uint32_t)((((unsigned long long)xnx << 32) | xhi) >> sh0) & 3u;
  uint32_t winN = (win >> 2) | (tok0 << (2 * (B - 1)));
  double eN[B], insEN = eins[(wN >> 15) & 0x1FFu];
#pragma unroll
  for (int b = 0; b < B; ++b) eN[b] = emis(wN, winN, b, d0 + b + 1 - l);

  // lanes are on column 1 at steps 0 .. G-1 (start term) and on their last column at steps endLo .. endHi (end terms)
  int endLo = active ? yLen + l - 1 : 0x7FFFFFFF, endHi = active ? yLen + l - 1 : -1;
  for (int o = 32; o; o >>= 1) {
    endLo = min(endLo, __shfl_xor(endLo, o));
    endHi = max(endHi, __shfl_xor(endHi, o));
  }
  endLo = __builtin_amdgcn_readfirstlane(endLo);
  endHi = __builtin_amdgcn_readfirstlane(endHi);

  int chunk = 0;
  for (int t0 = 0; t0 <= endHi; t0 += 16, ++chunk) {
    xlo = xhi; xhi = xnx; xnx = xword(q0 + chunk + 2);
    const unsigned long long xpair = ((unsigned long long)xhi << 32) | xlo;
#pragma unroll 1
    for (int s = 0; s < 16; ++s) {
      const int t = t0 + s;
      if (t > endHi) break;
      const int j = t - l + 1;
      const uint32_t w = wN;
      double e[B];
      const double insE = insEN;
#pragma unroll
      for (int b = 0; b < B; ++b) e[b] = eN[b];
      // step t+1's fetches (the token of step 16 of a chunk is the next chunk's first: still inside the 64-bit window)
      wN = wNN;
      wNN = ctxword(t + 2);
      winN = (winN >> 2) | (((uint32_t)(xpair >> (sh0 + 2 * (s + 1))) & 3u) << (2 * (B - 1)));
      insEN = eins[(wN >> 15) & 0x1FFu];
#pragma unroll
      for (int b = 0; b < B; ++b) eN[b] = emis(wN, winN, b, d0 + b + j + 1);
      const uint32_t gk = w >> 24;
      double m2m, m2i, m2d;
      if (GAPCTX) {
        const uint32_t gp = j <= 1 ? 0u : gkPrev;   // yIndelKmer is padded with a leading 0 (qmodel.cpp:1322)
        m2m = trans[gp]; m2i = trans[Kg + gp]; m2d = trans[2 * Kg + gk];
        gkPrev = gk;
      } else {
        m2m = c_m2m; m2i = c_m2i; m2d = c_m2d;
      }
      double prevM = dpp_from_below<G, false>(pubM), prevD = dpp_from_below<G, false>(pubD);
      double upM = 0, upI = 0;
      const bool colvalid = active && j >= 1 && j <= yLen;
      const bool startStep = t < G, endStep = t >= endLo;   // wave-uniform
      if (!startStep && !endStep) {
        // the common step: straight-line code over the B slots (no branches: the compiler overlaps the slots' table lookups)
#pragma unroll
        for (int b = 0; b < B; ++b) {
          // mat(i,j) = lse(lse(mat' + m2m, del' + d2m), ins' + i2m) + emit  (src/qmodel.cpp:1363-1372)
          const double nm = lseh(hs, lseh(hs, M[b] + m2m, D[b] + d2m), I[b] + i2m) + e[b];
          double srcM, srcI;
          if (b + 1 < B) { srcM = M[b + 1]; srcI = I[b + 1]; } else { srcM = upM; srcI = upI; }
          const double ni = insE + lseh(hs, srcI + i2i, srcM + m2i);
          const double ndl = lseh(hs, prevD + d2d, prevM + m2d);
          M[b] = nm; I[b] = ni; D[b] = ndl;
          prevM = nm; prevD = ndl;
          if (b == 0) { upM = dpp_from_above<G, false>(nm); upI = dpp_from_above<G, false>(ni); }
        }
      } else {
        // some lane is on its first column (start term) or its last (end terms, the exact last-column values)
#pragma unroll
        for (int b = 0; b < B; ++b) {
          double nm = lseh(hs, lseh(hs, M[b] + m2m, D[b] + d2m), I[b] + i2m);
          if (j == 1 && (d0 + b == 0 || local)) nm = lseh(hs, nm, 0.0);
          nm += e[b];
          double srcM, srcI;
          if (b + 1 < B) { srcM = M[b + 1]; srcI = I[b + 1]; } else { srcM = upM; srcI = upI; }
          const double ni = insE + lseh(hs, srcI + i2i, srcM + m2i);
          const double ndl = lseh(hs, prevD + d2d, prevM + m2d);
          M[b] = nm; I[b] = ni; D[b] = ndl;
          prevM = nm; prevD = ndl;
          if (colvalid && j == yLen) {
            if (b <= bmax) fwend[l * B + b] = nm;   // mat(i, yLen), exactly, for k_pair_forward
            if (local || d0 + b + j == xLen) endTerm[b] = nm + trans[3 * Kg + gk];   // (-inf outside band / matrix)
          }
          if (b == 0) { upM = dpp_from_above<G, false>(nm); upI = dpp_from_above<G, false>(ni); }
        }
      }
      pubM = prevM; pubD = prevD;
      if (colvalid && bmax >= 0) {
        // the step's row: anchor = the largest value (fp32), then the offsets of slot B-1 .. 0
        double anchor = M[0];
#pragma unroll
        for (int b = 0; b < B; ++b) anchor = vmax_f64(vmax_f64(anchor, M[b]), vmax_f64(I[b], D[b]));
        const float af = anchor > QF_NEG_INF ? (float)anchor : 0.f;
        const double ad = (double)af;
        float row[NF];
        row[0] = af;
#pragma unroll
        for (int b = 0; b < B; ++b) {
          row[fw_float_index(B, b, 0)] = (float)(M[b] - ad);
          row[fw_float_index(B, b, 1)] = (float)(I[b] - ad);
          row[fw_float_index(B, b, 2)] = (float)(D[b] - ad);
        }
#pragma unroll
        for (int k = 3 * B + 1; k < NF; ++k) row[k] = 0.f;
        float4* dst = fwrow + ((uint64_t)t * (NF / 4)) * G + l;
#pragma unroll
        for (int c = 0; c < NF / 4; ++c) {
          if (QF_FW_NT & 1) __builtin_nontemporal_store(f4v{row[4 * c], row[4 * c + 1], row[4 * c + 2], row[4 * c + 3]}, (f4v*)(dst + (uint64_t)c * G));
          else dst[(uint64_t)c * G] = make_float4(row[4 * c], row[4 * c + 1], row[4 * c + 2], row[4 * c + 3]);
        }
      }
    }
  }
  // end = lse(end, mat(i,yLen) + m2e) accumulated over rows in ascending order (src/qmodel.cpp:1379-1381): chain the
  // lanes one after the other
  double endv = QF_NEG_INF;
  for (int s = 0; s < G; ++s) {
    double v = endv;
#pragma unroll
    for (int b = 0; b < B; ++b) if (endTerm[b] > QF_NEG_INF) v = lseh(hs, v, endTerm[b]);
    endv = __shfl(l == s ? v : endv, s, G);
  }
  if (active && l == 0) a.units[uid].end_val = endv;
}

// Single-diagonal bands (diagonal 0 is in every envelope, so nearly every true pair carries one beside its seeded band):
// the gap states of a lone diagonal stay -inf and the match state is a serial chain, so one lane takes a band instead of
// a 16 x 2 wavefront group with 31 of its 32 slots idle.  The values go where slot 0 of lane 0 of the (16,2) layout keeps
// them (k_backward_fill<16,2> reads only the band's own diagonal), same arithmetic as k_forward_fill.
__global__ __launch_bounds__(256) void k_forward_single(FbArgs a) {
  __shared__ __attribute__((aligned(16))) double s_lseh[kLseDoubles];
  lseh_load(s_lseh, a.lse_h, threadIdx.x, 256);
  __syncthreads();
  const double* hs = s_lseh;
  constexpr int G = 16, B = 2;
  const uint32_t uidx = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = uidx < a.n_cls_units;
  uint32_t uid = 0;
  int d = 0, xLen = 0, yLen = 0;
  uint64_t xb = 0, yb = 0, fw_off = 0;
  if (active) {
    uid = a.cls_list[uidx];
    const Unit u = a.units[uid];
    const uint32_t r = u.pair / a.n_refs, x = u.pair % a.n_refs;
    xb = a.ref_off[x]; xLen = (int)(a.ref_off[x + 1] - xb);
    yb = a.read_off[r]; yLen = (int)(a.read_off[r + 1] - yb);
    d = u.dlo; fw_off = u.tb_off;
  }
  int T = active ? yLen : 0;
  for (int o = 32; o; o >>= 1) T = max(T, __shfl_xor(T, o));
  const double* __restrict__ ematch = a.dp.ematch;
  const double* __restrict__ trans = a.dp.trans;
  const uint32_t Kg = a.dp.Kg;
  const bool local = a.dp.local != 0;
  const uint8_t* __restrict__ xt = a.ref_tok + xb;
  const uint32_t* __restrict__ ctx = a.ctx + yb;
  constexpr int NF = fw_row_floats(B);
  float4* __restrict__ fwrow = (float4*)(a.fw + fw_off);
  double* __restrict__ fwend = a.fw + fw_off + (uint64_t)(yLen + G - 1) * G * NF / 2;
  double M = QF_NEG_INF, endTerm = QF_NEG_INF;
  uint32_t gkPrev = 0;
  for (int j = 1; j <= T; ++j) {
    const int i = d + j;
    const bool colvalid = active && j <= yLen, valid = colvalid && i >= 1 && i <= xLen;
    const uint32_t w = ctx[min(j - 1, yLen + 4)];
    const uint32_t erow4 = (w & 0x7FFFu) * 4u, gk = w >> 24;
    const uint32_t gp = j > 1 ? gkPrev : 0u;  // yIndelKmer is padded with a leading 0 (qmodel.cpp:1322)
    gkPrev = gk;
    const uint32_t tok = valid ? (uint32_t)xt[i - 1] : 0u;
    double nm = lseh(hs, lseh(hs, M + trans[gp], QF_NEG_INF), QF_NEG_INF);
    if (j == 1 && (i == 1 || local)) nm = lseh(hs, nm, 0.0);
    nm += ematch[erow4 + tok];
    if (!valid) nm = QF_NEG_INF;
    M = nm;
    if (colvalid) {   // the (16,2) row of step t = j - 1, lane 0: slot 0 holds the diagonal, everything else is -inf
      const float af = nm > QF_NEG_INF ? (float)nm : 0.f;
      float4* dst = fwrow + ((uint64_t)(j - 1) * (NF / 4)) * G;
      const float ninf = -__builtin_huge_valf();
      dst[0] = make_float4(af, ninf, ninf, ninf);                                  // anchor | slot 1: mat, ins, del
      dst[G] = make_float4((float)(nm - (double)af), ninf, ninf, 0.f);            // slot 0: mat, ins, del | pad
      if (j == yLen) fwend[0] = nm;
    }
    if (j == yLen && valid && (local || i == xLen)) endTerm = nm + trans[3 * Kg + gk];
  }
  if (active) a.units[uid].end_val = endTerm > QF_NEG_INF ? lseh(hs, QF_NEG_INF, endTerm) : QF_NEG_INF;
}

// Forward result of a pair: the reference keeps ONE running sum, `end = lse(end, mat(i,yLen) + m2e)`, down the whole last read
// column in ascending row order (src/qmodel.cpp:1379-1381).  The table log-sum-exp drops a term more than 10 below the
// running total, so summing band by band and combining the bands afterwards is a different number (by up to ~1e-4 when a
// pair has dozens of bands).  One thread per pair therefore walks the bands in ascending diagonal order and the cells of
// each band's last column in ascending row order, taking mat(i,yLen) from the Forward matrices, with the exact table.
// (Row-space bands are whole envelopes - one band per pair - and contribute their own end sum.)
__global__ void k_pair_forward(FinalArgs a, const double* __restrict__ tab) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.n_pairs) return;
  const uint32_t r = p / a.n_refs, x = p % a.n_refs;
  const int xLen = (int)(a.ref_off[x + 1] - a.ref_off[x]), yLen = (int)(a.read_off[r + 1] - a.read_off[r]);
  const double m2e = yLen > 0 ? a.trans[3 * a.Kg + (a.ctx[a.read_off[r] + (uint64_t)(yLen - 1)] >> 24)] : QF_NEG_INF;
  double v = QF_NEG_INF;
  int last = -2147483647 - 1;
  while (true) {  // bands in ascending diagonal (= ascending row) order, like the reference's row loop
    uint32_t pick = kNoUnit;
    int best = 2147483647;
    for (uint32_t uid = a.pair_head[p]; uid != kNoUnit; uid = a.units[uid].next) {
      const int dl = a.units[uid].dlo;
      if (dl > last && dl < best) { best = dl; pick = uid; }
    }
    if (pick == kNoUnit) break;
    last = best;
    const Unit u = a.units[pick];
    if (u.cls == (uint32_t)kRowClass) {
      if (u.end_val > QF_NEG_INF) v = lse2(tab, v, u.end_val);
      continue;
    }
    const double* __restrict__ fwend = a.fw + u.tb_off + unit_fw_steps_doubles((int)u.cls, (uint32_t)yLen);
    for (int d = u.dlo; d <= u.dhi; ++d) {
      const int i = d + yLen;
      if (i < 1 || i > xLen || !(a.local || i == xLen)) continue;
      const double m = fwend[d - u.dlo];   // mat(i, yLen), as the Forward kernel left it (fp64)
      const double term = m + m2e;
      if (term > QF_NEG_INF) v = lse2(tab, v, term);
    }
  }
  a.pair_score[p] = v;
}

// QuaffCountingTask::run, src/qmodel.cpp:2238-2271, the sequential part: running log-likelihood over the
// read's reference order, which references get a Backward pass (LL >= running - 20), the posterior weights
// exp(LL_x - yLogLike), and the next iteration's order (LL descending, cut at yLogLike - 20).
__global__ void k_count_plan(CountPlanArgs a) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.n_reads) return;
  const double* __restrict__ tab = a.lse;
  double ylog = a.use_null ? a.nll[r] : QF_NEG_INF;
  const uint32_t nord = a.order_in ? a.order_n_in[r] : a.n_refs;
  for (uint32_t k = 0; k < a.n_refs; ++k) a.weight[(uint64_t)r * a.n_refs + k] = 0.0;
  for (uint32_t k = 0; k < nord; ++k) {
    const uint32_t x = a.order_in ? a.order_in[(uint64_t)r * a.n_refs + k] : k;
    const double ll = a.pair_fwd[(uint64_t)r * a.n_refs + x];
    if (ll >= ylog - 20.0) a.weight[(uint64_t)r * a.n_refs + x] = 1.0;  // marks "run Backward"
    ylog = lse2(tab, ylog, ll);
  }
  // references outside the input order keep LL = -inf (xyLogLike initialisation, :2245)
  uint32_t* out = a.order_out + (uint64_t)r * a.n_refs;
  uint32_t n = 0;
  for (uint32_t x = 0; x < a.n_refs; ++x) {
    bool listed = !a.order_in;
    for (uint32_t k = 0; !listed && k < nord; ++k) listed = a.order_in[(uint64_t)r * a.n_refs + k] == x;
    const uint64_t p = (uint64_t)r * a.n_refs + x;
    const double ll = listed ? a.pair_fwd[p] : QF_NEG_INF;
    if (!listed) a.pair_fwd_out[p] = QF_NEG_INF; else a.pair_fwd_out[p] = ll;
    a.weight[p] = (a.weight[p] != 0.0 && ll > QF_NEG_INF) ? exp(ll - ylog) : 0.0;
    if (!(ll < ylog - 20.0)) {  // insertion sort, LL descending
      uint32_t k = n++;
      while (k > 0) {
        const uint32_t y = out[k - 1];
        const double lly = a.pair_fwd_out[(uint64_t)r * a.n_refs + y];
        if (lly > ll) break;   // ties: the later reference first (ascending stable sort, then reversed: util.h:115-124, qmodel.cpp:2264-2265)
        out[k] = y;
        --k;
      }
      out[k] = x;
    }
  }
  a.order_n_out[r] = n;
  a.read_loglike[r] = ylog;
}

// ------------------------------------------------------------------------------------------------
// Backward sweep with the E-step counts.  Lane l handles column j = yLen - (t - (G-1-l)); slots run from high diagonal to
// low, so
//   (i+1,j+1): same diagonal, previous step (own registers)
//   (i+1,j  ): diagonal d+1, same column   (own slot b+1 this step, or lane l+1's slot 0 from the previous step)
//   (i,  j+1): diagonal d-1, next column   (own slot b-1 from the previous step, or lane l-1's last slot, which that
//                                           lane finishes first in this very step)
// Each cell is treated as a SOURCE: its Backward values are the lse of (transition + emission + Backward of the
// destination), and the expected count of each of those transitions is exp(F_src + term - F_result) -- the same
// operands, in the same association, as transCount (src/qmodel.cpp:1504-1510).
//  * same LDS tables and one-step-ahead fetches (context word, token, emissions, and the cell's three Forward values) as
//    k_forward_fill, and no per-cell validity masks either: a destination above the band's last diagonal or outside the
//    reference has a -inf match emission.  By induction (down the rows from the band's top edge for the delete state, along
//    the row from the last column for the insert state) every source cell above the band or below the last reference row
//    then stays at -inf, which is all a cell of the band reads from them; source rows above row 1 are read by nobody.
//    Cells that do not exist read a -inf Forward value, so their counts are exp(-inf) = 0.  End transitions and the
//    start term sit in wave-uniform branches taken only while some lane is on its last / first column;
//  * count terms are formed in fp32 (hardware exp2; fp32 partial sums over the B cells of one step), converted once per
//    step and accumulated in fp64; the pair's posterior weight multiplies at the flush, not per term.
// ------------------------------------------------------------------------------------------------
template <int G, int B, bool GAPCTX, bool EMLDS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(G == 32 ? QF_BWD32_WAVES : B <= 5 ? QF_BWD_WAVES : 1, B <= 5 ? 8 : 1))) void k_backward_fill(FbArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds_fb[];
  const uint32_t Kg = a.dp.Kg, Km = a.Km;
  const uint32_t n_em = a.dp.ematch_ninf_off / 8 + 4;
  double* s_trans = lds_fb + kLseDoubles;
  double* s_acc_all = s_trans + ((4 * Kg + 1) & ~1u);             // [4 waves][4][64 lanes] i2m, d2m, i2i, d2d counts of each lane
  double* s_em = s_acc_all + 4 * 4 * 64;
  lseh_load(lds_fb, a.lse_h, threadIdx.x, 256);
  for (uint32_t k = threadIdx.x; k < 4 * Kg; k += 256) s_trans[k] = a.dp.trans[k];
  for (uint32_t k = threadIdx.x; k < 4 * 4 * 64; k += 256) s_acc_all[k] = 0.0;
  if (EMLDS) {
    for (uint32_t k = threadIdx.x; k < n_em; k += 256) s_em[k] = a.dp.ematch[k];
    for (uint32_t k = threadIdx.x; k < kInsRows; k += 256) s_em[n_em + k] = a.dp.eins[k];
  }
  __syncthreads();
  const double* hs = lds_fb;
  constexpr int UPW = 64 / G;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double* s_acc = s_acc_all + (size_t)wv * 4 * 64 + lane;   // this lane's four running sums, 64 doubles apart
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int grp = lane / G, l = lane % G, rl = G - 1 - l;
  const uint32_t uidx = wave * UPW + grp;
  const bool exists = uidx < a.n_cls_units;
  bool active = exists;
  uint32_t uid = 0;
  int dlo = 0, dhi = -1, xLen = 0, yLen = 0;
  uint64_t xb = 0, yb = 0, fw_off = 0;
  double Fres = 0, wgt = 0;
  if (active) {
    uid = a.cls_list[uidx];
    const Unit u = a.units[uid];
    const uint32_t r = u.pair / a.n_refs, x = u.pair % a.n_refs;
    xb = a.ref_off[x]; xLen = (int)(a.ref_off[x + 1] - xb);
    yb = a.read_off[r]; yLen = (int)(a.read_off[r + 1] - yb);
    dlo = u.dlo; dhi = u.dhi; fw_off = u.tb_off;
    Fres = a.pair_fwd[u.pair];
    wgt = a.pair_weight[u.pair];
    if (!(wgt > 0.0) || !(Fres > QF_NEG_INF)) active = false;  // pruned pair: no Backward (qmodel.cpp:2252)
    // Bands are disconnected from one another (a missing diagonal cannot be crossed), so every expected count of this
    // band is at most exp(band's Forward end - pair's Forward) x weight; count_exp flushes to zero below ~-87: such a
    // band (typically the lone diagonal 0 beside the seeded band) contributes exactly nothing.
    if (!a.no_band_shortcuts && u.end_val - Fres < -110.0) active = false;
  }
  if (exists && l == 0) a.units[uid].staged = active ? 1u : 0u;   // k_count_flush: this band's rows hold column sums
  int T = active ? yLen + G - 1 : 0;
  for (int o = 32; o; o >>= 1) T = max(T, __shfl_xor(T, o));
  if (T == 0) return;
  const int d0 = dlo + l * B;
  const int bmax = active ? dhi - d0 : -1;
  const double i2m = a.dp.i2m, d2m = a.dp.d2m, i2i = a.dp.i2i, d2d = a.dp.d2d;
  const double* __restrict__ ematch = EMLDS ? s_em : a.dp.ematch;
  const double* __restrict__ eins = EMLDS ? s_em + n_em : a.dp.eins;
  const double* __restrict__ trans = s_trans;
  const bool local = a.dp.local != 0;
  const double c_m2m = trans[0], c_m2i = trans[Kg], c_m2d = trans[2 * Kg];
  const uint8_t* __restrict__ xt = a.ref_tok + xb;
  const uint32_t* __restrict__ ctx = a.ctx + yb;
  unsigned long long* __restrict__ cnt = a.counts + 2 * (size_t)(blockIdx.x % kCountReplicas) * a.counts_stride;   // (lo, hi) per entry; contention: see kCountReplicas
  const uint64_t cMat = 4ull * kNQualDev, cTr = (4ull + 4ull * Km) * kNQualDev;

  double Bm[B], Bi[B], Bd[B];   // Backward values of this lane's diagonals at the column of its previous step
#pragma unroll
  for (int b = 0; b < B; ++b) Bm[b] = Bi[b] = Bd[b] = QF_NEG_INF;
  double pubD = QF_NEG_INF;     // slot 0's del after this lane's latest step (for lane l-1)
  double acc_m2e = 0, startv = QF_NEG_INF;
  double acc_m2m = 0, acc_m2i = 0, acc_m2d = 0;   // !GAPCTX only
  // running match-by-token[4] / insert / (GAPCTX) m2m, m2i, m2d sums of the column this lane is on: at most G x B terms each,
  // fp32 like the partials
  constexpr int NCOL = GAPCTX ? 8 : 5;
  float colsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t gkEnd = 0;

  // tokens of rows i+1 of the lane's B slots as a sliding 2-bit window (the rows move up by one per step)
  auto xtok = [&](int idx0) -> uint32_t { return (idx0 >= 0 && idx0 < xLen) ? (uint32_t)xt[idx0] : 0u; };
  const uint32_t ninf_off = a.dp.ematch_ninf_off;
  // match emission of the destination (row i + 1, word w of column j + 1): -inf above the band's last diagonal and outside
  // the reference.  With that, every source cell outside the band / below the last reference row stays at -inf by itself
  // (see the header comment); no per-cell masks.
  auto emis = [&](uint32_t w, unsigned long long window, int b, int idest) -> double {
    uint32_t off = ((w & 0x7FFFu) << 5) | (((uint32_t)(window >> (2 * b)) & 3u) << 3);
    if (b > bmax || (uint32_t)(idest - 1) >= (uint32_t)xLen) off = ninf_off;
    return *(const double*)((const char*)ematch + off);
  };
  constexpr unsigned long long winMask = (B < 32 ? (1ull << (2 * B)) : 0ull) - 1ull;
  unsigned long long win = 0;
#pragma unroll
  for (int b = 0; b < B; ++b) win |= (unsigned long long)xtok(d0 + b + (yLen + rl)) << (2 * b);
  uint32_t tokNext = xtok(d0 + (yLen + rl) - 1);   // slot 0's token at the next step
  // context words: the lane's column falls by one per step; the word of column j is ctx[j - 1], loaded a step ahead
  auto ctxword = [&](int t) -> uint32_t { return ctx[max(yLen - 1 - t + rl, -100)]; };
  uint32_t wA = ctxword(0);
  // Fq[c], e[b]: this step's operands, fetched during the previous step.  Fq = the cell row of the Forward storage (packed
  // fp32: anchor, then (mat, ins, del) offsets of slot B-1 .. 0, NF / 4 sixteen-byte chunks); a chunk is reloaded for the
  // next step as soon as the slot loop (which runs from slot B-1 down) has used its last value: one buffer, a whole step of
  // distance between load and use.  Forward values of cells that do not exist (not started / finished lanes, slots above
  // the band) read as -inf: their counts are 0.
  constexpr int NF = fw_row_floats(B);
  const float4* __restrict__ fwrow = (const float4*)(a.fw + fw_off);
  auto rowptr = [&](int j) -> const float4* { return fwrow + ((uint64_t)(j - 1 + l) * (NF / 4)) * G + l; };
  // (component by component: a whole-vector select of the 12-float row of B = 3 went through scratch memory)
  auto sel4 = [](bool ok, const float4& v) -> float4 {
    const float ni = -__builtin_huge_valf();
    return make_float4(ok ? v.x : ni, ok ? v.y : ni, ok ? v.z : ni, ok ? v.w : ni);
  };
  float4 Fq[NF / 4];
  {
    const int j = yLen + rl;
    const bool ok = active && bmax >= 0 && j <= yLen;
    const float4* src = ok ? rowptr(j) : fwrow;
#pragma unroll
    for (int c = 0; c < NF / 4; ++c) { const float4 v = ldrow(src + (uint64_t)c * G); Fq[c] = sel4(ok, v); }
  }
  double e[B], insE = eins[0];   // emissions of the destination column j+1: none before the first step (the Backward values there are -inf)
#pragma unroll
  for (int b = 0; b < B; ++b) e[b] = QF_NEG_INF;

  // lanes are on their last column (end transitions) at steps 0 .. G-1 and on column 1 (start) at steps startLo .. startHi
  int startLo = active ? yLen + rl - 1 : 0x7FFFFFFF, startHi = active ? yLen + rl - 1 : -1;
  for (int o = 32; o; o >>= 1) {
    startLo = min(startLo, __shfl_xor(startLo, o));
    startHi = max(startHi, __shfl_xor(startHi, o));
  }
  startLo = __builtin_amdgcn_readfirstlane(startLo);
  startHi = __builtin_amdgcn_readfirstlane(startHi);

#pragma unroll 1
  for (int t = 0; t <= startHi; ++t) {
    const int j = yLen - (t - rl);
    const uint32_t w = wA;
    wA = ctxword(t + 1);
    const uint32_t gk = w >> 24;
    double m2m, m2i, m2d;
    if (GAPCTX) { m2m = trans[gk]; m2i = trans[Kg + gk]; m2d = trans[2 * Kg + gk]; }
    else { m2m = c_m2m; m2i = c_m2i; m2d = c_m2d; }
    const unsigned long long winCur = win;
    // the next step's destination column is this step's column
    win = ((win << 2) | tokNext) & winMask;
    tokNext = xtok(d0 + j - 2);
    const bool more = active && j - 1 >= 1 && j - 1 <= yLen;
    const bool moreF = more && bmax >= 0;
    const float4* nsrc = moreF ? rowptr(j - 1) : fwrow;
    // anchor - Fres: a cell's F - Fres is this plus its fp32 offset
    const double adF = (double)Fq[0].x - Fres;
    auto Fval = [&](int b, int st) -> double {
      const int k = fw_float_index(B, b, st);
      const float4 q = Fq[k >> 2];
      const float o = (k & 3) == 0 ? q.x : (k & 3) == 1 ? q.y : (k & 3) == 2 ? q.z : q.w;
      return b <= bmax ? adF + (double)o : QF_NEG_INF;
    };
    auto refill = [&](int b) {   // slot b's operands for the next step, and the row chunks slot b was the last to use
      e[b] = emis(w, win, b, d0 + b + j);
#pragma unroll
      for (int c = 0; c < NF / 4; ++c) {
        // chunk c holds float indices 4c .. 4c+3; slot b's last index is fw_float_index(B, b, 2).  The chunk is free once
        // that index has passed its end; chunks that reach beyond slot 0's values are free after slot 0.
        const int hiK = fw_float_index(B, b, 2), prevHiK = b == B - 1 ? 0 : fw_float_index(B, b + 1, 2);
        const bool freed = 4 * c + 3 <= hiK && 4 * c + 3 > prevHiK;
        const bool tail = b == 0 && 4 * c + 3 > hiK;
        if (freed || tail) { const float4 v = ldrow(nsrc + (uint64_t)c * G); Fq[c] = sel4(moreF, v); }
      }
    };
    // (i+1, j) for the top slot: lane l+1's slot 0 at column j, finished in the previous step
    const double hiD = dpp_from_above<G, false>(pubD);
    const double mi_e = m2i + insE, ii_e = i2i + insE;
    insE = eins[(w >> 15) & 0x1FFu];
    float pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // match-by-token[4], insert, m2m, m2i, m2d partial counts (source column j)
    float pa[4] = {0, 0, 0, 0};               // i2m, d2m, i2i, d2d
    float pc0[4] = {0, 0, 0, 0};              // start -> mat(i,1) counts by token (source "column 0")
    double loI = QF_NEG_INF;                   // Bi(i, j+1) for slot 0: lane l-1's top slot, exchanged below
    double nextD = hiD;                        // Bd(i+1, j): slot b+1 of this step, or lane l+1's slot 0
    const bool colvalid = active && j >= 1 && j <= yLen;
    const bool endStep = t < G, startStep = t >= startLo;   // wave-uniform
    if (endStep && j == yLen) gkEnd = gk;
    // One cell as a source.  EDGE: some lane of the wavefront is on its last column (end transition) or on column 1 (start
    // term); the common step is straight-line code over the B slots, so the compiler overlaps their table lookups.
    auto cell = [&](int b, auto edge) {
      constexpr bool EDGE = decltype(edge)::value;
      const int i = d0 + b + j;
      const uint32_t tokN = (uint32_t)(winCur >> (2 * b)) & 3u;   // token of row i+1
      const double BmN = Bm[b];                                   // Bm(i+1, j+1), own diagonal, previous step
      const double BiN = b > 0 ? Bi[b - 1] : loI;                 // Bi(i, j+1), diagonal d-1 (not yet overwritten)
      const double T_mm = (m2m + e[b]) + BmN, T_im = (i2m + e[b]) + BmN, T_dm = (d2m + e[b]) + BmN;
      const double T_mi = mi_e + BiN, T_ii = ii_e + BiN;
      const double T_md = m2d + nextD, T_dd = d2d + nextD;
      // accumulation order of the reference's push-style sweep (columns descending, rows descending): the
      // contribution from mat(i+1,j+1) arrives first, then ins(i,j+1), then del(i+1,j), then the end transition.
      // The table log-sum-exp is not associative at the 1e-4 level (its x >= 10 cut-off drops up to 4.5e-5 per
      // call), so the order is part of the numerical contract.
      double nbm = lseh(hs, lseh(hs, T_mm, T_mi), T_md);
      const double Fm = Fval(b, 0), Fi = Fval(b, 1), Fd = Fval(b, 2);
      if (EDGE) {
        if (colvalid && j == yLen && b <= bmax && (uint32_t)(i - 1) < (uint32_t)xLen && (local || i == xLen)) {
          const double T_me = trans[3 * Kg + gk];
          nbm = lseh(hs, nbm, T_me);
          acc_m2e += count_exp(Fm + T_me);
        }
      }
      const double nbi = lseh(hs, T_im, T_ii);
      const double nbd = lseh(hs, T_dm, T_dd);
      // NB (F - Fres) + T differs from the reference's (F + T) - Fres only in rounding
      const float c_mm = count_expf(Fm + T_mm), c_im = count_expf(Fi + T_im), c_dm = count_expf(Fd + T_dm);
      const float c_mi = count_expf(Fm + T_mi), c_ii = count_expf(Fi + T_ii);
      const float c_md = count_expf(Fm + T_md), c_dd = count_expf(Fd + T_dd);
      const float cmat = c_mm + c_im + c_dm;
      pf[0] += tokN == 0 ? cmat : 0.f; pf[1] += tokN == 1 ? cmat : 0.f;
      pf[2] += tokN == 2 ? cmat : 0.f; pf[3] += tokN == 3 ? cmat : 0.f;
      pf[4] += c_mi + c_ii;
      pf[5] += c_mm; pf[6] += c_mi; pf[7] += c_md;
      pa[0] += c_im; pa[1] += c_dm; pa[2] += c_ii; pa[3] += c_dd;
      if (EDGE) {
        if (colvalid && j == 1 && b <= bmax && (uint32_t)(i - 1) < (uint32_t)xLen && (i == 1 || local)) {  // start -> mat(i,1), src/qmodel.cpp:1448-1454
          const uint32_t tok = xt[i - 1];
          const double S = ematch[(w & 0x7FFFu) * 4u + tok] + nbm;
          const float cs = count_expf(S - Fres);
          pc0[0] += tok == 0 ? cs : 0.f; pc0[1] += tok == 1 ? cs : 0.f;
          pc0[2] += tok == 2 ? cs : 0.f; pc0[3] += tok == 3 ? cs : 0.f;
          startv = lseh(hs, startv, S);
        }
      }
      refill(b);
      Bm[b] = nbm; Bi[b] = nbi; Bd[b] = nbd;
      nextD = nbd;
      if (b == B - 1) loI = dpp_from_below<G, false>(nbi);   // lane l-1 (on column j+1) has just produced Bi of its top slot
    };
    // (Two copies of the slot loop -- a straight-line one for the common step -- let the compiler overlap the slots'
    // lookups but cost ~50 more registers: measured 17.3 ms at one wavefront per SIMD against 13.9 ms for this single loop
    // with its wave-uniform EDGE branches at two.)
#pragma unroll
    for (int b = B - 1; b >= 0; --b) {
      if (endStep || startStep) cell(b, std::true_type());
      else cell(b, std::false_type());
    }
    pubD = Bd[0];
    // the context-free transition counts accumulate in fp64 in the lane's own LDS words (conflict-free; registers are what
    // limits this kernel's occupancy)
#pragma unroll
    for (int c = 0; c < 4; ++c) unsafeAtomicAdd(&s_acc[c * 64], (double)pa[c]);
    // ---- per-column partials travel with the column: lane l+1 was on column j one step ago and hands its running sums
    // to lane l; the unit's lane 0 is the last on every column and holds the complete sums (no LDS, no barriers)
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
      const float in = dpp_f32_from_above<G>(colsum[c]);
      colsum[c] = colvalid ? in + pf[c] : 0.f;
    }
    if (colvalid) {
      if (!GAPCTX) { acc_m2m += (double)pf[5]; acc_m2i += (double)pf[6]; acc_m2d += (double)pf[7]; }
      if (startStep && j == 1) {  // start -> mat(i,1): emission counts of column 1, once per lane
        const uint32_t er = w & 0x7FFFu;
        uint32_t mk, q;
        em_row_decode(a, er, mk, q);
        if (q < (uint32_t)kNQualDev) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            fx_add(cnt + 2 * (cMat + ((uint64_t)c * Km + mk) * kNQualDev + q), wgt * (double)pc0[c]);
        }
      }
    }
    // The column's sums do not go to the accumulators from here: that was ten global atomics and ~120 instructions per step
    // issued for one lane in sixteen, 3.8 of this kernel's 15.4 ms (and three more fixed-point adds per LANE and step for the
    // context-dependent transition counts, 0.8 ms).  Lane 0 leaves them in the Forward row of this step instead -- every lane of
    // the band has just consumed it (a row is one step's cells, fetched a step ahead), and chunks 0 and 1 of it are lane 0's
    // own -- as two 16-byte stores, and k_count_flush adds them up afterwards with every lane busy and an LDS table in front
    // of the global accumulators.
    if (colvalid && l == 0) {
      float4* dst = (float4*)rowptr(j);
      dst[0] = make_float4(colsum[0], colsum[1], colsum[2], colsum[3]);
      dst[G] = make_float4(colsum[4], colsum[5], colsum[6], colsum[7]);
    }
  }
  // context-free transitions, m2e and the Backward result (start), reduced over the unit's lanes
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  double acc_i2m = s_acc[0], acc_d2m = s_acc[64], acc_i2i = s_acc[128], acc_d2d = s_acc[192];
  for (int o = 1; o < G; o <<= 1) {
    acc_i2m += __shfl_xor(acc_i2m, o, G); acc_d2m += __shfl_xor(acc_d2m, o, G);
    acc_i2i += __shfl_xor(acc_i2i, o, G); acc_d2d += __shfl_xor(acc_d2d, o, G);
    acc_m2e += __shfl_xor(acc_m2e, o, G);
    acc_m2m += __shfl_xor(acc_m2m, o, G); acc_m2i += __shfl_xor(acc_m2i, o, G); acc_m2d += __shfl_xor(acc_m2d, o, G);
    startv = lseh(hs, startv, __shfl_xor(startv, o, G));
    gkEnd = max(gkEnd, (uint32_t)__shfl_xor((int)gkEnd, o, G));
  }
  if (!GAPCTX && active && l == 0) {
    fx_add_total(cnt + 2 * (cTr + 0), wgt * acc_m2m);
    fx_add_total(cnt + 2 * (cTr + 1), wgt * acc_m2i);
    fx_add_total(cnt + 2 * (cTr + 2), wgt * acc_m2d);
  }
  if (active && l == 0) {
    fx_add_total(cnt + 2 * (cTr + 3 * Kg + gkEnd), wgt * acc_m2e);
    fx_add_total(cnt + 2 * (cTr + 4 * Kg + 0), wgt * acc_d2d);
    fx_add_total(cnt + 2 * (cTr + 4 * Kg + 1), wgt * acc_d2m);
    fx_add_total(cnt + 2 * (cTr + 4 * Kg + 2), wgt * acc_i2i);
    fx_add_total(cnt + 2 * (cTr + 4 * Kg + 3), wgt * acc_i2m);
    a.units[uid].end_val = startv;  // Backward result of this band (diagnostic: should equal Forward's)
  }
}


// ------------------------------------------------------------------------------------------------
// Row-space Forward / Backward for bands wider than the diagonal-space kernels take (-kmatchoff, or the full-envelope
// fallback of a short read against a long reference).  Geometry of k_viterbi_rows (qf_kernels.hip): one wavefront per
// unit, stripes of 64 lanes x 8 rows, lane l one column behind lane l-1 (Forward) or lane l+1 (Backward), the stripe's
// edge row handed to the next stripe through a boundary buffer.  Forward cell (i,j) of stripe s is stored at
//   stripe_off[s] + (((j - jlo + l) * 8 + b) * 3 + state) * 64 + l          (l = lane, b = row slot)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_forward_rows(FbArgs a) {
  __shared__ __attribute__((aligned(16))) double s_lseh[kLseDoubles];
  lseh_load(s_lseh, a.lse_h, threadIdx.x, 64);
  __syncthreads();
  const double* hs = s_lseh;
  constexpr int G = 64, B = 8, S = kRowStripe;
  const uint32_t uidx = blockIdx.x;
  if (uidx >= a.n_cls_units) return;
  const int l = threadIdx.x;
  const uint32_t uid = a.cls_list[uidx];
  const Unit u = a.units[uid];
  const uint32_t r = u.pair / a.n_refs, x = u.pair % a.n_refs;
  const uint64_t xb = a.ref_off[x], yb = a.read_off[r];
  const int xLen = (int)(a.ref_off[x + 1] - xb), yLen = (int)(a.read_off[r + 1] - yb);
  const int dlo = u.dlo, dhi = u.dhi;
  const RowGeom g = row_geom(dlo, dhi, xLen, yLen);
  double* base = a.fw + u.tb_off;
  unsigned long long* stripe_off = (unsigned long long*)base;
  double* bnd = base + (g.nStripes + 1);
  double* cells = base + row_fw_header(g, yLen);
  const size_t bndStride = 3ull * (yLen + 2);
  if (l == 0) {
    unsigned long long w = 0;
    for (int s = 0; s < g.nStripes; ++s) {
      int jlo, jhi;
      row_stripe_cols(g, s, dlo, dhi, yLen, jlo, jhi);
      stripe_off[s] = w;
      if (jhi >= jlo) w += (unsigned long long)(jhi - jlo + 1 + 63) * 64 * B * 3;
    }
    stripe_off[g.nStripes] = w;
  }
  for (size_t c = l; c < bndStride; c += 64) bnd[c] = QF_NEG_INF;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");

  const double i2m = a.dp.i2m, d2m = a.dp.d2m, i2i = a.dp.i2i, d2d = a.dp.d2d;
  const double* __restrict__ ematch = a.dp.ematch;
  const double* __restrict__ eins = a.dp.eins;
  const double* __restrict__ trans = a.dp.trans;
  const uint32_t Kg = a.dp.Kg;
  const bool local = a.dp.local != 0;
  const uint8_t* __restrict__ xt = a.ref_tok + xb;
  const uint32_t* __restrict__ ctx = a.ctx + yb;
  double endv = QF_NEG_INF;   // lse over rows, ascending (src/qmodel.cpp:1379-1381)
  unsigned long long woff = 0;

  for (int s = 0; s < g.nStripes; ++s) {
    int jlo, jhi;
    row_stripe_cols(g, s, dlo, dhi, yLen, jlo, jhi);
    const int i0 = g.ilo + s * S + l * B;
    const double* __restrict__ bprev = bnd + (size_t)(s & 1) * bndStride;
    double* __restrict__ bnext = bnd + (size_t)((s + 1) & 1) * bndStride;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (size_t c = l; c < bndStride; c += 64) bnext[c] = QF_NEG_INF;
    if (jhi < jlo) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      continue;
    }
    uint32_t tk[B];
#pragma unroll
    for (int b = 0; b < B; ++b) tk[b] = (i0 + b >= 1 && i0 + b <= xLen) ? xt[i0 + b - 1] : 0u;
    double M[B], I[B], D[B], endTerm[B];
#pragma unroll
    for (int b = 0; b < B; ++b) { M[b] = I[b] = D[b] = QF_NEG_INF; endTerm[b] = QF_NEG_INF; }
    double p1M = QF_NEG_INF, p1I = QF_NEG_INF, p1D = QF_NEG_INF, p2M = QF_NEG_INF, p2I = QF_NEG_INF, p2D = QF_NEG_INF;
    const int steps = jhi - jlo + 1 + G - 1;
    for (int t = 0; t < steps; ++t) {
      const int j = jlo + t - l;
      const bool colvalid = j >= jlo && j <= jhi;
      const uint32_t w = ctx[min(max(j - 1, -kCtxPad + 1), yLen + 4)];
      const uint32_t erow4 = (w & 0x7FFFu) * 4u, insrow = (w >> 15) & 0x1FFu, gk = w >> 24;
      const uint32_t gp = j > 1 ? (ctx[min(j - 2, yLen + 4)] >> 24) : 0u;
      const double m2m = trans[gp], m2i = trans[Kg + gp], m2d = trans[2 * Kg + gk];
      const double insE = eins[insrow];
      double upM = __shfl_up(p1M, 1, G), upD = __shfl_up(p1D, 1, G);
      double dgM = __shfl_up(p2M, 1, G), dgI = __shfl_up(p2I, 1, G), dgD = __shfl_up(p2D, 1, G);
      if (l == 0) {
        const int jc = min(max(j, 0), yLen + 1), jp = min(max(j - 1, 0), yLen + 1);
        upM = bprev[jc]; upD = bprev[2 * (yLen + 2) + jc];
        dgM = bprev[jp]; dgI = bprev[(yLen + 2) + jp]; dgD = bprev[2 * (yLen + 2) + jp];
      }
      double aboveM = upM, aboveD = upD;
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int i = i0 + b, dgl = i - j;
        const bool valid = colvalid && i >= 1 && i <= xLen && dgl >= dlo && dgl <= dhi;
        const double e = ematch[erow4 + tk[b]];
        const double oM = M[b], oI = I[b], oD = D[b];   // (i, j-1)
        double nm = lseh(hs, lseh(hs, dgM + m2m, dgD + d2m), dgI + i2m);
        if (j == 1 && (i == 1 || local)) nm = lseh(hs, nm, 0.0);
        nm += e;
        double ni = insE + lseh(hs, oI + i2i, oM + m2i);
        double ndl = lseh(hs, aboveD + d2d, aboveM + m2d);
        if (!valid) { nm = QF_NEG_INF; ni = QF_NEG_INF; ndl = QF_NEG_INF; }
        M[b] = nm; I[b] = ni; D[b] = ndl;
        dgM = oM; dgI = oI; dgD = oD;
        aboveM = nm; aboveD = ndl;
        if (colvalid) {
          const unsigned long long at = woff + (((unsigned long long)t * B + b) * 3) * G + l;
          cells[at] = nm; cells[at + G] = ni; cells[at + 2 * G] = ndl;
        }
        if (j == yLen && valid && (local || i == xLen)) endTerm[b] = nm + trans[3 * Kg + gk];
      }
      p2M = p1M; p2I = p1I; p2D = p1D;
      p1M = M[B - 1]; p1I = I[B - 1]; p1D = D[B - 1];
      if (colvalid && l == G - 1) { bnext[j] = p1M; bnext[(yLen + 2) + j] = p1I; bnext[2 * (yLen + 2) + j] = p1D; }
    }
    woff += (unsigned long long)steps * G * B * 3;
    for (int q = 0; q < G; ++q) {   // this stripe's end terms, rows ascending
      double v = endv;
#pragma unroll
      for (int b = 0; b < B; ++b) if (endTerm[b] > QF_NEG_INF) v = lseh(hs, v, endTerm[b]);
      endv = __shfl(l == q ? v : endv, q, G);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  }
  if (l == 0) a.units[uid].end_val = endv;
}

// Backward over a row-space unit: stripes bottom-up, columns right-to-left, lane l one column behind lane l+1; the
// arithmetic, association order and count bookkeeping of k_backward_fill.
__global__ __launch_bounds__(64) void k_backward_rows(FbArgs a) {
  __shared__ __attribute__((aligned(16))) double s_lseh[kLseDoubles];
  lseh_load(s_lseh, a.lse_h, threadIdx.x, 64);
  __syncthreads();
  const double* hs = s_lseh;
  constexpr int G = 64, B = 8, S = kRowStripe;
  extern __shared__ unsigned long long s_tr[];   // [3 * Kg][2] context-dependent transition counts of this unit (fixed point: fx_add)
  const uint32_t uidx = blockIdx.x;
  if (uidx >= a.n_cls_units) return;
  const int l = threadIdx.x, rl = G - 1 - l;
  const uint32_t uid = a.cls_list[uidx];
  const Unit u = a.units[uid];
  const double Fres = a.pair_fwd[u.pair], wgt = a.pair_weight[u.pair];
  if (!(wgt > 0.0) || !(Fres > QF_NEG_INF)) return;   // pruned pair (qmodel.cpp:2252)
  const uint32_t r = u.pair / a.n_refs, x = u.pair % a.n_refs;
  const uint64_t xb = a.ref_off[x], yb = a.read_off[r];
  const int xLen = (int)(a.ref_off[x + 1] - xb), yLen = (int)(a.read_off[r + 1] - yb);
  const int dlo = u.dlo, dhi = u.dhi;
  const RowGeom g = row_geom(dlo, dhi, xLen, yLen);
  double* base = a.fw + u.tb_off;
  const unsigned long long* stripe_off = (const unsigned long long*)base;
  double* bnd = base + (g.nStripes + 1);            // reused: [2][2][yLen+2] Backward mat / del of a stripe's first row
  const double* __restrict__ cells = base + row_fw_header(g, yLen);
  const size_t bndStride = 2ull * (yLen + 2);
  for (uint32_t c = l; c < 3 * a.dp.Kg * 2; c += 64) s_tr[c] = 0ull;
  for (size_t c = l; c < 2 * bndStride; c += 64) bnd[c] = QF_NEG_INF;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
  __builtin_amdgcn_wave_barrier();

  const double i2m = a.dp.i2m, d2m = a.dp.d2m, i2i = a.dp.i2i, d2d = a.dp.d2d;
  const double* __restrict__ ematch = a.dp.ematch;
  const double* __restrict__ eins = a.dp.eins;
  const double* __restrict__ trans = a.dp.trans;
  const uint32_t Kg = a.dp.Kg, Km = a.Km;
  const bool local = a.dp.local != 0;
  const uint8_t* __restrict__ xt = a.ref_tok + xb;
  const uint32_t* __restrict__ ctx = a.ctx + yb;
  unsigned long long* __restrict__ cnt = a.counts + 2 * (size_t)(blockIdx.x % kCountReplicas) * a.counts_stride;   // (lo, hi) per entry; contention: see kCountReplicas
  const uint64_t cIns = 0, cMat = 4ull * kNQualDev, cTr = (4ull + 4ull * Km) * kNQualDev;
  double acc_i2m = 0, acc_d2m = 0, acc_i2i = 0, acc_d2d = 0, acc_m2e = 0, startv = QF_NEG_INF;
  const uint32_t gkEnd = ctx[yLen - 1] >> 24;

  for (int s = g.nStripes - 1; s >= 0; --s) {
    int jlo, jhi;
    row_stripe_cols(g, s, dlo, dhi, yLen, jlo, jhi);
    const int i0 = g.ilo + s * S + l * B;
    const int par = (g.nStripes - 1 - s) & 1;
    const double* __restrict__ bprev = bnd + (size_t)par * bndStride;          // first row of the stripe below
    double* __restrict__ bnext = bnd + (size_t)(par ^ 1) * bndStride;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (size_t c = l; c < bndStride; c += 64) bnext[c] = QF_NEG_INF;
    if (jhi < jlo) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      continue;
    }
    const unsigned long long woff = stripe_off[s];
    uint32_t tkN[B], tk0 = (i0 >= 1 && i0 <= xLen) ? xt[i0 - 1] : 0u;   // tokens of rows i+1; tk0: row i0 itself
#pragma unroll
    for (int b = 0; b < B; ++b) tkN[b] = (i0 + b >= 0 && i0 + b < xLen) ? xt[i0 + b] : 0u;
    double Bm[B], Bi[B], Bd[B];
#pragma unroll
    for (int b = 0; b < B; ++b) Bm[b] = Bi[b] = Bd[b] = QF_NEG_INF;
    double p1M = QF_NEG_INF, p1D = QF_NEG_INF, p2M = QF_NEG_INF;   // slot 0 after the previous / the one before
    double colsum[5] = {0, 0, 0, 0, 0};
    uint32_t wNext = 0;
    const int steps = jhi - jlo + 1 + G - 1;
    for (int t = 0; t < steps; ++t) {
      const int j = jhi - (t - rl);
      const bool colvalid = j >= jlo && j <= jhi;
      const uint32_t w = ctx[min(max(j - 1, -kCtxPad + 1), yLen + 4)];
      const uint32_t gk = w >> 24;
      const double m2m = trans[gk], m2i = trans[Kg + gk], m2d = trans[2 * Kg + gk];
      const uint32_t erowN4 = (wNext & 0x7FFFu) * 4u;
      const double insEN = eins[(wNext >> 15) & 0x1FFu];
      // row below the last slot: lane l+1's first row (column j one step ago, column j+1 two steps ago) or the boundary
      double belowD = __shfl_down(p1D, 1, G), dgM = __shfl_down(p2M, 1, G);
      if (l == G - 1) {
        const int jc = min(max(j, 0), yLen + 1), jn = min(max(j + 1, 0), yLen + 1);
        belowD = bprev[(yLen + 2) + jc];
        dgM = bprev[jn];
      }
      double pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pc0[4] = {0, 0, 0, 0};
#pragma unroll
      for (int b = B - 1; b >= 0; --b) {
        const int i = i0 + b, dgl = i - j;
        const bool valid = colvalid && i >= 1 && i <= xLen && dgl >= dlo && dgl <= dhi;
        const double eN = ematch[erowN4 + tkN[b]];
        const double oM = Bm[b];            // Bm(i, j+1): the diagonal neighbour of row i-1
        const double BmN = dgM;             // Bm(i+1, j+1)
        const double BiN = Bi[b];           // Bi(i, j+1)
        const double BdN = belowD;          // Bd(i+1, j)
        const double T_mm = (m2m + eN) + BmN, T_im = (i2m + eN) + BmN, T_dm = (d2m + eN) + BmN;
        const double T_mi = (m2i + insEN) + BiN, T_ii = (i2i + insEN) + BiN;
        const double T_md = m2d + BdN, T_dd = d2d + BdN;
        const bool isEnd = j == yLen && (local || i == xLen);
        const double T_me = isEnd ? trans[3 * Kg + gk] : QF_NEG_INF;
        double nbm = lseh(hs, lseh(hs, T_mm, T_mi), T_md);
        if (isEnd) nbm = lseh(hs, nbm, T_me);
        double nbi = lseh(hs, T_im, T_ii);
        double nbd = lseh(hs, T_dm, T_dd);
        if (!valid) { nbm = QF_NEG_INF; nbi = QF_NEG_INF; nbd = QF_NEG_INF; }
        if (valid) {
          const unsigned long long at = woff + (((unsigned long long)(j - jlo + l) * B + b) * 3) * G + l;
          const double Fm = cells[at] - Fres, Fi = cells[at + G] - Fres, Fd = cells[at + 2 * G] - Fres;
          const double c_mm = wgt * count_exp(Fm + T_mm), c_im = wgt * count_exp(Fi + T_im), c_dm = wgt * count_exp(Fd + T_dm);
          const double c_mi = wgt * count_exp(Fm + T_mi), c_ii = wgt * count_exp(Fi + T_ii);
          const double c_md = wgt * count_exp(Fm + T_md), c_dd = wgt * count_exp(Fd + T_dd);
          const double cmat = c_mm + c_im + c_dm;
          const uint32_t tokN = tkN[b];
          pc[0] += tokN == 0 ? cmat : 0.0; pc[1] += tokN == 1 ? cmat : 0.0;
          pc[2] += tokN == 2 ? cmat : 0.0; pc[3] += tokN == 3 ? cmat : 0.0;
          pc[4] += c_mi + c_ii;
          pc[5] += c_mm; pc[6] += c_mi; pc[7] += c_md;
          acc_i2m += c_im; acc_d2m += c_dm; acc_i2i += c_ii; acc_d2d += c_dd;
          if (isEnd) acc_m2e += wgt * count_exp(Fm + T_me);
          if (j == 1 && (i == 1 || local)) {
            const uint32_t tok = b > 0 ? tkN[b - 1] : tk0;
            const double Sv = ematch[(w & 0x7FFFu) * 4u + tok] + nbm;
            const double cs = wgt * count_exp(Sv - Fres);
            pc0[0] += tok == 0 ? cs : 0.0; pc0[1] += tok == 1 ? cs : 0.0;
            pc0[2] += tok == 2 ? cs : 0.0; pc0[3] += tok == 3 ? cs : 0.0;
            startv = lseh(hs, startv, Sv);
          }
        }
        Bm[b] = nbm; Bi[b] = nbi; Bd[b] = nbd;
        dgM = oM;
        belowD = nbd;
      }
      p2M = p1M;
      p1M = Bm[0]; p1D = Bd[0];
      if (colvalid && l == 0) { bnext[j] = p1M; bnext[(yLen + 2) + j] = p1D; }
      {
        double in[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          in[c] = __shfl_down(colsum[c], 1, G);
          if (l == G - 1) in[c] = 0.0;
        }
#pragma unroll
        for (int c = 0; c < 5; ++c) colsum[c] = colvalid ? in[c] + pc[c] : 0.0;
      }
      if (colvalid) {
        fx_add(s_tr + 2 * gk, pc[5]);
        fx_add(s_tr + 2 * (Kg + gk), pc[6]);
        fx_add(s_tr + 2 * (2 * Kg + gk), pc[7]);
        if (j == 1) {
          const uint32_t er = w & 0x7FFFu;
        uint32_t mk, q;
        em_row_decode(a, er, mk, q);
          if (q < (uint32_t)kNQualDev) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              fx_add(cnt + 2 * (cMat + ((uint64_t)c * Km + mk) * kNQualDev + q), pc0[c]);
          }
        }
        if (l == 0 && j < yLen) {
          const uint32_t er = wNext & 0x7FFFu;
        uint32_t mk, q;
        em_row_decode(a, er, mk, q);
          const uint32_t ytok = ((wNext >> 15) & 0x1FFu) / (kNQualDev + 1);
          if (q < (uint32_t)kNQualDev) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              fx_add(cnt + 2 * (cMat + ((uint64_t)c * Km + mk) * kNQualDev + q), colsum[c]);
            fx_add(cnt + 2 * (cIns + (uint64_t)ytok * kNQualDev + q), colsum[4]);
          }
        }
      }
      wNext = w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  }
  for (int o = 1; o < G; o <<= 1) {
    acc_i2m += __shfl_xor(acc_i2m, o, G); acc_d2m += __shfl_xor(acc_d2m, o, G);
    acc_i2i += __shfl_xor(acc_i2i, o, G); acc_d2d += __shfl_xor(acc_d2d, o, G);
    acc_m2e += __shfl_xor(acc_m2e, o, G);
    startv = lseh(hs, startv, __shfl_xor(startv, o, G));
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  for (uint32_t c = l; c < 3 * Kg; c += 64) fx_add_words(cnt + 2 * (cTr + c), s_tr[2 * c], s_tr[2 * c + 1]);
  if (l == 0) {
    fx_add_total(cnt + 2 * (cTr + 3 * Kg + gkEnd), acc_m2e);
    fx_add_total(cnt + 2 * (cTr + 4 * Kg + 0), acc_d2d);
    fx_add_total(cnt + 2 * (cTr + 4 * Kg + 1), acc_d2m);
    fx_add_total(cnt + 2 * (cTr + 4 * Kg + 2), acc_i2i);
    fx_add_total(cnt + 2 * (cTr + 4 * Kg + 3), acc_i2m);
    a.units[uid].end_val = startv;
  }
}


// ------------------------------------------------------------------------------------------------
// The column sums k_backward_fill left in the Forward rows (lane 0's chunks 0 and 1 of row j - 1: match counts by reference
// token[4], insert count, then the m2m, m2i, m2d counts of source column j) -> the count accumulators.  A workgroup takes
// `upw` bands of the class, a thread per column, and ONE SLICE of the match-emission rows (`rps` rows: what fits its LDS; the
// order-3 model of the bench has 2 560 rows in use, four slices): the terms of its slice meet in a fixed-point table in LDS,
// and each entry the workgroup touched goes to the global accumulators once -- a few million global atomics per launch
// instead of Backward's 173 M.  A column's context word says whose it is, so a record is read by the one workgroup that adds
// it; slice 0 also takes the insert and context-dependent transition counts.  The terms are the same fixed-point ones as
// ever (wgt x the fp32 column sum, fx_add): the totals do not depend on the order or on how the batch was cut.
// rps == 0: straight to the global accumulators (tables too large to slice usefully).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kFlushThreads = 1024;   // the loop is a chain of dependent loads per column (context word -> record): wavefronts hide it, two workgroups fill a CU's 32
constexpr uint32_t kFlushLdsMax = 46 * 1024, kFlushMaxSlices = 16, kFlushMaxUnits = 128;   // (kFlushMaxUnits: a power of two)
__global__ __launch_bounds__(kFlushThreads) void k_count_flush(FbArgs a, int G, int row_chunks, int gapctx, uint32_t upw, uint32_t n_slices, uint32_t rps) {
  extern __shared__ unsigned long long s_tab[];   // [rps x 4 match | 4 x 94 insert | 3 Kg transitions][2 words]
  const uint32_t Kg = a.dp.Kg, Km = a.Km;
  const uint32_t nI = 4 * kNQualDev, nT = gapctx ? 3 * Kg : 0;
  const uint32_t slice = rps ? blockIdx.x % n_slices : 0, grp = rps ? blockIdx.x / n_slices : blockIdx.x;
  const uint32_t erLo = slice * rps, nE = rps * 4, nAll = nE + (slice == 0 ? nI + nT : 0);
  const bool tab = rps != 0, rest = slice == 0;   // rest: this workgroup adds the insert / transition counts
  if (tab) {
    for (uint32_t k = threadIdx.x; k < 2 * nAll; k += kFlushThreads) s_tab[k] = 0ull;
    __syncthreads();
  }
  unsigned long long* __restrict__ cnt = a.counts + 2 * (size_t)(blockIdx.x % kCountReplicas) * a.counts_stride;
  const uint64_t cIns = 0, cMat = 4ull * kNQualDev, cTr = (4ull + 4ull * Km) * kNQualDev;
  // the workgroup's bands: offsets, weights and a prefix of their column counts in LDS, then one flat loop over all the columns
  // (a loop per band had every thread walk the same chain of dependent loads -- list, unit, offsets, weight -- upw times over)
  __shared__ uint64_t s_yb[kFlushMaxUnits], s_fw[kFlushMaxUnits];
  __shared__ double s_wgt[kFlushMaxUnits];
  __shared__ uint32_t s_pre[kFlushMaxUnits + 1];
  if (threadIdx.x < upw) {
    const uint32_t uidx = grp * upw + threadIdx.x;
    uint32_t cols = 0;
    if (uidx < a.n_cls_units) {
      const Unit u = a.units[a.cls_list[uidx]];
      const uint32_t r = u.pair / a.n_refs;
      const uint64_t yb = a.read_off[r];
      s_yb[threadIdx.x] = yb;
      s_fw[threadIdx.x] = u.tb_off;
      s_wgt[threadIdx.x] = a.pair_weight[u.pair];
      if (u.staged) cols = (uint32_t)(a.read_off[r + 1] - yb);   // (not staged: pruned pair or negligible band, Backward did not run)
    }
    s_pre[threadIdx.x + 1] = cols;
  }
  if (threadIdx.x == 0) s_pre[0] = 0;
  __syncthreads();
  if (threadIdx.x == 0) for (uint32_t q = 0; q < upw; ++q) s_pre[q + 1] += s_pre[q];
  __syncthreads();
  const uint32_t total = s_pre[upw];
  for (uint32_t cidx = threadIdx.x; cidx < total; cidx += kFlushThreads) {
    uint32_t q = 0;
    for (uint32_t step = kFlushMaxUnits / 2; step; step >>= 1)
      if (q + step < upw && s_pre[q + step] <= cidx) q += step;
    {
      const int j = 1 + (int)(cidx - s_pre[q]), yLen = (int)(s_pre[q + 1] - s_pre[q]);
      const double wgt = s_wgt[q];
      const float4* __restrict__ fwrow = (const float4*)(a.fw + s_fw[q]);
      const uint32_t* __restrict__ ctx = a.ctx + s_yb[q];
      const float4* rec = fwrow + (uint64_t)(j - 1) * row_chunks * G;
      float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rest) r1 = rec[G];
      if (rest && gapctx) {
        const uint32_t gk = ctx[j - 1] >> 24;   // context of source column j
        if (tab) {
          fx_add(s_tab + 2 * (nE + nI + gk), wgt * (double)r1.y);
          fx_add(s_tab + 2 * (nE + nI + Kg + gk), wgt * (double)r1.z);
          fx_add(s_tab + 2 * (nE + nI + 2 * Kg + gk), wgt * (double)r1.w);
        } else {
          fx_add(cnt + 2 * (cTr + gk), wgt * (double)r1.y);
          fx_add(cnt + 2 * (cTr + Kg + gk), wgt * (double)r1.z);
          fx_add(cnt + 2 * (cTr + 2 * Kg + gk), wgt * (double)r1.w);
        }
      }
      if (j < yLen) {   // emission rows belong to the destination column j + 1 (context word index j); column yLen has none
        const uint32_t wN = ctx[j], er = wN & 0x7FFFu;
        uint32_t mk, qq;
        em_row_decode(a, er, mk, qq);
        if (qq < (uint32_t)kNQualDev) {
          if (!tab || er - erLo < rps) {
            const float4 r0 = rec[0];
            const float m[4] = {r0.x, r0.y, r0.z, r0.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (tab) fx_add(s_tab + 2 * ((er - erLo) * 4 + c), wgt * (double)m[c]);
              else fx_add(cnt + 2 * (cMat + ((uint64_t)c * Km + mk) * kNQualDev + qq), wgt * (double)m[c]);
            }
          }
          if (rest) {
            const uint32_t ytok = ((wN >> 15) & 0x1FFu) / (kNQualDev + 1);
            if (tab) fx_add(s_tab + 2 * (nE + ytok * kNQualDev + qq), wgt * (double)r1.x);
            else fx_add(cnt + 2 * (cIns + (uint64_t)ytok * kNQualDev + qq), wgt * (double)r1.x);
          }
        }
      }
    }
  }
  if (tab) {
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < nAll; e += kFlushThreads) {
      const unsigned long long w0 = s_tab[2 * e], w1 = s_tab[2 * e + 1];
      if (!(w0 | w1)) continue;
      uint64_t idx;
      if (e < nE) {
        uint32_t mk, qq;
        em_row_decode(a, erLo + (e >> 2), mk, qq);
        idx = cMat + ((uint64_t)(e & 3u) * Km + mk) * kNQualDev + qq;
      } else if (e < nE + nI) idx = cIns + (e - nE);
      else idx = cTr + (e - nE - nI);
      fx_add_words(cnt + 2 * idx, w0, w1);
    }
  }
}
static void launch_count_flush(const FbArgs& a, int G, int B, hipStream_t s) {
  const int gap = a.dp.Kg > 1;
  const uint32_t rows = a.dp.ematch_ninf_off / 32, rest = 4 * kNQualDev + (gap ? 3 * a.dp.Kg : 0);
  // rows per slice: what the LDS budget leaves beside the insert / transition entries (64 bytes per row)
  const uint32_t budget = a.flush_lds ? a.flush_lds : kFlushLdsMax;   // (tests: a small table = many slices; 1 = none)
  uint32_t rps = rest * 16 < budget ? (budget - rest * 16) / 64 : 0;
  rps = std::min(rps, rows);
  uint32_t n_slices = rps ? (rows + rps - 1) / rps : 1;
  if (n_slices > kFlushMaxSlices) { rps = 0; n_slices = 1; }
  else if (rps) rps = (rows + n_slices - 1) / n_slices;   // equal slices
  // bands per workgroup: what the global accumulators take per second is the bound (~10 G atomics/s measured: a workgroup's table
  // is ~6 000 of them), so as many bands as still leave one round of workgroups for the chip (two fit a CU)
  const uint32_t upw = std::max<uint32_t>(4, std::min<uint32_t>(kFlushMaxUnits, (uint32_t)((uint64_t)a.n_cls_units * n_slices / 512)));
  const uint32_t blocks = (a.n_cls_units + upw - 1) / upw * n_slices;
  const size_t lds = rps ? ((size_t)rps * 4 + rest) * 16 : 0;
  hipLaunchKernelGGL(k_count_flush, dim3(blocks), dim3(kFlushThreads), lds, s, a, G, fw_row_floats(B) / 4, gap, upw, n_slices, rps);
}

// dynamic LDS of the diagonal-space kernels: lse pieces | transition scores | (Backward) the lanes' context-free transition
// counts | (EMLDS) emission tables.  The emission tables go to LDS when three workgroups still fit a CU (<= 52 KB each).
static uint32_t fb_lds_bytes(const FbArgs& a, bool backward, bool& emlds) {
  const uint32_t Kg = a.dp.Kg;
  uint32_t d = kLseDoubles + ((4 * Kg + 1) & ~1u);
  if (backward) d += 4 * 4 * 64;
  const uint32_t em = a.dp.ematch_ninf_off / 8 + 4 + kInsRows;
  emlds = (d + em) * 8 <= (a.lds_limit ? a.lds_limit : 48u * 1024u);
  return (d + (emlds ? em : 0)) * 8;
}
template <int G, int B>
static void launch_fwd_gb(const FbArgs& a, hipStream_t s) {
  const uint32_t upw = 64 / G, waves = (a.n_cls_units + upw - 1) / upw, blocks = (waves + 3) / 4;
  bool emlds;
  const uint32_t lds = fb_lds_bytes(a, false, emlds);
  const bool gap = a.dp.Kg > 1;
  if (emlds && lds > 48 * 1024) {   // above the default cap of dynamic LDS
    if (gap) (void)hipFuncSetAttribute((const void*)k_forward_fill<G, B, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else (void)hipFuncSetAttribute((const void*)k_forward_fill<G, B, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (gap && emlds) hipLaunchKernelGGL((k_forward_fill<G, B, true, true>), dim3(blocks), dim3(256), lds, s, a);
  else if (gap) hipLaunchKernelGGL((k_forward_fill<G, B, true, false>), dim3(blocks), dim3(256), lds, s, a);
  else if (emlds) hipLaunchKernelGGL((k_forward_fill<G, B, false, true>), dim3(blocks), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((k_forward_fill<G, B, false, false>), dim3(blocks), dim3(256), lds, s, a);
}
template <int G, int B>
static void launch_bwd_gb(const FbArgs& a, hipStream_t s) {
  const uint32_t upw = 64 / G, waves = (a.n_cls_units + upw - 1) / upw, blocks = (waves + 3) / 4;
  bool emlds;
  const uint32_t lds = fb_lds_bytes(a, true, emlds);
  const bool gap = a.dp.Kg > 1;
  if (emlds && lds > 48 * 1024) {   // above the default cap of dynamic LDS
    if (gap) (void)hipFuncSetAttribute((const void*)k_backward_fill<G, B, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    else (void)hipFuncSetAttribute((const void*)k_backward_fill<G, B, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (gap && emlds) hipLaunchKernelGGL((k_backward_fill<G, B, true, true>), dim3(blocks), dim3(256), lds, s, a);
  else if (gap) hipLaunchKernelGGL((k_backward_fill<G, B, true, false>), dim3(blocks), dim3(256), lds, s, a);
  else if (emlds) hipLaunchKernelGGL((k_backward_fill<G, B, false, true>), dim3(blocks), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((k_backward_fill<G, B, false, false>), dim3(blocks), dim3(256), lds, s, a);
  launch_count_flush(a, G, B, s);
}
#define QF_FB_DISPATCH(FN)                         \
  switch (cls) {                                   \
    case 1: FN<16, 2>(a, s); break;                \
    case 2: FN<16, 3>(a, s); break;                \
    case 3: FN<16, 4>(a, s); break;                \
    case 4: FN<16, 5>(a, s); break;                \
    case 5: FN<16, 6>(a, s); break;                \
    case 6: FN<16, 8>(a, s); break;                \
    case 7: FN<64, 3>(a, s); break;                \
    case 8: FN<64, 4>(a, s); break;                \
    case 9: FN<64, 6>(a, s); break;                \
    case 10: FN<64, 8>(a, s); break;               \
    case 11: FN<64, 12>(a, s); break;              \
    case 12: FN<64, 16>(a, s); break;              \
    case 14: FN<32, 3>(a, s); break;               \
  }
void launch_forward_fill(int cls, const FbArgs& a, hipStream_t s) {
  if (!a.n_cls_units) return;
  if (cls == 0 && a.no_band_shortcuts) { launch_fwd_gb<16, 2>(a, s); return; }
  if (cls == 0) { hipLaunchKernelGGL(k_forward_single, dim3((a.n_cls_units + 255) / 256), dim3(256), 0, s, a); return; }
  if (cls == kRowClass) { hipLaunchKernelGGL(k_forward_rows, dim3(a.n_cls_units), dim3(64), 0, s, a); return; }
  QF_FB_DISPATCH(launch_fwd_gb)
}
void launch_backward_fill(int cls, const FbArgs& a, hipStream_t s) {
  if (!a.n_cls_units) return;
  if (cls == 0) { launch_bwd_gb<16, 2>(a, s); return; }   // single diagonals keep the (16,2) Forward layout
  if (cls == kRowClass) { hipLaunchKernelGGL(k_backward_rows, dim3(a.n_cls_units), dim3(64), (size_t)3 * a.dp.Kg * 16, s, a); return; }
  QF_FB_DISPATCH(launch_bwd_gb)
}
// Sum of the replicas of every accumulator (exact: 128-bit integer addition of word 1 2^32 + word 0), as 64.64 fixed-point
// (low, high) words and as a double
__global__ void k_sum_count_replicas(const unsigned long long* fx, uint32_t n, uint64_t stride, unsigned long long* out_fx, double* out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned __int128 tot = 0;
  for (int r = 0; r < kCountReplicas; ++r) {
    const unsigned long long w0 = fx[2 * ((size_t)r * stride + i)], w1 = fx[2 * ((size_t)r * stride + i) + 1];
    tot += ((unsigned __int128)w1 << 32) + w0;
  }
  const unsigned long long lo = (unsigned long long)tot, hi = (unsigned long long)(tot >> 64);
  out_fx[2 * (size_t)i] = lo;
  out_fx[2 * (size_t)i + 1] = hi;
  out[i] = (double)hi + (double)lo * 5.421010862427522170e-20;   // 2^-64
}
void launch_sum_count_replicas(const unsigned long long* fx, uint32_t n, uint64_t stride, unsigned long long* out_fx, double* out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_sum_count_replicas, dim3((n + 255) / 256), dim3(256), 0, s, fx, n, stride, out_fx, out);
}
void launch_pair_forward(const FinalArgs& a, const double* lse, hipStream_t s) {
  if (a.n_pairs) hipLaunchKernelGGL(k_pair_forward, dim3((a.n_pairs + 255) / 256), dim3(256), 0, s, a, lse);
}
void launch_count_plan(const CountPlanArgs& a, hipStream_t s) {
  if (a.n_reads) hipLaunchKernelGGL(k_count_plan, dim3((a.n_reads + 63) / 64), dim3(64), 0, s, a);
}

}  // namespace qf
