// qf_fb.hip — Forward and Backward fills with E-step counts (the `quaff train` / `count` hot path):
// QuaffForwardMatrix ctor src/qmodel.cpp:1343-1391, QuaffBackwardMatrix ctor + transCount :1393-1510,
// QuaffCountingTask::run :2238-2271.  Same skewed G x B wavefront as the Viterbi fill (qf_kernels.hip); max
// is replaced by the reference's table log-sum-exp (src/logsumexp.cpp:34-103: 1e-4-step table, linear
// interpolation, cut-off at 10), whose table is built on the host and shared.  The Forward matrix IS
// materialised (24 B/cell, step-major so every store/load is a coalesced segment); Backward re-reads it and
// never stores its own matrix.  Results are compared with 1e-4 relative tolerance (north_star), so sums may
// be re-associated: per-column count partials are combined in an LDS ring and flushed with fp64 atomics.
#include <hip/hip_runtime.h>

#include "qf_kernels.hpp"

namespace qf {

#define QF_NEG_INF (-__builtin_huge_val())
#ifndef QF_BWD_WAVES
#define QF_BWD_WAVES 3
#endif

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
__device__ __forceinline__ double count_exp(double x) { return (double)__expf((float)x); }


// Forward / Backward are compared at 1e-4 relative, so their log(1 + exp(-x)) need not be the reference's table
// interpolant bit for bit.  The 800 KB table is a 64-way L2 gather per call (a third of Forward's time, measured); the
// same function as a cubic Hermite spline on a 1/64 grid (641 nodes of value and slope, 10 KB) sits in LDS.  It is
// within 2e-11 of log1p(exp(-x)); the reference's own 1e-4-step linear interpolant is within 3e-10 of it.  The x >= 10
// cut-off and the a == b rule are the reference's (src/logsumexp.cpp:34-50, :84-103).
constexpr int kLseNodes = 641;
__device__ __forceinline__ void lseh_load(double* s_h, const double* __restrict__ g_h, int tid, int nthreads) {
  for (int k = tid; k < 2 * kLseNodes; k += nthreads) s_h[k] = g_h[k];
}
__device__ __forceinline__ double lseh(const double* hs, double a, double b) {
  // branch-free: the spline is evaluated at a clamped argument and discarded when the cut-off (or -inf - -inf = NaN)
  // applies, so the several calls of a cell overlap instead of each taking its own divergent branch
  const double mx = fmax(a, b), mn = fmin(a, b);
  const double diff = mx - mn;
  const double u = fmin(diff * 64.0, 639.984375);   // NaN -> the bound
  const int n = (int)u;
  const double t = u - (double)n, s = 1.0 - t;
  const double g0 = hs[2 * n], d0 = hs[2 * n + 1], g1 = hs[2 * n + 2], d1 = hs[2 * n + 3];
  const double t2 = t * t, s2 = s * s;
  const double p = (g0 * (1.0 + 2.0 * t) + d0 * (t * (1.0 / 64.0))) * s2 + (g1 * (3.0 - 2.0 * t) - d1 * (s * (1.0 / 64.0))) * t2;
  return mx + (diff < 10.0 ? p : (a == b ? hs[0] : 0.0));
}

template <int G, int B>
__global__ __launch_bounds__(256) void k_forward_fill(FbArgs a) {
  __shared__ double s_lseh[2 * kLseNodes];
  lseh_load(s_lseh, a.lse_h, threadIdx.x, 256);
  __syncthreads();
  const double* hs = s_lseh;
  constexpr int UPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int grp = lane / G, l = lane % G;
  const uint32_t uidx = wave * UPW + grp;
  const bool active = uidx < a.n_cls_units;
  uint32_t uid = 0;
  int dlo = 0, dhi = -1, xLen = 0, yLen = 0;
  uint64_t xb = 0, yb = 0, fw_off = 0;
  if (active) {
    uid = a.cls_list[uidx];
    const Unit u = a.units[uid];
    const uint32_t r = u.pair / a.n_refs, x = u.pair % a.n_refs;
    xb = a.ref_off[x]; xLen = (int)(a.ref_off[x + 1] - xb);
    yb = a.read_off[r]; yLen = (int)(a.read_off[r + 1] - yb);
    dlo = u.dlo; dhi = u.dhi; fw_off = u.tb_off;
  }
  int T = active ? yLen + G - 1 : 0;
  for (int o = 32; o; o >>= 1) T = max(T, __shfl_xor(T, o));
  const int d0 = dlo + l * B;
  const double i2m = a.dp.i2m, d2m = a.dp.d2m, i2i = a.dp.i2i, d2d = a.dp.d2d;
  const double* __restrict__ ematch = a.dp.ematch;
  const double* __restrict__ eins = a.dp.eins;
  const double* __restrict__ trans = a.dp.trans;
  const uint32_t Kg = a.dp.Kg;
  const bool local = a.dp.local != 0;
  const uint8_t* __restrict__ xt = a.ref_tok + xb;
  const uint32_t* __restrict__ ctx = a.ctx + yb;
  double* __restrict__ fw = a.fw + fw_off;

  double M[B], I[B], D[B];
#pragma unroll
  for (int b = 0; b < B; ++b) M[b] = I[b] = D[b] = QF_NEG_INF;
  double pubM = QF_NEG_INF, pubD = QF_NEG_INF;
  double endTerm[B];  // mat(i,yLen) + m2e for this lane's rows of the last column
#pragma unroll
  for (int b = 0; b < B; ++b) endTerm[b] = QF_NEG_INF;
  uint32_t gkPrev = 0;
  // reference tokens of the lane's B rows as a sliding 2-bit window: the rows move down by one per step, so one new token
  // per step (fetched a step ahead) instead of B byte gathers
  auto xtok = [&](int idx0) -> uint32_t { return (idx0 >= 0 && idx0 < xLen) ? (uint32_t)xt[idx0] : 0u; };
  unsigned long long win = 0;
#pragma unroll
  for (int b = 0; b < B; ++b) win |= (unsigned long long)xtok(d0 + b + (0 - l + 1) - 1) << (2 * b);
  uint32_t tokNext = xtok(d0 + B - 1 + (0 - l + 1));   // slot B-1's token at the next step
  for (int t = 0; t < T; ++t) {
    const int j = t - l + 1;
    const bool colvalid = active && j >= 1 && j <= yLen;
    const uint32_t w = ctx[min(max(j - 1, -kCtxPad + 1), yLen + 4)];
    const uint32_t erow4 = (w & 0x7FFFu) * 4u, insrow = (w >> 15) & 0x1FFu, gk = w >> 24;
    const uint32_t gp = j > 1 ? gkPrev : 0u;  // yIndelKmer is padded with a leading 0 (qmodel.cpp:1322)
    const double m2m = trans[gp], m2i = trans[Kg + gp], m2d = trans[2 * Kg + gk];
    gkPrev = gk;
    const double insE = eins[insrow];
    double lowM = __shfl_up(pubM, 1, G), lowD = __shfl_up(pubD, 1, G);
    if (l == 0) { lowM = QF_NEG_INF; lowD = QF_NEG_INF; }
    double upM = 0, upI = 0, prevM = lowM, prevD = lowD;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const int d = d0 + b, i = d + j;
      const bool valid = colvalid && d <= dhi && i >= 1 && i <= xLen;
      const uint32_t tok = (uint32_t)(win >> (2 * b)) & 3u;
      const double e = ematch[erow4 + tok];
      // mat(i,j) = lse(lse(mat' + m2m, del' + d2m), ins' + i2m) [lse with start at column 1] + emit
      double nm = lseh(hs, lseh(hs, M[b] + m2m, D[b] + d2m), I[b] + i2m);
      if (j == 1 && (i == 1 || local)) nm = lseh(hs, nm, 0.0);
      nm += e;
      double srcM, srcI;
      if (b + 1 < B) { srcM = M[b + 1]; srcI = I[b + 1]; } else { srcM = upM; srcI = upI; }
      double ni = insE + lseh(hs, srcI + i2i, srcM + m2i);
      double ndl = lseh(hs, prevD + d2d, prevM + m2d);
      if (!valid) { nm = QF_NEG_INF; ni = QF_NEG_INF; ndl = QF_NEG_INF; }
      M[b] = nm; I[b] = ni; D[b] = ndl;
      prevM = nm; prevD = ndl;
      if (colvalid) {
        const uint64_t base = ((uint64_t)t * B + b) * 3 * G + l;
        fw[base] = nm; fw[base + G] = ni; fw[base + 2 * G] = ndl;
      }
      if (j == yLen && valid && (local || i == xLen)) endTerm[b] = nm + trans[3 * Kg + gk];
      if (b == 0) {
        upM = __shfl_down(nm, 1, G); upI = __shfl_down(ni, 1, G);
        if (l == G - 1) { upM = QF_NEG_INF; upI = QF_NEG_INF; }
      }
    }
    pubM = prevM; pubD = prevD;
    win = (win >> 2) | ((unsigned long long)tokNext << (2 * (B - 1)));
    tokNext = xtok(d0 + B - 1 + j + 1);
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
  __shared__ double s_lseh[2 * kLseNodes];
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
  double* __restrict__ fw = a.fw + fw_off;
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
    if (colvalid) {
      const uint64_t base = (uint64_t)(j - 1) * B * 3 * G;   // step t = j - 1 of lane 0, slot 0
      fw[base] = nm; fw[base + G] = QF_NEG_INF; fw[base + 2 * G] = QF_NEG_INF;
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
    const FillClass fc = fill_class(fb_class((int)u.cls));
    const double* __restrict__ fw = a.fw + u.tb_off;
    for (int d = u.dlo; d <= u.dhi; ++d) {
      const int i = d + yLen;
      if (i < 1 || i > xLen || !(a.local || i == xLen)) continue;
      const int l = (d - u.dlo) / fc.B, b = (d - u.dlo) % fc.B;
      const double m = fw[((uint64_t)(yLen - 1 + l) * fc.B + b) * 3 * fc.G + l];   // mat(i, yLen): step yLen-1+l of lane l, slot b
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

// Backward sweep.  Lane l handles column j = yLen - (t - (G-1-l)); slots run from high diagonal to low, so
//   (i+1,j+1): same diagonal, previous step (own registers)
//   (i+1,j  ): diagonal d+1, same column   (own slot b+1 this step, or lane l+1's slot 0 from the previous step)
//   (i,  j+1): diagonal d-1, next column   (own slot b-1 from the previous step, or lane l-1's last slot, which that
//                                           lane finishes first in this very step)
// Each cell is treated as a SOURCE: its Backward values are the lse of (transition + emission + Backward of the
// destination), and the expected count of each of those transitions is exp(F_src + term - F_result) — the same
// operands, in the same association, as transCount (src/qmodel.cpp:1504-1510).
template <int G, int B>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(B <= 5 ? QF_BWD_WAVES : 1))) void k_backward_fill(FbArgs a) {
  __shared__ double s_lseh[2 * kLseNodes];
  lseh_load(s_lseh, a.lse_h, threadIdx.x, 256);
  __syncthreads();
  const double* hs = s_lseh;
  constexpr int UPW = 64 / G;
  // context-dependent transition counts (m2m / m2i / m2d by indel context): a handful of addresses that every column of
  // every band would hit with a global atomic; they are summed per wavefront in LDS (registers when there is one context)
  // and flushed once at the end
  extern __shared__ double s_tr_all[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int grp = lane / G, l = lane % G, rl = G - 1 - l;
  const uint32_t uidx = wave * UPW + grp;
  bool active = uidx < a.n_cls_units;
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
  int T = active ? yLen + G - 1 : 0;
  for (int o = 32; o; o >>= 1) T = max(T, __shfl_xor(T, o));
  if (T == 0) return;
  double* s_tr = s_tr_all + (size_t)wv * 3 * a.dp.Kg;
  if (a.dp.Kg > 1) for (uint32_t c = lane; c < 3 * a.dp.Kg; c += 64) s_tr[c] = 0.0;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();

  const int d0 = dlo + l * B;
  const double i2m = a.dp.i2m, d2m = a.dp.d2m, i2i = a.dp.i2i, d2d = a.dp.d2d;
  const double* __restrict__ ematch = a.dp.ematch;
  const double* __restrict__ eins = a.dp.eins;
  const double* __restrict__ trans = a.dp.trans;
  const uint32_t Kg = a.dp.Kg, Km = a.Km;
  const bool local = a.dp.local != 0;
  const uint8_t* __restrict__ xt = a.ref_tok + xb;
  const uint32_t* __restrict__ ctx = a.ctx + yb;
  const double* __restrict__ fw = a.fw + fw_off;
  double* __restrict__ cnt = a.counts + (size_t)(blockIdx.x % kCountReplicas) * a.counts_stride;   // contention: see kCountReplicas
  const uint64_t cIns = 0, cMat = 4ull * kNQualDev, cTr = (4ull + 4ull * Km) * kNQualDev;

  double Bm[B], Bi[B], Bd[B];   // Backward values of this lane's diagonals at the column of its previous step
#pragma unroll
  for (int b = 0; b < B; ++b) Bm[b] = Bi[b] = Bd[b] = QF_NEG_INF;
  double pubD = QF_NEG_INF;     // slot 0's del after this lane's latest step (for lane l-1)
  double acc_i2m = 0, acc_d2m = 0, acc_i2i = 0, acc_d2d = 0, acc_m2e = 0, startv = QF_NEG_INF;
  double acc_m2m = 0, acc_m2i = 0, acc_m2d = 0;   // Kg == 1 only
  double colsum[5] = {0, 0, 0, 0, 0};             // running match-by-token[4] / insert sums of the column this lane is on
  uint32_t wNext = 0;           // context word of column j+1 (this lane's previous step)
  uint32_t gkEnd = 0;

  // tokens of rows i+1 of the lane's B slots as a sliding 2-bit window (the rows move up by one per step)
  auto xtok = [&](int idx0) -> uint32_t { return (idx0 >= 0 && idx0 < xLen) ? (uint32_t)xt[idx0] : 0u; };
  unsigned long long win = 0;
#pragma unroll
  for (int b = 0; b < B; ++b) win |= (unsigned long long)xtok(d0 + b + (yLen + rl)) << (2 * b);
  uint32_t tokNext = xtok(d0 + (yLen + rl) - 1);   // slot 0's token at the next step
  for (int t = 0; t < T; ++t) {
    const int j = yLen - (t - rl);
    const bool colvalid = active && j >= 1 && j <= yLen;
    const uint32_t w = ctx[min(max(j - 1, -kCtxPad + 1), yLen + 4)];
    const uint32_t gk = w >> 24;
    if (j == yLen) gkEnd = gk;
    const double m2m = trans[gk], m2i = trans[Kg + gk], m2d = trans[2 * Kg + gk];
    const uint32_t erowN4 = (wNext & 0x7FFFu) * 4u;
    const double insEN = eins[(wNext >> 15) & 0x1FFu];
    // (i+1, j) for the top slot: lane l+1's slot 0 at column j, finished in the previous step
    double hiD = __shfl_down(pubD, 1, G);
    if (l == G - 1) hiD = QF_NEG_INF;
    double pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // match-by-token[4], insert, m2m, m2i, m2d partial counts (source column j)
    double pc0[4] = {0, 0, 0, 0};             // start -> mat(i,1) counts by token (source "column 0")
    double loI = QF_NEG_INF;                   // Bi(i, j+1) for slot 0: lane l-1's top slot, exchanged below
    double nextD = hiD;                        // Bd(i+1, j): slot b+1 of this step, or lane l+1's slot 0
#pragma unroll
    for (int b = B - 1; b >= 0; --b) {
      const int d = d0 + b, i = d + j;
      const bool valid = colvalid && d <= dhi && i >= 1 && i <= xLen;
      const uint32_t tokN = (uint32_t)(win >> (2 * b)) & 3u;   // token of row i+1
      const double eN = ematch[erowN4 + tokN];
      const double BmN = Bm[b];                                   // Bm(i+1, j+1), own diagonal, previous step
      const double BiN = b > 0 ? Bi[b - 1] : loI;                 // Bi(i, j+1), diagonal d-1 (not yet overwritten)
      const double BdN = nextD;                                   // Bd(i+1, j), diagonal d+1
      const double T_mm = (m2m + eN) + BmN, T_im = (i2m + eN) + BmN, T_dm = (d2m + eN) + BmN;
      const double T_mi = (m2i + insEN) + BiN, T_ii = (i2i + insEN) + BiN;
      const double T_md = m2d + BdN, T_dd = d2d + BdN;
      const bool isEnd = j == yLen && (local || i == xLen);
      const double T_me = isEnd ? trans[3 * Kg + gk] : QF_NEG_INF;
      // accumulation order of the reference's push-style sweep (columns descending, rows descending): the
      // contribution from mat(i+1,j+1) arrives first, then ins(i,j+1), then del(i+1,j), then the end transition.
      // The table log-sum-exp is not associative at the 1e-4 level (its x >= 10 cut-off drops up to 4.5e-5 per
      // call), so the order is part of the numerical contract.
      double nbm = lseh(hs, lseh(hs, T_mm, T_mi), T_md);
      if (isEnd) nbm = lseh(hs, nbm, T_me);
      double nbi = lseh(hs, T_im, T_ii);
      double nbd = lseh(hs, T_dm, T_dd);
      if (!valid) { nbm = QF_NEG_INF; nbi = QF_NEG_INF; nbd = QF_NEG_INF; }
      if (valid) {
        const uint64_t base = ((uint64_t)(j - 1 + l) * B + b) * 3 * G + l;
        const double Fm = fw[base] - Fres, Fi = fw[base + G] - Fres, Fd = fw[base + 2 * G] - Fres;
        // NB (F - Fres) + T differs from the reference's (F + T) - Fres only in rounding
        const double c_mm = wgt * count_exp(Fm + T_mm), c_im = wgt * count_exp(Fi + T_im), c_dm = wgt * count_exp(Fd + T_dm);
        const double c_mi = wgt * count_exp(Fm + T_mi), c_ii = wgt * count_exp(Fi + T_ii);
        const double c_md = wgt * count_exp(Fm + T_md), c_dd = wgt * count_exp(Fd + T_dd);
        const double cmat = c_mm + c_im + c_dm;
        pc[0] += tokN == 0 ? cmat : 0.0; pc[1] += tokN == 1 ? cmat : 0.0;
        pc[2] += tokN == 2 ? cmat : 0.0; pc[3] += tokN == 3 ? cmat : 0.0;
        pc[4] += c_mi + c_ii;
        pc[5] += c_mm; pc[6] += c_mi; pc[7] += c_md;
        acc_i2m += c_im; acc_d2m += c_dm; acc_i2i += c_ii; acc_d2d += c_dd;
        if (isEnd) acc_m2e += wgt * count_exp(Fm + T_me);
        if (j == 1 && (i == 1 || local)) {  // start -> mat(i,1), src/qmodel.cpp:1448-1454
          const uint32_t tok = xt[i - 1];
          const double S = ematch[(w & 0x7FFFu) * 4u + tok] + nbm;
          const double cs = wgt * count_exp(S - Fres);
          pc0[0] += tok == 0 ? cs : 0.0; pc0[1] += tok == 1 ? cs : 0.0;
          pc0[2] += tok == 2 ? cs : 0.0; pc0[3] += tok == 3 ? cs : 0.0;
          startv = lseh(hs, startv, S);
        }
      }
      Bm[b] = nbm; Bi[b] = nbi; Bd[b] = nbd;
      nextD = nbd;
      if (b == B - 1) {  // lane l-1 (one column behind) has just produced Bi of its top slot at column j+1
        loI = __shfl_up(nbi, 1, G);
        if (l == 0) loI = QF_NEG_INF;
      }
    }
    pubD = Bd[0];
    // ---- per-column partials travel with the column: lane l+1 was on column j one step ago and hands its running sums
    // to lane l; the unit's lane 0 is the last on every column and flushes the complete sums (no LDS, no barriers)
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
      if (Kg == 1) { acc_m2m += pc[5]; acc_m2i += pc[6]; acc_m2d += pc[7]; }
      else {
        if (pc[5] != 0.0) unsafeAtomicAdd(&s_tr[gk], pc[5]);
        if (pc[6] != 0.0) unsafeAtomicAdd(&s_tr[Kg + gk], pc[6]);
        if (pc[7] != 0.0) unsafeAtomicAdd(&s_tr[2 * Kg + gk], pc[7]);
      }
      if (j == 1) {  // start -> mat(i,1): emission counts of column 1, once per lane
        const uint32_t er = w & 0x7FFFu, mk = er / (kNQualDev + 1), q = er % (kNQualDev + 1);
        if (q < (uint32_t)kNQualDev) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (pc0[c] != 0.0) unsafeAtomicAdd(&cnt[cMat + ((uint64_t)c * Km + mk) * kNQualDev + q], pc0[c]);
        }
      }
      if (l == 0 && j < yLen) {
        // emission rows belong to the destination column j+1 (context word index j); column yLen has no destination
        const uint32_t er = wNext & 0x7FFFu, mk = er / (kNQualDev + 1), q = er % (kNQualDev + 1);
        const uint32_t ytok = ((wNext >> 15) & 0x1FFu) / (kNQualDev + 1);
        if (q < (uint32_t)kNQualDev) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (colsum[c] != 0.0) unsafeAtomicAdd(&cnt[cMat + ((uint64_t)c * Km + mk) * kNQualDev + q], colsum[c]);
          if (colsum[4] != 0.0) unsafeAtomicAdd(&cnt[cIns + (uint64_t)ytok * kNQualDev + q], colsum[4]);
        }
      }
    }
    wNext = w;
    win = ((win << 2) | tokNext) & ((B < 32 ? (1ull << (2 * B)) : 0ull) - 1ull);
    tokNext = xtok(d0 + j - 2);
  }
  // context-free transitions, m2e and the Backward result (start), reduced over the unit's lanes
  for (int o = 1; o < G; o <<= 1) {
    acc_i2m += __shfl_xor(acc_i2m, o, G); acc_d2m += __shfl_xor(acc_d2m, o, G);
    acc_i2i += __shfl_xor(acc_i2i, o, G); acc_d2d += __shfl_xor(acc_d2d, o, G);
    acc_m2e += __shfl_xor(acc_m2e, o, G);
    acc_m2m += __shfl_xor(acc_m2m, o, G); acc_m2i += __shfl_xor(acc_m2i, o, G); acc_m2d += __shfl_xor(acc_m2d, o, G);
    startv = lseh(hs, startv, __shfl_xor(startv, o, G));
    gkEnd = max(gkEnd, (uint32_t)__shfl_xor((int)gkEnd, o, G));
  }
  if (Kg > 1) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t c = lane; c < 3 * Kg; c += 64) if (s_tr[c] != 0.0) unsafeAtomicAdd(&cnt[cTr + c], s_tr[c]);
  } else if (active && l == 0) {
    if (acc_m2m != 0.0) unsafeAtomicAdd(&cnt[cTr + 0], acc_m2m);
    if (acc_m2i != 0.0) unsafeAtomicAdd(&cnt[cTr + 1], acc_m2i);
    if (acc_m2d != 0.0) unsafeAtomicAdd(&cnt[cTr + 2], acc_m2d);
  }
  if (active && l == 0) {
    if (acc_m2e != 0.0) unsafeAtomicAdd(&cnt[cTr + 3 * Kg + gkEnd], acc_m2e);
    if (acc_d2d != 0.0) unsafeAtomicAdd(&cnt[cTr + 4 * Kg + 0], acc_d2d);
    if (acc_d2m != 0.0) unsafeAtomicAdd(&cnt[cTr + 4 * Kg + 1], acc_d2m);
    if (acc_i2i != 0.0) unsafeAtomicAdd(&cnt[cTr + 4 * Kg + 2], acc_i2i);
    if (acc_i2m != 0.0) unsafeAtomicAdd(&cnt[cTr + 4 * Kg + 3], acc_i2m);
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
  __shared__ double s_lseh[2 * kLseNodes];
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
  __shared__ double s_lseh[2 * kLseNodes];
  lseh_load(s_lseh, a.lse_h, threadIdx.x, 64);
  __syncthreads();
  const double* hs = s_lseh;
  constexpr int G = 64, B = 8, S = kRowStripe;
  extern __shared__ double s_tr[];   // [3 * Kg] context-dependent transition counts of this unit
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
  for (uint32_t c = l; c < 3 * a.dp.Kg; c += 64) s_tr[c] = 0.0;
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
  double* __restrict__ cnt = a.counts + (size_t)(blockIdx.x % kCountReplicas) * a.counts_stride;   // contention: see kCountReplicas
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
        if (pc[5] != 0.0) unsafeAtomicAdd(&s_tr[gk], pc[5]);
        if (pc[6] != 0.0) unsafeAtomicAdd(&s_tr[Kg + gk], pc[6]);
        if (pc[7] != 0.0) unsafeAtomicAdd(&s_tr[2 * Kg + gk], pc[7]);
        if (j == 1) {
          const uint32_t er = w & 0x7FFFu, mk = er / (kNQualDev + 1), q = er % (kNQualDev + 1);
          if (q < (uint32_t)kNQualDev) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (pc0[c] != 0.0) unsafeAtomicAdd(&cnt[cMat + ((uint64_t)c * Km + mk) * kNQualDev + q], pc0[c]);
          }
        }
        if (l == 0 && j < yLen) {
          const uint32_t er = wNext & 0x7FFFu, mk = er / (kNQualDev + 1), q = er % (kNQualDev + 1);
          const uint32_t ytok = ((wNext >> 15) & 0x1FFu) / (kNQualDev + 1);
          if (q < (uint32_t)kNQualDev) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (colsum[c] != 0.0) unsafeAtomicAdd(&cnt[cMat + ((uint64_t)c * Km + mk) * kNQualDev + q], colsum[c]);
            if (colsum[4] != 0.0) unsafeAtomicAdd(&cnt[cIns + (uint64_t)ytok * kNQualDev + q], colsum[4]);
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
  for (uint32_t c = l; c < 3 * Kg; c += 64) if (s_tr[c] != 0.0) unsafeAtomicAdd(&cnt[cTr + c], s_tr[c]);
  if (l == 0) {
    if (acc_m2e != 0.0) unsafeAtomicAdd(&cnt[cTr + 3 * Kg + gkEnd], acc_m2e);
    if (acc_d2d != 0.0) unsafeAtomicAdd(&cnt[cTr + 4 * Kg + 0], acc_d2d);
    if (acc_d2m != 0.0) unsafeAtomicAdd(&cnt[cTr + 4 * Kg + 1], acc_d2m);
    if (acc_i2i != 0.0) unsafeAtomicAdd(&cnt[cTr + 4 * Kg + 2], acc_i2i);
    if (acc_i2m != 0.0) unsafeAtomicAdd(&cnt[cTr + 4 * Kg + 3], acc_i2m);
    a.units[uid].end_val = startv;
  }
}


template <int G, int B>
static void launch_fwd_gb(const FbArgs& a, hipStream_t s) {
  const uint32_t upw = 64 / G, waves = (a.n_cls_units + upw - 1) / upw, blocks = (waves + 3) / 4;
  hipLaunchKernelGGL((k_forward_fill<G, B>), dim3(blocks), dim3(256), 0, s, a);
}
template <int G, int B>
static void launch_bwd_gb(const FbArgs& a, hipStream_t s) {
  const uint32_t upw = 64 / G, waves = (a.n_cls_units + upw - 1) / upw, blocks = (waves + 3) / 4;
  hipLaunchKernelGGL((k_backward_fill<G, B>), dim3(blocks), dim3(256), (size_t)4 * 3 * a.dp.Kg * 8, s, a);
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
  if (cls == kRowClass) { hipLaunchKernelGGL(k_backward_rows, dim3(a.n_cls_units), dim3(64), (size_t)3 * a.dp.Kg * 8, s, a); return; }
  QF_FB_DISPATCH(launch_bwd_gb)
}
// counts[0][i] += counts[1..R-1][i]
__global__ void k_sum_count_replicas(double* counts, uint32_t n, uint64_t stride) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = counts[i];
  for (int r = 1; r < kCountReplicas; ++r) v += counts[(size_t)r * stride + i];
  counts[i] = v;
}
void launch_sum_count_replicas(double* counts, uint32_t n, uint64_t stride, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_sum_count_replicas, dim3((n + 255) / 256), dim3(256), 0, s, counts, n, stride);
}
void launch_pair_forward(const FinalArgs& a, const double* lse, hipStream_t s) {
  if (a.n_pairs) hipLaunchKernelGGL(k_pair_forward, dim3((a.n_pairs + 255) / 256), dim3(256), 0, s, a, lse);
}
void launch_count_plan(const CountPlanArgs& a, hipStream_t s) {
  if (a.n_reads) hipLaunchKernelGGL(k_count_plan, dim3((a.n_reads + 63) / 64), dim3(64), 0, s, a);
}

}  // namespace qf
