// qf_overlap.hip — read-vs-read overlap Viterbi (quaff overlap): QuaffOverlapViterbiMatrix ctor + alignment(),
// src/qoverlap.cpp:77-290, on the same skewed G x B diagonal wavefront as the align fill.  Differences from the
// align model: both sequences are reads (pair-emission table indexed by both reads' context k-mer and quality),
// the gap states mix through a table log-sum-exp before the max (:145-151), both ends are free (:141,:153), the
// traceback takes three candidates for insert and delete states (6 traceback bits per cell -> one byte), and the
// reference's accessor swaps (src/qoverlap.h:46-50) and fill/traceback inconsistency (:148 vs :222) are reproduced
// literally.  Bit-exact: the log-sum-exp uses the host-built table with IEEE division, as src/logsumexp.cpp does.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "qf_dpp.hpp"
#include "qf_kernels.hpp"

namespace qf {

#define QF_NEG_INF (-__builtin_huge_val())

__device__ __forceinline__ int tokc(int c) {
  c &= ~0x20;
  return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0;
}

// x / 1e-4 as IEEE division rounds it, in three operations: q0 = x * RN(1 / c), r = x - c * q0 (exact in an fma),
// q = q0 + r * RN(1 / c).  With RN(1 / c) the correctly rounded reciprocal and q0 within an ulp, q is the correctly rounded
// quotient (Markstein's division theorem; the divisor's significand is not all ones); checked against `/` on every
// x = n * 1e-4 +- 40 ulps for n <= 100 001 and 4e8 random x in [0, 10) (tools/dev/divtest.c).  The compiler's own
// expansion of `/` is ~12 instructions with two of the slow ones (v_div_scale, v_rcp).
__device__ __forceinline__ double div_1e4(double x) {
  const double c = .0001, rc = 1.0 / .0001;
  const double q0 = x * rc;
  const double r = fma(-c, q0, x);
  return fma(r, rc, q0);
}
struct __attribute__((packed, aligned(8))) D2 { double v[2]; };   // 16-byte load from an 8-byte aligned address

// Entries n and n + 1 of the exact table from its packed form in LDS (qf_device.hpp: kLsePack*): three 16-byte reads and one
// 8-byte read of the piece, two Horner evaluations that share them, one 8-byte read of the correction stream.
__device__ __forceinline__ void lse_pack_pair(const char* pk, int n, double& f0, double& f1) {
  const int t = n & (kLsePackSpan - 1), p = n >> 8;
  const double2 c01 = *(const double2*)(pk + kLsePackC01 + p * 16), c23 = *(const double2*)(pk + kLsePackC23 + p * 16),
                c45 = *(const double2*)(pk + kLsePackC45 + p * 16);
  const uint2 bw = *(const uint2*)(pk + kLsePackMeta + p * 8);
  const double u0 = (double)(t - kLsePackSpan / 2), u1 = u0 + 1.0;
  double v0 = c45.y, v1 = c45.y;
  v0 = fma(v0, u0, c45.x); v1 = fma(v1, u1, c45.x);
  v0 = fma(v0, u0, c23.y); v1 = fma(v1, u1, c23.y);
  v0 = fma(v0, u0, c23.x); v1 = fma(v1, u1, c23.x);
  v0 = fma(v0, u0, c01.y); v1 = fma(v1, u1, c01.y);
  v0 = fma(v0, u0, c01.x); v1 = fma(v1, u1, c01.x);
  const uint32_t o = bw.x + (uint32_t)t * bw.y;
  const uint32_t* wp = (const uint32_t*)(pk + kLsePackStream) + (o >> 5);
  const uint32_t bits = __builtin_amdgcn_alignbit(wp[1], wp[0], o & 31u);   // both fields: 2 * width <= 32
  const long long k0 = (int)__builtin_amdgcn_sbfe(bits, 0u, bw.y), k1 = (int)__builtin_amdgcn_sbfe(bits, bw.y, bw.y);   // (the builtin's type is unsigned)
  f0 = __longlong_as_double(__double_as_longlong(v0) + k0);
  f1 = __longlong_as_double(__double_as_longlong(v1) + k1);
}

// max(log_sum_exp(a, b), c) as the fills use it (std::max(lse, c): the first unless it is smaller); log_sum_exp(a, b) is
// src/logsumexp.cpp:34-50,84-103 bit for bit (the reference's divisions, no contraction), without branches; x >= 10, NaN and
// infinities take max + 0 (:86-87).  The table term is at most its entry 0 = log 2 and rounding is monotonic, so
// log_sum_exp(a, b) <= RN(max(a, b) + 0.694): a lane whose c is above that takes c without a look-up (it reads entry 0 like
// the x >= 10 lanes).  PACKED: the two entries come from the packed table in LDS; otherwise they are one 16-byte gather of the
// 800 KB table, which costs a 128-byte line through the L1 per lane (2 clocks of the CU's L2 port each, measured:
// tools/dev/l2_gather_bench.hip) -- that, three times per cell with the pair emission, is what bounds the global variant.
template <bool PACKED>
__device__ __forceinline__ double max_lse_exact(const void* __restrict__ tab, double a, double b, double c) {
  double mx;                                                // v_max_f64 as it is (fmax would canonicalise both operands first)
  asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(a), "v"(b));
  const double diff = fabs(a - b);                          // NaN for -inf - -inf: not "small", the result is max + 0 = -inf as in :86-87
  const bool skip = c > mx + 0.694;
  const bool small = diff < 10.0;
  const double x = small && !skip ? diff : 0.0;
  const int n = (int)div_1e4(x);
  const double dx = x - (n * .0001);
  double f0, f1;
  if (PACKED) lse_pack_pair((const char*)tab, n, f0, f1);
  else { const D2 f = *(const D2*)((const double*)tab + n); f0 = f.v[0]; f1 = f.v[1]; }
  const double df = f1 - f0;
  const double r = mx + (f0 + df * div_1e4(dx));
  const double l = small ? r : mx;
  return !(l > c) ? c : l;                                  // (a skipped lane has l <= RN(mx + log 2) < c)
}

// Rebuilds every table entry from the packed form, both ways it can be reached (as entry n of its own piece and as the 257th
// entry of the piece below), and counts the ones that differ from the table: the library uses the packed form only if none does.
__global__ __launch_bounds__(256) void k_lse_pack_check(const uint8_t* __restrict__ pack, uint32_t pack_bytes, const double* __restrict__ tab,
                                                        uint32_t* __restrict__ bad) {
  extern __shared__ __attribute__((aligned(16))) char s_pack[];
  for (uint32_t k = threadIdx.x; k < pack_bytes / 16; k += blockDim.x) ((uint4*)s_pack)[k] = ((const uint4*)pack)[k];
  __syncthreads();
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < kLseEntriesDev - 1; n += gridDim.x * blockDim.x) {
    double f0, f1;
    lse_pack_pair(s_pack, n, f0, f1);
    if (__double_as_longlong(f0) != __double_as_longlong(tab[n]) || __double_as_longlong(f1) != __double_as_longlong(tab[n + 1])) atomicAdd(bad, 1u);
  }
}

// Context words of the reverse-complement strand in this sequence's orientation (src/qoverlap.cpp:91-98: the arrays
// of revcomp(y), reversed): complemented token, context k-mers read right-to-left, padded with revcomp's most
// frequent token.  One wavefront per sequence.
__global__ __launch_bounds__(64) void k_prep_overlap(PrepArgs a) {
  const uint32_t r = blockIdx.x, lane = threadIdx.x;
  const uint64_t b = a.off[r];
  const uint32_t L = (uint32_t)(a.off[r + 1] - b);
  uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (uint32_t i = lane; i < L; i += 64) {
    const int t = 3 - tokc((unsigned char)a.seq[b + i]);
    c0 += t == 0; c1 += t == 1; c2 += t == 2; c3 += t == 3;
  }
  for (int o = 32; o; o >>= 1) {
    c0 += __shfl_xor(c0, o); c1 += __shfl_xor(c1, o); c2 += __shfl_xor(c2, o); c3 += __shfl_xor(c3, o);
  }
  uint32_t padTok = 0, best = c0;
  if (c1 > best) { best = c1; padTok = 1; }
  if (c2 > best) { best = c2; padTok = 2; }
  if (c3 > best) { best = c3; padTok = 3; }
  auto ctok = [&](int64_t p) -> uint32_t {  // complemented token at position p of this sequence; beyond the end = pad
    if (p >= (int64_t)L) return padTok;
    return 3u - (uint32_t)tokc((unsigned char)a.seq[b + p]);
  };
  for (uint32_t i = lane; i < L; i += 64) {
    uint32_t mk = 0, gk = 0;
    for (uint32_t c = 0; c < a.match_len; ++c) mk = mk * 4 + ctok((int64_t)i + (a.match_len - 1) - c);
    for (uint32_t c = 0; c < a.gap_len; ++c) gk = gk * 4 + ctok((int64_t)i + (a.gap_len - 1) - c);
    uint32_t q = kNQualDev;
    if (a.qual) {
      const int v = (int)(signed char)a.qual[b + i] - '!';
      q = (uint32_t)max(0, min(kNQualDev - 1, v));
    }
    a.ctxc[b + i] = ctx_pack(mk * (kNQualDev + 1) + q, ctok(i) * (kNQualDev + 1) + q, gk);
  }
}

// per-sequence sequential sums: insert scores in both orientations (src/qoverlap.cpp:105-113) and the null
// log-likelihood of the reverse complement (scoreAdjustedAlignment, :292-302).  One lane per sequence.
__global__ __launch_bounds__(64) void k_overlap_sums(PrepArgs a, const uint32_t* __restrict__ ctx, uint32_t n) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const uint64_t b = a.off[r];
  const uint32_t L = (uint32_t)(a.off[r + 1] - b);
  double s0 = 0, s1 = 0;
  for (uint32_t i = 0; i < L; ++i) {
    s0 += a.eins[(ctx[b + i] >> 15) & 0x1FFu];
    s1 += a.eins[(a.ctxc[b + i] >> 15) & 0x1FFu];
  }
  a.ins_sum[r] = s0;
  a.ins_sum_c[r] = s1;
  double ll = 0;
  if (a.has_null) {  // null LL of revcomp(seq): bases and qualities visited from the far end
    ll = (double)L * a.null_logEmit + a.null_log1mEmit;
    for (uint32_t k = 0; k < L; ++k) {
      const uint32_t i = L - 1 - k;
      const uint32_t t = 3u - a.tok[b + i];
      ll += a.null_logSym[t];
      if (a.qual) {
        const int v = (int)(signed char)a.qual[b + i] - '!';
        ll += a.null_logQual[t * kNQualDev + max(0, min(kNQualDev - 1, v))];
      }
    }
  }
  a.nll_c[r] = ll;
}

// The banded fill.  Every cell carries two dependent look-ups of the exact log-sum-exp table (800 KB: L2), and the delete
// state's chains through the lane's B slots within a step (del(i,j) needs del(i-1,j)): what bounds the kernel is that
// chain of L2 latencies, so it is written for occupancy (few registers: the three-operation division above, DPP lane
// exchanges) and keeps everything that does not depend on the chain out of its way -- the pair emissions of step t+1 and the
// context words of step t+2 are fetched at step t, and the insert state's look-ups (which read only the previous column)
// are issued before the delete chain starts.
#ifndef QF_OV_PACK_WAVES
#define QF_OV_PACK_WAVES 8     // wavefronts of the one workgroup per CU that shares the packed table (8: two per SIMD)
#endif
// (32, 3): twelve wavefronts share the table -- three per SIMD -- which its 168 registers allow
constexpr int ov_pack_waves(int G, int B) { return G == 32 && B == 3 ? 12 : QF_OV_PACK_WAVES; }
template <int G, int B, bool GAPCTX, bool PACKED>
__global__ __launch_bounds__(PACKED ? 64 * ov_pack_waves(G, B) : 256) __attribute__((amdgpu_waves_per_eu(PACKED ? ov_pack_waves(G, B) / 4 : B <= 5 ? 3 : 2)))
void k_overlap_fill(OvArgs a) {
  constexpr int UPW = 64 / G;
  extern __shared__ __attribute__((aligned(16))) char s_pack[];
  if (PACKED) {
    for (uint32_t k = threadIdx.x; k < a.lse_pack_bytes / 16; k += blockDim.x) ((uint4*)s_pack)[k] = ((const uint4*)a.lse_pack)[k];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int grp = lane / G, l = lane % G;
  const uint32_t uidx = wave * UPW + grp;
  const bool active = uidx < a.n_cls_units;
  uint32_t uid = 0, comp = 0;
  int dlo = 0, dhi = -1, xLen = 0, yLen = 0;
  uint64_t xb = 0, yb = 0, tb_off = 0;
  if (active) {
    uid = a.cls_list[uidx];
    const Unit u = a.units[uid];
    const uint32_t x = a.pair_x[u.pair], y = a.pair_y[u.pair];
    comp = a.pair_comp[u.pair];
    xb = a.seq_off[x]; xLen = (int)(a.seq_off[x + 1] - xb);
    yb = a.seq_off[y]; yLen = (int)(a.seq_off[y + 1] - yb);
    dlo = u.dlo; dhi = u.dhi; tb_off = u.tb_off;
  }
  // the columns this band crosses inside the rectangle (qf_device.hpp: band_col0): step t of lane l is column j0 + t - l + 1
  const int j0 = band_col0(dhi), jEnd = band_last_col(dlo, xLen, yLen);
  int T = active && jEnd > j0 ? jEnd - j0 + G - 1 : 0;
  for (int o = 32; o; o >>= 1) T = max(T, __shfl_xor(T, o));
  const int d0 = dlo + l * B;
  const double* __restrict__ mmi = a.mmi[comp];
  const double* __restrict__ gap = a.gap[comp];
  const void* tab = PACKED ? (const void*)s_pack : (const void*)a.lse;
  const uint32_t Kg = a.Kg, KQ = a.Km * (kNQualDev + 1);
  const double* gsc = gap + 3ull * Kg * Kg;
  // accessor swaps of src/qoverlap.h:46-50, literally
  const double i2mS = gsc[1], i2iS = gsc[0], i2dS = gsc[2], d2mS = gsc[4], d2iS = gsc[3], d2dS = gsc[5];
  const double c_m2m = gap[0], c_m2i = gap[(size_t)Kg * Kg], c_m2d = gap[2ull * Kg * Kg];
  const uint32_t* __restrict__ xc = a.ctx + xb;
  const uint32_t* __restrict__ yc = (comp ? a.ctxc : a.ctx) + yb;
  uint32_t* __restrict__ tb = a.tb + tb_off;

  double M[B], I[B], D[B];
#pragma unroll
  for (int b = 0; b < B; ++b) M[b] = I[b] = D[b] = QF_NEG_INF;
  double pubM = QF_NEG_INF, pubI = QF_NEG_INF, pubD = QF_NEG_INF;
  double colBest = QF_NEG_INF, rowBest = QF_NEG_INF;
  uint32_t colI = 0, rowJ = 0;
  // x-side context words of the rows this lane's slots are on: xw[b] = word of row d0 + b + j - 1 (slot b is on row
  // i = d0 + b + j: xw[b] is row i-1, xw[b+1] row i).  The rows slide down by one per step: a step shifts the window and
  // takes one new word, loaded two steps ahead (the next step's emissions are fetched a step ahead).
  auto xword = [&](int row) -> uint32_t { return (row >= 1 && row <= xLen) ? xc[row - 1] : 0u; };
  auto yword = [&](int j) -> uint32_t { return yc[min(max(j - 1, -kCtxPad + 1), yLen + 4)]; };
  uint32_t xw[B + 1];
#pragma unroll
  for (int b = 0; b <= B; ++b) xw[b] = xword(d0 + b + j0 - l);
  uint32_t xwN = xword(d0 + B + 1 + j0 - l), xwNN = xword(d0 + B + 2 + j0 - l);   // the words entering at steps 1 and 2
  uint32_t wy = yword(j0 + 1 - l), wyN = yword(j0 + 2 - l), wyNN = yword(j0 + 3 - l);
  uint32_t gkyPrev = yword(j0 - l) >> 24;
  // (a 32-bit element offset from the table's uniform base: (Km x 95)^2 entries are far below 2^32)
  auto emis = [&](uint32_t wxrow, uint32_t wycol) -> double { return mmi[(wxrow & 0x7FFFu) * KQ + (wycol & 0x7FFFu)]; };
  const int bmax = active ? dhi - d0 : -1;                            // slots b > bmax are outside the band
  double e[B];
#pragma unroll
  for (int b = 0; b < B; ++b) { const double en = emis(xw[b + 1], wy); e[b] = b > bmax ? QF_NEG_INF : en; }

  // FAST steps (wave-uniform range [fastLo, fastHi]): every lane that has a band is on a column >= 2 inside its band's column range
  // and all its band's slots are on rows 2 ... xLen.  Such a step needs no start candidate and no row / column validity; the
  // slots above the band's last diagonal are kept at -inf by a -inf emission (match: its sums; insert: by induction, its sources
  // lie further outside) and one select on the delete state (its source is the band's last diagonal).  Values come from v_max,
  // the traceback bits straight from the compares (first maximum in the reference's order: bit 0 = second candidate beats the
  // first, bit 1 = third beats both; "both" reads as 3, which the traceback takes for the third candidate away from row / column 1).
  int fastLo = 0, fastHi = 0x7FFFFFFF;
  if (active && bmax >= 0) {
    const int bt = bmax < B - 1 ? bmax : B - 1;
    const int jmin = max(max(2, j0 + 1), 2 - d0), jmax = min(jEnd, xLen - d0 - bt);
    fastLo = jmin - j0 + l - 1;
    fastHi = jmax - j0 + l - 1;
  }
  for (int o = 32; o; o >>= 1) { fastLo = max(fastLo, __shfl_xor(fastLo, o)); fastHi = min(fastHi, __shfl_xor(fastHi, o)); }
  fastLo = __builtin_amdgcn_readfirstlane(fastLo);
  fastHi = __builtin_amdgcn_readfirstlane(fastHi);
  if (a.no_fast_steps) fastHi = -1;

  for (int t = 0; t < T; ++t) {
    const int j = j0 + t - l + 1;
    const bool colvalid = active && j > j0 && j <= jEnd;   // (the wavefront runs as long as its longest band: the others stop storing)
    const uint32_t gky = wy >> 24;
    const uint32_t gkyP = j > 1 ? gkyPrev : 0u;   // yIndelKmer[j-1], padded with a leading 0
    gkyPrev = gky;
    double prevM = dpp_from_below<G, false>(pubM), prevI = dpp_from_below<G, false>(pubI), prevD = dpp_from_below<G, false>(pubD);
    double upM = 0, upI = 0, upD = 0;
    uint32_t tbw0 = 0, tbw1 = 0;
    if (t >= fastLo && t <= fastHi) {
      uint32_t acc0 = 0, acc1 = 0;
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const uint32_t wx = xw[b + 1];                       // row i
        double m2m, m2i, m2d;
        if (GAPCTX) {
          const uint32_t gkx = wx >> 24, gkxP = xw[b] >> 24;   // xIndelKmer[i], xIndelKmer[i-1] (i, j >= 2 here)
          m2m = gap[gkxP * Kg + gkyP];
          m2i = gap[(size_t)Kg * Kg + gkx * Kg + gkyP];
          m2d = gap[2ull * Kg * Kg + gkxP * Kg + gky];
        } else { m2m = c_m2m; m2i = c_m2i; m2d = c_m2d; }
        const double eb = e[b];
        {
          const double en = emis(b + 1 < B ? xw[b + 2] : xwN, wyN);
          e[b] = b > bmax ? QF_NEG_INF : en;                 // (next step's emission; -inf above the band)
        }
        const double tM = (M[b] + m2m) + eb, tI = (I[b] + i2mS) + eb, tD = (D[b] + d2mS) + eb;
        double m1, nm;
        asm("v_max_f64 %0, %1, %2" : "=v"(m1) : "v"(tM), "v"(tI));
        asm("v_max_f64 %0, %1, %2" : "=v"(nm) : "v"(m1), "v"(tD));
        double sM, sI, sD;
        if (b + 1 < B) { sM = M[b + 1]; sI = I[b + 1]; sD = D[b + 1]; } else { sM = upM; sI = upI; sD = upD; }
        const double iM = sM + m2i, iI = sI + i2iS, iD = sD + d2iS;
        const double ni = max_lse_exact<PACKED>(tab, iI, iD, iM);
        double mi;
        asm("v_max_f64 %0, %1, %2" : "=v"(mi) : "v"(iM), "v"(iI));
        const double dM = prevM + m2d, dD = prevD + d2dS, dIfill = prevI + d2iS, dItb = prevI + i2dS;
        double ndl = max_lse_exact<PACKED>(tab, dD, dIfill, dM);
        double md;
        asm("v_max_f64 %0, %1, %2" : "=v"(md) : "v"(dM), "v"(dItb));
        if (b > bmax) ndl = QF_NEG_INF;
        uint32_t& acc = b < 4 ? acc0 : acc1;
        acc = shift_in_gt(acc, tI, tM);
        acc = shift_in_gt(acc, tD, m1);
        acc = shift_in_gt(acc, iI, iM);
        acc = shift_in_gt(acc, iD, mi);
        acc = shift_in_gt(acc, dItb, dM);
        acc = shift_in_gt(acc, dD, md);
        acc <<= 2;
        M[b] = nm; I[b] = ni; D[b] = ndl;
        prevM = nm; prevI = ni; prevD = ndl;
        if (b == 0) {
          upM = dpp_from_above<G, false>(nm); upI = dpp_from_above<G, false>(ni); upD = dpp_from_above<G, false>(ndl);
        }
        if (!PACKED) __builtin_amdgcn_sched_barrier(0);
      }
      tbw0 = __builtin_bitreverse32(acc0) >> (32 - 8 * (B < 4 ? B : 4));
      if (B > 4) tbw1 = __builtin_bitreverse32(acc1) >> (32 - 8 * (B - 4));
    } else
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const int d = d0 + b, i = d + j;
      const bool valid = colvalid && d <= dhi && i >= 1 && i <= xLen;
      const uint32_t wx = xw[b + 1];                       // row i
      double m2m, m2i, m2d;
      if (GAPCTX) {
        const uint32_t gkx = wx >> 24;                       // xIndelKmer[i]
        const uint32_t gkxP = i > 1 ? (xw[b] >> 24) : 0u;    // xIndelKmer[i-1]
        m2m = gap[gkxP * Kg + gkyP];                         // m2mScore(i-1, j-1)
        m2i = gap[(size_t)Kg * Kg + gkx * Kg + gkyP];        // m2iScore(i,   j-1)
        m2d = gap[2ull * Kg * Kg + gkxP * Kg + gky];         // m2dScore(i-1, j)
      } else { m2m = c_m2m; m2i = c_m2i; m2d = c_m2d; }
      const double eb = e[b];
      // this slot's emission for the next step: row i + 1 (the word above it in the window), column j + 1 (-inf above the band:
      // the FAST steps rely on it)
      {
        const double en = emis(b + 1 < B ? xw[b + 2] : xwN, wyN);
        e[b] = b > bmax ? QF_NEG_INF : en;
      }
      // match state; traceback candidate order M, I, D, Start (strict >), src/qoverlap.cpp:204-209
      const double tM = (M[b] + m2m) + eb, tI = (I[b] + i2mS) + eb, tD = (D[b] + d2mS) + eb;
      double nm = tM;
      uint32_t sm = 0;
      if (tI > nm) { nm = tI; sm = 1; }
      if (tD > nm) { nm = tD; sm = 2; }
      if ((j == 1 || i == 1) && eb > nm) { nm = eb; sm = 3; }
      // insert state: sources at (i, j-1) = diagonal d+1, previous column
      double sM, sI, sD;
      if (b + 1 < B) { sM = M[b + 1]; sI = I[b + 1]; sD = D[b + 1]; } else { sM = upM; sI = upI; sD = upD; }
      const double iM = sM + m2i, iI = sI + i2iS, iD = sD + d2iS;
      double ni = max_lse_exact<PACKED>(tab, iI, iD, iM);   // max(lse(ins + i2i, del + d2i), mat + m2i)
      uint32_t si = 0;                     // traceback: M, I, D on the individual terms (:215-217)
      { double sx = iM; if (iI > sx) { sx = iI; si = 1; } if (iD > sx) { sx = iD; si = 2; } }
      // delete state: sources at (i-1, j) = diagonal d-1, this column
      const double dM = prevM + m2d, dD = prevD + d2dS, dIfill = prevI + d2iS, dItb = prevI + i2dS;
      double ndl = max_lse_exact<PACKED>(tab, dD, dIfill, dM);
      uint32_t sd = 0;                     // traceback: M, then ins + i2dScore(), then D (:221-223)
      { double sx = dM; if (dItb > sx) { sx = dItb; sd = 1; } if (dD > sx) { sx = dD; sd = 2; } }
      if (!valid) { nm = QF_NEG_INF; ni = QF_NEG_INF; ndl = QF_NEG_INF; }
      M[b] = nm; I[b] = ni; D[b] = ndl;
      prevM = nm; prevI = ni; prevD = ndl;
      const uint32_t byte = sm | (si << 2) | (sd << 4);
      if (b < 4) tbw0 |= byte << (8 * b); else tbw1 |= byte << (8 * (b - 4));
      if (b == 0) {
        upM = dpp_from_above<G, false>(nm); upI = dpp_from_above<G, false>(ni); upD = dpp_from_above<G, false>(ndl);
      }
      if (!PACKED) __builtin_amdgcn_sched_barrier(0);   // global table: one slot at a time, registers (occupancy) matter more than overlap inside a wavefront
    }
    pubM = prevM; pubI = prevI; pubD = prevD;
    // free ends (src/qoverlap.cpp:141,153): best match cell of the last column / last row.  Outside the slot loop, so that the
    // loop is one basic block (the slots' look-ups overlap), and behind a branch few steps take.
    if (colvalid && (j == yLen || (xLen - j >= d0 && xLen - j < d0 + B))) {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int d = d0 + b, i = d + j;
        const double nm = M[b];
        const bool valid = d <= dhi && i >= 1 && i <= xLen;
        if (valid && j == yLen && nm >= colBest) { colBest = nm; colI = (uint32_t)i; }
        if (valid && i == xLen && (nm > rowBest || (nm == rowBest && (uint32_t)j > rowJ))) { rowBest = nm; rowJ = (uint32_t)j; }
      }
    }
#pragma unroll
    for (int b = 0; b < B; ++b) xw[b] = xw[b + 1];
    xw[B] = xwN;
    xwN = xwNN;
    xwNN = xword(d0 + B + j + 2);
    wy = wyN; wyN = wyNN;
    wyNN = yword(j + 3);
    if (colvalid) {
      tb_store(&tb[((uint64_t)t * G + l) * 2], tbw0);
      tb_store(&tb[((uint64_t)t * G + l) * 2 + 1], tbw1);
    }
  }
  for (int o = 1; o < G; o <<= 1) {
    const double ov = __shfl_xor(colBest, o, G);
    const uint32_t oi = __shfl_xor(colI, o, G);
    if (ov > colBest || (ov == colBest && oi > colI)) { colBest = ov; colI = oi; }
    const double rv = __shfl_xor(rowBest, o, G);
    const uint32_t rj = __shfl_xor(rowJ, o, G);
    if (rv > rowBest || (rv == rowBest && rj > rowJ)) { rowBest = rv; rowJ = rj; }
  }
  if (active && l == 0) {
    Unit* u = &a.units[uid];
    u->end_val = colBest; u->end_i = colI;
    u->end2_val = rowBest; u->end2_j = rowJ;
  }
}

// Single-diagonal bands (most read pairs do not overlap: only diagonal 0 is in the envelope): with no neighbouring
// diagonal the gap states stay -inf and the match state is a serial chain.  One lane per band, eight columns per round
// with the context words and emissions fetched ahead of the chain.  Traceback: every cell's match state comes from the
// match state of the cell before it, except that the diagonal's first cell (the only one with i == 1 or j == 1) may start
// the alignment (src/qoverlap.cpp:204-209): that one flag is all a single-diagonal band keeps (in its unit's tb_off, which
// it has no other use for), instead of a nibble per cell that would be zero everywhere else.
__global__ __launch_bounds__(256) void k_overlap_single(OvArgs a) {
  struct __attribute__((packed, aligned(4))) W4 { uint32_t v[4]; };
  const uint32_t uidx = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = uidx < a.n_cls_units;
  uint32_t uid = 0, comp = 0;
  int d = 0, xLen = 0, yLen = 0;
  uint64_t xb = 0, yb = 0;
  if (active) {
    uid = a.cls_list[uidx];
    const Unit u = a.units[uid];
    const uint32_t x = a.pair_x[u.pair], y = a.pair_y[u.pair];
    comp = a.pair_comp[u.pair];
    xb = a.seq_off[x]; xLen = (int)(a.seq_off[x + 1] - xb);
    yb = a.seq_off[y]; yLen = (int)(a.seq_off[y + 1] - yb);
    d = u.dlo;
  }
  int T = active ? yLen : 0;
  for (int o = 32; o; o >>= 1) T = max(T, __shfl_xor(T, o));
  const double* __restrict__ mmi = a.mmi[comp];
  const double* __restrict__ gap = a.gap[comp];
  const uint32_t Kg = a.Kg, KQ = a.Km * (kNQualDev + 1);
  const uint32_t* __restrict__ xc = a.ctx + xb;
  const uint32_t* __restrict__ yc = (comp ? a.ctxc : a.ctx) + yb;
  double M = QF_NEG_INF, colBest = QF_NEG_INF, rowBest = QF_NEG_INF;
  uint32_t colI = 0, rowJ = 0, gxPrev = 0, gyPrev = 0, startFlag = 0;
  for (int j0 = 1; j0 <= T; j0 += 8) {
    const int yi = min(j0 - 1, yLen);                       // both context arrays are padded by kCtxPad words
    const int xi = min(max(d + j0 - 1, -kCtxPad + 8), xLen);
    const W4 ya = *(const W4*)(yc + yi), yb4 = *(const W4*)(yc + yi + 4);
    const W4 xa = *(const W4*)(xc + xi), xb4 = *(const W4*)(xc + xi + 4);
    uint32_t wy[8], wx[8];
    double e[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      wy[c] = c < 4 ? ya.v[c] : yb4.v[c - 4];
      wx[c] = c < 4 ? xa.v[c] : xb4.v[c - 4];
      e[c] = mmi[(size_t)(wx[c] & 0x7FFFu) * KQ + (wy[c] & 0x7FFFu)];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int j = j0 + c, i = d + j;
      const bool valid = active && j <= yLen && i >= 1 && i <= xLen;
      const uint32_t gxP = i > 1 ? gxPrev : 0u, gyP = j > 1 ? gyPrev : 0u;  // xIndelKmer[i-1], yIndelKmer[j-1] (padded 0)
      gxPrev = wx[c] >> 24; gyPrev = wy[c] >> 24;
      const double tM = (M + gap[gxP * Kg + gyP]) + e[c];
      double nm = tM;
      if ((j == 1 || i == 1) && e[c] > nm) { nm = e[c]; if (valid) startFlag = 3; }
      if (!valid) nm = QF_NEG_INF;
      M = nm;
      if (valid && j == yLen && nm >= colBest) { colBest = nm; colI = (uint32_t)i; }
      if (valid && i == xLen && (nm > rowBest || (nm == rowBest && (uint32_t)j > rowJ))) { rowBest = nm; rowJ = (uint32_t)j; }
    }
  }
  if (active) {
    Unit* u = &a.units[uid];
    u->end_val = colBest; u->end_i = colI;
    u->end2_val = rowBest; u->end2_j = rowJ;
    u->tb_off = startFlag;
  }
}

// The same with the pair-emission rows staged through LDS.  The class list follows the x-major pair list, so the bands of
// a workgroup nearly always share x, the diagonal and the strand flag: at column j they all read row (context, quality) of
// x's base d + j of the emission table and differ only in the column (y's base j).  Per block of kSingleSub columns the
// workgroup copies those rows (Km * 95 doubles each, coalesced) into LDS and every lane picks its entries there; a lane
// whose band differs from the first band's (x, diagonal, strand) gathers from global memory as before.  Each lane also
// takes a whole 128-byte line (32 columns) of its y context words at a time.  (The per-lane gathers of the plain kernel are
// bound by L1 line fills: one 128-byte line per 8-byte entry.)
constexpr int kSingleSub = 8;    // columns per block (one barrier per block); each of the four wavefronts stages kSingleSub / 4 rows.
                                 // (8 measured the same 24 ms on config 3 at either 2 or 3 wavefronts per SIMD: the kernel streams ~30 GB of
                                 // y context words and traceback at 2.7 TB/s; barriers are not what it waits for.)
template <bool GAPCTX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void k_overlap_single_lds(OvArgs a) {
  struct __attribute__((packed, aligned(4))) W4 { uint32_t v[4]; };
  typedef double D2 __attribute__((ext_vector_type(2)));
  extern __shared__ double s_rows[];   // [2][kSingleSub][KQ]
  __shared__ unsigned long long s_xb0;
  __shared__ int s_d0, s_xLen0, s_T;
  __shared__ uint32_t s_x0, s_comp0, s_first;
  // Which band is this lane's?  Plain list: 256 consecutive entries.  Slotted list (SeedArgs::slot_list): workgroup b takes y
  // chunk (b / 8 / rows) * 8 + b % 8 and x row (b / 8) % rows -- the workgroups of one y chunk, one per x row, follow each
  // other on one XCD (workgroup ids go round the eight XCDs), so the chunk's context words (2 MB for 256 sequences of 2 kb)
  // come from that XCD's L2 for every x but the first instead of from HBM once per x.
  uint32_t uid = ~0u;
  if (a.slot_list) {
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, chunk = (slot / a.slot_rows) * 8 + xcd, row = slot % a.slot_rows;
    if (chunk < a.slot_ychunks) uid = a.slot_list[((uint64_t)chunk * a.slot_rows + row) * 256 + threadIdx.x];
  } else {
    const uint32_t uidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (uidx < a.n_cls_units) uid = a.cls_list[uidx];
  }
  const bool active = uid != ~0u;
  if (threadIdx.x == 0) { s_first = 256; s_T = 0; }
  __syncthreads();
  if (active) atomicMin(&s_first, threadIdx.x);
  __syncthreads();
  if (s_first == 256) return;                              // an empty slot group (below the diagonal of the pair triangle)
  uint32_t comp = 0, x = 0;
  int d = 0, xLen = 0, yLen = 0;
  uint64_t xb = 0, yb = 0;
  if (active) {
    const Unit u = a.units[uid];
    x = a.pair_x[u.pair];
    const uint32_t y = a.pair_y[u.pair];
    comp = a.pair_comp[u.pair];
    xb = a.seq_off[x]; xLen = (int)(a.seq_off[x + 1] - xb);
    yb = a.seq_off[y]; yLen = (int)(a.seq_off[y + 1] - yb);
    d = u.dlo;
  }
  if (threadIdx.x == s_first) { s_x0 = x; s_comp0 = comp; s_d0 = d; s_xb0 = xb; s_xLen0 = xLen; }   // the group's first band sets the shared rows
  __syncthreads();
  atomicMax(&s_T, active ? yLen : 0);
  __syncthreads();
  const int T = s_T;
  const bool shared_row = active && x == s_x0 && d == s_d0 && comp == s_comp0;
  const double* __restrict__ mmi = a.mmi[comp];
  const double* __restrict__ mmi0 = a.mmi[s_comp0];
  const double* __restrict__ gap = a.gap[comp];
  const double gap0 = gap[0];
  const uint32_t Kg = a.Kg, KQ = a.Km * (kNQualDev + 1);
  const uint32_t* __restrict__ xc = a.ctx + xb;
  const uint32_t* __restrict__ xc0 = a.ctx + s_xb0;
  const int d0 = s_d0, xLen0 = s_xLen0;
  const uint32_t* __restrict__ yc = (comp ? a.ctxc : a.ctx) + yb;
  double M = QF_NEG_INF, colBest = QF_NEG_INF, rowBest = QF_NEG_INF;
  uint32_t colI = 0, rowJ = 0, gxPrev = 0, gyPrev = 0, startFlag = 0;
  const uint32_t srow = threadIdx.x >> 6, scol = threadIdx.x & 63;   // staging: one wavefront per row of the block
  // Software pipeline, one barrier per block: block g's rows are in registers (loaded during block g-1's arithmetic), go to
  // LDS buffer g & 1, and the loads of block g+1's rows are issued before block g's arithmetic; the x context word that
  // names a row is fetched two blocks ahead, the lane's line of y context words one line ahead.
  constexpr int kRowRegs = 3;                                         // 16-byte chunks per lane per round: 64 x 3 x 2 = 384 >= 380 doubles
  const uint32_t rowChunks = KQ / 2;                                  // (KQ = Km * 95 with Km a power of four: even)
  const int rowRounds = (int)((rowChunks + 64 * kRowRegs - 1) / (64 * kRowRegs));   // 1 for order 0
  constexpr int kRowsPerWave = kSingleSub / 4;                        // this wavefront stages rows srow, srow + 4, ...
  struct Lead { uint32_t w[kRowsPerWave]; };
  auto lead_word = [&](int g) -> Lead {                               // context words of the first band's x at block g, this wavefront's rows
    Lead L;
#pragma unroll
    for (int rr = 0; rr < kRowsPerWave; ++rr)                         // base i = d0 + j, j = 1 + kSingleSub g + row -> index i - 1
      L.w[rr] = xc0[min(max(d0 + kSingleSub * g + (int)srow + 4 * rr, -kCtxPad + 8), xLen0)];
    return L;
  };
  D2 R[kRowsPerWave][kRowRegs];
  auto load_rows = [&](const Lead& L, int round) {
#pragma unroll
    for (int rr = 0; rr < kRowsPerWave; ++rr) {
      const D2* __restrict__ src = (const D2*)(mmi0 + (size_t)(L.w[rr] & 0x7FFFu) * KQ);
#pragma unroll
      for (int q = 0; q < kRowRegs; ++q) R[rr][q] = src[min((uint32_t)(round * kRowRegs + q) * 64 + scol, rowChunks - 1)];
    }
  };
  auto store_rows = [&](int b, int round) {
#pragma unroll
    for (int rr = 0; rr < kRowsPerWave; ++rr) {
      D2* dst = (D2*)(s_rows + ((size_t)b * kSingleSub + srow + 4 * rr) * KQ);
#pragma unroll
      for (int q = 0; q < kRowRegs; ++q) {
        const uint32_t ch = (uint32_t)(round * kRowRegs + q) * 64 + scol;
        if (ch < rowChunks) dst[ch] = R[rr][q];
      }
    }
  };
  const int nBlocks = (T + kSingleSub - 1) / kSingleSub;
  Lead wordCur = lead_word(0), wordNext = lead_word(1);
  if (rowRounds == 1) load_rows(wordCur, 0);
  W4 yw[8], ywNext[8];
  {
    const int yi = min(0, yLen);
#pragma unroll
    for (int q = 0; q < 8; ++q) ywNext[q] = *(const W4*)(yc + yi + 4 * q);
  }
  for (int j0 = 1; j0 <= T; j0 += 32) {
#pragma unroll
    for (int q = 0; q < 8; ++q) yw[q] = ywNext[q];
    {
      const int yi = min(j0 + 31, yLen);                    // next line; both context arrays are padded by kCtxPad words
#pragma unroll
      for (int q = 0; q < 8; ++q) ywNext[q] = *(const W4*)(yc + yi + 4 * q);
    }
#pragma unroll
    for (int sub = 0; sub < 32 / kSingleSub; ++sub) {
      const int js = j0 + sub * kSingleSub;                 // first column of the block
      const int g = (js - 1) / kSingleSub, buf = g & 1;
      if (g < nBlocks) {                                    // (uniform over the workgroup)
        if (rowRounds == 1) {
          store_rows(buf, 0);
        } else {                                            // long rows (order-1 contexts): not pipelined
          for (int round = 0; round < rowRounds; ++round) { load_rows(wordCur, round); store_rows(buf, round); }
        }
        __syncthreads();
        wordCur = wordNext;
        wordNext = lead_word(g + 2);
        if (rowRounds == 1 && g + 1 < nBlocks) load_rows(wordCur, 0);
        uint32_t wyc[kSingleSub];
#pragma unroll
        for (int c = 0; c < kSingleSub; ++c) { const int col = sub * kSingleSub + c; wyc[c] = yw[col >> 2].v[col & 3]; }
        const double* __restrict__ rows = s_rows + (size_t)buf * kSingleSub * KQ;
        // interior block: every column of it is inside this lane's band and away from the matrix edges, and the lane shares
        // the staged rows -> the recurrence is two additions per cell (no start, no end candidates, traceback nibble 0)
        const bool interior = shared_row && js > 1 && d + js > 1 && js + kSingleSub - 1 < yLen && d + js + kSingleSub - 1 < xLen;
        if (!GAPCTX && __all(interior)) {
#pragma unroll
          for (int c = 0; c < kSingleSub; ++c) M = (M + gap0) + rows[(size_t)c * KQ + (wyc[c] & 0x7FFFu)];
        } else {
          const int xi = min(max(d + js - 1, -kCtxPad + 8), xLen);
          W4 xa4[kSingleSub / 4];
#pragma unroll
          for (int q = 0; q < kSingleSub / 4; ++q) xa4[q] = *(const W4*)(xc + min(xi + 4 * q, xLen + 4));
          uint32_t xav[kSingleSub];
#pragma unroll
          for (int c = 0; c < kSingleSub; ++c) xav[c] = xa4[c >> 2].v[c & 3];
          double e[kSingleSub];
#pragma unroll
          for (int c = 0; c < kSingleSub; ++c) {
            const uint32_t ey = wyc[c] & 0x7FFFu;
            e[c] = shared_row ? rows[(size_t)c * KQ + ey] : mmi[(size_t)(xav[c] & 0x7FFFu) * KQ + ey];
          }
#pragma unroll
          for (int c = 0; c < kSingleSub; ++c) {
            const int j = js + c, i = d + j;
            const bool valid = active && j <= yLen && i >= 1 && i <= xLen;
            const uint32_t gxP = i > 1 ? gxPrev : 0u, gyP = j > 1 ? gyPrev : 0u;  // xIndelKmer[i-1], yIndelKmer[j-1] (padded 0)
            gxPrev = xav[c] >> 24; gyPrev = wyc[c] >> 24;
            const double tM = (M + (GAPCTX ? gap[gxP * Kg + gyP] : gap0)) + e[c];
            double nm = tM;
            if ((j == 1 || i == 1) && e[c] > nm) { nm = e[c]; if (valid) startFlag = 3; }   // the band's first cell only
            if (!valid) nm = QF_NEG_INF;
            M = nm;
            if (valid && j == yLen && nm >= colBest) { colBest = nm; colI = (uint32_t)i; }
            if (valid && i == xLen && (nm > rowBest || (nm == rowBest && (uint32_t)j > rowJ))) { rowBest = nm; rowJ = (uint32_t)j; }
          }
        }
      }
    }
  }
  if (active) {
    Unit* u = &a.units[uid];
    u->end_val = colBest; u->end_i = colI;
    u->end2_val = rowBest; u->end2_j = rowJ;
    u->tb_off = startFlag;   // the band's whole traceback (see k_overlap_single)
  }
}


// ------------------------------------------------------------------------------------------------
// Single-diagonal bands of the scheduler's row blocks, rebuilt around what bounded k_overlap_single_lds (profiles/r03_pmc_
// overlap.json: 12.9 instructions per cell for two additions; the L1 busy with one 128-byte line per lane for the y words and
// 12 bytes per cell of row staging; one barrier per 8 columns of a 256-pair workgroup):
//  * the pair-emission table is kept COMPACT -- only the quality values the resident sequences use, quality-major, rows
//    P doubles apart (k_mmi_compact) -- so a staged row is Km x nq doubles (672 bytes for 21 quality values) instead of 3 040,
//    and the rows reach LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write, one instruction per KB);
//  * what a lane needs of its y per column is the LDS byte offset of that base's column, two bytes, in arrays TRANSPOSED per
//    64 consecutive sequences (k_overlap_cols): a wavefront's fetch of 8 columns for its 64 bands is one contiguous KB;
//  * a lane runs TWO bands (y and y + 2048: the same XCD's chunks), so a workgroup is one x against 512 y: half the staging and
//    barriers per cell, and two independent add chains per lane;
//  * a cell is one mask-or-shift (the offset), one ds_read_b64 whose row and buffer sit in the instruction's offset field,
//    and the two additions of the recurrence.  Only the 8-column blocks in which some lane's diagonal starts or ends (its
//    first cell may start the alignment, src/qoverlap.cpp:141; its last is the band's only end cell, :153-170) run the
//    general form that captures them.
// Applies to the slotted list (x rows of the scheduler, band >= 2 so that the only single diagonal is the forced diagonal 0,
// src/diagenv.cpp:53), context-free gap scores and compact rows of at most 512 entries; everything else keeps the kernels above.
__global__ void k_mmi_compact(const double* __restrict__ mmi, uint32_t Km, uint32_t qmin, uint32_t nq, uint32_t pitch, double* __restrict__ out) {
  const uint32_t RS = Km * nq, KQ = Km * (kNQualDev + 1);
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= RS * RS) return;
  const uint32_t r = idx / RS, c = idx % RS;
  const uint32_t fr = (r % Km) * (kNQualDev + 1) + qmin + r / Km, fc = (c % Km) * (kNQualDev + 1) + qmin + c / Km;
  out[(size_t)r * pitch + c] = mmi[(size_t)fr * KQ + fc];
}
// compact index of a context word's emission row: (quality - qmin) * Km + context k-mer
__device__ __forceinline__ uint32_t ctx_compact(uint32_t word, uint32_t Km, uint32_t qmin) {
  const uint32_t erow = word & 0x7FFFu, k = erow / (kNQualDev + 1), q = erow - k * (kNQualDev + 1);
  return (q - qmin) * Km + k;
}
// per base of every sequence: byte offset of its row in the compact table (x side)
__global__ void k_overlap_xrows(const uint32_t* __restrict__ ctx, uint64_t total, uint32_t Km, uint32_t qmin, uint32_t pitch, uint32_t* __restrict__ out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x)
    out[i] = ctx_compact(ctx[i], Km, qmin) * pitch * 8u;
}
// per base of every sequence, both strands: byte offset of its column inside a row (y side), 16 bits, transposed per group of
// 64 consecutive sequences: chunk (goff[g] + b * 64 + l) holds columns 8b + 1 ... 8b + 8 of sequence 64g + l (zeros past its end)
__global__ __launch_bounds__(256) void k_overlap_cols(const uint32_t* __restrict__ ctx, const uint32_t* __restrict__ ctxc, const uint64_t* __restrict__ off,
                                                      uint32_t n_seqs, const uint64_t* __restrict__ goff, uint32_t Km, uint32_t qmin,
                                                      uint4* __restrict__ out0, uint4* __restrict__ out1) {
  const uint32_t g = blockIdx.x, l = threadIdx.x & 63u, y = g * 64 + l;
  const uint64_t g0 = goff[g];
  const uint32_t blocks = (uint32_t)((goff[g + 1] - g0) >> 6);
  const uint64_t yb = y < n_seqs ? off[y] : 0;
  const uint32_t len = y < n_seqs ? (uint32_t)(off[y + 1] - yb) : 0;
  for (uint32_t b = blockIdx.y * 4 + (threadIdx.x >> 6); b < blocks; b += gridDim.y * 4) {
    uint32_t v0[4] = {0, 0, 0, 0}, v1[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t c = 0; c < 8; ++c) {
      const uint32_t i = b * 8 + c;
      if (i < len) {
        v0[c >> 1] |= (ctx_compact(ctx[yb + i], Km, qmin) * 8u) << (16 * (c & 1));
        v1[c >> 1] |= (ctx_compact(ctxc[yb + i], Km, qmin) * 8u) << (16 * (c & 1));
      }
    }
    out0[g0 + (uint64_t)b * 64 + l] = make_uint4(v0[0], v0[1], v0[2], v0[3]);
    out1[g0 + (uint64_t)b * 64 + l] = make_uint4(v1[0], v1[1], v1[2], v1[3]);
  }
}

#ifndef QF_SR_SUB
#define QF_SR_SUB 8
#endif
#ifndef QF_SR_WAVES
#define QF_SR_WAVES 4
#endif
constexpr int kRowsSub = QF_SR_SUB;            // columns per staged block
template <int P> constexpr uint32_t single_rows_lds() { return 2u * kRowsSub * P * 8u + 64u; }
template <int P>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(QF_SR_WAVES))) void k_overlap_single_rows(OvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char s_crow[];     // [2][kRowsSub][P] doubles, then control words
  constexpr uint32_t kBufBytes = kRowsSub * P * 8u;
  int* s_ctl = (int*)(s_crow + 2 * kBufBytes);
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup b: XCD b % 8 (workgroup ids go round the XCDs), y chunks 16 s + xcd and 16 s + 8 + xcd of chunk set s = b / 8 / rows,
  // x row (b / 8) % rows: the rows of one set follow each other on one XCD, whose L2 then serves the set's column offsets
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, set = slot / a.slot_rows, row = slot % a.slot_rows;
  const uint32_t x = a.slot_x0 + row;
  const uint64_t xb = a.seq_off[x];
  const int xLen = (int)(a.seq_off[x + 1] - xb);
  uint32_t uid[2], comp[2] = {0, 0}, yblocks[2] = {1, 1};
  int yLen[2] = {0, 0};
  const uint4* yp[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const uint32_t ch = set * 16 + 8 * k + xcd;
    uid[k] = ch < a.slot_ychunks ? a.slot_list[((uint64_t)ch * a.slot_rows + row) * 256 + tid] : kNoUnit;
    yp[k] = a.ycolT[0] + lane;                                      // (a lane without a band reads somewhere harmless)
    if (uid[k] != kNoUnit) {
      const uint32_t pair = a.units[uid[k]].pair, y = a.pair_y[pair];   // y = ch * 256 + tid: lane == y % 64
      comp[k] = a.pair_comp[pair] ? 1u : 0u;
      yLen[k] = (int)(a.seq_off[y + 1] - a.seq_off[y]);
      const uint64_t g0 = a.ygoff[y >> 6];
      yblocks[k] = (uint32_t)((a.ygoff[(y >> 6) + 1] - g0) >> 6);
      yp[k] = a.ycolT[comp[k]] + g0 + lane;
    }
  }
  if (!__syncthreads_or(uid[0] != kNoUnit || uid[1] != kNoUnit)) return;   // an empty slot group (below the diagonal of the pair triangle)
  const uint32_t* __restrict__ xro = a.xrowoff + xb;
  for (uint32_t cp = 0; cp < 2; ++cp) {                              // the strand flag selects the table: one pass per flag present
    int nn[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) nn[k] = (uid[k] != kNoUnit && comp[k] == cp) ? min(xLen, yLen[k]) : 0;   // cells of the band (diagonal 0)
    int gEnd[2] = {(nn[0] - 1) >> 3, (nn[1] - 1) >> 3};              // block of the band's last cell (-1: no band in this pass)
    int T = max(nn[0], nn[1]), gFirst = min(gEnd[0] < 0 ? 0x7FFFFFFF : gEnd[0], gEnd[1] < 0 ? 0x7FFFFFFF : gEnd[1]);
#pragma unroll
    for (int o = 32; o; o >>= 1) { T = max(T, __shfl_xor(T, o)); gFirst = min(gFirst, __shfl_xor(gFirst, o)); }
    __syncthreads();
    if (tid == 0) s_ctl[0] = 0;
    __syncthreads();
    if (lane == 0) atomicMax(&s_ctl[0], T);
    __syncthreads();
    T = s_ctl[0];
    if (T == 0) continue;
    gFirst = __builtin_amdgcn_readfirstlane(gFirst);
    const int nBlocks = (T + kRowsSub - 1) / kRowsSub;
    const char* __restrict__ tab = (const char*)a.mmic[cp];
    const double g0 = a.gap[cp][0];
    // x's row offsets for block g, this wavefront's two rows (fetched a block ahead of the staging that uses them)
    struct Ro { uint32_t v[kRowsSub / 4]; };
    auto row_offsets = [&](int g) -> Ro {
      Ro r;
#pragma unroll
      for (uint32_t rr = 0; rr < kRowsSub / 4; ++rr) r.v[rr] = xro[min(kRowsSub * g + (int)(w + 4 * rr), xLen - 1)];
      return r;
    };
    auto stage = [&](const Ro& r, uint32_t buf) {                     // this wavefront's two rows of a block -> LDS buffer `buf`
#pragma unroll
      for (uint32_t rr = 0; rr < kRowsSub / 4; ++rr) {
        const uint32_t c = w + 4 * rr;
        const uint32_t ro = __builtin_amdgcn_readfirstlane(r.v[rr]);
        const char* src = tab + ro + lane * 16;
        char* dst = s_crow + buf * kBufBytes + c * (P * 8u);
#pragma unroll
        for (uint32_t i = 0; i < P / 128; ++i)
          if (lane + 64 * i < a.mmic_cpr)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
      }
    };
    uint4 yr[2][2];                                                  // [block parity][band]: column offsets of the block
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      yr[0][k] = yp[k][0];
      yr[1][k] = yp[k][(size_t)min(1u, yblocks[k] - 1) * 64];
    }
    double M[2] = {0, 0}, res[2] = {QF_NEG_INF, QF_NEG_INF};
    uint32_t flag[2] = {0, 0};
    stage(row_offsets(0), 0);
    Ro roNext = row_offsets(1);
    auto block = [&](auto BUFC, int g) {
      constexpr uint32_t BUF = decltype(BUFC)::value;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wavefront's rows of block g have landed ...
      __syncthreads();                                              // ... and everybody's; buffer BUF ^ 1 is no longer being read
      uint32_t ad[2][kRowsSub];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const uint32_t wd[4] = {yr[BUF][k].x, yr[BUF][k].y, yr[BUF][k].z, yr[BUF][k].w};
#pragma unroll
        for (int c = 0; c < kRowsSub; ++c) ad[k][c] = (c & 1) ? wd[c >> 1] >> 16 : wd[c >> 1] & 0xFFFFu;
      }
      const bool general = g == 0 || (g >= gFirst && __any(gEnd[0] == g || gEnd[1] == g));
      // the block's sixteen emissions in two halves of eight (all sixteen at once cost the registers of a fourth wavefront per SIMD)
      constexpr int H = kRowsSub / 2;
      double e[2][H];
      auto fetch = [&](int h) {
#pragma unroll
        for (int c = 0; c < H; ++c)
#pragma unroll
          for (int k = 0; k < 2; ++k) e[k][c] = *(const double*)(s_crow + BUF * kBufBytes + (h * H + c) * (P * 8u) + ad[k][h * H + c]);
      };
      auto chain = [&](int h) {
        if (!general) {
#pragma unroll
          for (int c = 0; c < H; ++c)
#pragma unroll
            for (int k = 0; k < 2; ++k) M[k] = (M[k] + g0) + e[k][c];
        } else {
#pragma unroll
          for (int c = 0; c < H; ++c)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              double m = (M[k] + g0) + e[k][c];
              if (g == 0 && h == 0 && c == 0) {                     // the diagonal's first cell: Start -> Match beats -inf + ... (src/qoverlap.cpp:141)
                m = e[k][c];
                flag[k] = e[k][c] > QF_NEG_INF ? 3u : 0u;
              }
              M[k] = m;
              if (kRowsSub * g + h * H + c + 1 == nn[k]) res[k] = m;   // the band's last cell, its only end cell
            }
        }
      };
      fetch(0);
      if (g + 1 < nBlocks) stage(roNext, BUF ^ 1u);
      roNext = row_offsets(g + 2);
#pragma unroll
      for (int k = 0; k < 2; ++k) yr[BUF][k] = yp[k][(size_t)min((uint32_t)g + 2, yblocks[k] - 1) * 64];
      chain(0);
      fetch(1);
      chain(1);
    };
    for (int g = 0; g < nBlocks; g += 2) {
      block(std::integral_constant<uint32_t, 0>(), g);
      if (g + 1 < nBlocks) block(std::integral_constant<uint32_t, 1>(), g + 1);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (nn[k] > 0) {
        Unit* u = &a.units[uid[k]];
        const bool colEnd = yLen[k] <= xLen, rowEnd = xLen <= yLen[k];   // the last cell is in y's last column / x's last row
        u->end_val = colEnd ? res[k] : QF_NEG_INF; u->end_i = colEnd ? (uint32_t)nn[k] : 0u;
        u->end2_val = rowEnd ? res[k] : QF_NEG_INF; u->end2_j = rowEnd ? (uint32_t)nn[k] : 0u;
        u->tb_off = flag[k];   // the band's whole traceback (see k_overlap_single)
      }
  }
}


// Row-space overlap Viterbi for bands wider than 512 diagonals (-kmatchoff, or a sequence shorter than 2(k+threshold):
// full envelope).  Geometry of k_viterbi_rows (qf_kernels.hip): one wavefront per unit, stripes of 64 lanes x 8 rows,
// lane l one column behind lane l-1, the stripe's last row handed on through a boundary buffer.  Arithmetic, candidate
// order and traceback bytes of k_overlap_fill.
__global__ __launch_bounds__(64) void k_overlap_rows(OvArgs a) {
  constexpr int G = 64, B = 8, S = kRowStripe;
  const uint32_t uidx = blockIdx.x;
  if (uidx >= a.n_cls_units) return;
  const int l = threadIdx.x;
  const uint32_t uid = a.cls_list[uidx];
  const Unit u = a.units[uid];
  const uint32_t x = a.pair_x[u.pair], y = a.pair_y[u.pair], comp = a.pair_comp[u.pair];
  const uint64_t xb = a.seq_off[x], yb = a.seq_off[y];
  const int xLen = (int)(a.seq_off[x + 1] - xb), yLen = (int)(a.seq_off[y + 1] - yb);
  const int dlo = u.dlo, dhi = u.dhi;
  const RowGeom g = row_geom(dlo, dhi, xLen, yLen);
  uint32_t* base = a.tb + u.tb_off;
  unsigned long long* stripe_off = (unsigned long long*)base;
  double* bnd = (double*)(base + 2ull * (g.nStripes + 1));
  uint32_t* tbw = base + row_header_words(g, yLen);
  const size_t bndStride = 3ull * (yLen + 2);
  if (l == 0) {
    unsigned long long w = 0;
    for (int s = 0; s < g.nStripes; ++s) {
      int jlo, jhi;
      row_stripe_cols(g, s, dlo, dhi, yLen, jlo, jhi);
      stripe_off[s] = w;
      if (jhi >= jlo) w += (unsigned long long)(jhi - jlo + 1 + 63) * 64 * 2;
    }
    stripe_off[g.nStripes] = w;
  }
  for (size_t c = l; c < bndStride; c += 64) bnd[c] = QF_NEG_INF;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");

  const double* __restrict__ mmi = a.mmi[comp];
  const double* __restrict__ gap = a.gap[comp];
  const double* __restrict__ tab = a.lse;
  const uint32_t Kg = a.Kg, KQ = a.Km * (kNQualDev + 1);
  const double* gsc = gap + 3ull * Kg * Kg;
  const double i2mS = gsc[1], i2iS = gsc[0], i2dS = gsc[2], d2mS = gsc[4], d2iS = gsc[3], d2dS = gsc[5];
  const uint32_t* __restrict__ xc = a.ctx + xb;
  const uint32_t* __restrict__ yc = (comp ? a.ctxc : a.ctx) + yb;
  double colBest = QF_NEG_INF, rowBest = QF_NEG_INF;
  uint32_t colI = 0, rowJ = 0;
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
    uint32_t erX[B], gkx[B], gkxP[B];   // row i: emission row, xIndelKmer[i], xIndelKmer[i-1] (padded 0)
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const int i = i0 + b;
      const uint32_t wx = (i >= 1 && i <= xLen) ? xc[i - 1] : 0u;
      const uint32_t wp = (i >= 2 && i <= xLen + 1) ? xc[i - 2] : 0u;
      erX[b] = wx & 0x7FFFu; gkx[b] = wx >> 24; gkxP[b] = i > 1 ? (wp >> 24) : 0u;
    }
    double M[B], I[B], D[B];
#pragma unroll
    for (int b = 0; b < B; ++b) M[b] = I[b] = D[b] = QF_NEG_INF;
    double p1M = QF_NEG_INF, p1I = QF_NEG_INF, p1D = QF_NEG_INF, p2M = QF_NEG_INF, p2I = QF_NEG_INF, p2D = QF_NEG_INF;
    const int steps = jhi - jlo + 1 + G - 1;
    for (int t = 0; t < steps; ++t) {
      const int j = jlo + t - l;
      const bool colvalid = j >= jlo && j <= jhi;
      const uint32_t wy = yc[min(max(j - 1, -kCtxPad + 1), yLen + 4)];
      const uint32_t erowY = wy & 0x7FFFu, gky = wy >> 24;
      const uint32_t gkyP = j > 1 ? (yc[min(j - 2, yLen + 4)] >> 24) : 0u;
      double upM = __shfl_up(p1M, 1, G), upI = __shfl_up(p1I, 1, G), upD = __shfl_up(p1D, 1, G);
      double dgM = __shfl_up(p2M, 1, G), dgI = __shfl_up(p2I, 1, G), dgD = __shfl_up(p2D, 1, G);
      if (l == 0) {
        const int jc = min(max(j, 0), yLen + 1), jp = min(max(j - 1, 0), yLen + 1);
        upM = bprev[jc]; upI = bprev[(yLen + 2) + jc]; upD = bprev[2 * (yLen + 2) + jc];
        dgM = bprev[jp]; dgI = bprev[(yLen + 2) + jp]; dgD = bprev[2 * (yLen + 2) + jp];
      }
      uint32_t tbw0 = 0, tbw1 = 0;
      double abM = upM, abI = upI, abD = upD;
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int i = i0 + b, dgl = i - j;
        const bool valid = colvalid && i >= 1 && i <= xLen && dgl >= dlo && dgl <= dhi;
        const double e = mmi[(size_t)erX[b] * KQ + erowY];
        const double m2m = gap[gkxP[b] * Kg + gkyP];
        const double m2i = gap[(size_t)Kg * Kg + gkx[b] * Kg + gkyP];
        const double m2d = gap[2ull * Kg * Kg + gkxP[b] * Kg + gky];
        const double oM = M[b], oI = I[b], oD = D[b];   // (i, j-1)
        const double tM = (dgM + m2m) + e, tI = (dgI + i2mS) + e, tD = (dgD + d2mS) + e;
        double nm = tM;
        uint32_t sm = 0;
        if (tI > nm) { nm = tI; sm = 1; }
        if (tD > nm) { nm = tD; sm = 2; }
        if ((j == 1 || i == 1) && e > nm) { nm = e; sm = 3; }
        const double iM = oM + m2i, iI = oI + i2iS, iD = oD + d2iS;
        double ni = max_lse_exact<false>(tab, iI, iD, iM);
        uint32_t si = 0;
        { double sv = iM; if (iI > sv) { sv = iI; si = 1; } if (iD > sv) { sv = iD; si = 2; } }
        const double dM = abM + m2d, dD = abD + d2dS, dIfill = abI + d2iS, dItb = abI + i2dS;
        double ndl = max_lse_exact<false>(tab, dD, dIfill, dM);
        uint32_t sd = 0;
        { double sv = dM; if (dItb > sv) { sv = dItb; sd = 1; } if (dD > sv) { sv = dD; sd = 2; } }
        if (!valid) { nm = QF_NEG_INF; ni = QF_NEG_INF; ndl = QF_NEG_INF; }
        M[b] = nm; I[b] = ni; D[b] = ndl;
        dgM = oM; dgI = oI; dgD = oD;
        abM = nm; abI = ni; abD = ndl;
        const uint32_t byte = sm | (si << 2) | (sd << 4);
        if (b < 4) tbw0 |= byte << (8 * b); else tbw1 |= byte << (8 * (b - 4));
        if (valid && j == yLen && nm >= colBest) { colBest = nm; colI = (uint32_t)i; }
        if (valid && i == xLen && (nm > rowBest || (nm == rowBest && (uint32_t)j > rowJ))) { rowBest = nm; rowJ = (uint32_t)j; }
      }
      p2M = p1M; p2I = p1I; p2D = p1D;
      p1M = M[B - 1]; p1I = I[B - 1]; p1D = D[B - 1];
      if (colvalid) {
        tb_store(&tbw[woff + ((unsigned long long)t * G + l) * 2], tbw0);
        tb_store(&tbw[woff + ((unsigned long long)t * G + l) * 2 + 1], tbw1);
        if (l == G - 1) { bnext[j] = p1M; bnext[(yLen + 2) + j] = p1I; bnext[2 * (yLen + 2) + j] = p1D; }
      }
    }
    woff += (unsigned long long)steps * G * 2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  }
  for (int o = 1; o < G; o <<= 1) {
    const double ov = __shfl_xor(colBest, o, G);
    const uint32_t oi = __shfl_xor(colI, o, G);
    if (ov > colBest || (ov == colBest && oi > colI)) { colBest = ov; colI = oi; }
    const double rv = __shfl_xor(rowBest, o, G);
    const uint32_t rj = __shfl_xor(rowJ, o, G);
    if (rv > rowBest || (rv == rowBest && rj > rowJ)) { rowBest = rv; rowJ = rj; }
  }
  if (l == 0) {
    Unit* uu = &a.units[uid];
    uu->end_val = colBest; uu->end_i = colI;
    uu->end2_val = rowBest; uu->end2_j = rowJ;
  }
}

// End cell (src/qoverlap.cpp:164-182): start from mat(xLen,yLen), scan the last read column downwards, then the last
// reference row, replacing only on strict '>'.  result = end + xInsertScore + yInsertScore (:157); adjusted score
// subtracts both reads' null log-likelihoods (:292-302).
// 64-bit sum over the wavefront's lanes; the result is valid in every lane
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += (unsigned long long)__shfl_xor((long long)v, m, 64);
  return v;
}

// (A fixed grid striding over the pairs: the totals are three words, and one atomic per wavefront on each -- 800 000 of them per
// 2^24-pair row block -- serialised in L2 for 6 ms; now one per workgroup of the fixed grid.)
constexpr int kFinalizeBlocks = 2048;
__global__ __launch_bounds__(256) void k_overlap_finalize(OvArgs a) {
  __shared__ unsigned long long s_tot[3];
  if (threadIdx.x < 3) s_tot[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long t_finite = 0, t_ndiag = 0, t_bits = 0;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < a.n_pairs; p += gridDim.x * blockDim.x) {
    const uint32_t x = a.pair_x[p], y = a.pair_y[p], comp = a.pair_comp[p];
    const uint32_t xLen = (uint32_t)(a.seq_off[x + 1] - a.seq_off[x]), yLen = (uint32_t)(a.seq_off[y + 1] - a.seq_off[y]);
    double cb = QF_NEG_INF, rb = QF_NEG_INF;
    uint32_t ci = 0, cu = kNoUnit, rj = 0, ru = kNoUnit;
    for (uint32_t uid = a.pair_head[p]; uid != kNoUnit; uid = a.units[uid].next) {
      const Unit& u = a.units[uid];
      if (u.end_val > cb || (u.end_val == cb && u.end_val > QF_NEG_INF && u.end_i > ci)) { cb = u.end_val; ci = u.end_i; cu = uid; }
      if (u.end2_val > rb || (u.end2_val == rb && u.end2_val > QF_NEG_INF && u.end2_j > rj)) { rb = u.end2_val; rj = u.end2_j; ru = uid; }
    }
    double end = cb;
    uint32_t ei = ci, ej = yLen, eu = cu;
    if (rb > cb) { end = rb; ei = xLen; ej = rj; eu = ru; }
    const double yins = comp ? a.ins_sum_c[y] : a.ins_sum[y];
    const double result = end + a.ins_sum[x] + yins;
    double score = result - a.nll[x];
    score -= comp ? a.nll_c[y] : a.nll[y];
    const bool keep = end > QF_NEG_INF && score >= a.min_score;
    if (a.per_pair) {   // the per-pair arrays of the pair-list entry point; a row block reports totals and the kept alignments only
      a.pair_result[p] = result;
      a.pair_score[p] = score;
      a.pair_end_unit[p] = end > QF_NEG_INF ? eu : kNoUnit;
    }
    if (a.per_pair || keep) {   // (the traceback starts from here)
      a.pair_end_ij[2 * p] = ei;
      a.pair_end_ij[2 * p + 1] = ej;
    }
    t_ndiag += a.pair_ndiag[p];
    if (end > QF_NEG_INF) {
      t_finite += 1;
      t_bits += (unsigned long long)__double_as_longlong(result);
    }
    if (keep) {
      const Unit& u = a.units[eu];
      const uint32_t cap = xLen + yLen + (uint32_t)(u.dhi - u.dlo + 1) + 4;
      const uint32_t idx = atomicAdd(&a.bc->n_align, 1u);
      AlignRec rec{};
      rec.read = p;  // pair index
      rec.ref = x;
      rec.unit = eu;
      rec.viterbi = result;
      rec.score = score;
      rec.tmp_off = atomicAdd(&a.bc->n_runs, (unsigned long long)cap);
      a.recs[idx] = rec;
    }
  }
  // totals over the pairs (what the row-block entry point reports instead of per-pair arrays): one atomic per workgroup each
  t_finite = wave_sum_u64(t_finite);
  t_ndiag = wave_sum_u64(t_ndiag);
  t_bits = wave_sum_u64(t_bits);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&s_tot[0], t_finite);
    atomicAdd(&s_tot[1], t_ndiag);
    atomicAdd(&s_tot[2], t_bits);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_tot[0]) atomicAdd(&a.bc->n_finite, s_tot[0]);
    if (s_tot[1]) atomicAdd(&a.bc->sum_ndiag, s_tot[1]);
    if (s_tot[2]) atomicAdd(&a.bc->result_sum, s_tot[2]);
  }
}

// The scheduler's pair enumeration for rows [x0, x0 + rows) (QuaffOverlapScheduler::advance, src/qoverlap.cpp:475-480): row nx
// holds ny = nx + 1 ... n_seqs - 1, yComplemented = ny >= nOriginals (:540).  blockIdx.y = row; pair index = pairs of the
// earlier rows + (ny - nx - 1).
__global__ void k_overlap_row_pairs(uint32_t x0, uint32_t n_seqs, uint32_t n_orig, uint32_t* __restrict__ px,
                                    uint32_t* __restrict__ py, uint8_t* __restrict__ pc) {
  const uint32_t r = blockIdx.y, x = x0 + r;
  const uint32_t len = n_seqs - 1 - x;
  const unsigned long long base = (unsigned long long)r * (n_seqs - 1 - x0) - (unsigned long long)r * (r - 1) / 2;
  for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < len; q += gridDim.x * blockDim.x) {
    const uint32_t y = x + 1 + q;
    px[base + q] = x;
    py[base + q] = y;
    pc[base + q] = y >= n_orig;
  }
}

// QuaffOverlapViterbiMatrix::alignment traceback (src/qoverlap.cpp:184-268) from the per-cell bytes; raw M/I/D
// state runs are emitted, the indel "squashing" of :231-267 is a host-side re-pairing of adjacent runs.
__global__ void k_overlap_traceback(OvArgs a) {
  const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.n_recs) return;
  AlignRec rec = a.recs[idx];
  const uint32_t p = rec.read;
  const Unit u = a.units[rec.unit];
  const FillClass fc = fill_class((int)u.cls);
  const uint32_t* __restrict__ tb = a.tb + u.tb_off;
  const uint32_t px = a.pair_x[p], py = a.pair_y[p];
  const int xLenR = (int)(a.seq_off[px + 1] - a.seq_off[px]), yLenR = (int)(a.seq_off[py + 1] - a.seq_off[py]);
  const RowGeom rg = u.cls == (uint32_t)kRowClass ? row_geom(u.dlo, u.dhi, xLenR, yLenR) : RowGeom{0, 0, 0};
  auto cellbyte = [&](int i, int j) -> uint32_t {
    if (u.cls == 0) return (i == 1 || j == 1) ? (uint32_t)u.tb_off & 0x3u : 0u;  // single diagonal: the first cell's start flag is all there is
    if (u.cls == (uint32_t)kRowClass) {
      const int rr = i - rg.ilo, s = rr / kRowStripe, li = (rr % kRowStripe) / 8, b = rr % 8;
      int jlo, jhi;
      row_stripe_cols(rg, s, u.dlo, u.dhi, yLenR, jlo, jhi);
      const unsigned long long* so = (const unsigned long long*)tb;
      const uint32_t* words = tb + row_header_words(rg, yLenR);
      return (words[so[s] + ((unsigned long long)(j - jlo + li) * 64 + li) * 2 + (b >> 2)] >> (8 * (b & 3))) & 0xFFu;
    }
    const int dd = (i - j) - u.dlo, l = dd / fc.B, b = dd % fc.B;
    const uint64_t w = ((uint64_t)(j - 1 - band_col0(u.dhi) + l) * fc.G + l) * 2 + (b >> 2);
    return (tb[w] >> (8 * (b & 3))) & 0xFFu;
  };
  uint32_t* tmp = a.runs_tmp + rec.tmp_off;
  int i = (int)a.pair_end_ij[2 * p], j = (int)a.pair_end_ij[2 * p + 1];
  const uint32_t xEnd = (uint32_t)i, yEnd = (uint32_t)j;
  uint32_t n = 0, ncol = 0, curOp = 3, curLen = 0;
  int state = 1;  // 0 Start, 1 Match, 2 Insert, 3 Delete
  while (state != 0 && i >= 0 && j >= 0 && (i > 0 || j > 0)) {
    const uint32_t byte = (i >= 1 && j >= 1) ? cellbyte(i, j) : 0u;
    uint32_t op, s;
    // (match source 3 = Start on row / column 1, the only place a start candidate exists; elsewhere the FAST steps' raw compare bits
    // "I > M and D > max(M, I)", i.e. D)
    if (state == 1) { op = 0; s = byte & 3u; const bool edge = i == 1 || j == 1; --i; --j; state = s == 0 ? 1 : s == 1 ? 2 : (s == 2 || !edge) ? 3 : 0; }
    else if (state == 2) { op = 1; s = (byte >> 2) & 3u; --j; state = s == 0 ? 1 : s == 1 ? 2 : 3; }
    else { op = 2; s = (byte >> 4) & 3u; --i; state = s == 0 ? 1 : s == 1 ? 2 : 3; }
    ++ncol;
    if (op == curOp) ++curLen;
    else {
      if (curLen) tmp[n++] = (curLen << 2) | curOp;
      curOp = op; curLen = 1;
    }
  }
  if (curLen) tmp[n++] = (curLen << 2) | curOp;
  const unsigned long long off = atomicAdd(&a.bc->total_runs_out, (unsigned long long)n);
  for (uint32_t c = 0; c < n; ++c) a.runs_out[off + c] = tmp[n - 1 - c];
  rec.x_start = (uint32_t)(i + 1);
  rec.x_end = xEnd;
  rec.y_start = (uint32_t)(j + 1);
  rec.y_end = yEnd;
  rec.n_columns = ncol;
  rec.n_runs = n;
  rec.run_off = off;
  rec.ok = state == 0;
  a.recs[idx] = rec;
}

template <int G, int B>
static void launch_ov_gb(const OvArgs& a, hipStream_t s) {
  const uint32_t upw = 64 / G, waves = (a.n_cls_units + upw - 1) / upw;
  if (a.lse_pack) {   // the packed table takes most of a CU's LDS: one workgroup of eight wavefronts per CU
    auto fn = a.Kg > 1 ? k_overlap_fill<G, B, true, true> : k_overlap_fill<G, B, false, true>;
    (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)a.lse_pack_bytes);
    constexpr uint32_t W = ov_pack_waves(G, B);
    hipLaunchKernelGGL(fn, dim3((waves + W - 1) / W), dim3(64 * W), a.lse_pack_bytes, s, a);
  } else {
    auto fn = a.Kg > 1 ? k_overlap_fill<G, B, true, false> : k_overlap_fill<G, B, false, false>;
    hipLaunchKernelGGL(fn, dim3((waves + 3) / 4), dim3(256), 0, s, a);
  }
}
uint32_t lse_pack_mismatches(const uint8_t* pack, uint32_t pack_bytes, const double* tab, uint32_t* d_bad, hipStream_t s) {
  (void)hipFuncSetAttribute((const void*)k_lse_pack_check, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pack_bytes);
  (void)hipMemsetAsync(d_bad, 0, 4, s);
  hipLaunchKernelGGL(k_lse_pack_check, dim3(64), dim3(256), pack_bytes, s, pack, pack_bytes, tab, d_bad);
  uint32_t bad = ~0u;
  if (hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return ~0u;
  return bad;
}
void launch_overlap_fill(int cls, const OvArgs& a, hipStream_t s) {
  if (!a.n_cls_units) return;
  switch (cls) {
    case 0: {
      if (a.slot_list && a.mmic_pitch && a.Kg == 1 && !a.no_lds_rows) {
        const uint32_t blocks = ((a.slot_ychunks + 15) / 16) * 8 * a.slot_rows;
        auto go = [&](auto fn, uint32_t lds) {
          (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), lds, s, a);
        };
        if (a.mmic_pitch == 128) go(k_overlap_single_rows<128>, single_rows_lds<128>());
        else if (a.mmic_pitch == 256) go(k_overlap_single_rows<256>, single_rows_lds<256>());
        else go(k_overlap_single_rows<512>, single_rows_lds<512>());
        break;
      }
      const size_t row_lds = 2ull * kSingleSub * a.Km * (kNQualDev + 1) * 8;   // order 0 / 1 emission rows fit, longer contexts do not
      if (row_lds <= 64 * 1024 && !a.no_lds_rows) {
        auto fn = a.Kg > 1 ? k_overlap_single_lds<true> : k_overlap_single_lds<false>;
        if (row_lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)row_lds);
        const uint32_t blocks = a.slot_list ? ((a.slot_ychunks + 7) / 8) * 8 * a.slot_rows : (a.n_cls_units + 255) / 256;
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), row_lds, s, a);
      } else
        hipLaunchKernelGGL(k_overlap_single, dim3((a.n_cls_units + 255) / 256), dim3(256), 0, s, a);
      break;
    }
    case 1: launch_ov_gb<16, 2>(a, s); break;
    case 2: launch_ov_gb<16, 3>(a, s); break;
    case 3: launch_ov_gb<16, 4>(a, s); break;
    case 4: launch_ov_gb<16, 5>(a, s); break;
    case 5: launch_ov_gb<16, 6>(a, s); break;
    case 6: launch_ov_gb<16, 8>(a, s); break;
    case 7: launch_ov_gb<64, 3>(a, s); break;
    case 8: launch_ov_gb<64, 4>(a, s); break;
    case 9: launch_ov_gb<64, 6>(a, s); break;
    case 10: launch_ov_gb<64, 8>(a, s); break;
    case 13: hipLaunchKernelGGL(k_overlap_rows, dim3(a.n_cls_units), dim3(64), 0, s, a); break;
    case 14: launch_ov_gb<32, 3>(a, s); break;
  }
}
uint32_t overlap_compact_pitch(uint32_t Km, uint32_t nq) {
  const uint32_t rs = Km * nq;
  return rs > 512 || (rs & 1u) ? 0u : rs <= 128 ? 128u : rs <= 256 ? 256u : 512u;
}
void launch_mmi_compact(const double* mmi, uint32_t Km, uint32_t qmin, uint32_t nq, uint32_t pitch, double* out, hipStream_t s) {
  const uint32_t n = Km * nq * Km * nq;
  hipLaunchKernelGGL(k_mmi_compact, dim3((n + 255) / 256), dim3(256), 0, s, mmi, Km, qmin, nq, pitch, out);
}
void launch_overlap_cols(const uint32_t* ctx, const uint32_t* ctxc, const uint64_t* off, uint32_t n_seqs, uint64_t total, const uint64_t* goff,
                         uint32_t max_blocks, uint32_t Km, uint32_t qmin, uint32_t pitch, uint32_t* xrowoff, uint4* col0, uint4* col1, hipStream_t s) {
  if (!n_seqs) return;
  hipLaunchKernelGGL(k_overlap_xrows, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, 65536)), dim3(256), 0, s, ctx, total, Km, qmin, pitch, xrowoff);
  const uint32_t groups = (n_seqs + 63) / 64;
  hipLaunchKernelGGL(k_overlap_cols, dim3(groups, std::max(1u, std::min(64u, (max_blocks + 3) / 4))), dim3(256), 0, s, ctx, ctxc, off, n_seqs, goff, Km, qmin,
                     col0, col1);
}
bool overlap_single_stages_rows(uint32_t Km) { return 2ull * kSingleSub * Km * (kNQualDev + 1) * 8 <= 64 * 1024; }
void launch_prep_overlap(const PrepArgs& a, uint32_t n, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_prep_overlap, dim3(n), dim3(64), 0, s, a);
  hipLaunchKernelGGL(k_overlap_sums, dim3((n + 63) / 64), dim3(64), 0, s, a, (const uint32_t*)a.ctx, n);
}
void launch_overlap_finalize(const OvArgs& a, hipStream_t s) {
  if (a.n_pairs) hipLaunchKernelGGL(k_overlap_finalize, dim3(std::min<uint32_t>((a.n_pairs + 255) / 256, kFinalizeBlocks)), dim3(256), 0, s, a);
}
void launch_overlap_row_pairs(uint32_t x0, uint32_t rows, uint32_t n_seqs, uint32_t n_orig, uint32_t* px, uint32_t* py, uint8_t* pc,
                              hipStream_t s) {
  if (!rows) return;
  const uint32_t longest = n_seqs - 1 - x0;
  const uint32_t bx = std::max(1u, std::min(64u, (longest + 255) / 256));
  hipLaunchKernelGGL(k_overlap_row_pairs, dim3(bx, rows), dim3(256), 0, s, x0, n_seqs, n_orig, px, py, pc);
}
void launch_overlap_traceback(const OvArgs& a, hipStream_t s) {
  if (a.n_recs) hipLaunchKernelGGL(k_overlap_traceback, dim3((a.n_recs + 63) / 64), dim3(64), 0, s, a);
}

}  // namespace qf
