// qf_kernels.hip — hand-written HIP kernels for gfx950 (MI355X): read/ref preparation, k-mer index,
// diagonal seeding, banded Viterbi fill (anti-diagonal-skewed wavefront), end-cell reduction and
// traceback.  fp64 throughout; compiled with -ffp-contract=off so every add/mul rounds exactly as the
// reference's scalar code does.  See DESIGN.md for the layout and the roofline of each kernel.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "qf_dpp.hpp"
#include "qf_kernels.hpp"

namespace qf {

#define QF_NEG_INF (-__builtin_huge_val())

struct __attribute__((packed, aligned(4))) U32x4 { uint32_t v[4]; };   // 16-byte load from a 4-byte aligned address
struct __attribute__((packed, aligned(1))) U8x16 { uint32_t v[4]; };   // 16-byte load from any address

// ------------------------------------------------------------------------------------------------
// Sequence preparation
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tokenize_char(int c) {  // tokenize(), src/fastseq.cpp:11-16
  c &= ~0x20;  // toupper for letters
  return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1;
}

// One thread per reference base: token bytes; one thread per 16 bases: 2-bit packed words.
__global__ void k_prep_ref(const char* __restrict__ seq, uint64_t total, uint8_t* __restrict__ tok,
                           BatchCounters* bc) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int t = tokenize_char((unsigned char)seq[i]);
  if (t < 0) {
    atomicOr(&bc->error, 4u);
    bc->error_detail = (uint32_t)i;
    t = 0;
  }
  tok[i] = (uint8_t)t;
}

__global__ void k_pack_ref(const uint8_t* __restrict__ tok, const uint64_t* __restrict__ off,
                           const uint64_t* __restrict__ woff, uint32_t n_refs, uint32_t* __restrict__ packed) {
  const uint32_t x = blockIdx.y;
  const uint64_t b = off[x], len = off[x + 1] - b;
  const uint64_t nw = (len + 15) / 16 + 2;  // two zero words of slack
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nw) return;
  uint32_t v = 0;
  for (int a = 0; a < 16; ++a) {
    const uint64_t p = w * 16 + a;
    if (p < len) v |= (uint32_t)tok[b + p] << (2 * a);
  }
  packed[woff[x] + w] = v;
}

// k-mer index of each reference: counting sort of positions by k-mer value (KmerIndex, src/fastseq.cpp:240-256,
// built on the reference side once instead of on every read: the (i,j) match set is symmetric).
__global__ void k_ref_kmer_count(const uint8_t* __restrict__ tok, const uint64_t* __restrict__ off, uint32_t k,
                                 uint32_t nbuckets, uint32_t* __restrict__ counts) {
  const uint32_t x = blockIdx.y;
  const uint64_t b = off[x], len = off[x + 1] - b;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (len < k || i > len - k) return;
  uint32_t km = 0;
  for (uint32_t a = 0; a < k; ++a) km = km * 4 + tok[b + i + a];
  atomicAdd(&counts[(uint64_t)x * (nbuckets + 1) + km], 1u);
}

// exclusive scan of each reference's bucket counts (one block per reference)
// pad: every bucket's count rounded up to a multiple of `pad` first (1 = as counted): the padded chunk index of k_seed_rows_lds
__global__ __launch_bounds__(1024) void k_bucket_scan(uint32_t* __restrict__ counts, uint32_t nbuckets, uint32_t pad = 1) {
  __shared__ uint32_t part[1024];
  uint32_t* c = counts + (uint64_t)blockIdx.x * (nbuckets + 1);
  const uint32_t tid = threadIdx.x, per = (nbuckets + 1 + 1023) / 1024;
  const uint32_t lo = tid * per, hi = min(lo + per, nbuckets + 1);
  uint32_t s = 0;
  if (pad > 1)
    for (uint32_t a = lo; a < hi; ++a) c[a] = (c[a] + pad - 1) / pad * pad;
  for (uint32_t a = lo; a < hi; ++a) s += c[a];
  part[tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    uint32_t v = tid >= d ? part[tid - d] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  uint32_t run = tid ? part[tid - 1] : 0;
  for (uint32_t a = lo; a < hi; ++a) {
    const uint32_t v = c[a];
    c[a] = run;
    run += v;
  }
}

__global__ void k_ref_kmer_scatter(const uint8_t* __restrict__ tok, const uint64_t* __restrict__ off, uint32_t k,
                                   uint32_t nbuckets, const uint32_t* __restrict__ starts,
                                   uint32_t* __restrict__ cursor, uint32_t* __restrict__ pos) {
  const uint32_t x = blockIdx.y;
  const uint64_t b = off[x], len = off[x + 1] - b;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (len < k || i > len - k) return;
  uint32_t km = 0;
  for (uint32_t a = 0; a < k; ++a) km = km * 4 + tok[b + i + a];
  const uint64_t bi = (uint64_t)x * (nbuckets + 1) + km;
  const uint32_t slot = starts[bi] + atomicAdd(&cursor[bi], 1u);
  pos[b + slot] = (uint32_t)i;
}

// k_prep_reads2: one wavefront per read, sixteen bases per lane: tokens, packed per-column context words and seeding k-mers
// (FastSeq::tokens/kmers/qualScores src/fastseq.cpp:71-109; QuaffDPMatrix ctor src/qmodel.cpp:1308-1324).  A
// byte-at-a-time version spends its time in ~10 single-byte gathers per base (every base re-reads its context and
// seed-k-mer neighbours); here each base is tokenised once (SWAR on 4 characters), staged in LDS, and the context /
// seeding k-mers roll over a 48-token register window.  One wavefront per read, 1024 bases per tile.  Seeding k-mers are produced by their LAST base (the k-mer that
// starts at i is stored when base i + k - 1 is reached), so the window only looks backwards.
struct __attribute__((packed, aligned(1))) C16 { uint32_t v[4]; };
struct __attribute__((packed, aligned(4))) W4a { uint32_t v[4]; };
struct __attribute__((packed, aligned(4))) L4a { unsigned long long v[2]; };

__device__ __forceinline__ uint32_t zero_bytes(uint32_t v) {  // 0x80 in every byte of v that is zero (exact)
  const uint32_t t = (v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
  return ~(t | v | 0x7F7F7F7Fu);
}

template <bool K64>
__global__ __launch_bounds__(64) void k_prep_reads2(PrepArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t s_tok[(32 + 1024) / 4];
  const uint32_t r = blockIdx.x, lane = threadIdx.x;
  const uint64_t b = a.off[r];
  const uint32_t L = (uint32_t)(a.off[r + 1] - b);
  const uint32_t mlen = a.match_len, glen = a.gap_len, K = a.seed_k;
  const uint32_t maskM = (1u << (2 * mlen)) - 1u, maskG = glen ? (1u << (2 * glen)) - 1u : 0u;
  const unsigned long long maskK = K >= 32 ? ~0ull : ((1ull << (2 * K)) - 1ull);
  uint32_t cnt1 = 0, cnt2 = 0, cnt3 = 0, nvalid = 0;
  if (lane < 2) ((uint4*)s_tok)[lane] = make_uint4(0, 0, 0, 0);   // no bases before the read (pad fixed up below)
  for (uint32_t T0 = 0; T0 < L; T0 += 1024) {
    const uint32_t p0 = T0 + lane * 16;
    // ---- tokenise this lane's 16 characters
    uint32_t tw[4] = {0, 0, 0, 0};
    if (p0 < L) {
      const C16 ch = *(const C16*)(a.seq + b + p0);
      bool bad = false;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t cw = ch.v[w], up = cw & 0xDFDFDFDFu;
        const uint32_t ok = zero_bytes(up ^ 0x41414141u) | zero_bytes(up ^ 0x43434343u) | zero_bytes(up ^ 0x47474747u) |
                            zero_bytes(up ^ 0x54545454u);
        const uint32_t x = (cw >> 1) & 0x03030303u;            // A 0, C 1, T 2, G 3
        uint32_t t = x ^ ((x >> 1) & 0x01010101u);             // A 0, C 1, G 2, T 3
        uint32_t live = 0;                                     // 0x80 in the bytes that are inside the read
#pragma unroll
        for (int c = 0; c < 4; ++c) if (p0 + w * 4 + c < L) live |= 0x80u << (8 * c);
        bad |= (live & ~ok) != 0;
        const uint32_t keep = ok & live;
        t &= (keep >> 7) | (keep >> 6);
        tw[w] = t;
        const uint32_t lo = t & 0x01010101u, hi = (t >> 1) & 0x01010101u;
        cnt1 += __popc(lo & ~hi); cnt2 += __popc(hi & ~lo); cnt3 += __popc(lo & hi);
        nvalid += __popc(live);
      }
      if (bad) {
        atomicOr(&a.bc->error, 4u);
        a.bc->error_detail = r;
      }
    }
    if (T0 && lane < 2) ((uint4*)s_tok)[lane] = ((const uint4*)s_tok)[64 + lane];  // last 32 tokens of the previous tile
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    ((uint4*)s_tok)[2 + lane] = make_uint4(tw[0], tw[1], tw[2], tw[3]);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (p0 < L) {
      // window: tokens of bases p0-32 .. p0+15 (byte o <-> base p0 - 32 + o)
      uint32_t W[12];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const uint4 v = ((const uint4*)s_tok)[lane + q];
        W[4 * q] = v.x; W[4 * q + 1] = v.y; W[4 * q + 2] = v.z; W[4 * q + 3] = v.w;
      }
      auto tk = [&](int o) -> uint32_t { return (W[o >> 2] >> (8 * (o & 3))) & 3u; };
      C16 qc{};
      if (a.qual) qc = *(const C16*)(a.qual + b + p0);
      uint32_t mk = 0, gk = 0;
#pragma unroll
      for (int o = 28; o < 32; ++o) { mk = ((mk << 2) | tk(o)) & maskM; gk = ((gk << 2) | tk(o)) & maskG; }
      uint32_t cw[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const uint32_t t = tk(32 + e);
        mk = ((mk << 2) | t) & maskM;
        gk = ((gk << 2) | t) & maskG;
        uint32_t q = kNQualDev;
        if (a.qual) {
          const int v = (int)(signed char)((qc.v[e >> 2] >> (8 * (e & 3))) & 0xFFu) - '!';
          q = (uint32_t)max(0, min(kNQualDev - 1, v));
        }
        cw[e] = ctx_pack(a.em_qmajor_Km ? (q - a.em_qmin) * a.em_qmajor_Km + mk : mk * (kNQualDev + 1) + q, t * (kNQualDev + 1) + q, gk);
      }
      const bool whole = p0 + 16 <= L;
      uint32_t* cdst = a.ctx + b + p0;
      uint8_t* tdst = a.tok + b + p0;
      if (whole) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { W4a o4; o4.v[0] = cw[4 * q]; o4.v[1] = cw[4 * q + 1]; o4.v[2] = cw[4 * q + 2]; o4.v[3] = cw[4 * q + 3]; *(W4a*)(cdst + 4 * q) = o4; }
        C16 t16; t16.v[0] = tw[0]; t16.v[1] = tw[1]; t16.v[2] = tw[2]; t16.v[3] = tw[3];
        *(C16*)tdst = t16;
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (p0 + e < L) { cdst[e] = cw[e]; tdst[e] = (uint8_t)((tw[e >> 2] >> (8 * (e & 3))) & 3u); }
      }
      // ---- seeding k-mers by their last base
      if (K) {
        if (K64) {
          unsigned long long sk = 0;
#pragma unroll
          for (int o = 0; o < 32; ++o) sk = (sk << 2) | tk(o);
          unsigned long long kv[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) { sk = (sk << 2) | tk(32 + e); kv[e] = sk & maskK; }
          const long long i0 = (long long)p0 - (long long)K + 1;   // start of the k-mer that ends at p0
          if (whole && i0 >= 0) {
            unsigned long long* dst = a.skmer64 + b + i0;
#pragma unroll
            for (int q = 0; q < 8; ++q) { L4a o2; o2.v[0] = kv[2 * q]; o2.v[1] = kv[2 * q + 1]; *(L4a*)(dst + 2 * q) = o2; }
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e)
              if (i0 + e >= 0 && p0 + e < L) a.skmer64[b + i0 + e] = kv[e];
          }
        } else {
          uint32_t sk = 0;
#pragma unroll
          for (int o = 16; o < 32; ++o) sk = (sk << 2) | tk(o);
          uint32_t kv[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) { sk = (sk << 2) | tk(32 + e); kv[e] = sk & (uint32_t)maskK; }
          const long long i0 = (long long)p0 - (long long)K + 1;
          if (whole && i0 >= 0) {
            uint32_t* dst = a.skmer + b + i0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { W4a o4; o4.v[0] = kv[4 * q]; o4.v[1] = kv[4 * q + 1]; o4.v[2] = kv[4 * q + 2]; o4.v[3] = kv[4 * q + 3]; *(W4a*)(dst + 4 * q) = o4; }
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e)
              if (i0 + e >= 0 && p0 + e < L) a.skmer[b + i0 + e] = kv[e];
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  // ---- the first bases' context k-mers reach before the read: left-pad with the most frequent token, first maximum
  // on ties (FastSeq::kmers, src/fastseq.cpp:85-99)
  for (int o = 32; o; o >>= 1) {
    cnt1 += __shfl_xor(cnt1, o); cnt2 += __shfl_xor(cnt2, o); cnt3 += __shfl_xor(cnt3, o); nvalid += __shfl_xor(nvalid, o);
  }
  const uint32_t cnt0 = nvalid - cnt1 - cnt2 - cnt3;
  uint32_t padTok = 0, best = cnt0;
  if (cnt1 > best) { best = cnt1; padTok = 1; }
  if (cnt2 > best) { best = cnt2; padTok = 2; }
  if (cnt3 > best) { best = cnt3; padTok = 3; }
  const uint32_t H = max(mlen, glen ? glen : 1u) - 1u;
  if (lane < H && lane < L) {
    const uint32_t i = lane;
    auto tokAt = [&](int64_t p) -> uint32_t {
      if (p < 0) return padTok;
      const int t = tokenize_char((unsigned char)a.seq[b + p]);
      return t < 0 ? 0u : (uint32_t)t;
    };
    uint32_t mk = 0, gk = 0;
    for (uint32_t c = 0; c < mlen; ++c) mk = mk * 4 + tokAt((int64_t)i - (mlen - 1) + c);
    for (uint32_t c = 0; c < glen; ++c) gk = gk * 4 + tokAt((int64_t)i - (glen - 1) + c);
    uint32_t q = kNQualDev;
    if (a.qual) {
      const int v = (int)(signed char)a.qual[b + i] - '!';
      q = (uint32_t)max(0, min(kNQualDev - 1, v));
    }
    a.ctx[b + i] = ctx_pack(a.em_qmajor_Km ? (q - a.em_qmin) * a.em_qmajor_Km + mk : mk * (kNQualDev + 1) + q, tokAt(i) * (kNQualDev + 1) + q, gk);
  }
}

// Null-model log-likelihood (QuaffNullParams::logLikelihood, src/qmodel.cpp:1875-1890): a strictly
// sequential fp64 sum per read, so one LANE per read (64 reads per wavefront) rather than one wave.
__global__ __launch_bounds__(64) void k_null_ll(PrepArgs a, uint32_t n_reads) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_reads) return;
  double ll = 0;
  if (a.has_null) {
    const uint64_t b = a.off[r];
    const uint32_t L = (uint32_t)(a.off[r + 1] - b);
    ll = (double)L * a.null_logEmit + a.null_log1mEmit;
    // 16 bases per round: token / quality bytes and the table values are independent loads, only the adds are serial
    // (buffers are allocated with 16 bytes of slack, so the last round may read past the read)
    for (uint32_t i0 = 0; i0 < L; i0 += 16) {
      const U8x16 tw = *(const U8x16*)(a.tok + b + i0);
      U8x16 qw{};
      if (a.qual) qw = *(const U8x16*)(a.qual + b + i0);
      double ls[16], lq[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const uint32_t t = (tw.v[c >> 2] >> (8 * (c & 3))) & 3u;
        ls[c] = a.null_logSym[t];
        lq[c] = 0;
        if (a.qual) {
          const int v = (int)(signed char)((qw.v[c >> 2] >> (8 * (c & 3))) & 0xFFu) - '!';
          lq[c] = a.null_logQual[t * kNQualDev + max(0, min(kNQualDev - 1, v))];
        }
      }
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (i0 + c < L) {
          ll += ls[c];
          if (a.qual) ll += lq[c];
        }
    }
  }
  a.nll[r] = ll;
}

// ------------------------------------------------------------------------------------------------
// Seeding: DiagonalEnvelope::initSparse / initFull, src/diagenv.cpp:11-106
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long band_cells(int dlo, int dhi, int xLen, int yLen) {
  unsigned long long c = 0;  // sum_d #{j in [1,yLen] : 1 <= d+j <= xLen}, diagenv.h:75-85
  for (int d = dlo; d <= dhi; ++d) {
    const int jlo = max(1, 1 - d), jhi = min(yLen, xLen - d);
    if (jhi >= jlo) c += (unsigned)(jhi - jlo + 1);
  }
  return c;
}

__device__ __forceinline__ void pair_rx(const SeedArgs& a, uint32_t pair, uint32_t& r, uint32_t& x) {
  if (a.pair_x) { x = a.pair_x[pair]; r = a.pair_y[pair]; }
  else { r = pair / a.n_refs; x = pair % a.n_refs; }
}

// Index lookup: positions of x-sequence `x` whose k-mer equals km are pos[s..e).  Direct-addressed buckets (k <= 8) or
// binary search in the sorted k-mer array (k > 8).
__device__ __forceinline__ void bucket_range(const SeedArgs& a, uint32_t x, uint64_t xb, int xLen, unsigned long long km,
                                             uint32_t& s, uint32_t& e) {
  if (!a.ref_skeys) {
    const uint32_t* st = a.ref_bucket + (uint64_t)x * (a.nbuckets + 1);
    s = st[km]; e = st[km + 1];
    return;
  }
  const unsigned long long* __restrict__ keys = a.ref_skeys + xb;
  const int n = xLen - a.kmer_len + 1;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < km) lo = mid + 1; else hi = mid;
  }
  int up = lo;
  while (up < n && keys[up] == km) ++up;
  s = (uint32_t)lo; e = (uint32_t)up;
}

// Seeding records each band of a pair in a fixed per-pair slot (an uncontended per-pair counter);
// k_bin_units then classifies the bands and allocates unit ids, class-list slots and traceback space with
// workgroup-aggregated atomics.  (Allocating straight from the seeding kernel put ~10^6 returning atomics
// on three words: 10 ms at config 2.)
__device__ __forceinline__ void record_band(const SeedArgs& a, uint32_t pair, int dlo, int dhi) {
  const uint32_t slot = atomicAdd(&a.pair_nbands[pair], 1u);
  if (slot < (uint32_t)kMaxBandsPerPair) {
    a.pair_bands[(uint64_t)pair * kMaxBandsPerPair + slot] = make_int2(dlo, dhi);
  } else {  // rare (very low thresholds / narrow bands): spill to the overflow list
    const uint32_t o = atomicAdd(&a.bc->n_ovf, 1u);
    if (o < a.ovf_cap) a.ovf_bands[o] = make_int4((int)pair, dlo, dhi, 0);
    else atomicOr(&a.bc->error, 8u);
  }
}

// One 256-thread workgroup per (read, ref) pair.  LDS: a dense histogram of k-mer matches per diagonal
// (two 16-bit counters per dword) and a per-diagonal membership array.
// W32: one 32-bit counter per diagonal instead of two 16-bit halves per word (a diagonal can collect 65 536+ matches only
// when both sequences are that long; the reference counts in int, diagenv.cpp:33-40)
template <bool MEM, bool W32 = false>
__device__ void seed_pair(const SeedArgs& a, uint32_t pair, uint32_t* lds, uint32_t* s_red) {
  const uint32_t tid = threadIdx.x;
  if (a.pair_skip && a.pair_skip[pair]) return;
  uint32_t r, x;
  pair_rx(a, pair, r, x);
  const uint64_t xb = a.ref_off[x], yb = a.read_off[r];
  const int xLen = (int)(a.ref_off[x + 1] - xb), yLen = (int)(a.read_off[r + 1] - yb);
  const int minD = 1 - yLen, maxD = xLen - 1, nd = xLen + yLen - 1;
  const int k = a.kmer_len;

  bool full = !a.sparse;
  if (!full && a.threshold >= 0) {  // diagenv.cpp:23-29
    const uint32_t minLen = 2u * (uint32_t)(k + a.threshold);
    if ((uint32_t)xLen < minLen || (uint32_t)yLen < minLen) full = true;
  }
  if (full) {  // initFull, diagenv.cpp:11-18
    if (tid == 0) {
      record_band(a, pair, minD, maxD);
      a.pair_ndiag[pair] = (uint32_t)nd;
    }
    if (a.dump_cover)
      for (int b = tid; b < nd; b += 256) a.dump_cover[b] = 1;
    return;
  }

  const int histWords = W32 ? a.max_nd : (a.max_nd + 1) / 2;
  uint32_t* hist = lds;
  // threshold mode: cover = u8[nd]; memory mode: cover = u16[nd] level stamps, st = u16[nd+2] storage stamps
  uint8_t* cover8 = (uint8_t*)(lds + histWords);
  uint16_t* cover16 = (uint16_t*)(lds + histWords);
  uint16_t* st16 = cover16 + ((a.max_nd + 3) & ~1);

  for (int w = tid; w < (W32 ? nd : (nd + 1) / 2); w += 256) hist[w] = 0;
  if (MEM) {
    for (int b = tid; b < nd; b += 256) cover16[b] = 0;
    for (int b = tid; b < nd + 2; b += 256) st16[b] = 0;
  } else {
    for (int b = tid; b < nd; b += 256) cover8[b] = 0;
  }
  __syncthreads();

  // histogram of matching k-mer pairs per diagonal (diagenv.cpp:33-40), reads' k-mers against the
  // reference's k-mer index
  if (xLen >= k && yLen >= k) {
    const uint32_t* pos = a.ref_pos + xb;
    for (int j = tid; j <= yLen - k; j += 256) {
      const unsigned long long km = a.skmer64 ? a.skmer64[yb + j] : (unsigned long long)a.skmer[yb + j];
      uint32_t s, e;
      bucket_range(a, x, xb, xLen, km, s, e);
      for (uint32_t p = s; p < e; ++p) {
        const int bin = (int)pos[p] - j + yLen - 1;
        if (W32) atomicAdd(&hist[bin], 1u);
        else atomicAdd(&hist[bin >> 1], 1u << (16 * (bin & 1)));
      }
    }
  }
  __syncthreads();
  auto count = [&](int bin) -> uint32_t { return W32 ? hist[bin] : (hist[bin >> 1] >> (16 * (bin & 1))) & 0xFFFFu; };
  const int half = a.band / 2;

  uint32_t accepted = 1;  // highest accepted level stamp (memory mode)
  if (!MEM) {
    const uint32_t thr = a.threshold > 1 ? (uint32_t)a.threshold : 1u;  // map holds only counts >= 1
    for (int b = tid; b < nd; b += 256)
      if (count(b) >= thr) {
        const int seed = b + minD;
        const int lo = max(minD, seed - half), hi = min(maxD, seed + half);
        for (int d = lo; d <= hi; ++d) cover8[d - minD] = 1;
      }
    if (tid == 0) cover8[0 - minD] = 1;  // diagonal 0 always present, diagenv.cpp:52-54
  } else {
    // memory mode, diagenv.cpp:68-96: add whole count-levels in descending order while
    // |storage diagonals| * diagSize < maxSize.
    if (tid == 0) { cover16[0 - minD] = 1; st16[0 - minD + 1] = 1; }
    uint32_t mx = 0;
    for (int b = tid; b < nd; b += 256) mx = max(mx, count(b));
    s_red[tid] = mx;
    __syncthreads();
    for (int o = 128; o; o >>= 1) { if (tid < o) s_red[tid] = max(s_red[tid], s_red[tid + o]); __syncthreads(); }
    const uint32_t maxc = s_red[0];
    __syncthreads();
    const unsigned long long diagSize = (unsigned long long)min(xLen, yLen) * a.cell_size;
    unsigned long long nst = 1;
    uint32_t level = 1;
    uint32_t c = maxc;
    while (c >= 1) {
      // mark every diagonal with exactly c matches under the next level stamp; find the next lower
      // populated count on the way (levels = distinct counts, visited in descending order)
      ++level;
      uint32_t nextc = 0;
      for (int b = tid; b < nd; b += 256) {
        const uint32_t cb = count(b);
        if (cb == c) {
          const int seed = b + minD;
          const int lo = max(minD, seed - half), hi = min(maxD, seed + half);
          for (int d = lo; d <= hi; ++d) if (cover16[d - minD] == 0) cover16[d - minD] = (uint16_t)level;
          for (int d = lo - 1; d <= hi + 1; ++d) if (st16[d - minD + 1] == 0) st16[d - minD + 1] = (uint16_t)level;
        } else if (cb < c)
          nextc = max(nextc, cb);
      }
      __syncthreads();
      uint32_t added = 0;
      for (int b = tid; b < nd + 2; b += 256) added += st16[b] == level;
      s_red[tid] = added;
      __syncthreads();
      for (int o = 128; o; o >>= 1) { if (tid < o) s_red[tid] += s_red[tid + o]; __syncthreads(); }
      added = s_red[0];
      __syncthreads();
      s_red[tid] = nextc;
      __syncthreads();
      for (int o = 128; o; o >>= 1) { if (tid < o) s_red[tid] = max(s_red[tid], s_red[tid + o]); __syncthreads(); }
      nextc = s_red[0];
      __syncthreads();
      if ((nst + added) * diagSize >= a.max_size) break;  // diagenv.cpp:88-89 (level rejected, stop)
      nst += added;
      accepted = level;
      c = nextc;
    }
  }
  __syncthreads();
  auto member = [&](int b) -> bool {
    if (MEM) { const uint32_t s = cover16[b]; return s != 0 && s <= accepted; }
    return cover8[b] != 0;
  };

  // contiguous runs -> units; diagonal and cell counts
  uint32_t nmem = 0;
  for (int b = tid; b < nd; b += 256) {
    const bool in = member(b);
    nmem += in;
    if (a.dump_cover) a.dump_cover[b] = in;
    if (in && (b == 0 || !member(b - 1))) {
      int e = b;
      while (e + 1 < nd && member(e + 1)) ++e;
      record_band(a, pair, b + minD, e + minD);
    }
  }
  s_red[tid] = nmem;
  __syncthreads();
  for (int o = 128; o; o >>= 1) { if (tid < o) s_red[tid] += s_red[tid + o]; __syncthreads(); }
  if (tid == 0) a.pair_ndiag[pair] = s_red[0];
}
template <bool MEM>
__global__ __launch_bounds__(256) void k_seed(SeedArgs a) {
  extern __shared__ uint32_t lds[];
  __shared__ uint32_t s_red[256];
  seed_pair<MEM>(a, a.pair_base + blockIdx.x, lds, s_red);
}
// The same per-pair procedure with the histogram and membership arrays in a global-memory workspace (one per resident
// workgroup, reused pair after pair): references too long for the LDS histogram (genome scale).
template <bool MEM, bool W32>
__global__ __launch_bounds__(256) void k_seed_global(SeedArgs a, uint32_t n_pairs) {
  __shared__ uint32_t s_red[256];
  uint32_t* ws = a.ws + (uint64_t)blockIdx.x * a.ws_words;
  for (uint32_t p = blockIdx.x; p < n_pairs; p += gridDim.x) {
    seed_pair<MEM, W32>(a, a.pair_base + p, ws, s_red);
    __threadfence();
    __syncthreads();
  }
}
template __global__ void k_seed<false>(SeedArgs);
template __global__ void k_seed<true>(SeedArgs);

// Threshold-mode seeding, one WAVEFRONT per (read, ref) pair (the common case: `quaff align/overlap/train`
// defaults).  Same semantics as k_seed<false>; restructured for memory-level parallelism: no workgroup
// barriers, the read's k-mers and the reference bucket ranges are fetched as batches of independent loads,
// bucket entries four at a time, the threshold scan reads eight 16-bit counters per lane per LDS read, and
// envelope membership is a bitmap (atomicOr of band masks, run detection by bit tricks).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
struct __attribute__((packed, aligned(4))) U32x4u { uint32_t v[4]; };

// LDSIDX: the reference's k-mer index (bucket starts and positions, 16 bits each) sits in LDS (sb, sp) instead of being
// gathered from global memory.
// WIDE: 32-bit coarse counters (a 32-diagonal bin of a read of 2 048+ bases can collect 65 536 matches and wrap 16 bits)
// CB: log2 of the coarse bin width (5, 4 or 3).  Unrelated sequences still share length/4^k k-mers per diagonal by
// chance; when 32 diagonals' worth of those comes close to the threshold (overlap defaults: 2 kb reads, k = 6, n = 14)
// nearly every pair would have "candidate" bins and pay the second walk, so the host picks narrower bins then.
// RR: read positions per lane per round of the walk (4; 2 in the explicit-pair-list kernel: measured, see there)
template <bool LDSIDX, bool WIDE, int CB, int RR = 4>
__device__ __forceinline__ void seed_wave_pair(const SeedArgs& a, uint32_t pair, uint32_t r, uint32_t x, uint32_t* wlds,
                                               const uint16_t* sb, const uint16_t* sp) {
  // Two-level histogram.  Pass 1 counts k-mer matches per COARSE bin of 2^CB diagonals; a diagonal can reach
  // the threshold only inside a coarse bin that does, so pass 2 re-walks the matches and keeps exact
  // per-diagonal counters for those candidate bins alone (none at all for unrelated / wrong-strand pairs).
  // ~4 KB of LDS per wavefront instead of a dense 2 B/diagonal histogram.
  constexpr int kCand = 32;  // candidate coarse bins refined per round
  constexpr int CW = 1 << CB, FW = CW / 2;  // diagonals per coarse bin; dwords of 16-bit fine counters per candidate bin
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t xb = a.ref_off[x], yb = a.read_off[r];
  const int xLen = (int)(a.ref_off[x + 1] - xb), yLen = (int)(a.read_off[r + 1] - yb);
  const int minD = 1 - yLen, maxD = xLen - 1, nd = xLen + yLen - 1;
  const int k = a.kmer_len;
  const uint32_t minLen = 2u * (uint32_t)(k + a.threshold);
  if ((uint32_t)xLen < minLen || (uint32_t)yLen < minLen) {  // diagenv.cpp:23-29 -> initFull
    if (lane == 0) {
      record_band(a, pair, minD, maxD);
      a.pair_ndiag[pair] = (uint32_t)nd;
    }
    if (a.dump_cover)
      for (int b = lane; b < nd; b += 64) a.dump_cover[b] = 1;
    return;
  }
  const int nCoarse = (nd + CW - 1) >> CB, coarseWords = WIDE ? nCoarse : (nCoarse + 1) / 2, bmWords = (nd + 31) >> 5;
  const int maxCoarse = (a.max_nd + CW - 1) >> CB;
  uint32_t* coarse = wlds;                                   // two 16-bit counters per dword (one if WIDE); later: slot map
  uint32_t* bm = coarse + (WIDE ? maxCoarse + 2 : (maxCoarse + 1) / 2 + 1);  // membership bitmap, one bit per diagonal
  uint32_t* fine = bm + ((a.max_nd + 31) / 32 + 1);          // [kCand][FW] dwords = CW x 16-bit counters each (16 dwords reserved)
  uint32_t* misc = fine + kCand * 16;                        // [0] candidate count, [1..kCand] candidate bins

  for (int w = lane; w < coarseWords; w += 64) coarse[w] = 0;
  for (int w = lane; w < bmWords; w += 64) bm[w] = 0;
  if (lane == 0) misc[0] = 0;
  wave_lds_sync();

  const uint32_t* __restrict__ pos = a.ref_pos + xb;
  const uint32_t* __restrict__ sk = a.skmer + yb;
  const unsigned long long* __restrict__ sk64 = a.skmer64 ? a.skmer64 + yb : nullptr;
  const int nk = yLen - k + 1;  // read k-mers (>= 1 here)
  const bool few = a.few_hits != 0;   // (uniform)
  // visit(bin) for every (i, j) with equal k-mers, bin = i - j + yLen - 1 (diagenv.cpp:33-40).
  // R read positions per lane per round; the three dependent loads are issued as batches.
  auto walk = [&](auto&& visit) {
    constexpr int R = RR;
    for (int j0 = 0; j0 < nk; j0 += 64 * R) {
      unsigned long long km[R];
      uint32_t s[R], e[R];
#pragma clang loop unroll(full)
      for (int c = 0; c < R; ++c) {
        const int j = j0 + c * 64 + (int)lane;
        km[c] = j < nk ? (sk64 ? sk64[j] : (unsigned long long)sk[j]) : 0ull;
      }
#pragma clang loop unroll(full)
      for (int c = 0; c < R; ++c) {
        const int j = j0 + c * 64 + (int)lane;
        s[c] = e[c] = 0;
        if (j < nk) {
          if (LDSIDX) { s[c] = sb[km[c]]; e[c] = sb[km[c] + 1]; }
          else bucket_range(a, x, xb, xLen, km[c], s[c], e[c]);
        }
      }
      // the first entries of each bucket are fetched unconditionally, as a batch (the index has 4 words of slack): four of
      // them, or two when a k-mer of the read is expected to occur less than once in x (`few`: 2 kb reads against each
      // other; the third and later entries then take the rare loop)
      uint32_t pa[R], pb[R], pc[R], pd[R];
#pragma clang loop unroll(full)
      for (int c = 0; c < R; ++c) {
        pc[c] = pd[c] = 0;
        if (LDSIDX) {
          const uint16_t* q4 = sp + s[c];
          pa[c] = q4[0]; pb[c] = q4[1];
          if (!few) { pc[c] = q4[2]; pd[c] = q4[3]; }
        } else {
          const uint32_t* q4 = pos + s[c];
          pa[c] = q4[0]; pb[c] = q4[1];
          if (!few) { pc[c] = q4[2]; pd[c] = q4[3]; }
        }
      }
#pragma clang loop unroll(full)
      for (int c = 0; c < R; ++c) {
        const int j = j0 + c * 64 + (int)lane;
        const uint32_t n = e[c] - s[c];
        if (n > 0) visit((int)pa[c] - j + yLen - 1);
        if (n > 1) visit((int)pb[c] - j + yLen - 1);
        uint32_t q = 2;
        if (!few) {
          if (n > 2) visit((int)pc[c] - j + yLen - 1);
          if (n > 3) visit((int)pd[c] - j + yLen - 1);
          q = 4;
        }
        for (; q < n; ++q) visit((int)(LDSIDX ? (uint32_t)sp[s[c] + q] : pos[s[c] + q]) - j + yLen - 1);  // long buckets
      }
    }
  };
  walk([&](int bin) {
    if (WIDE) atomicAdd(&coarse[bin >> CB], 1u);
    else atomicAdd(&coarse[bin >> (CB + 1)], 1u << (16 * ((bin >> CB) & 1)));
  });
  wave_lds_sync();

  // candidate coarse bins; afterwards the coarse array holds each bin's candidate slot (0xFFFF = none)
  const uint32_t thr = a.threshold > 1 ? (uint32_t)a.threshold : 1u;
  uint32_t ncand = 0;
  for (int w0 = 0; w0 < coarseWords; w0 += 64) {
    const int w = w0 + (int)lane;
    uint32_t hv = w < coarseWords ? coarse[w] : 0u, outv = 0xFFFFFFFFu;
    if (WIDE) {
      if (hv >= thr) {
        const uint32_t slot = atomicAdd(&misc[0], 1u);
        if (slot < 0xFFFFu) outv = slot;
        if (slot < (uint32_t)kCand) misc[1 + slot] = (uint32_t)w;
      } else outv = 0xFFFFu;
    } else {
#pragma unroll
      for (int c = 0; c < 2; ++c)
        if (((hv >> (16 * c)) & 0xFFFFu) >= thr) {
          const uint32_t slot = atomicAdd(&misc[0], 1u);
          if (slot < 0xFFFFu) outv = (outv & ~(0xFFFFu << (16 * c))) | (slot << (16 * c));
          if (slot < (uint32_t)kCand) misc[1 + slot] = (uint32_t)(2 * w + c);   // first round's bins, for the seed scan
        }
    }
    if (w < coarseWords) coarse[w] = outv;
  }
  wave_lds_sync();
  ncand = min(misc[0], 0xFFFEu);

  const int half = a.band / 2;
  auto mark = [&](int b) {
    const int lo = max(0, b - half), hi = min(nd - 1, b + half);
    for (int w = lo >> 5; w <= (hi >> 5); ++w) {
      const int blo = max(lo, w * 32) & 31, bhi = min(hi, w * 32 + 31) & 31;
      atomicOr(&bm[w], (0xFFFFFFFFu >> (31 - bhi)) & (0xFFFFFFFFu << blo));
    }
  };
  auto slotOf = [&](int cb) -> uint32_t { return WIDE ? (coarse[cb] & 0xFFFFu) : (coarse[cb >> 1] >> (16 * (cb & 1))) & 0xFFFFu; };
  for (uint32_t base = 0; base < ncand; base += kCand) {  // almost always one round
    for (int w = lane; w < kCand * FW; w += 64) fine[w] = 0;
    wave_lds_sync();
    walk([&](int bin) {
      const uint32_t slot = slotOf(bin >> CB) - base;  // 0xFFFF - base is never < kCand
      if (slot < (uint32_t)kCand) atomicAdd(&fine[slot * FW + ((bin & (CW - 1)) >> 1)], 1u << (16 * (bin & 1)));
    });
    wave_lds_sync();
    // seeds: diagonals of this round's candidate bins reaching the threshold (diagenv.cpp:68-96)
    if (base == 0) {
      // one lane per (candidate bin, diagonal): the first round's bins are listed in misc[1..]
      const uint32_t nslots = min(ncand, (uint32_t)kCand);
      for (uint32_t idx = lane; idx < nslots * CW; idx += 64) {
        const uint32_t slot = idx >> CB, c = idx & (CW - 1);
        const int cb = (int)misc[1 + slot];
        const uint32_t cnt = (fine[slot * FW + (c >> 1)] >> (16 * (c & 1))) & 0xFFFFu;
        if (cnt >= thr && cb * CW + (int)c < nd) mark(cb * CW + (int)c);
      }
    } else {
      for (int cb = lane; cb < nCoarse; cb += 64) {
        const uint32_t slot = slotOf(cb) - base;
        if (slot >= (uint32_t)kCand) continue;
        for (int c = 0; c < CW; ++c) {
          const uint32_t cnt = (fine[slot * FW + (c >> 1)] >> (16 * (c & 1))) & 0xFFFFu;
          if (cnt >= thr && cb * CW + c < nd) mark(cb * CW + c);
        }
      }
    }
    wave_lds_sync();
  }
  if (lane == 0) atomicOr(&bm[(yLen - 1) >> 5], 1u << ((yLen - 1) & 31));  // diagonal 0, diagenv.cpp:52-54
  wave_lds_sync();

  // runs of consecutive member diagonals -> bands
  auto member = [&](int b) -> bool { return (bm[b >> 5] >> (b & 31)) & 1u; };
  uint32_t nmem = 0;
  for (int w = lane; w < bmWords; w += 64) {
    const uint32_t bits = bm[w];
    nmem += __popc(bits);
    if (a.dump_cover)
      for (int c = 0; c < 32 && w * 32 + c < nd; ++c) a.dump_cover[w * 32 + c] = (bits >> c) & 1u;
    const uint32_t prev = w ? bm[w - 1] >> 31 : 0u;
    uint32_t st = bits & ~((bits << 1) | prev);
    while (st) {
      const int b = w * 32 + __ffs(st) - 1;
      st &= st - 1;
      // end of the run: the first clear bit after b, a word at a time
      int e2;
      {
        int ww = b >> 5;
        uint32_t clr = ~bm[ww] & (0xFFFFFFFEu << (b & 31));   // clear bits above b in its word (0 if b is bit 31)
        if ((b & 31) == 31) clr = 0;
        while (clr == 0 && ww + 1 < bmWords) { ++ww; clr = ~bm[ww]; }
        e2 = clr ? ww * 32 + __ffs(clr) - 2 : nd - 1;
        if (e2 > nd - 1) e2 = nd - 1;
      }
      (void)member;
      record_band(a, pair, b + minD, e2 + minD);
    }
  }
  for (int o = 32; o; o >>= 1) nmem += __shfl_xor(nmem, o);
  if (lane == 0) a.pair_ndiag[pair] = nmem;
}


// One wavefront per (read, ref) pair, four pairs per workgroup; the index is gathered from global memory (any reference
// set, explicit pair lists).
template <bool WIDE, int CB>
__global__ __launch_bounds__(256) void k_seed_wave(SeedArgs a, uint32_t n_pairs, uint32_t wave_lds_words) {
  extern __shared__ uint32_t lds[];
  const uint32_t wv = threadIdx.x >> 6;
  const uint32_t pidx = blockIdx.x * (blockDim.x >> 6) + wv;
  if (pidx >= n_pairs) return;
  const uint32_t pair = a.pair_base + pidx;
  if (a.pair_skip && a.pair_skip[pair]) return;
  uint32_t r, x;
  pair_rx(a, pair, r, x);
  seed_wave_pair<false, WIDE, CB>(a, pair, r, x, lds + (size_t)wv * wave_lds_words, nullptr, nullptr);
}

// Short references (k-mer index of one reference <= 48 KB as 16-bit entries): a workgroup of eight wavefronts copies one
// reference's index to LDS and seeds kSeedReadsPerBlock reads against it.  The global-memory version spends two thirds
// of its cycles waiting on L1 misses of those gathers (measured); here the only global traffic is the reads' k-mers.
#ifndef QF_SEED_RPB
#define QF_SEED_RPB 32
#endif
#ifndef QF_SEED_PPB
#define QF_SEED_PPB 256
#endif
constexpr uint32_t kSeedReadsPerBlock = QF_SEED_RPB;   // k_seed_wave_lds: reads per workgroup (one reference's index in LDS)
constexpr uint32_t kSeedPairsPerBlock = QF_SEED_PPB;   // k_seed_wave_lds_pairs: consecutive pairs per workgroup
template <bool WIDE, int CB>
__global__ __launch_bounds__(512) void k_seed_wave_lds(SeedArgs a, uint32_t n_reads, uint32_t wave_lds_words, uint32_t idx_words) {
  extern __shared__ uint32_t lds[];
  const uint32_t wv = threadIdx.x >> 6;
  const uint32_t x = blockIdx.x % a.n_refs, r0 = (blockIdx.x / a.n_refs) * kSeedReadsPerBlock;
  const uint64_t xb = a.ref_off[x];
  const uint32_t xLen = (uint32_t)(a.ref_off[x + 1] - xb), nb1 = a.nbuckets + 1;
  uint16_t* sb = (uint16_t*)lds;
  uint16_t* sp = sb + ((nb1 + 1) & ~1u);
  const uint32_t* gb = a.ref_bucket + (uint64_t)x * nb1;
  const uint32_t* gp = a.ref_pos + xb;
  for (uint32_t q = threadIdx.x; q < nb1; q += 512) sb[q] = (uint16_t)gb[q];
  for (uint32_t q = threadIdx.x; q < xLen + 4; q += 512) sp[q] = q < xLen ? (uint16_t)gp[q] : (uint16_t)0;
  __syncthreads();
  uint32_t* wlds = lds + idx_words + (size_t)wv * wave_lds_words;
  for (uint32_t r = r0 + wv; r < min(r0 + kSeedReadsPerBlock, n_reads); r += 8) {
    const uint32_t pair = a.pair_base + r * a.n_refs + x;
    if (a.pair_skip && a.pair_skip[pair]) continue;
    seed_wave_pair<true, WIDE, CB>(a, pair, r, x, wlds, sb, sp);
    wave_lds_sync();
  }
}

// The same for explicit pair lists (read-vs-read overlap: x is a read too).  The scheduler's list is x-major
// (src/qoverlap.cpp:528-547), so the 64 consecutive pairs of a workgroup nearly always share their x: its index goes to
// LDS; a pair with another x takes the global-memory path.
template <bool WIDE, int CB>
__global__ __launch_bounds__(512) void k_seed_wave_lds_pairs(SeedArgs a, uint32_t n_pairs, uint32_t wave_lds_words, uint32_t idx_words) {
  extern __shared__ uint32_t lds[];
  const uint32_t wv = threadIdx.x >> 6;
  const uint32_t p0 = blockIdx.x * kSeedPairsPerBlock;
  uint32_t r0, x0;
  pair_rx(a, a.pair_base + p0, r0, x0);
  const uint64_t xb = a.ref_off[x0];
  const uint32_t xLen = (uint32_t)(a.ref_off[x0 + 1] - xb), nb1 = a.nbuckets + 1;
  uint16_t* sb = (uint16_t*)lds;
  uint16_t* sp = sb + ((nb1 + 1) & ~1u);
  const uint32_t* gb = a.ref_bucket + (uint64_t)x0 * nb1;
  const uint32_t* gp = a.ref_pos + xb;
  for (uint32_t q = threadIdx.x; q < nb1; q += 512) sb[q] = (uint16_t)gb[q];
  for (uint32_t q = threadIdx.x; q < xLen + 4; q += 512) sp[q] = q < xLen ? (uint16_t)gp[q] : (uint16_t)0;
  __syncthreads();
  uint32_t* wlds = lds + idx_words + (size_t)wv * wave_lds_words;
  const uint32_t pend = min(p0 + kSeedPairsPerBlock, n_pairs);
  for (uint32_t p = p0 + wv; p < pend; p += 8) {
    const uint32_t pair = a.pair_base + p;
    if (a.pair_skip && a.pair_skip[pair]) continue;
    uint32_t r, x;
    pair_rx(a, pair, r, x);
    // two read positions per lane per round: reads against reads rarely share a k-mer more than once (19.0 ms per 3.4 M
    // pairs of 2 kb reads against 21.0 with four; one and three are worse; the align kernels are best with four)
    if (x == x0) seed_wave_pair<true, WIDE, CB, 2>(a, pair, r, x, wlds, sb, sp);
    else seed_wave_pair<false, WIDE, CB, 2>(a, pair, r, x, wlds, nullptr, nullptr);
    wave_lds_sync();
  }
}

// Row prefilter for explicit pair lists (read-vs-read overlap).  The scheduler's list is x-major (src/qoverlap.cpp:528-547):
// long runs (x, y0), (x, y0 + 1), ...  Nearly all of those pairs are unrelated reads whose only envelope diagonal is the
// forced one (diagenv.cpp:52-54), and finding that out one pair per wavefront is what the seeding spends its time on: a
// read position meets 0.5 k-mer matches in the other read, so most lanes of every counting instruction are idle.  Here one
// workgroup takes x against a whole chunk of 2^cl consecutive y at once: the chunk has its own k-mer index (built once per
// upload: launch_chunk_index), an x position meets ~30 matches in it, and the lanes walk those lists into per-y coarse
// counters in LDS (the coarse pass of seed_wave_pair for 2^cl pairs).  A pair none of whose coarse bins reaches the threshold
// has no diagonal that does: it gets its one band (diagonal 0) here and is marked in row_skip; the others go through the
// per-pair kernels as before, which skip the marked ones.
// Sixteen wavefronts per workgroup: a position's three dependent loads (its k-mer, the bucket, the bucket's entries four at a
// time) are all the latency there is to hide, and the counters take 64 KB of LDS whatever the workgroup's size.
constexpr int kSeedRowThreads = 1024;
// The row prefilter's item list for the scheduler's triangle -- whole rows X0 .. X0 + R - 1, each x against x + 1 ... n_seqs - 1
// (src/qoverlap.cpp:528-547) -- formed on the device: chunk-major (every row of one chunk of y before the next chunk), then dealt
// out in segments of kRowSeg items to XCD 0, 1, ... 7 like the host's list for arbitrary runs (overlap_chunk).  A row block of
// 2^24 pairs against 100 k sequences is 1.1 M items: built and copied by the host that was 2 - 3 ms of idle GPU per block.
// Chunk ch0 + c holds rows X0 .. min(X0 + R, ((ch0 + c + 1) << cl) - 1) - 1: A + c x 2^cl of them while that is below R (the
// first nPart chunks, firstFull items in all), R from there on.
__global__ void k_row_items_tri(uint32_t X0, uint32_t R, uint32_t n_seqs, int cl, uint32_t ch0, uint32_t A, uint32_t nPart, uint32_t firstFull,
                                uint32_t n_items, uint32_t n_padded, RowItem* __restrict__ items) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n_padded) return;
  const uint32_t xcd = d % kRowXcd, t = d / kRowXcd, pos = t % kRowSeg, seg = (t / kRowSeg) * kRowXcd + xcd, k = seg * kRowSeg + pos;
  RowItem it{~0u, 0u, 0u, 0u, 0u};
  if (k < n_items) {
    uint32_t c, r;
    if (k >= firstFull) { c = nPart + (k - firstFull) / R; r = (k - firstFull) % R; }
    else {
      uint32_t first = 0, cnt = A;
      for (c = 0; first + cnt <= k; ++c) { first += cnt; cnt += 1u << cl; }
      r = k - first;
    }
    const uint32_t x = X0 + r, chunk = ch0 + c;
    const uint32_t ylo = max(x + 1, chunk << cl), yhi = (uint32_t)min((uint64_t)n_seqs, ((uint64_t)chunk + 1) << cl);
    const uint64_t p0 = (uint64_t)r * (n_seqs - 1 - X0) - (uint64_t)r * (r ? r - 1 : 0) / 2;   // pairs of the rows before x
    it = RowItem{x, chunk, ylo, yhi, (uint32_t)(p0 + (ylo - (x + 1)))};
  }
  items[d] = it;
}
uint32_t launch_row_items_tri(uint32_t X0, uint32_t R, uint32_t n_seqs, int cl, RowItem* items, size_t capacity, hipStream_t s) {
  const uint32_t ch0 = (X0 + 1) >> cl, n_chunks = (uint32_t)(((uint64_t)n_seqs + (1u << cl) - 1) >> cl);
  const uint32_t A = ((ch0 + 1) << cl) - 1 - X0;
  uint64_t firstFull = 0;
  uint32_t nPart = 0;
  for (uint64_t cnt = A; cnt < R && ch0 + nPart < n_chunks; cnt += 1u << cl) { firstFull += cnt; ++nPart; }
  const uint64_t n_items = firstFull + (uint64_t)(n_chunks - (ch0 + nPart)) * R;
  const uint64_t n_seg = (n_items + kRowSeg - 1) / kRowSeg, n_padded = (n_seg + kRowXcd - 1) / kRowXcd * kRowXcd * kRowSeg;
  if (!items) return (uint32_t)std::min<uint64_t>(n_padded, 0xFFFFFFFFull);   // (size query)
  if (!n_items || n_padded > capacity) return 0;
  hipLaunchKernelGGL(k_row_items_tri, dim3((uint32_t)((n_padded + 255) / 256)), dim3(256), 0, s, X0, R, n_seqs, cl, ch0, A, nPart, (uint32_t)firstFull,
                     (uint32_t)n_items, (uint32_t)n_padded, items);
  return (uint32_t)n_padded;
}

struct __attribute__((packed, aligned(4))) W2a { uint32_t v[2]; };
// E16: the index entries are 16 bits, (sequence in chunk) << pb | (len - 1 - j) with 2^pb > the longest sequence, and a sequence's
// counters span 2^(pb + 1) diagonals, so that an entry e names the counter position L = e + (e & ~(2^pb - 1)) = sequence << (pb + 1) |
// position without a multiply.  What bounds this kernel is not arithmetic but the texture addresser: every look-up is a 64-lane
// gather (`TA_BUSY` 74 %, 4.7 G cache accesses per launch in round 3's form: per x position two 4-byte bucket bounds and four
// 16-byte entry loads, each lane on its own line).  16-bit entries halve the entry loads (a 16-byte load holds 8), the two bucket
// bounds come as one 8-byte load: 6.25 -> 3.25 accesses per position.
template <int CB, bool E16>
__global__ __launch_bounds__(kSeedRowThreads) void k_seed_rows(SeedArgs a, uint32_t stride) {
  extern __shared__ uint32_t cnt[];   // [2^cl][stride]: two 16-bit coarse counters per word
  const RowItem it = a.row_items[blockIdx.x];
  if (it.x == ~0u) return;   // padding of the dealt-out list
  const int cl = a.chunk_log2, k = a.kmer_len;
  const uint32_t tid = threadIdx.x, csize = 1u << cl;
  for (uint32_t w = tid; w < csize * stride; w += kSeedRowThreads) cnt[w] = 0;
  __syncthreads();
  const uint64_t xb = a.ref_off[it.x];
  const int xLen = (int)(a.ref_off[it.x + 1] - xb), nkx = xLen - k + 1;
  const uint32_t* __restrict__ cs = a.chunk_start + (uint64_t)it.chunk * (a.nbuckets + 1);
  const uint32_t* __restrict__ ce = a.chunk_entries + (E16 ? 0 : a.read_off[(uint64_t)it.chunk << cl]);
  const uint16_t* __restrict__ ce16 = (const uint16_t*)a.chunk_entries + (E16 ? chunk_base16(a.read_off, it.chunk, cl, a.nbuckets) : 0);
  const uint2* __restrict__ cb2 = E16 ? a.chunk_bounds + (uint64_t)it.chunk * a.nbuckets : nullptr;
  const uint32_t* __restrict__ xk = a.skmer + xb;
  const uint32_t y0 = it.chunk << cl, yylo = it.ylo - y0, yyhi = it.yhi - y0;
  const bool whole = yylo == 0 && yyhi == csize;            // the item takes every sequence of the chunk (all but a row's two end chunks)
  // A 32-bit index entry is L = (sequence in chunk) x Wd + (len - 1 - j), Wd = the diagonals a sequence's counters span (stride x 2
  // bins of 2^CB: launch_chunk_index): the 16-bit counter of a match of x position i is number (L + i) >> CB of the whole array,
  // i.e. word ((L + i) >> (CB + 1)), half (L + i) >> CB & 1 -- bin = i - j + yLen - 1 (diagenv.cpp:33-40) -- with no per-sequence
  // multiply.  (Round 3 kept (sequence << 26 | position) and spent ~15 instructions per match unpacking it.)
  const uint32_t Wd = (stride * 2u) << CB, Llo = yylo * Wd, Lhi = yyhi * Wd;
  const uint32_t himask = E16 ? (0xFFFFu & ~((1u << a.chunk_pb) - 1u)) : 0u;
  auto count = [&](uint32_t i, uint32_t L) {
    if (E16) L += L & himask;
    if (!whole && (L < Llo || L >= Lhi)) return;
    const uint32_t t = L + i;
    atomicAdd(&cnt[t >> (CB + 1)], 1u << (16 * ((t >> CB) & 1u)));
  };
  const uint32_t cnt_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)cnt;   // LDS address of the counters
  // two positions per thread and round, their buckets' first entries fetched as one batch of independent 16-byte loads
  constexpr int kRowPos = 2, kRowBatch = E16 ? 3 : 5, kPerLoad = E16 ? 8 : 4;
  for (int i0 = (int)tid; i0 < nkx; i0 += kRowPos * kSeedRowThreads) {
    uint32_t km[kRowPos], sq[kRowPos], eq[kRowPos];
#pragma unroll
    for (int c = 0; c < kRowPos; ++c) km[c] = i0 + c * kSeedRowThreads < nkx ? xk[i0 + c * kSeedRowThreads] : 0u;
#pragma unroll
    for (int c = 0; c < kRowPos; ++c) {
      if (E16) {                                           // (first entry: even, entries) as one 8-byte load
        const uint2 se = cb2[km[c]];
        sq[c] = se.x;
        eq[c] = se.x + (i0 + c * kSeedRowThreads < nkx ? se.y : 0u);
      } else {
        const W2a se = *(const W2a*)(cs + km[c]);          // the bucket's bounds as one 8-byte load
        sq[c] = se.v[0];
        eq[c] = i0 + c * kSeedRowThreads < nkx ? se.v[1] : sq[c];
      }
    }
    W4a v[kRowPos][kRowBatch];
#pragma unroll
    for (int c = 0; c < kRowPos; ++c)
#pragma unroll
      for (int b = 0; b < kRowBatch; ++b) {
        // (the entry array has 16 words of slack; a lane whose bucket ends earlier holds zeros it never counts; a 16-bit bucket
        // starts on an even element of an even base, so its loads are 4-byte aligned)
        const uint32_t first = sq[c];
        if (first + kPerLoad * b < eq[c]) v[c][b] = E16 ? *(const W4a*)(ce16 + first + kPerLoad * b) : *(const W4a*)(ce + first + kPerLoad * b);
        else v[c][b] = W4a{{0u, 0u, 0u, 0u}};
      }
#pragma unroll
    for (int c = 0; c < kRowPos; ++c) {
      const uint32_t i = (uint32_t)(i0 + c * kSeedRowThreads);
      const uint32_t ip = i + (cnt_base << (CB - 1)), nrem = eq[c] - sq[c];
      if (whole && !E16) {
        // Four entries at a time, written out: per entry the sum L + i (the counters' LDS address rides in i, shifted up by the
        // CB - 1 bits the word address drops), the word address, the increment 1 << 16 x (bin parity), and the count under an execution mask
        // set by the compare itself (lanes whose bucket still has this entry) and put back by a scalar move: seven vector
        // instructions and no branch per match.
#pragma unroll
        for (int b = 0; b < kRowBatch; ++b) {
          uint32_t t0, t1, t2, t3, a0, a1, a2, a3;
          unsigned long long sv;
          asm volatile(
              "v_add_u32 %[t0], %[e0], %[ip]\n v_add_u32 %[t1], %[e1], %[ip]\n v_add_u32 %[t2], %[e2], %[ip]\n v_add_u32 %[t3], %[e3], %[ip]\n"
              "v_lshrrev_b32 %[a0], %[shw], %[t0]\n v_lshrrev_b32 %[a1], %[shw], %[t1]\n v_lshrrev_b32 %[a2], %[shw], %[t2]\n v_lshrrev_b32 %[a3], %[shw], %[t3]\n"
              "v_and_b32 %[a0], -4, %[a0]\n v_and_b32 %[a1], -4, %[a1]\n v_and_b32 %[a2], -4, %[a2]\n v_and_b32 %[a3], -4, %[a3]\n"
              "v_bfe_u32 %[t0], %[t0], %[cb], 1\n v_bfe_u32 %[t1], %[t1], %[cb], 1\n v_bfe_u32 %[t2], %[t2], %[cb], 1\n v_bfe_u32 %[t3], %[t3], %[cb], 1\n"
              "v_lshlrev_b32 %[t0], 4, %[t0]\n v_lshlrev_b32 %[t1], 4, %[t1]\n v_lshlrev_b32 %[t2], 4, %[t2]\n v_lshlrev_b32 %[t3], 4, %[t3]\n"
              "v_lshlrev_b32_e64 %[t0], %[t0], 1\n v_lshlrev_b32_e64 %[t1], %[t1], 1\n v_lshlrev_b32_e64 %[t2], %[t2], 1\n v_lshlrev_b32_e64 %[t3], %[t3], 1\n"
              "s_mov_b64 %[sv], exec\n"
              "v_cmpx_lt_u32 vcc, %[k0], %[n]\n ds_add_u32 %[a0], %[t0]\n s_mov_b64 exec, %[sv]\n"
              "v_cmpx_lt_u32 vcc, %[k1], %[n]\n ds_add_u32 %[a1], %[t1]\n s_mov_b64 exec, %[sv]\n"
              "v_cmpx_lt_u32 vcc, %[k2], %[n]\n ds_add_u32 %[a2], %[t2]\n s_mov_b64 exec, %[sv]\n"
              "v_cmpx_lt_u32 vcc, %[k3], %[n]\n ds_add_u32 %[a3], %[t3]\n s_mov_b64 exec, %[sv]\n"
              : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3),
                [sv] "=&s"(sv)
              : [e0] "v"(v[c][b].v[0]), [e1] "v"(v[c][b].v[1]), [e2] "v"(v[c][b].v[2]), [e3] "v"(v[c][b].v[3]), [ip] "v"(ip), [n] "v"(nrem),
                [shw] "n"(CB - 1), [cb] "n"(CB), [k0] "n"(4 * b), [k1] "n"(4 * b + 1), [k2] "n"(4 * b + 2), [k3] "n"(4 * b + 3)
              : "vcc", "memory");
        }
      } else if (whole && E16) {
        // The same for 16-bit entries, a dword = two entries at a time: entry e -> L = e + (e & himask); the increment is
        // 1 + 0xFFFF x (bin parity) by one multiply-add.  Nine vector instructions per entry.
#pragma unroll
        for (int b = 0; b < kRowBatch; ++b)
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            uint32_t e0, e1, t0, t1, a0, a1;
            unsigned long long sv;
            asm volatile(
                "v_and_b32 %[e0], 0xffff, %[wd]\n v_lshrrev_b32 %[e1], 16, %[wd]\n"
                "v_and_b32 %[t0], %[hm], %[e0]\n v_and_b32 %[t1], %[hm], %[e1]\n"
                "v_add3_u32 %[t0], %[e0], %[t0], %[ip]\n v_add3_u32 %[t1], %[e1], %[t1], %[ip]\n"
                "v_lshrrev_b32 %[a0], %[shw], %[t0]\n v_lshrrev_b32 %[a1], %[shw], %[t1]\n"
                "v_and_b32 %[a0], -4, %[a0]\n v_and_b32 %[a1], -4, %[a1]\n"
                "v_bfe_u32 %[t0], %[t0], %[cb], 1\n v_bfe_u32 %[t1], %[t1], %[cb], 1\n"
                "v_mad_u32_u24 %[t0], %[t0], %[ffff], 1\n v_mad_u32_u24 %[t1], %[t1], %[ffff], 1\n"
                "s_mov_b64 %[sv], exec\n"
                "v_cmpx_lt_u32 vcc, %[k0], %[n]\n ds_add_u32 %[a0], %[t0]\n s_mov_b64 exec, %[sv]\n"
                "v_cmpx_lt_u32 vcc, %[k1], %[n]\n ds_add_u32 %[a1], %[t1]\n s_mov_b64 exec, %[sv]\n"
                : [e0] "=&v"(e0), [e1] "=&v"(e1), [t0] "=&v"(t0), [t1] "=&v"(t1), [a0] "=&v"(a0), [a1] "=&v"(a1), [sv] "=&s"(sv)
                : [wd] "v"(v[c][b].v[w]), [hm] "v"(himask), [ip] "v"(ip), [n] "v"(nrem), [ffff] "v"(0xFFFFu), [shw] "n"(CB - 1), [cb] "n"(CB),
                  [k0] "n"(8 * b + 2 * w), [k1] "n"(8 * b + 2 * w + 1)
                : "vcc", "memory");
          }
      } else if (E16) {
#pragma unroll
        for (int b = 0; b < kRowBatch; ++b)
#pragma unroll
          for (int w = 0; w < 8; ++w)
            if ((uint32_t)(8 * b + w) < nrem) count(i, (v[c][b].v[w >> 1] >> (16 * (w & 1))) & 0xFFFFu);
      } else {
#pragma unroll
        for (int b = 0; b < kRowBatch; ++b)
#pragma unroll
          for (int w = 0; w < 4; ++w)
            if (sq[c] + 4 * b + w < eq[c]) count(i, v[c][b].v[w]);
      }
      // longer buckets: rare
      if (E16) for (uint32_t q = sq[c] + kPerLoad * kRowBatch; q < eq[c]; ++q) count(i, ce16[q]);
      else for (uint32_t q = sq[c] + kPerLoad * kRowBatch; q < eq[c]; ++q) count(i, ce[q]);
    }
  }
  __syncthreads();
  const uint32_t thr = a.threshold > 1 ? (uint32_t)a.threshold : 1u, minLen = 2u * (uint32_t)(k + a.threshold);
  const uint32_t lane = tid & 63;
  for (uint32_t yy = yylo + (tid >> 6); yy < yyhi; yy += kSeedRowThreads / 64) {   // one wavefront per y
    const uint32_t y = y0 + yy;
    const int yLen = (int)(a.read_off[y + 1] - a.read_off[y]);
    const uint32_t nwords = (uint32_t)((((xLen + yLen - 1 + (1 << CB) - 1) >> CB) + 1) / 2);
    bool hit = false;
    for (uint32_t w = lane; w < nwords; w += 64) {
      const uint32_t v = cnt[yy * stride + w];
      hit |= (v & 0xFFFFu) >= thr || (v >> 16) >= thr;
    }
    // (sequences shorter than 2 (k + threshold) take the full envelope, diagenv.cpp:23-29: left to the per-pair kernel)
    const bool cand = __any(hit) || (uint32_t)xLen < minLen || (uint32_t)yLen < minLen;
    if (lane == 0) {
      const uint32_t p = it.pbase + (y - it.ylo);
      if (!cand) {
        record_band(a, a.pair_base + p, 0, 0);
        a.pair_ndiag[a.pair_base + p] = 1;
      }
      a.row_skip[p] = cand ? 0 : 1;
    }
  }
}

// The same prefilter with the chunk's k-mer index IN LDS and one 32-bit counter per coarse bin.  What bounded k_seed_rows was its
// instruction count at a third of the lanes (profiles/r03_pmc_overlap.json; 49 lane-instructions per k-mer match): ~15
// instructions to turn an index entry into a counter address and a 16-bit increment, behind 16-byte gathers of the entries,
// each lane from its own 128-byte line.  Here a workgroup takes one chunk and MANY x rows (a piece of the chunk-major item
// list).  It copies the chunk's bucket starts and entries to LDS once, each entry already as
//     L = (sequence in chunk) x W + (len - 1 - j)        W = a sequence's counters in units of 2^(CB-2) diagonals
// so that the counter of a match of x position i is at byte address ((L + i) >> (CB - 2)) & ~3: with CB = 2 (the overlap
// default: bins of 4 diagonals, a byte of address per diagonal) and 16-bit entries a match is one sdwa add, one mask, one
// compare that sets the execution mask, and a ds_add_u32.  Everything a match touches is LDS; the only global loads left are
// x's own k-mers (coalesced, fetched one x ahead).  The threshold scan of x's counters also clears them for the next x.
// The index is built with every bucket padded to an even number of entries (launch_chunk_index: estride > 0), so that a lane
// reads its lists four entries at a time from 4-byte aligned addresses; a pad entry (0xFFFFFFFF in the index) becomes
// L = 2^cl x W, whose increments land in a dummy zone behind the counters.  16 sequences of 2 kb: 64 KB of counters, 72 KB of
// 16-bit entries, 16 KB of bucket starts -- one workgroup of 16 wavefronts per CU.
__host__ __device__ inline uint32_t seed_rows_dummy_bytes(uint32_t max_len, int cb) { return (((max_len >> (cb - 2)) + 8u) + 15u) & ~15u; }
template <int CB, typename ET>
__global__ __launch_bounds__(kSeedRowThreads) void k_seed_rows_lds(SeedArgs a, uint32_t stride) {
  extern __shared__ uint32_t cnt[];   // [2^cl][stride] counters | dummy zone | bucket starts [nbuckets + 1] | sequence lengths [2^cl] | entries
  static_assert(CB >= 2, "an entry carries its position in units of 2^(CB-2) diagonals");
  // a piece: up to kSeedRowPiece consecutive x rows against one chunk.  Either items of the chunk-major list (row_sorted), or --
  // the scheduler's triangle (row x = pairs (x, x + 1 ... n_seqs - 1), src/qoverlap.cpp:475-480), where the items are a function
  // of (x, chunk) -- just (chunk, first row, rows) and the items are formed here (a 2^24-pair block has a million of them: built and
  // uploaded by the host they cost more than this kernel)
  const uint4 pd = a.row_pieces4[blockIdx.x];
  const bool tri = a.row_sorted == nullptr;
  const RowItemL* __restrict__ items = tri ? nullptr : (const RowItemL*)a.row_sorted + pd.x;
  const uint32_t n_items = pd.y;
  const int cl = a.chunk_log2, k = a.kmer_len;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, csize = 1u << cl, nb1 = a.nbuckets + 1, chunk = tri ? pd.z : items[0].chunk, y0 = chunk << cl;
  auto item_at = [&](uint32_t q) -> RowItemL {
    if (!tri) return items[q];
    const uint32_t x = pd.x + q, r = x - a.tri_x0;
    const uint64_t xb = a.read_off[x];
    RowItemL it;
    it.x = x;
    it.ylo = max(x + 1, y0);
    it.yhi = min(a.row_n_seqs, y0 + csize);
    it.pbase = (uint32_t)((uint64_t)r * (a.row_n_seqs - 1 - a.tri_x0) - (uint64_t)r * (r - 1) / 2) + (it.ylo - (x + 1));
    it.xlen = (uint32_t)(a.read_off[x + 1] - xb);
    it.xb_lo = (uint32_t)xb; it.xb_hi = (uint32_t)(xb >> 32);
    it.chunk = chunk;
    return it;
  };
  uint32_t* st = cnt + csize * stride + seed_rows_dummy_bytes(a.max_read_len, CB) / 4;
  uint32_t* ylen = st + nb1;
  ET* ents = (ET*)(ylen + csize);
  const uint32_t* __restrict__ cs = a.chunk_start + (uint64_t)chunk * nb1;
  const uint32_t* __restrict__ ce = a.chunk_entries + (uint64_t)chunk * a.chunk_estride;
  const uint32_t nent = cs[a.nbuckets];
  const uint32_t W = stride << CB;                                  // a sequence's counters in address units (4 x 2^(2 - CB) per byte)
  for (uint32_t w = tid; w < csize * stride; w += kSeedRowThreads) cnt[w] = 0;
  for (uint32_t q = tid; q < nb1; q += kSeedRowThreads) st[q] = cs[q];
  for (uint32_t q = tid; q < nent + 8; q += kSeedRowThreads) {      // (+ the slack the four-at-a-time reads may touch)
    const uint32_t e = q < nent ? ce[q] : 0xFFFFFFFFu;
    ents[q] = (ET)(e == 0xFFFFFFFFu ? csize * W : (e >> 26) * W + (e & 0x3FFFFFFu));
  }
  if (tid < csize) ylen[tid] = y0 + tid < a.row_n_seqs ? (uint32_t)(a.read_off[y0 + tid + 1] - a.read_off[y0 + tid]) : 0u;
  __syncthreads();
  constexpr int R = 2;                                               // x positions per thread whose k-mers are fetched one x ahead
  auto load_km = [&](const RowItemL& it, uint32_t* km) {
    const uint32_t* __restrict__ xk = a.skmer + (((uint64_t)it.xb_hi << 32) | it.xb_lo);
    const int nkx = (int)it.xlen - k + 1;
#pragma unroll
    for (int r = 0; r < R; ++r) km[r] = (int)(tid + r * kSeedRowThreads) < nkx ? xk[tid + r * kSeedRowThreads] : 0u;
  };
  const uint32_t thr = a.threshold > 1 ? (uint32_t)a.threshold : 1u, minLen = 2u * (uint32_t)(k + a.threshold);
  const uint32_t cnt_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)cnt;   // LDS address of the counters
  const uint32_t ents_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) ET*)ents;
  // Two x in flight: a counter word holds x (even item)'s count in its low half and x (odd item)'s in its high half (a bin holds
  // fewer than 65 536 matches: seed_row_stride_bytes), so while the wavefronts walk item p they also scan and clear item p - 1's
  // halves -- one workgroup barrier per x, and each wavefront has the other activity to issue while one waits on LDS.
  auto settle = [&](const RowItemL& it, uint32_t half) {            // scan + clear the counters of `it`, one wavefront per y
    const uint32_t yylo = it.ylo - y0, yyhi = it.yhi - y0;
    const int xLen = (int)it.xlen;
    const unsigned long long keep = half ? 0x0000FFFF0000FFFFull : 0xFFFF0000FFFF0000ull;
    for (uint32_t yy = yylo + (tid >> 6); yy < yyhi; yy += kSeedRowThreads / 64) {
      const int yLen = (int)ylen[yy];
      const uint32_t nwords = (uint32_t)(((xLen + yLen - 1 + (1 << CB) - 1) >> CB) + 1);
      bool hit = false;
      for (uint32_t w = lane * 4; w < nwords; w += 256) {           // (rows are a multiple of four words long)
        uint32_t* q = &cnt[yy * stride + w];
        const uint4 v = *(const uint4*)q;
        const uint32_t sh = 16 * half;
        hit |= ((v.x >> sh) & 0xFFFFu) >= thr || ((v.y >> sh) & 0xFFFFu) >= thr || ((v.z >> sh) & 0xFFFFu) >= thr || ((v.w >> sh) & 0xFFFFu) >= thr;
        atomicAnd((unsigned long long*)q, keep);                   // (the walkers of the other x are adding to the other halves)
        atomicAnd((unsigned long long*)q + 1, keep);
      }
      // (sequences shorter than 2 (k + threshold) take the full envelope, diagenv.cpp:23-29: left to the per-pair kernel)
      const bool cand = __any(hit) || (uint32_t)xLen < minLen || (uint32_t)yLen < minLen;
      if (lane == 0) {
        const uint32_t p = it.pbase + (y0 + yy - it.ylo);
        if (!cand) {   // the pair's one band, the forced diagonal 0 (nobody else records bands of a settled pair)
          a.pair_bands[(uint64_t)(a.pair_base + p) * kMaxBandsPerPair] = make_int2(0, 0);
          a.pair_nbands[a.pair_base + p] = 1;
          a.pair_ndiag[a.pair_base + p] = 1;
        }
        a.row_skip[p] = cand ? 0 : 1;
      }
    }
  };
  RowItemL cur = item_at(0), prev = cur;
  uint32_t kmc[R];
  load_km(cur, kmc);
  for (uint32_t itn = 0; itn < n_items; ++itn) {
    const RowItemL nxt = item_at(min(itn + 1, n_items - 1));
    uint32_t kmn[R];
    load_km(nxt, kmn);
    const int xLen = (int)cur.xlen, nkx = xLen - k + 1;
    const uint32_t yylo = cur.ylo - y0, yyhi = cur.yhi - y0;
    const bool whole = yylo == 0 && yyhi == csize;                   // the item takes every sequence of the chunk (all but a row's end chunks)
    const uint32_t Llo = yylo * W, Lhi = yyhi * W;                   // (a pad entry, L = csize x W, is outside every such range)
    const uint32_t inc = (itn & 1u) ? 0x10000u : 1u;
    auto count = [&](uint32_t i, uint32_t L) {                       // bin = i - j + yLen - 1 (diagenv.cpp:33-40)
      atomicAdd((uint32_t*)((char*)cnt + (((L + i) >> (CB - 2)) & ~3u)), inc);
    };
    uint32_t s[R], n[R], nmax = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool in = (int)(tid + r * kSeedRowThreads) < nkx;
      s[r] = st[kmc[r]];
      n[r] = in ? st[kmc[r] + 1] - s[r] : 0u;                        // (even: the pad entry is counted into the dummy zone)
      nmax = max(nmax, n[r]);
    }
    if (whole && CB == 2 && sizeof(ET) == 2) {
      // The thread's first two positions side by side, four entries of each list per round, the next round's entries fetched
      // before this round's are counted.  Written out: two 4-byte reads per list (the lists start on even entries), then per
      // entry the address (sdwa add of the 16-bit entry to the position, which also carries the counters' LDS address; mask)
      // and the count under an execution mask set by the compare itself (lanes whose list still has entry t + u) and put back
      // by one scalar move.  LDS returns in order: with the four reads of the next round and at most eight counts behind them
      // in flight, lgkmcnt(8) at the top of a round says this round's entries have arrived.
      static_assert(R == 2, "the counting block is written out for two lists");
      const uint32_t ipos0 = tid + cnt_base, ipos1 = tid + kSeedRowThreads + cnt_base;
      uint32_t p0 = ents_base + s[0] * 2, p1 = ents_base + s[1] * 2;
      int rem0 = (int)n[0], rem1 = (int)n[1];
      uint32_t w00, w01, w10, w11;
      asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:4\n ds_read_b32 %2, %5\n ds_read_b32 %3, %5 offset:4\n"
                   : "=&v"(w00), "=&v"(w01), "=&v"(w10), "=&v"(w11) : "v"(p0), "v"(p1) : "memory");
      if (itn) settle(prev, (itn - 1) & 1u);                         // (the first round's entries arrive meanwhile)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for (uint32_t t = 0; t < nmax; t += 4) {
        uint32_t a0, a1, a2, a3, a4, a5, a6, a7, c00, c01, c10, c11;
        unsigned long long sv;
        p0 += 8; p1 += 8;
        asm volatile(
            "s_waitcnt lgkmcnt(8)\n"
            "v_mov_b32 %[c00], %[w00]\n v_mov_b32 %[c01], %[w01]\n v_mov_b32 %[c10], %[w10]\n v_mov_b32 %[c11], %[w11]\n"
            "ds_read_b32 %[w00], %[p0]\n ds_read_b32 %[w01], %[p0] offset:4\n ds_read_b32 %[w10], %[p1]\n ds_read_b32 %[w11], %[p1] offset:4\n"
            "v_add_u32_sdwa %[a0], %[i0], %[c00] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
            "v_add_u32_sdwa %[a1], %[i0], %[c00] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
            "v_add_u32_sdwa %[a2], %[i0], %[c01] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
            "v_add_u32_sdwa %[a3], %[i0], %[c01] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
            "v_add_u32_sdwa %[a4], %[i1], %[c10] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
            "v_add_u32_sdwa %[a5], %[i1], %[c10] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
            "v_add_u32_sdwa %[a6], %[i1], %[c11] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
            "v_add_u32_sdwa %[a7], %[i1], %[c11] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
            "v_and_b32 %[a0], -4, %[a0]\n v_and_b32 %[a1], -4, %[a1]\n v_and_b32 %[a2], -4, %[a2]\n v_and_b32 %[a3], -4, %[a3]\n"
            "v_and_b32 %[a4], -4, %[a4]\n v_and_b32 %[a5], -4, %[a5]\n v_and_b32 %[a6], -4, %[a6]\n v_and_b32 %[a7], -4, %[a7]\n"
            "s_mov_b64 %[sv], exec\n"
            "v_cmpx_lt_i32 vcc, 0, %[r0]\n ds_add_u32 %[a0], %[inc]\n s_mov_b64 exec, %[sv]\n"
            "v_cmpx_lt_i32 vcc, 0, %[r1]\n ds_add_u32 %[a4], %[inc]\n s_mov_b64 exec, %[sv]\n"
            "v_cmpx_lt_i32 vcc, 1, %[r0]\n ds_add_u32 %[a1], %[inc]\n s_mov_b64 exec, %[sv]\n"
            "v_cmpx_lt_i32 vcc, 1, %[r1]\n ds_add_u32 %[a5], %[inc]\n s_mov_b64 exec, %[sv]\n"
            "v_cmpx_lt_i32 vcc, 2, %[r0]\n ds_add_u32 %[a2], %[inc]\n s_mov_b64 exec, %[sv]\n"
            "v_cmpx_lt_i32 vcc, 2, %[r1]\n ds_add_u32 %[a6], %[inc]\n s_mov_b64 exec, %[sv]\n"
            "v_cmpx_lt_i32 vcc, 3, %[r0]\n ds_add_u32 %[a3], %[inc]\n s_mov_b64 exec, %[sv]\n"
            "v_cmpx_lt_i32 vcc, 3, %[r1]\n ds_add_u32 %[a7], %[inc]\n s_mov_b64 exec, %[sv]\n"
            : [w00] "+&v"(w00), [w01] "+&v"(w01), [w10] "+&v"(w10), [w11] "+&v"(w11), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2),
              [a3] "=&v"(a3), [a4] "=&v"(a4), [a5] "=&v"(a5), [a6] "=&v"(a6), [a7] "=&v"(a7), [c00] "=&v"(c00), [c01] "=&v"(c01),
              [c10] "=&v"(c10), [c11] "=&v"(c11), [sv] "=&s"(sv)
            : [p0] "v"(p0), [p1] "v"(p1), [i0] "v"(ipos0), [i1] "v"(ipos1), [r0] "v"(rem0), [r1] "v"(rem1), [inc] "v"(inc)
            : "vcc", "memory");
        rem0 -= 4; rem1 -= 4;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // (the last round's look-ahead reads land in registers nobody uses)
    } else {
      if (itn) settle(prev, (itn - 1) & 1u);
      for (uint32_t t = 0; t < nmax; ++t) {
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (t < n[r]) {
            const uint32_t L = ents[s[r] + t];
            if (L >= Llo && L < Lhi) count(tid + r * kSeedRowThreads, L);
          }
      }
    }
    const uint32_t* __restrict__ xk = a.skmer + (((uint64_t)cur.xb_hi << 32) | cur.xb_lo);
    for (int i = (int)tid + R * kSeedRowThreads; i < nkx; i += kSeedRowThreads) {   // sequences longer than R x 1024 bases
      const uint32_t km = xk[i], s0 = st[km], e0 = st[km + 1];
      for (uint32_t q = s0; q < e0; ++q) {
        const uint32_t L = ents[q];
        if (L >= Llo && L < Lhi) count((uint32_t)i, L);
      }
    }
    __syncthreads();
    prev = cur;
    cur = nxt;
#pragma unroll
    for (int r = 0; r < R; ++r) kmc[r] = kmn[r];
  }
  settle(prev, (n_items - 1) & 1u);
}

__global__ void k_chunk_kmer_count(const uint8_t* __restrict__ tok, const uint64_t* __restrict__ off, uint32_t k, uint32_t nbuckets,
                                   int cl, uint32_t* __restrict__ counts) {
  const uint32_t x = blockIdx.y;
  const uint64_t b = off[x], len = off[x + 1] - b;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (len < k || i > len - k) return;
  uint32_t km = 0;
  for (uint32_t c = 0; c < k; ++c) km = km * 4 + tok[b + i + c];
  atomicAdd(&counts[(uint64_t)(x >> cl) * (nbuckets + 1) + km], 1u);
}
__global__ void k_chunk_kmer_scatter(const uint8_t* __restrict__ tok, const uint64_t* __restrict__ off, uint32_t k, uint32_t nbuckets,
                                     int cl, const uint32_t* __restrict__ starts, uint32_t* __restrict__ cursor,
                                     uint32_t* __restrict__ entries, uint64_t estride, uint32_t wd, int pb) {
  const uint32_t x = blockIdx.y;
  const uint64_t b = off[x], len = off[x + 1] - b;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (len < k || i > len - k) return;
  uint32_t km = 0;
  for (uint32_t c = 0; c < k; ++c) km = km * 4 + tok[b + i + c];
  const uint64_t bi = (uint64_t)(x >> cl) * (nbuckets + 1) + km;
  const uint32_t slot = starts[bi] + atomicAdd(&cursor[bi], 1u);
  // a chunk's entries: behind those of the chunks before it, or (padded index) `estride` entries per chunk
  const uint64_t cbase = estride ? (uint64_t)(x >> cl) * estride : pb ? chunk_base16(off, x >> cl, cl, nbuckets) : off[(uint64_t)(x >> cl) << cl];
  // (k_seed_rows: sequence x wd + position, wd = the diagonals a sequence's counters span; k_seed_rows_lds re-packs (sequence, position) itself)
  if (pb) ((uint16_t*)entries)[cbase + slot] = (uint16_t)(((x & ((1u << cl) - 1)) << pb) | (uint32_t)(len - 1 - i));   // k_seed_rows<., true>
  else entries[cbase + slot] = wd ? (x & ((1u << cl) - 1)) * wd + (uint32_t)(len - 1 - i) : ((x & ((1u << cl) - 1)) << 26) | (uint32_t)(len - 1 - i);
}

// 16-bit index: a bucket's (first entry, entries) side by side, so that k_seed_rows fetches both with one 8-byte load (the cursor
// array holds the counts once the scatter is done; buckets are padded to even length, so starts are even)
__global__ void k_chunk_bounds(const uint32_t* __restrict__ starts, const uint32_t* __restrict__ cursor, uint32_t nbuckets, uint2* __restrict__ bounds) {
  const uint32_t km = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (km >= nbuckets) return;
  const uint64_t bi = (uint64_t)c * (nbuckets + 1) + km;
  bounds[(uint64_t)c * nbuckets + km] = make_uint2(starts[bi], cursor[bi]);
}

// Bands -> units: class, unit id, class-list slot, traceback offset, cell counts.  One thread per
// (pair, band slot); one global atomic per workgroup and counter.
// (1 024 threads per workgroup: the handful of global atomics a workgroup makes all go to the same few words, and at 256
// threads a 2^24-pair row block made 800 000 of them -- 5 ms of serialised L2 atomics per block)
constexpr int kBinThreads = 1024;
__global__ __launch_bounds__(kBinThreads) void k_bin_units(SeedArgs a, uint32_t n_pairs, uint32_t n_ovf) {
  __shared__ uint32_t s_cnt[kNumClasses], s_base[kNumClasses], s_nact, s_ubase;
  __shared__ unsigned long long s_cells[kNumClasses], s_wtot[kBinThreads / 64], s_wbase[kBinThreads / 64], s_tb_total, s_tb_base;
  const uint32_t tid = threadIdx.x;
  const uint64_t idx = (uint64_t)blockIdx.x * kBinThreads + tid;
  const uint32_t pidx = (uint32_t)(idx / kMaxBandsPerPair), slot = (uint32_t)(idx % kMaxBandsPerPair);
  if (tid < kNumClasses) { s_cnt[tid] = 0; s_cells[tid] = 0; }
  if (tid == 0) s_nact = 0;
  __syncthreads();
  bool act = false, have = false;
  uint32_t pair = 0, lrank = 0, urank = 0;
  int dlo = 0, dhi = 0, cls = 0, yLen = 0, xLen = 0;
  unsigned long long tbw = 0, cells = 0;
  if (n_ovf) {  // overflow pass: one thread per spilled band
    if (idx < n_ovf) {
      const int4 ob = a.ovf_bands[idx];
      pair = (uint32_t)ob.x; dlo = ob.y; dhi = ob.z;
      have = true;
    }
  } else if (pidx < n_pairs) {
    pair = a.pair_base + pidx;
    if (slot < min(a.pair_nbands[pair], (uint32_t)kMaxBandsPerPair)) {
      const int2 bd = a.pair_bands[(uint64_t)pair * kMaxBandsPerPair + slot];
      dlo = bd.x; dhi = bd.y;
      have = true;
    }
  }
  {
    if (have) {
      uint32_t r, x;
  pair_rx(a, pair, r, x);
      xLen = (int)(a.ref_off[x + 1] - a.ref_off[x]);
      yLen = (int)(a.read_off[r + 1] - a.read_off[r]);
      cls = classify_width(dhi - dlo + 1);
      if (a.storage_mode == 2 && cls > 10) cls = kRowClass;    // overlap kernels take up to 8 diagonals per lane
      if (a.storage_mode == 2 && a.ov_use_32x3 && dhi - dlo + 1 > 64 && dhi - dlo + 1 <= 96) cls = kOv32Class;   // (qf_device.hpp)
      if (a.storage_mode == 1 && dhi - dlo + 1 > (a.fb_use_32x3 ? 64 : 80) && dhi - dlo + 1 <= 96) cls = kOv32Class;   // E-step: (32, 3) instead of (16, 6) [-kmatchband 80: Backward 23.8 -> 20.7 ms per 20 k reads] -- and, as an A/B, of (16, 5) [15.0 -> 20.0]
      // overlap bands of 97 .. 128 diagonals are few (a hundred per thousand x rows) and each is a chain of 2 000 steps: on 64
      // lanes x 3 diagonals a step is three dependent look-ups deep instead of eight on 16 lanes x 8
      if (a.storage_mode == 2 && cls == 6 && !a.ov_wide_on_16x8) cls = 7;
      if (cls < 0) {
        atomicOr(&a.bc->error, 2u);
        a.bc->error_detail = (uint32_t)(dhi - dlo + 1);
      } else {
        act = true;
        lrank = atomicAdd(&s_cnt[cls], 1u);
        urank = atomicAdd(&s_nact, 1u);
        tbw = a.storage_mode == 2 ? (cls == 0 ? 0ull   // a single-diagonal band's traceback is one flag, kept in the unit
                                     : cls == kRowClass ? row_ov_words(dlo, dhi, xLen, yLen)
                                               : (unsigned long long)(band_cols(dlo, dhi, xLen, yLen) + fill_class(cls).G - 1) * fill_class(cls).G * 2)
              : a.storage_mode == 1 ? (cls == kRowClass ? row_fw_doubles(dlo, dhi, xLen, yLen) : unit_fw_doubles(cls, (uint32_t)yLen))
              : cls == kRowClass  ? row_unit_words(dlo, dhi, xLen, yLen)
                                  : unit_tb_words(cls, (uint32_t)yLen);
        tbw = (tbw + 1) & ~1ull;  // keep every unit 8-byte aligned (row-space units hold doubles)
        cells = (dhi - dlo + 1 == xLen + yLen - 1) ? (unsigned long long)xLen * (unsigned long long)yLen
                                                    : band_cells(dlo, dhi, xLen, yLen);
        atomicAdd(&s_cells[cls], cells);
      }
    }
  }
  // exclusive prefix of the units' storage over the workgroup: inside a wavefront by shifts, across wavefronts through LDS
  unsigned long long incl = tbw;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long v = (unsigned long long)__shfl_up((long long)incl, o, 64);
    if ((int)(tid & 63u) >= o) incl += v;
  }
  if ((tid & 63u) == 63u) s_wtot[tid >> 6] = incl;
  __syncthreads();
  if (tid < 64) {
    const unsigned long long t = tid < kBinThreads / 64 ? s_wtot[tid] : 0ull;
    unsigned long long ti = t;
#pragma unroll
    for (int o = 1; o < kBinThreads / 64; o <<= 1) {
      const unsigned long long v = (unsigned long long)__shfl_up((long long)ti, o, 64);
      if ((int)tid >= o) ti += v;
    }
    if (tid < kBinThreads / 64) s_wbase[tid] = ti - t;
    if (tid == kBinThreads / 64 - 1) s_tb_total = ti;
  }
  __syncthreads();
  const unsigned long long excl = s_wbase[tid >> 6] + incl - tbw;
  if (tid < kNumClasses && s_cnt[tid]) {
    s_base[tid] = atomicAdd(&a.bc->cls_count[tid], s_cnt[tid]);
    atomicAdd(&a.bc->cls_cells[tid], s_cells[tid]);
    atomicAdd(&a.bc->total_cells, s_cells[tid]);
  }
  if (tid == 0 && s_nact) {
    s_ubase = atomicAdd(&a.bc->n_units, s_nact);
    s_tb_base = atomicAdd(&a.bc->tb_words, s_tb_total);
  }
  __syncthreads();
  if (!act) return;
  const uint32_t uid = s_ubase + urank;
  if (uid >= a.max_units) {
    atomicOr(&a.bc->error, 1u);
    return;
  }
  if (a.slot_list && cls == 0) {   // overlap: single-diagonal bands by (y chunk, x row, y), so that a workgroup is one x against 256 consecutive y
    const uint32_t x = a.pair_x[pair], y = a.pair_y[pair];
    // The host admits the slotted list only for lists in which (x, y) is unique and a pair has one single-diagonal band; should
    // that ever not hold, the second claimant of a slot raises error bit 4 and the host redoes the chunk with the plain list.
    // (and that the band is the forced diagonal 0, src/diagenv.cpp:53: bands are at least 2 wide otherwise)
    if (dlo != 0 || atomicCAS(&a.slot_list[((uint64_t)(y >> 8) * a.slot_rows + (x - a.slot_x0)) * 256 + (y & 255u)], kNoUnit, uid) != kNoUnit)
      atomicOr(&a.bc->error, 16u);
  } else
    a.cls_list[(uint64_t)cls * a.max_units + s_base[cls] + lrank] = uid;
  // sort key of the class lists (descending): read length = steps of the fill; overlap: the columns the band crosses, and its
  // single-diagonal list is put back into pair order instead, so that the bands of a workgroup share their x (k_overlap_single_lds)
  if (a.cls_key)
    a.cls_key[(uint64_t)cls * a.max_units + s_base[cls] + lrank] =
        a.storage_mode != 2 ? (uint32_t)yLen : cls == 0 ? ~pair : (uint32_t)band_cols(dlo, dhi, xLen, yLen);
  Unit u;
  u.pair = pair;
  u.dlo = dlo;
  u.dhi = dhi;
  u.tb_off = s_tb_base + excl;
  u.end_val = QF_NEG_INF;
  u.end_i = 0;
  u.cls = (uint32_t)cls;
  u.end2_val = QF_NEG_INF;
  u.end2_j = 0;
  u.staged = 0;
  u.next = atomicExch(&a.pair_head[pair], uid);
  a.units[uid] = u;
  atomicAdd(&a.pair_cells[pair], cells);
}

// ------------------------------------------------------------------------------------------------
// Banded Viterbi fill: QuaffViterbiMatrix ctor, src/qmodel.cpp:1512-1560
//
// Lane l of a G-lane group owns B adjacent diagonals d0 = dlo + l*B ... and at step t works on read
// column j = t - l + 1 (a skew of one column per lane).  With that skew every dependency is either in
// the lane's own registers or one lane away:
//   mat(i,j) <- (i-1,j-1): same diagonal, previous step               (own registers)
//   ins(i,j) <- (i,  j-1): diagonal d+1, previous column              (own slot b+1, or lane l+1's slot 0,
//                                                                       which lane l+1 finishes this very step)
//   del(i,j) <- (i-1,j  ): diagonal d-1, same column                  (own slot b-1 this step, or lane l-1's
//                                                                       last slot from the previous step)
// so a group advances all its lanes every step.  Each cell records 4 traceback bits chosen exactly as
// QuaffViterbiMatrix::alignment's updateMax sequence would choose them (src/qmodel.cpp:1590-1616).
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// k_viterbi_fill: the wavefront described above with a trimmed instruction stream.
//  * Every step is classified wave-uniformly.  A FAST step has no lane on its first or last read column and no lane
//    whose band pokes above reference row 1; it needs no start candidate, no end tracking and no row/column validity
//    masking (cells before a lane's first column are -inf by construction, cells below the last reference row or
//    after the last column are never read by a valid cell).  Everything else takes the general SLOW step.  For a
//    1 kb read ~95 % of the steps are FAST.
//  * Values come from v_max_f64; the traceback flags from compares of the same candidates (first maximum in the
//    reference's M, I, D order) instead of value/flag select chains.
//  * Lane exchange by DPP row/wave shifts (a -inf "old" value fills the group's edge lane) instead of ds_bpermute.
// ------------------------------------------------------------------------------------------------
template <int G, int B, bool GAPCTX, bool EMLDS>
__global__ __launch_bounds__(256) void k_viterbi_fill(FillArgs a) {
  // EMLDS: the match-emission table (+ its -inf row) and the insert-emission table are copied to LDS once per workgroup.
  // Every lane of a wavefront is on a different read column, so the B emission fetches of a step are 64-way gathers;
  // through the vector L1 those gathers, not the arithmetic, bound the kernel (measured), LDS serves them far faster.
  extern __shared__ double lds_tab[];
  const uint32_t n_em = a.dp.ematch_ninf_off / 8 + 4;
  if (EMLDS) {
    for (uint32_t k = threadIdx.x; k < n_em; k += 256) lds_tab[k] = a.dp.ematch[k];
    for (uint32_t k = threadIdx.x; k < kInsRows; k += 256) lds_tab[n_em + k] = a.dp.eins[k];
    __syncthreads();
  }
  constexpr int UPW = 64 / G;
  constexpr int WPL = B > 8 ? 2 : 1;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int grp = lane / G, l = lane % G;
  const uint32_t uidx = wave * UPW + grp;
  const bool active = uidx < a.n_cls_units;

  uint32_t uid = 0;
  int dlo = 0, dhi = -1, xLen = 0, yLen = 0;
  uint64_t xb = 0, yb = 0, xw = 0, tb_off = 0;
  if (active) {
    uid = a.cls_list[uidx];
    const Unit u = a.units[uid];
    const uint32_t r = u.pair / a.n_refs, x = u.pair % a.n_refs;
    xb = a.ref_off[x]; xLen = (int)(a.ref_off[x + 1] - xb); xw = a.ref_woff[x];
    yb = a.read_off[r]; yLen = (int)(a.read_off[r + 1] - yb);
    dlo = u.dlo; dhi = u.dhi; tb_off = u.tb_off;
  }
  int T = active ? yLen + G - 1 : 0;
  for (int o = 32; o; o >>= 1) T = max(T, __shfl_xor(T, o));

  const int d0 = dlo + l * B;
  const int bmax = active ? dhi - d0 : -1;  // slots b > bmax are outside the band
  const double i2m = a.dp.i2m, d2m = a.dp.d2m, i2i = a.dp.i2i, d2d = a.dp.d2d;
  const double* __restrict__ ematch = EMLDS ? lds_tab : a.dp.ematch;
  const double* __restrict__ eins = EMLDS ? lds_tab + n_em : a.dp.eins;
  const double* __restrict__ trans = a.dp.trans;
  const uint32_t Kg = a.dp.Kg;
  const bool local = a.dp.local != 0;
  const double c_m2m = trans[0], c_m2i = trans[Kg], c_m2d = trans[2 * Kg], c_m2e = trans[3 * Kg];

  double M[B], I[B], D[B];
#pragma unroll
  for (int b = 0; b < B; ++b) M[b] = I[b] = D[b] = QF_NEG_INF;
  double pubM = QF_NEG_INF, pubD = QF_NEG_INF;
  double bestEnd = QF_NEG_INF;
  uint32_t bestI = 0;

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
  U32x4 cwn = *(const U32x4*)(ctx + (0 - l));
  uint32_t gkPrev = 0;
  uint32_t* __restrict__ tb = a.tb + tb_off;

  // Emission scores are fetched one step ahead (context word, reference-token window and the B match rows of step t+1
  // are all known at step t), so a step never waits for its own loads.
  // Slots above the band's last diagonal read a -inf emission instead: their match and insert states stay -inf, and
  // the delete state they pick up from inside the band never reaches a valid cell.
  const uint32_t ninf_off = a.dp.ematch_ninf_off;
  // LDS tables are addressed by their 32-bit LDS offset (the table base folded into the per-step row offset)
  typedef __attribute__((address_space(3))) const double lds_cdouble;
  const uint32_t em_base = EMLDS ? (uint32_t)(uintptr_t)(__attribute__((address_space(3))) double*)lds_tab : 0u;
  auto emis = [&](uint32_t w, uint32_t window, int b) -> double {
    if (EMLDS) {  // em_base is 32-byte aligned (dynamic LDS starts at 0; rows are 32 bytes), so the or below is an add
      uint32_t off = (((w & 0x7FFFu) << 5) + em_base) | (((window >> (2 * b)) & 3u) << 3);
      if (b > bmax) off = ninf_off + em_base;
      return *(lds_cdouble*)(uintptr_t)off;
    }
    uint32_t off = ((w & 0x7FFFu) << 5) | (((window >> (2 * b)) & 3u) << 3);
    if (b > bmax) off = ninf_off;
    return *(const double*)((const char*)ematch + off);
  };
  uint32_t wN = cwn.v[0];
  const uint32_t tok0 = (uint32_t)((((unsigned long long)xnx << 32) | xhi) >> sh0) & 3u;  // (xhi, xnx) become chunk 0's window
  uint32_t winN = (win >> 2) | (tok0 << (2 * (B - 1)));
  double eN[B], insEN = eins[(wN >> 15) & 0x1FFu];
#pragma unroll
  for (int b = 0; b < B; ++b) eN[b] = emis(wN, winN, b);

  // Steps are specialised four at a time (one traceback tile).  A FAST tile has no lane on its first or last read column
  // and no lane whose lowest row is above reference row 1; the wave-uniform bounds below are conservative:
  //   [0, slowA]      some lane is on column 1 or still has rows above row 1
  //   [slowB0, slowB1] some lane is on its last column
  int slowA = active ? max(l, l - d0 - 1) : -1, slowB0 = active ? yLen + l - 1 : 0x7FFFFFFF, slowB1 = active ? yLen + l - 1 : -1;
  for (int o = 32; o; o >>= 1) {
    slowA = max(slowA, __shfl_xor(slowA, o));
    slowB0 = min(slowB0, __shfl_xor(slowB0, o));
    slowB1 = max(slowB1, __shfl_xor(slowB1, o));
  }
  slowA = __builtin_amdgcn_readfirstlane(slowA);
  slowB0 = __builtin_amdgcn_readfirstlane(slowB0);
  slowB1 = __builtin_amdgcn_readfirstlane(slowB1);
  // lane 0 of a group: in FAST tiles the exchange from below zero-fills it and these constants turn the zero into -inf
  const double m2dEdge = l == 0 ? QF_NEG_INF : c_m2d, d2dEdge = l == 0 ? QF_NEG_INF : d2d;

  int chunk = 0;
  for (int t0 = 0; t0 < T; t0 += 16, ++chunk) {
    xlo = xhi; xhi = xnx; xnx = xword(q0 + chunk + 2);
    const unsigned long long xpair = ((unsigned long long)xhi << 32) | xlo;
    for (int s4 = 0; s4 < 16; s4 += 4) {
      const U32x4 cw = cwn;
      cwn = *(const U32x4*)(ctx + min(t0 + s4 + 4 - l, yLen + 4));
      U32x4 tile;   // this lane's traceback words of the four steps (one 16-byte store)
      const int ts = t0 + s4;
      const bool tileFast = (ts > slowA && ts + 3 < slowB0) || ts > slowB1;   // wave-uniform
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bool SLOW = !tileFast;
        const int t = t0 + s4 + s;
        const int j = t - l + 1;
        const bool colvalid = active && j >= 1 && j <= yLen;
        const uint32_t w = wN;
        double e[B];
        const double insE = insEN;
#pragma unroll
        for (int b = 0; b < B; ++b) e[b] = eN[b];
        // step t+1's fetch (the token of step 16 of a chunk is the next chunk's first: still inside the 64-bit window)
        wN = s < 3 ? cw.v[s + 1] : cwn.v[0];
        winN = (winN >> 2) | (((uint32_t)(xpair >> (sh0 + 2 * (s4 + s + 1))) & 3u) << (2 * (B - 1)));
        insEN = eins[(wN >> 15) & 0x1FFu];
#pragma unroll
        for (int b = 0; b < B; ++b) eN[b] = emis(wN, winN, b);
        const uint32_t gk = w >> 24;
        double m2m, m2i, m2d;
        if (GAPCTX) {
          const uint32_t gp = j <= 1 ? 0u : gkPrev;
          m2m = trans[gp]; m2i = trans[Kg + gp]; m2d = trans[2 * Kg + gk];
          gkPrev = gk;
        } else {
          m2m = c_m2m; m2i = c_m2i; m2d = c_m2d;
        }
        uint32_t tbw0 = 0, tbw1 = 0;
        double upM = 0, upI = 0;
        if (!SLOW) {
          // ---------------- FAST step
          // Lane exchange zero-fills the group's edge lanes; the edge constants (-inf there) make the sums -inf.
          // Flags are shifted in as raw compare bits, one v_addc each (first slot ends up highest, bits I>M, D>max(M,I),
          // ins-from-I, del-from-D from the top of its nibble); a bit reverse restores "slot b at nibble b" with the usual
          // bit order.  Match source "I>M and D>max" reads as 3, which the traceback takes as D away from column 1.
          constexpr bool EDGE = !GAPCTX;   // with gap contexts the transition scores change per step: keep the -inf fill
          double prevM = EDGE ? dpp_from_below<G, true>(pubM) : dpp_from_below<G, false>(pubM);
          double prevD = EDGE ? dpp_from_below<G, true>(pubD) : dpp_from_below<G, false>(pubD);
          uint32_t acc0 = 0, acc1 = 0;
#pragma unroll
          for (int b = 0; b < B; ++b) {
            const double tM = (M[b] + m2m) + e[b], tI = (I[b] + i2m) + e[b], tD = (D[b] + d2m) + e[b];
            const double m1 = fmax(tM, tI);
            const double nm = fmax(m1, tD);
            double cM, cI;
            if (b + 1 < B) { cM = (M[b + 1] + m2i) + insE; cI = (I[b + 1] + i2i) + insE; }
            else { cM = (upM + m2i) + insE; cI = (upI + i2i) + insE; }
            const double ni = fmax(cM, cI);
            const double gM = prevM + ((EDGE && b == 0) ? m2dEdge : m2d), gD = prevD + ((EDGE && b == 0) ? d2dEdge : d2d);
            const double ndl = fmax(gM, gD);
            uint32_t& acc = b < 8 ? acc0 : acc1;
            acc = shift_in_gt(acc, tI, tM);
            acc = shift_in_gt(acc, tD, m1);
            acc = shift_in_gt(acc, cI, cM);
            acc = shift_in_gt(acc, gD, gM);
            M[b] = nm; I[b] = ni; D[b] = ndl;
            prevM = nm; prevD = ndl;
            if (b == 0) {
              upM = dpp_from_above<G, false>(nm);
              upI = dpp_from_above<G, false>(ni);
            }
          }
          tbw0 = __builtin_bitreverse32(acc0) >> (32 - 4 * (B < 8 ? B : 8));
          if (B > 8) tbw1 = __builtin_bitreverse32(acc1) >> (B > 8 ? 32 - 4 * (B - 8) : 0);
          pubM = prevM; pubD = prevD;
        } else {
          // ---------------- general step (start candidate, end tracking, full validity masking)
          double prevM = dpp_from_below<G, false>(pubM), prevD = dpp_from_below<G, false>(pubD);
          const bool startCol = j == 1, endCol = j == yLen;
#pragma unroll
          for (int b = 0; b < B; ++b) {
            const int d = d0 + b, i = d + j;
            const bool valid = colvalid && d <= dhi && i >= 1 && i <= xLen;
            const double tM = (M[b] + m2m) + e[b], tI = (I[b] + i2m) + e[b], tD = (D[b] + d2m) + e[b];
            double nm = tM;
            uint32_t sm = 0;
            if (tI > nm) { nm = tI; sm = 1; }
            if (tD > nm) { nm = tD; sm = 2; }
            if (startCol && (i == 1 || local) && e[b] > nm) { nm = e[b]; sm = 3; }
            double srcM, srcI;
            if (b + 1 < B) { srcM = M[b + 1]; srcI = I[b + 1]; } else { srcM = upM; srcI = upI; }
            const double cM = (srcM + m2i) + insE, cI = (srcI + i2i) + insE;
            double ni = cM;
            uint32_t si = 0;
            if (cI > ni) { ni = cI; si = 1; }
            const double gM = prevM + m2d, gD = prevD + d2d;
            double ndl = gM;
            uint32_t sd = 0;
            if (gD > ndl) { ndl = gD; sd = 1; }
            // idle lanes (before their first column) stay -inf by themselves; cells below the last reference row or
            // past the last column may hold anything (never read by a valid cell) but are forced to -inf here too
            if (!valid) { nm = QF_NEG_INF; ni = QF_NEG_INF; ndl = QF_NEG_INF; }
            M[b] = nm; I[b] = ni; D[b] = ndl;
            prevM = nm; prevD = ndl;
            const uint32_t nib = sm | (si << 2) | (sd << 3);
            if (b < 8) tbw0 |= nib << (4 * (b & 7)); else tbw1 |= nib << (4 * (b & 7));
            if (endCol && valid && (local || i == xLen)) {
              const double ev = nm + (GAPCTX ? trans[3 * Kg + gk] : c_m2e);
              if (ev >= bestEnd) { bestEnd = ev; bestI = (uint32_t)i; }
            }
            if (b == 0) { upM = dpp_from_above<G, false>(nm); upI = dpp_from_above<G, false>(ni); }
          }
          pubM = prevM; pubD = prevD;
        }
        if (WPL == 1) tile.v[s] = tbw0;
        else if (colvalid) { tb_store(&tb[((uint64_t)t * G + l) * 2], tbw0); tb_store(&tb[((uint64_t)t * G + l) * 2 + 1], tbw1); }
      }
      // steps t0+s4 .. +3 hold columns j0 .. j0+3 of this lane; stored when any of them is a real column
      if (WPL == 1) {
        const int j0 = t0 + s4 - l + 1;
        if (active && j0 + 3 >= 1 && j0 <= yLen) tb_store4(tb + tb_word_index(t0 + s4, l, G), tile.v[0], tile.v[1], tile.v[2], tile.v[3]);
      }
    }
  }
  for (int o = 1; o < G; o <<= 1) {
    const double ov = __shfl_xor(bestEnd, o, G);
    const uint32_t oi = __shfl_xor(bestI, o, G);
    if (ov > bestEnd || (ov == bestEnd && oi > bestI)) { bestEnd = ov; bestI = oi; }
  }
  if (active && l == 0) {
    a.units[uid].end_val = bestEnd;
    a.units[uid].end_i = bestI;
  }
}

// Single-diagonal units (the lone diagonal 0, wrong-strand pairs): no neighbours, so ins = del = -inf and
// the match state is a serial chain.  One lane per unit, 8 traceback nibbles per word.
template <bool EMLDS>
__global__ __launch_bounds__(256) void k_viterbi_single(FillArgs a) {
  // EMLDS: match-emission table in LDS (see k_viterbi_fill).  The chain of a lane is serial, so what a round costs is the
  // latency of its eight emission gathers: ~100 cycles from LDS against ~800 from L2.
  extern __shared__ double lds_tab[];
  if (EMLDS) {
    const uint32_t n_em = a.dp.ematch_ninf_off / 8;
    for (uint32_t q = threadIdx.x; q < n_em; q += 256) lds_tab[q] = a.dp.ematch[q];
    __syncthreads();
  }
  const uint32_t uidx = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = uidx < a.n_cls_units;
  uint32_t uid = 0;
  int d = 0, xLen = 0, yLen = 0;
  uint64_t xb = 0, yb = 0, tb_off = 0;
  if (active) {
    uid = a.cls_list[uidx];
    const Unit u = a.units[uid];
    const uint32_t r = u.pair / a.n_refs, x = u.pair % a.n_refs;
    xb = a.ref_off[x]; xLen = (int)(a.ref_off[x + 1] - xb);
    yb = a.read_off[r]; yLen = (int)(a.read_off[r + 1] - yb);
    d = u.dlo; tb_off = u.tb_off;
  }
  int T = active ? yLen : 0;
  for (int o = 32; o; o >>= 1) T = max(T, __shfl_xor(T, o));
  const double* __restrict__ ematch = EMLDS ? lds_tab : a.dp.ematch;
  const double* __restrict__ trans = a.dp.trans;
  const uint32_t Kg = a.dp.Kg;
  const bool local = a.dp.local != 0;
  const uint32_t* __restrict__ ctx = a.ctx + yb;
  uint32_t* __restrict__ tb = a.tb + tb_off;
  double M = QF_NEG_INF, bestEnd = QF_NEG_INF;
  uint32_t bestI = 0, gkPrev = 0;
  // Thirty-two columns per round.  The context words and reference tokens of a round do not depend on the DP chain: they are
  // fetched a whole round (eight 16-byte loads + three token words per lane) ahead of the serial adds, so the global-memory
  // latency is paid once per 32 columns; the emission scores of eight columns at a time come from the table (LDS).
  const uint32_t* __restrict__ xp = a.ref_packed + (active ? a.ref_woff[a.units[uid].pair % a.n_refs] : 0);
  const int nxw = (xLen + 15) / 16 + 2;
  auto xword = [&](int q) -> uint32_t { return xp[min(max(q, 0), nxw - 1)]; };
  U32x4 cur[8], nxt[8];
  uint32_t nx0, nx1, nx2;
  auto fetch = [&](int j0) {   // columns j0 .. j0+31
    const int base = min(j0 - 1, yLen);   // ctx is padded: indices up to yLen + kCtxPad are readable
#pragma unroll
    for (int q = 0; q < 8; ++q) nxt[q] = *(const U32x4*)(ctx + min(base + 4 * q, yLen + 4));
    const int r0 = d + j0 - 1;            // 0-based reference index of row i-1 at column j0
    nx0 = xword(r0 >> 4); nx1 = xword((r0 >> 4) + 1); nx2 = xword((r0 >> 4) + 2);
  };
  fetch(1);
  for (int j0 = 1; j0 <= T; j0 += 32) {
#pragma unroll
    for (int q = 0; q < 8; ++q) cur[q] = nxt[q];
    const int sh = 2 * ((d + j0 - 1) & 15);
    const unsigned long long xlo = (((unsigned long long)nx1 << 32) | nx0) >> sh;   // tokens of columns j0 .. j0+15
    const unsigned long long xhi = (((unsigned long long)nx2 << 32) | nx1) >> sh;   // ... j0+16 .. j0+31
    fetch(j0 + 32);
    U32x4 words4;
#pragma unroll
    for (int g8 = 0; g8 < 4; ++g8) {
      uint32_t w[8];
      double e[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int cc = g8 * 8 + c;
        w[c] = cur[cc >> 2].v[cc & 3];
        const uint32_t tok = (uint32_t)((cc < 16 ? xlo : xhi) >> (2 * (cc & 15))) & 3u;
        e[c] = ematch[(w[c] & 0x7FFFu) * 4u + tok];
      }
      uint32_t word = 0;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int j = j0 + g8 * 8 + c, i = d + j;
        const bool valid = active && j <= yLen && i >= 1 && i <= xLen;
        const uint32_t gk = w[c] >> 24;
        const double tM = (M + trans[j <= 1 ? 0u : gkPrev]) + e[c];
        gkPrev = gk;
        double nm = tM;
        uint32_t sm = 0;
        if (j == 1 && (i == 1 || local) && e[c] > nm) { nm = e[c]; sm = 3; }
        if (!valid) nm = QF_NEG_INF;
        M = nm;
        word |= sm << (4 * c);
        if (j == yLen && valid && (local || i == xLen)) { bestEnd = nm + trans[3 * Kg + gk]; bestI = (uint32_t)i; }
      }
      words4.v[g8] = word;
    }
    // four traceback words (one per eight columns); the unit's words are allocated in whole rounds
#pragma unroll
    for (int g8 = 0; g8 < 4; ++g8)
      if (active && j0 + 8 * g8 <= yLen) tb_store(&tb[((j0 - 1) >> 3) + g8], words4.v[g8]);
  }
  if (active) {
    a.units[uid].end_val = bestEnd;
    a.units[uid].end_i = bestI;
  }
}

// ------------------------------------------------------------------------------------------------
// Row-space Viterbi fill for bands wider than 1024 diagonals (-kmatchoff, or the full-envelope fallback of short
// sequences against long references).  One workgroup of kVitWaves wavefronts per unit; the band's rows are cut into stripes
// of kVitLanes lanes x 8 rows.  Lane L owns 8 consecutive rows and at step t is at column jlo + t - vit_skew(L), so
//   ins(i,j) <- (i,  j-1): own registers (previous step)
//   del(i,j) <- (i-1,j  ): own slot b-1 this step, or lane L-1's last row, which it finished one step ago
//   mat(i,j) <- (i-1,j-1): own slot b-1 before this step's update, or lane L-1's last row two steps ago
// Inside a wavefront the neighbour's values come by shuffle; between wavefronts through an LDS ring: each wavefront lags
// the previous one by kVitLag extra columns, so a workgroup barrier every kVitLag steps is enough.  The last row of a stripe is kept in a global boundary buffer for the next stripe
// (ping-pong).  Same arithmetic, candidate order and 4-bit traceback records as k_viterbi_fill.
// ------------------------------------------------------------------------------------------------
template <bool EMLDS>
__global__ __launch_bounds__(kVitLanes) void k_viterbi_rows(FillArgs a) {
  constexpr int G = kVitLanes, B = 8, S = kVitStripe, NW = kVitWaves;
  // EMLDS: emission tables in LDS (see k_viterbi_fill): the eight emission fetches of a step are gathers
  extern __shared__ double lds_tab[];
  const uint32_t n_em = a.dp.ematch_ninf_off / 8 + 4;
  if (EMLDS) {
    for (uint32_t q = threadIdx.x; q < n_em; q += G) lds_tab[q] = a.dp.ematch[q];
    for (uint32_t q = threadIdx.x; q < kInsRows; q += G) lds_tab[n_em + q] = a.dp.eins[q];
    for (uint32_t q = threadIdx.x; q < 4 * a.dp.Kg; q += G) lds_tab[n_em + kInsRows + q] = a.dp.trans[q];
    __syncthreads();
  }
  constexpr int K = kVitLag, R = 2 * K + 4;
  __shared__ double s_x[NW][R][3];      // [wave][step % R][M, I, D] of the wave's lane 63, last row
  __shared__ double s_best[NW];
  __shared__ uint32_t s_bi[NW];
  const uint32_t uidx = blockIdx.x;
  if (uidx >= a.n_cls_units) return;
  const int L = threadIdx.x, wv = L >> 6, l = L & 63;
  const uint32_t uid = a.cls_list[uidx];
  const Unit u = a.units[uid];
  const uint32_t r = u.pair / a.n_refs, x = u.pair % a.n_refs;
  const uint64_t xb = a.ref_off[x], yb = a.read_off[r];
  const int xLen = (int)(a.ref_off[x + 1] - xb), yLen = (int)(a.read_off[r + 1] - yb);
  const int dlo = u.dlo, dhi = u.dhi;
  const RowGeom g = row_geom(dlo, dhi, xLen, yLen, S);
  uint32_t* base = a.tb + u.tb_off;
  unsigned long long* stripe_off = (unsigned long long*)base;                 // [nStripes+1]
  double* bnd = (double*)(base + 2ull * (g.nStripes + 1));                   // [2][3][yLen+2]
  uint32_t* tbw = base + row_header_words(g, yLen);
  asm volatile("" : "+v"(tbw));
  const size_t bndStride = 3ull * (yLen + 2);
  // stripe offsets (thread 0) and the row-0 boundary (-inf)
  if (L == 0) {
    unsigned long long w = 0;
    for (int s = 0; s < g.nStripes; ++s) {
      int jlo, jhi;
      row_stripe_cols(g, s, dlo, dhi, yLen, jlo, jhi, S);
      stripe_off[s] = w;
      if (jhi >= jlo) w += (unsigned long long)(jhi - jlo + 1 + kVitSkewMax) * G;
    }
    stripe_off[g.nStripes] = w;
  }
  for (size_t c = L; c < bndStride; c += G) bnd[c] = QF_NEG_INF;
  __threadfence();
  __syncthreads();
  const int skew = vit_skew(L);

  // (kept in vector registers: the kernel has fifty of those to spare at two wavefronts per SIMD and no scalar ones -- the
  // compiler was parking scalars in the lanes of a vector register and reading them back ~25 times per step)
  double i2m = a.dp.i2m, d2m = a.dp.d2m, i2i = a.dp.i2i, d2d = a.dp.d2d;
  asm volatile("" : "+v"(i2m), "+v"(d2m), "+v"(i2i), "+v"(d2d));
  const double* __restrict__ ematch = EMLDS ? lds_tab : a.dp.ematch;
  const double* __restrict__ eins = EMLDS ? lds_tab + n_em : a.dp.eins;
  const double* __restrict__ trans = EMLDS ? lds_tab + n_em + kInsRows : a.dp.trans;   // (a step's first cell waits on these)
  const uint32_t Kg = a.dp.Kg;
  const bool local = a.dp.local != 0;
  const uint8_t* __restrict__ xt = a.ref_tok + xb;
  const uint32_t* __restrict__ ctx = a.ctx + yb;
  double bestEnd = QF_NEG_INF;
  uint32_t bestI = 0;
  unsigned long long woff = 0;
  auto ctxword = [&](int j) -> uint32_t { return ctx[min(max(j - 1, -kCtxPad + 1), yLen + 4)]; };   // word of column j

  for (int s = 0; s < g.nStripes; ++s) {
    int jlo, jhi;
    row_stripe_cols(g, s, dlo, dhi, yLen, jlo, jhi, S);
    const int i0 = g.ilo + s * S + L * B;           // this lane's first row
    const double* __restrict__ bprev = bnd + (size_t)(s & 1) * bndStride;      // last row of the previous stripe
    double* __restrict__ bnext = bnd + (size_t)((s + 1) & 1) * bndStride;
    asm volatile("" : "+v"(bprev), "+v"(bnext));                              // (vector registers: see above)
    for (size_t c = L; c < bndStride; c += G) bnext[c] = QF_NEG_INF;
    if (L < NW * R * 3) (&s_x[0][0][0])[L] = QF_NEG_INF;
    __threadfence();
    __syncthreads();
    if (jhi < jlo) continue;  // no cell of this stripe is inside the band: its last row is all -inf
    uint32_t tk[B];
#pragma unroll
    for (int b = 0; b < B; ++b) tk[b] = (i0 + b >= 1 && i0 + b <= xLen) ? xt[i0 + b - 1] : 0u;
    double M[B], I[B], D[B];
#pragma unroll
    for (int b = 0; b < B; ++b) M[b] = I[b] = D[b] = QF_NEG_INF;
    double p1M = QF_NEG_INF, p1I = QF_NEG_INF, p1D = QF_NEG_INF;   // this lane's last row, column of the previous step
    // the row above slot 0 at the previous step = this step's diagonal neighbour (it is lane L-1's last row two steps ago)
    double dgMnext = QF_NEG_INF, dgInext = QF_NEG_INF, dgDnext = QF_NEG_INF;
    if (L == 0) {   // the stripe's first lane starts on column jlo: its first diagonal neighbour is the previous stripe's last row at jlo - 1
      const int jp = min(max(jlo - 1, 0), yLen + 1);
      dgMnext = bprev[jp]; dgInext = bprev[(yLen + 2) + jp]; dgDnext = bprev[2 * (yLen + 2) + jp];
    }
    const int steps = jhi - jlo + 1 + kVitSkewMax;
    // FAST steps (below) of this stripe, for the whole workgroup's lane pattern: a lane is on an inner column with all eight
    // cells inside the band and the matrix for a contiguous range of steps; the wavefront's range is the intersection
    int fastLo, fastHi;
    {
      const bool rowsIn = i0 >= 1 && i0 + B - 1 <= xLen;
      const int jA = max(max(jlo, 2), i0 + B - 1 - dhi), jB = min(min(jhi, yLen - 1), i0 - dlo);   // laneFast <=> jA <= j <= jB
      int lo = rowsIn ? jA - jlo + skew : 1, hi = rowsIn ? jB - jlo + skew : 0;                      // in steps t (j = jlo + t - skew)
      for (int o = 32; o; o >>= 1) { lo = max(lo, __shfl_xor(lo, o)); hi = min(hi, __shfl_xor(hi, o)); }
      fastLo = lo; fastHi = hi;
    }
    // the column's context word is fetched a step ahead (two wavefronts per SIMD do not hide a load per step), and the
    // previous column's word is simply the previous step's
    uint32_t wPrev = ctxword(jlo - skew - 1), wCur = ctxword(jlo - skew);
    // The row above a wavefront's lane 0 does not come by DPP: the stripe's first lane reads the previous stripe's last row from
    // the global boundary buffer, the first lane of the other wavefronts the previous wavefront's ring in LDS.  Both are fetched
    // ONE STEP AHEAD (the boundary row is final before the stripe starts; a ring slot is read K steps after it was written,
    // across a workgroup barrier, instead of K + 1), so that wavefront 0 does not begin every step with a round trip to
    // global memory: 377 -> 366 ms per 256 reads of config 5.  (Fetching the step's eight emissions a step ahead as well cost
    // 24 registers and bought nothing: 373 ms.)
    double nbM = QF_NEG_INF, nbI = QF_NEG_INF, nbD = QF_NEG_INF;
    auto fetch_above = [&](int tt) {   // what lane 0 of this wavefront needs at step tt (column jlo + tt - skew)
      if (wv == 0) {
        const int jc = min(max(jlo + tt - skew, 0), yLen + 1);
        nbM = bprev[jc]; nbI = bprev[(yLen + 2) + jc]; nbD = bprev[2 * (yLen + 2) + jc];
      } else {
        // the previous wavefront's last lane was on that column at step tt - K - 1
        const double* x1 = s_x[wv - 1][(tt + 2 * R - K - 1) % R];
        nbM = x1[0]; nbI = x1[1]; nbD = x1[2];
      }
    };
    if (l == 0) fetch_above(0);
    for (int t = 0; t < steps; ++t) {
      const int j = jlo + t - skew;
      const bool colvalid = j >= jlo && j <= jhi;
      const uint32_t w = wCur;
      const uint32_t wNext = ctxword(j + 1);
      const uint32_t erow4 = (w & 0x7FFFu) * 4u, insrow = (w >> 15) & 0x1FFu, gk = w >> 24;
      const uint32_t gp = j > 1 ? (wPrev >> 24) : 0u;  // yIndelKmer[j-1]; padded 0 for j == 1
      wPrev = w; wCur = wNext;
      const double m2m = trans[gp], m2i = trans[Kg + gp], m2d = trans[2 * Kg + gk];
      const double insE = eins[insrow];
      // row above slot 0: lane L-1's last row (column j one step ago, column j-1 two steps ago): by shuffle inside the
      // wavefront, from the previous wavefront's ring, or (lane 0 of the stripe) from the boundary buffer
      double upM = dpp_from_below<64, false>(p1M), upI = dpp_from_below<64, false>(p1I), upD = dpp_from_below<64, false>(p1D);   // DPP wave shift, not ds_bpermute
      if (l == 0) {
        upM = nbM; upI = nbI; upD = nbD;
        fetch_above(t + 1);
      }
      double dgM = dgMnext, dgI = dgInext, dgD = dgDnext;
      dgMnext = upM; dgInext = upI; dgDnext = upD;
      uint32_t tbword = 0;
      double aboveM = upM, aboveD = upD;
      // FAST step (wave-uniform): every lane is on an inner column (not the first or last read column) and all eight of
      // its cells are inside the band and the matrix: no start candidate, no end tracking, no masking; values by
      // v_max_f64, flags as raw compare bits (see k_viterbi_fill)
      if (t >= fastLo && t <= fastHi) {
        uint32_t acc = 0;
#pragma unroll
        for (int b = 0; b < B; ++b) {
          const double e = ematch[erow4 + tk[b]];
          const double oM = M[b], oI = I[b], oD = D[b];   // (i, j-1)
          const double tM = (dgM + m2m) + e, tI = (dgI + i2m) + e, tD = (dgD + d2m) + e;
          const double m1 = fmax(tM, tI), nm = fmax(m1, tD);
          const double cM = (oM + m2i) + insE, cI = (oI + i2i) + insE;
          const double ni = fmax(cM, cI);
          const double gM = aboveM + m2d, gD = aboveD + d2d;
          const double ndl = fmax(gM, gD);
          acc = shift_in_gt(acc, tI, tM);
          acc = shift_in_gt(acc, tD, m1);
          acc = shift_in_gt(acc, cI, cM);
          acc = shift_in_gt(acc, gD, gM);
          M[b] = nm; I[b] = ni; D[b] = ndl;
          dgM = oM; dgI = oI; dgD = oD;
          aboveM = nm; aboveD = ndl;
        }
        tbword = __builtin_bitreverse32(acc);
      } else {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const int i = i0 + b, dgl = i - j;
        const bool valid = colvalid && i >= 1 && i <= xLen && dgl >= dlo && dgl <= dhi;
        const double e = ematch[erow4 + tk[b]];
        const double oM = M[b], oI = I[b], oD = D[b];   // (i, j-1)
        const double tM = (dgM + m2m) + e, tI = (dgI + i2m) + e, tD = (dgD + d2m) + e;
        double nm = tM;
        uint32_t sm = 0;
        if (tI > nm) { nm = tI; sm = 1; }
        if (tD > nm) { nm = tD; sm = 2; }
        if (j == 1 && (i == 1 || local) && e > nm) { nm = e; sm = 3; }
        const double cM = (oM + m2i) + insE, cI = (oI + i2i) + insE;
        double ni = cM;
        uint32_t si = 0;
        if (cI > ni) { ni = cI; si = 1; }
        const double gM = aboveM + m2d, gD = aboveD + d2d;
        double ndl = gM;
        uint32_t sd = 0;
        if (gD > ndl) { ndl = gD; sd = 1; }
        if (!valid) { nm = QF_NEG_INF; ni = QF_NEG_INF; ndl = QF_NEG_INF; }
        M[b] = nm; I[b] = ni; D[b] = ndl;
        dgM = oM; dgI = oI; dgD = oD;       // (i, j-1) is the diagonal neighbour of row i+1
        aboveM = nm; aboveD = ndl;
        tbword |= (sm | (si << 2) | (sd << 3)) << (4 * b);
        if (j == yLen && valid && (local || i == xLen)) {
          const double ev = nm + trans[3 * Kg + gk];
          if (ev >= bestEnd) { bestEnd = ev; bestI = (uint32_t)i; }
        }
      }
      }
      p1M = M[B - 1]; p1I = I[B - 1]; p1D = D[B - 1];
      if (l == 63) { double* xo = s_x[wv][t % R]; xo[0] = p1M; xo[1] = p1I; xo[2] = p1D; }
      if (colvalid) {
        tb_store(&tbw[woff + (unsigned long long)t * G + L], tbword);
        if (L == G - 1) { bnext[j] = p1M; bnext[(yLen + 2) + j] = p1I; bnext[2 * (yLen + 2) + j] = p1D; }
      }
      // Ring slots are read K / K+1 steps after they are written.  The barrier orders LDS only (LDS operations complete in
      // order: lgkmcnt(0)); __syncthreads() would also wait for every global access in flight -- the traceback store just
      // issued and the loads fetched a step ahead -- i.e. a round trip to HBM every K steps.  The boundary row written to
      // global memory is read by the next stripe, behind the full barrier at the end of this one.
      if (t % K == K - 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __syncthreads();   // everybody is done with the ring and the boundary row before the next stripe resets them
    woff += (unsigned long long)steps * G;
  }
  // end cell of the unit: max value, largest row on ties
  for (int o = 1; o < 64; o <<= 1) {
    const double ov = __shfl_xor(bestEnd, o, 64);
    const uint32_t oi = __shfl_xor(bestI, o, 64);
    if (ov > bestEnd || (ov == bestEnd && oi > bestI)) { bestEnd = ov; bestI = oi; }
  }
  if (l == 0) { s_best[wv] = bestEnd; s_bi[wv] = bestI; }
  __syncthreads();
  if (L == 0) {
    for (int q = 1; q < NW; ++q)
      if (s_best[q] > bestEnd || (s_best[q] == bestEnd && s_bi[q] > bestI)) { bestEnd = s_best[q]; bestI = s_bi[q]; }
    a.units[uid].end_val = bestEnd;
    a.units[uid].end_i = bestI;
  }
}

// ------------------------------------------------------------------------------------------------
// Pair-level result, best reference per read, traceback
// ------------------------------------------------------------------------------------------------
// result = max over the pair's bands; end cell = largest row among the maxima (the reference scans rows
// downwards replacing only on strict '>', src/qmodel.cpp:1565-1575).
__global__ void k_finalize_pairs(FinalArgs a) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.n_pairs) return;
  double best = QF_NEG_INF;
  uint32_t bu = kNoUnit, bi = 0;
  for (uint32_t uid = a.pair_head[p]; uid != kNoUnit; uid = a.units[uid].next) {
    const double v = a.units[uid].end_val;
    const uint32_t ei = a.units[uid].end_i;
    if (v > best || (v == best && v > QF_NEG_INF && ei > bi)) { best = v; bu = uid; bi = ei; }
  }
  a.pair_score[p] = best;
  a.pair_end_unit[p] = bu;
}

// QuaffAlignmentTask::run, src/qmodel.cpp:2764-2778: keep the best score-adjusted alignment per read,
// the earlier reference on ties; or every finite one (-printall).
__global__ void k_select(FinalArgs a) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= a.n_reads) return;
  const double nll = a.nll[r];
  const uint32_t yLen = (uint32_t)(a.read_off[r + 1] - a.read_off[r]);
  double bestAdj = QF_NEG_INF;
  uint32_t bestX = kNoUnit;
  for (uint32_t x = 0; x < a.n_refs; ++x) {
    const uint32_t p = r * a.n_refs + x;
    const double v = a.pair_score[p];
    if (!(v > QF_NEG_INF)) continue;
    const double adj = v - nll;
    if (a.all) {
      if (!(adj >= a.min_score)) continue;
      const Unit& u = a.units[a.pair_end_unit[p]];
      const uint32_t cap = 2 * yLen + (uint32_t)(u.dhi - u.dlo + 1) + 4;
      const uint32_t idx = atomicAdd(&a.bc->n_align, 1u);
      AlignRec rec{r, x, a.pair_end_unit[p], 0, v, adj, atomicAdd(&a.bc->n_runs, (unsigned long long)cap), 0, 0, 0, 0};
      a.recs[idx] = rec;
    } else if (bestX == kNoUnit || adj > bestAdj) {
      bestAdj = adj;
      bestX = x;
    }
  }
  if (!a.all && bestX != kNoUnit && !(bestAdj >= a.min_score)) bestX = kNoUnit;   // the read's best alignment is below the threshold
  if (!a.all && bestX != kNoUnit) {
    const uint32_t p = r * a.n_refs + bestX;
    const Unit& u = a.units[a.pair_end_unit[p]];
    const uint32_t cap = 2 * yLen + (uint32_t)(u.dhi - u.dlo + 1) + 4;
    const uint32_t idx = atomicAdd(&a.bc->n_align, 1u);
    AlignRec rec{r, bestX, a.pair_end_unit[p], 0, a.pair_score[p], bestAdj,
                 atomicAdd(&a.bc->n_runs, (unsigned long long)cap), 0, 0, 0, 0};
    a.recs[a.dense ? r : idx] = rec;   // dense: record r is read r's, so the results need no ordering on the host
  } else if (!a.all && a.dense) {
    AlignRec rec{};
    rec.read = r;
    rec.unit = kNoUnit;
    a.recs[r] = rec;
    AlignOut o{};
    o.read = r + a.read_base;
    o.n_runs = kAlignHole;
    a.out_align[r] = o;
  }
}

// QuaffViterbiMatrix::alignment, src/qmodel.cpp:1576-1622, replayed from the packed traceback bits.
// One thread per alignment; CIGAR runs are produced end-to-start, then written start-to-end.
// One lane per alignment.  A wavefront advances at the pace of its slowest lane, and a lane that has to fetch a traceback
// word from HBM stalls all 64, so the words are fetched for everybody at a fixed cadence instead: every 8 moves each lane
// loads the 3 x 8 words around its position (its own lane-of-the-fill and both neighbours, this step and the seven
// before: a path cannot leave that window in 8 moves) as 24 independent loads and parks them in LDS.  The position is
// tracked as (fill lane, slot, step) incrementally: no divisions on the path.
#ifndef QF_TB_LANES
#define QF_TB_LANES 32
#endif
constexpr uint32_t kTbLanes = QF_TB_LANES;  // alignments per wavefront: few, so that many wavefronts interleave on a SIMD
constexpr int kTbStride = 49;  // 3 lanes x 2 tiles x 8 words per lane in LDS (odd stride: conflict-free)
// WINDOWED = true handles the alignments of the one-word diagonal classes (everything a banded run produces), false the
// rest (single diagonals, row-space units, two-word classes) with one dependent load per move; each alignment is taken
// by exactly one of the two launches.
template <bool WINDOWED>
__global__ __launch_bounds__(kTbLanes) void k_traceback(FinalArgs a) {
  __shared__ uint32_t s_win[WINDOWED ? kTbLanes * kTbStride : 1];
  const uint32_t idx = blockIdx.x * kTbLanes + threadIdx.x;
  bool live = idx < a.n_recs;
  AlignRec rec{};
  Unit u{};
  uint32_t yLen = 0;
  int xLenR = 0;
  if (live) {
    rec = a.recs[idx];
    live = rec.unit != kNoUnit;   // dense mode: a read without any alignment
  }
  if (live) {
    u = a.units[rec.unit];
    const bool win_cls = u.cls != 0 && u.cls != (uint32_t)kRowClass && fill_class((int)u.cls).B <= 8;
    live = win_cls == WINDOWED;
  }
  if (!WINDOWED && !live) return;
  if (live) {
    yLen = (uint32_t)(a.read_off[rec.read + 1] - a.read_off[rec.read]);
    xLenR = (int)(a.ref_off[rec.ref + 1] - a.ref_off[rec.ref]);
  }
  const FillClass fc = fill_class((int)u.cls);
  const int G = fc.G, B = fc.B;
  const uint32_t* __restrict__ tb = a.tb + u.tb_off;
  uint32_t* tmp = a.runs_tmp + rec.tmp_off;
  int i = (int)u.end_i, j = (int)yLen;
  const uint32_t xEnd = u.end_i;
  uint32_t n = 0, ncol = 0, curOp = 3, curLen = 0;
  int state = live ? 1 : 0;  // 0 Start, 1 Match, 2 Insert, 3 Delete
  auto walking = [&]() { return state != 0 && i >= 0 && j >= 0 && (i > 0 || j > 0); };
  // one move, branch-free (the lanes of a wavefront are in different states): M steps to (i-1,j-1) on the same diagonal,
  // I to (i,j-1) on diagonal d+1, D to (i-1,j) on diagonal d-1.  Match source 3 = Start on column 1 (the only place a start
  // candidate exists); elsewhere the trimmed fill's raw compare bits "I>M and D>max(M,I)", i.e. D.
  auto step = [&](uint32_t nib, int& db) {
    const bool isM = state == 1, isI = state == 2;
    const uint32_t op = isM ? 0u : isI ? 1u : 2u;
    i -= isI ? 0 : 1;
    j -= (isM || isI) ? 1 : 0;
    db = isM ? 0 : isI ? 1 : -1;
    const uint32_t s = nib & 3u;
    const int fromM = s == 0 ? 1 : s == 1 ? 2 : (s == 2 || j > 0) ? 3 : 0;
    const int fromI = (nib >> 2) & 1u ? 2 : 1, fromD = (nib >> 3) & 1u ? 3 : 1;
    state = isM ? fromM : isI ? fromI : fromD;
    ++ncol;
    if (op == curOp) ++curLen;
    else {
      if (curLen) tmp[n++] = (curLen << 2) | curOp;
      curOp = op; curLen = 1;
    }
  };
  if (WINDOWED) {
    uint32_t* win = s_win + threadIdx.x * kTbStride;
    // position inside the unit: fill lane l, slot b; the word of cell (i,j) is tb[tb_word_index(j-1+l, l, G)]
    int l = 0, b = 0, cl = 0, ct = 0;
    if (live) { const int dd = (i - j) - u.dlo; l = dd / B; b = dd % B; }
    while (__builtin_amdgcn_ballot_w64(walking())) {
      if (walking() && i >= 1 && j >= 1) {
        cl = l; ct = (j - 1 + l) >> 3;   // window: tiles ct, ct-1 (16 steps) of fill lanes cl-1, cl, cl+1
#pragma unroll
        for (int dl = 0; dl < 3; ++dl) {
          const int ll = min(max(cl - 1 + dl, 0), G - 1);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const uint32_t* src = tb + ((uint64_t)max(ct - q, 0) * G + ll) * 8;
            const U32x4 v0 = *(const U32x4*)src, v1 = *(const U32x4*)(src + 4);
            uint32_t* dst = win + (dl * 2 + q) * 8;
            dst[0] = v0.v[0]; dst[1] = v0.v[1]; dst[2] = v0.v[2]; dst[3] = v0.v[3];
            dst[4] = v1.v[0]; dst[5] = v1.v[1]; dst[6] = v1.v[2]; dst[7] = v1.v[3];
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      // eight moves from the window.  A path that leaves the window early (8 same-direction indels in a row) just waits for
      // the next refill, so the loop has no global-memory fallback; the window index of (lane l, step t) is
      // ((l - cl + 1) * 2 + ct - t/8) * 8 + t%8  =  16 l + kw - t + 2 (t % 8)
      const int kw = (1 - cl) * 16 + 8 * ct;
#pragma unroll
      for (int mv = 0; mv < 8; ++mv) {
        const int t = j - 1 + l;
        const bool cell = i >= 1 && j >= 1;
        const bool inwin = (unsigned)(l - cl + 1) < 3u && (unsigned)(ct - (t >> 3)) < 2u;
        const int widx = (cell && inwin) ? 16 * l + kw - t + 2 * (t & 7) : 0;
        const uint32_t wv = win[widx];
        if (walking() && (inwin || !cell)) {
          // A match run: if every remaining step of this tile (t, t-1, ... down to the tile's first) came from the match
          // state, take them all at once (the words of one fill lane's tile are consecutive in the window).
          bool skipped = false;
          if (state == 1 && cell && inwin) {
            const int tt = t & 7;
            uint32_t orw = 0;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
              const uint32_t wm = win[widx - tt + m];
              orw |= m <= tt ? wm : 0u;
            }
            const int kk = tt + 1;
            if (kk > 1 && ((orw >> (4 * b)) & 3u) == 0u && kk <= i && kk <= j) {
              i -= kk; j -= kk; ncol += (uint32_t)kk;
              if (curOp == 0u) curLen += (uint32_t)kk;
              else {
                if (curLen) tmp[n++] = (curLen << 2) | curOp;
                curOp = 0u; curLen = (uint32_t)kk;
              }
              skipped = true;
            }
          }
          if (!skipped) {
            int db;
            step(cell ? (wv >> (4 * b)) & 0xFu : 0u, db);
            b += db;
            if (b == B) { b = 0; ++l; }
            if (b < 0) { b = B - 1; --l; }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  } else {
    const RowGeom rg = u.cls == (uint32_t)kRowClass ? row_geom(u.dlo, u.dhi, xLenR, (int)yLen, kVitStripe) : RowGeom{0, 0, 0};
    while (walking()) {
      uint32_t nib = 0;
      if (i >= 1 && j >= 1) {
        if (u.cls == 0) nib = (tb[(j - 1) >> 3] >> (4 * ((j - 1) & 7))) & 0xFu;
        else if (u.cls == (uint32_t)kRowClass) {
          const int rr = i - rg.ilo, s = rr / kVitStripe, li = (rr % kVitStripe) / 8, b = rr % 8;
          int jlo, jhi;
          row_stripe_cols(rg, s, u.dlo, u.dhi, (int)yLen, jlo, jhi, kVitStripe);
          const unsigned long long* so = (const unsigned long long*)tb;
          const uint32_t* words = tb + row_header_words(rg, (int)yLen);
          nib = (words[so[s] + (unsigned long long)(j - jlo + vit_skew(li)) * kVitLanes + li] >> (4 * b)) & 0xFu;
        } else {
          const int dd = (i - j) - u.dlo, l = dd / B, b = dd % B, t = j - 1 + l;
          nib = (tb[((uint64_t)t * G + l) * 2 + (b >> 3)] >> (4 * (b & 7))) & 0xFu;
        }
      }
      int db;
      step(nib, db);
    }
  }
  if (!live) return;
  if (curLen) tmp[n++] = (curLen << 2) | curOp;
  const unsigned long long off = atomicAdd(&a.bc->total_runs_out, (unsigned long long)n);
  uint32_t c = 0;
  for (; c + 4 <= n; c += 4) {   // reversed copy, four independent loads at a time
    const uint32_t r0 = tmp[n - 1 - c], r1 = tmp[n - 2 - c], r2 = tmp[n - 3 - c], r3 = tmp[n - 4 - c];
    a.runs_out[off + c] = r0; a.runs_out[off + c + 1] = r1; a.runs_out[off + c + 2] = r2; a.runs_out[off + c + 3] = r3;
  }
  for (; c < n; ++c) a.runs_out[off + c] = tmp[n - 1 - c];
  rec.x_start = (uint32_t)(i + 1);
  rec.x_end = xEnd;
  rec.n_columns = ncol;
  rec.n_runs = n;
  rec.run_off = off;
  rec.ok = state == 0;
  a.recs[idx] = rec;
  if (!rec.ok) atomicOr(&a.bc->error, 16u);   // the traceback did not reach the start state
  if (a.out_align) {
    AlignOut o;
    o.read = rec.read + a.read_base; o.ref = rec.ref;
    o.viterbi = rec.viterbi; o.score = rec.score;
    o.x_start = rec.x_start; o.x_end = rec.x_end; o.n_columns = ncol; o.n_runs = n;
    o.run_offset = off;
    a.out_align[idx] = o;
  }
}

// ------------------------------------------------------------------------------------------------
// launch helpers (called from qf_api.cpp, which contains no device code)
// ------------------------------------------------------------------------------------------------
template <int G, int B>
static void launch_fill_gb(const FillArgs& a, bool gapctx, hipStream_t s) {
  const uint32_t upw = 64 / G, waves = (a.n_cls_units + upw - 1) / upw, blocks = (waves + 3) / 4;
  {
    // emission tables in LDS when three workgroups' copies fit a CU's 160 KB (match contexts of up to 2 bases)
    const uint32_t lds_bytes = a.dp.ematch_ninf_off + 32 + kInsRows * 8;
    if (lds_bytes <= 52 * 1024 && !a.no_lds_tables) {
      if (gapctx) hipLaunchKernelGGL((k_viterbi_fill<G, B, true, true>), dim3(blocks), dim3(256), lds_bytes, s, a);
      else hipLaunchKernelGGL((k_viterbi_fill<G, B, false, true>), dim3(blocks), dim3(256), lds_bytes, s, a);
    } else {
      if (gapctx) hipLaunchKernelGGL((k_viterbi_fill<G, B, true, false>), dim3(blocks), dim3(256), 0, s, a);
      else hipLaunchKernelGGL((k_viterbi_fill<G, B, false, false>), dim3(blocks), dim3(256), 0, s, a);
    }
  }
}

void launch_viterbi_fill(int cls, const FillArgs& a, bool gapctx, hipStream_t s) {
  if (a.n_cls_units == 0) return;
  switch (cls) {
    case 0: {
      const uint32_t lds_bytes = a.dp.ematch_ninf_off;
      if (lds_bytes <= 52 * 1024 && !a.no_lds_tables)
        hipLaunchKernelGGL(k_viterbi_single<true>, dim3((a.n_cls_units + 255) / 256), dim3(256), lds_bytes, s, a);
      else hipLaunchKernelGGL(k_viterbi_single<false>, dim3((a.n_cls_units + 255) / 256), dim3(256), 0, s, a);
      break;
    }
    case 1: launch_fill_gb<16, 2>(a, gapctx, s); break;
    case 2: launch_fill_gb<16, 3>(a, gapctx, s); break;
    case 3: launch_fill_gb<16, 4>(a, gapctx, s); break;
    case 4: launch_fill_gb<16, 5>(a, gapctx, s); break;
    case 5: launch_fill_gb<16, 6>(a, gapctx, s); break;
    case 6: launch_fill_gb<16, 8>(a, gapctx, s); break;
    case 7: launch_fill_gb<64, 3>(a, gapctx, s); break;
    case 8: launch_fill_gb<64, 4>(a, gapctx, s); break;
    case 9: launch_fill_gb<64, 6>(a, gapctx, s); break;
    case 10: launch_fill_gb<64, 8>(a, gapctx, s); break;
    case 11: launch_fill_gb<64, 12>(a, gapctx, s); break;
    case 12: launch_fill_gb<64, 16>(a, gapctx, s); break;
    case 13: {
      const uint32_t lds_bytes = a.dp.ematch_ninf_off + 32 + kInsRows * 8 + 4 * a.dp.Kg * 8;
      if (lds_bytes <= 52 * 1024 && !a.no_lds_tables) hipLaunchKernelGGL(k_viterbi_rows<true>, dim3(a.n_cls_units), dim3(kVitLanes), lds_bytes, s, a);
      else hipLaunchKernelGGL(k_viterbi_rows<false>, dim3(a.n_cls_units), dim3(kVitLanes), 0, s, a);
      break;
    }
  }
}

// Workgroups of the row-space Viterbi kernel the device holds at once (one band each): a batch of bands that is not a multiple of
// this leaves the last round of workgroups on a partly idle chip (624 bands on 512 slots run as long as 1024).
uint32_t viterbi_rows_resident_workgroups(const FillArgs& a) {
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  const uint32_t lds_bytes = a.dp.ematch_ninf_off + 32 + kInsRows * 8 + 4 * a.dp.Kg * 8;
  const bool lds = lds_bytes <= 52 * 1024 && !a.no_lds_tables;
  const hipError_t e = lds ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_viterbi_rows<true>, kVitLanes, lds_bytes)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_viterbi_rows<false>, kVitLanes, 0);
  if (e != hipSuccess || per_cu <= 0 || cus <= 0) { (void)hipGetLastError(); return 0; }
  return (uint32_t)(per_cu * cus);
}

// The chip's fp64 vector issue rate as it is, not as specified: eight independent v_add_f64 chains per wavefront, four wavefronts
// per SIMD, every CU busy (tools/dev/valu_rate_bench.hip has the other instructions).  The Viterbi roofline's 39.3 T op/s assumes
// one fp64 wave-instruction per 4 clocks at 2.4 GHz; measured, the instruction takes ~5 clocks and the clock under load is ~2.1 GHz.
__global__ __launch_bounds__(256) void k_f64_add_rate(double* out, int iters) {
  double d[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) d[c] = 1.0 + threadIdx.x * 1e-3 + c;
  const double k = 1.0000001;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 8; ++c) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(k));
  }
  double acc = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) acc += d[c];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// lane-operations per second (wave-instructions x 64), or 0 on error
double measure_f64_add_rate(double* d_out /* >= 256 K doubles */, hipStream_t s) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
  const int blocks = std::min(cus * 4, 1024), iters = 40000;   // 4 workgroups of 4 wavefronts per CU
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 0;
  hipLaunchKernelGGL(k_f64_add_rate, dim3(blocks), dim3(256), 0, s, d_out, 2000);   // clocks ramp up
  (void)hipEventRecord(e0, s);
  hipLaunchKernelGGL(k_f64_add_rate, dim3(blocks), dim3(256), 0, s, d_out, iters);
  (void)hipEventRecord(e1, s);
  float ms = 0;
  const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return ok ? (double)blocks * 256.0 * iters * 8.0 / (ms * 1e-3) : 0.0;
}

__global__ void k_qual_range(const char* __restrict__ qual, uint64_t total, uint32_t* out) {
  uint32_t lo = 0xFFFFFFFFu, hi = 0;
  auto take = [&](uint32_t c) {
    const uint32_t q = (uint32_t)max(0, min(kNQualDev - 1, (int)(signed char)c - '!'));
    lo = min(lo, q); hi = max(hi, q);
  };
  // 16 characters per lane and load (the buffer is a hipMalloc allocation: 16-byte aligned, and reserved 16 bytes past `total`)
  const uint64_t n16 = total / 16;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint4 v = ((const uint4*)qual)[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int b = 0; b < 4; ++b) take((w[k] >> (8 * b)) & 0xFFu);
  }
  if (blockIdx.x == 0 && threadIdx.x < total - n16 * 16) take((uint32_t)(unsigned char)qual[n16 * 16 + threadIdx.x]);
  for (int o = 32; o; o >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, o)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, o)); }
  if ((threadIdx.x & 63) == 0 && lo != 0xFFFFFFFFu) { atomicMin(out, lo); atomicMax(out + 1, hi); }
}
void launch_qual_range(const char* qual, uint64_t total, uint32_t* out, hipStream_t s) {
  if (!total) return;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>(2048, (total / 16 + 255) / 256 + 1);
  hipLaunchKernelGGL(k_qual_range, dim3(blocks), dim3(256), 0, s, qual, total, out);
}

void launch_prep_ref(const char* seq, uint64_t total, uint8_t* tok, BatchCounters* bc, hipStream_t s) {
  if (!total) return;
  hipLaunchKernelGGL(k_prep_ref, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, seq, total, tok, bc);
}
void launch_pack_ref(const uint8_t* tok, const uint64_t* off, const uint64_t* woff, uint32_t n_refs, uint64_t max_len,
                     uint32_t* packed, hipStream_t s) {
  const uint64_t nw = (max_len + 15) / 16 + 2;
  hipLaunchKernelGGL(k_pack_ref, dim3((uint32_t)((nw + 255) / 256), n_refs), dim3(256), 0, s, tok, off, woff, n_refs, packed);
}
void launch_ref_index(const uint8_t* tok, const uint64_t* off, uint32_t n_refs, uint64_t max_len, uint32_t k,
                      uint32_t nbuckets, uint32_t* starts, uint32_t* cursor, uint32_t* pos, hipStream_t s) {
  const dim3 grid((uint32_t)((max_len + 255) / 256), n_refs);
  hipLaunchKernelGGL(k_ref_kmer_count, grid, dim3(256), 0, s, tok, off, k, nbuckets, starts);
  hipLaunchKernelGGL(k_bucket_scan, dim3(n_refs), dim3(1024), 0, s, starts, nbuckets, 1u);
  hipLaunchKernelGGL(k_ref_kmer_scatter, grid, dim3(256), 0, s, tok, off, k, nbuckets, starts, cursor, pos);
}
// estride = 0: a chunk's entries follow the chunks before it; > 0 (k_seed_rows_lds): buckets padded to even length (the pad
// entry stays 0xFFFFFFFF: the caller fills the array with it first), `estride` entries per chunk
void launch_chunk_index(const uint8_t* tok, const uint64_t* off, uint32_t n_seqs, uint64_t max_len, uint32_t k, uint32_t nbuckets,
                        int chunk_log2, uint32_t* starts, uint32_t* cursor, uint32_t* entries, uint64_t estride, uint32_t wd, int pb, uint2* bounds, hipStream_t s) {
  const dim3 grid((uint32_t)((max_len + 255) / 256), n_seqs);
  const uint32_t n_chunks = (n_seqs + (1u << chunk_log2) - 1) >> chunk_log2;
  hipLaunchKernelGGL(k_chunk_kmer_count, grid, dim3(256), 0, s, tok, off, k, nbuckets, chunk_log2, starts);
  const bool e16 = !estride && pb;
  hipLaunchKernelGGL(k_bucket_scan, dim3(n_chunks), dim3(1024), 0, s, starts, nbuckets, estride || e16 ? 2u : 1u);
  hipLaunchKernelGGL(k_chunk_kmer_scatter, grid, dim3(256), 0, s, tok, off, k, nbuckets, chunk_log2, starts, cursor, entries, estride, estride ? 0u : wd, estride ? 0 : pb);
  if (e16) hipLaunchKernelGGL(k_chunk_bounds, dim3((nbuckets + 255) / 256, n_chunks), dim3(256), 0, s, starts, cursor, nbuckets, bounds);
}
void launch_prep_reads(const PrepArgs& a, uint32_t n_reads, hipStream_t s) {
  if (!n_reads) return;
  if (a.skmer64) hipLaunchKernelGGL(k_prep_reads2<true>, dim3(n_reads), dim3(64), 0, s, a);
  else hipLaunchKernelGGL(k_prep_reads2<false>, dim3(n_reads), dim3(64), 0, s, a);
}
void launch_null_ll(const PrepArgs& a, uint32_t n_reads, hipStream_t s) {
  if (n_reads) hipLaunchKernelGGL(k_null_ll, dim3((n_reads + 63) / 64), dim3(64), 0, s, a, n_reads);
}
// A diagonal holds at most min(xLen, yLen) - k + 1 matches: 16-bit counters are exact below 65 536 of them.
bool seed_needs_deep_counters(const SeedArgs& a) {
  const uint32_t m = a.max_ref_len < a.max_read_len ? a.max_ref_len : a.max_read_len;
  return a.sparse && (a.max_ref_len == 0 ? a.max_read_len : m) >= 65535u + (uint32_t)a.kmer_len;
}
size_t seed_lds_bytes(int max_nd, bool mem, bool deep) {
  const size_t hist = deep ? (size_t)max_nd * 4 : (size_t)((max_nd + 1) / 2) * 4;
  return mem ? hist + (size_t)((max_nd + 3) & ~1) * 2 + (size_t)(max_nd + 4) * 2 : hist + (size_t)max_nd + 4;
}
// Coarse bin width of the wavefront seeding (log2): the widest of 32 / 16 / 8 diagonals whose chance k-mer matches between
// unrelated sequences (min(len) / 4^k per diagonal, Poisson) stay four standard deviations below the threshold.
static int seed_coarse_bits(const SeedArgs& a) {
  const double len = a.max_ref_len ? (double)(a.max_ref_len < a.max_read_len ? a.max_ref_len : a.max_read_len) : (double)a.max_read_len;
  const double per_diag = a.kmer_len < 32 ? len / (double)(1ull << (2 * a.kmer_len)) : 0.0;
  for (int cb = 5; cb > 3; --cb) {
    const double lam = per_diag * (1 << cb);
    if (lam + 4.0 * sqrt(lam) + 1.0 <= (double)a.threshold) return cb;
  }
  return 3;
}
// LDS words one wavefront of the wavefront seeding needs; `wide` = 32-bit coarse counters (a bin could pass 65 535)
static uint32_t seed_wave_words(const SeedArgs& a, int cb, bool& wide) {
  const uint32_t nc = (uint32_t)((a.max_nd + (1 << cb) - 1) >> cb), nb = (uint32_t)((a.max_nd + 31) / 32);
  wide = ((uint64_t)a.max_read_len << cb) >= 65280u;
  return (wide ? nc + 2 : (nc + 1) / 2 + 1) + (nb + 1) + 32 * 16 + 40;
}
// The prefilter's coarse bins are half as wide as the per-pair kernel's: k-mer matches come in runs (a chance 8-mer is three
// of them on one diagonal), and with bins of 8 diagonals 17 % of unrelated 2 kb read pairs still had a bin at the threshold.
static int seed_row_bits(const SeedArgs& a) { return seed_coarse_bits(a) - 1; }
// k_seed_rows<., E16>: bytes of 16-bit counters per sequence when a sequence's counters span 2^(pb + 1) diagonals
size_t seed_row_stride_bytes_e16(int pb, int cb) { return ((size_t)2 << (pb + 1)) >> cb; }
int seed_row_bits_of(const SeedArgs& a) { return seed_row_bits(a); }
// diagonals the counters of one sequence of a chunk span in k_seed_rows (its index entries are sequence x this + position)
uint32_t seed_row_entry_span(const SeedArgs& a) { return (uint32_t)(seed_row_stride_bytes(a) / 4 * 2) << seed_row_bits(a); }
size_t seed_row_stride_bytes(const SeedArgs& a) {
  if (!a.sparse || a.threshold < 0 || !a.nbuckets || a.ref_skeys || a.dump_cover || a.force_block_kernel || seed_needs_deep_counters(a)) return 0;
  const int cb = seed_row_bits(a);
  if (((uint64_t)a.max_read_len << cb) >= 65280u || a.max_read_len >= (1u << 26)) return 0;   // 16-bit coarse counters, 26-bit positions
  return (size_t)((((a.max_nd + (1 << cb) - 1) >> cb) + 1) / 2 + 1) * 4;
}
// k_seed_rows_lds: bytes of 32-bit coarse counters per sequence of a chunk (0: the prefilter does not apply)
size_t seed_rows_lds_stride_bytes(const SeedArgs& a) {
  if (!seed_row_stride_bytes(a) || seed_row_bits(a) < 2) return 0;
  const int cb = seed_row_bits(a);
  return (((size_t)(((a.max_nd + (1 << cb) - 1) >> cb) + 1) + 3) & ~(size_t)3) * 4;   // (whole 16-byte groups: the scan reads four words at a time)
}
// LDS of one k_seed_rows_lds workgroup for chunks of 2^cl sequences holding at most max_entries k-mer positions (every bucket
// padded to even: + one entry per bucket); *e16: entries fit 16 bits (all the chunk's counters, in the entries' address units,
// below 65 536).  0: the kernel does not apply.
size_t seed_rows_lds_fit(const SeedArgs& a, int cl, uint64_t max_entries, bool* e16) {
  const size_t stride = seed_rows_lds_stride_bytes(a);
  if (!stride) return 0;
  *e16 = (((uint64_t)stride << (seed_row_bits(a) - 2)) << cl) < 65536;
  return seed_rows_lds_bytes(a, stride, cl, max_entries, *e16);
}
size_t seed_rows_lds_bytes(const SeedArgs& a, size_t stride, int cl, uint64_t max_entries, bool e16) {
  return (stride << cl) + seed_rows_dummy_bytes(a.max_read_len, seed_row_bits(a)) + ((size_t)a.nbuckets + 1 + ((size_t)1 << cl)) * 4 +
         ((size_t)max_entries + a.nbuckets + 8) * (e16 ? 2 : 4) + 16;
}
bool seed_needs_workspace(const SeedArgs& a, bool mem) {
  if (!a.sparse) return false;
  if (seed_needs_deep_counters(a)) return true;
  if (!mem && a.threshold >= 0 && !a.force_block_kernel) {
    bool wide;
    const uint32_t words = seed_wave_words(a, seed_coarse_bits(a), wide);
    if ((size_t)words * 4 * 4 <= 150 * 1024) return false;
  }
  return seed_lds_bytes(a.max_nd, mem, false) > 150 * 1024;
}
namespace {
template <class F>
void with_seed_variant(bool wide, int cb, F&& f) {   // f(std::bool_constant<WIDE>, std::integral_constant<int, CB>)
  auto pick = [&](auto w) {
    if (cb == 5) f(w, std::integral_constant<int, 5>());
    else if (cb == 4) f(w, std::integral_constant<int, 4>());
    else f(w, std::integral_constant<int, 3>());
  };
  if (wide) pick(std::true_type()); else pick(std::false_type());
}
}  // namespace
int launch_seed(const SeedArgs& a_in, uint32_t n_pairs, bool mem, hipStream_t s) {
  if (!n_pairs) return 0;
  SeedArgs a = a_in;
  if (a.row_pieces4 && a.n_row_pieces && !mem) {   // the same with the chunk's index in LDS, one workgroup per chunk and piece of x rows
    const size_t stride = seed_rows_lds_stride_bytes(a);
    const size_t lds = seed_rows_lds_bytes(a, stride, a.chunk_log2, a.row_max_entries, a.row_e16 != 0);
    if (stride && lds <= kSeedRowLdsBig) {
      const int cb = seed_row_bits(a);
      auto go = [&](auto fn) {
        (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(fn, dim3(a.n_row_pieces), dim3(kSeedRowThreads), lds, s, a, (uint32_t)(stride / 4));
      };
      if (a.row_e16) { if (cb == 4) go(k_seed_rows_lds<4, uint16_t>); else if (cb == 3) go(k_seed_rows_lds<3, uint16_t>); else go(k_seed_rows_lds<2, uint16_t>); }
      else { if (cb == 4) go(k_seed_rows_lds<4, uint32_t>); else if (cb == 3) go(k_seed_rows_lds<3, uint32_t>); else go(k_seed_rows_lds<2, uint32_t>); }
      static_assert(sizeof(RowItemL) == 32, "host and device agree on the item layout");
      a.pair_skip = a.row_skip;
    }
  } else if (a.row_items && a.n_row_items && !mem) {   // settle the pairs with nothing but the forced diagonal a chunk of y at a time
    const int cb = seed_row_bits(a);
    const size_t stride = a.chunk_pb ? seed_row_stride_bytes_e16(a.chunk_pb, cb) : seed_row_stride_bytes(a);
    const size_t lds = stride << a.chunk_log2;
    if (seed_row_stride_bytes(a) && lds <= kSeedRowLdsMax) {
      auto go = [&](auto fn) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(fn, dim3(a.n_row_items), dim3(kSeedRowThreads), lds, s, a, (uint32_t)(stride / 4));
      };
      if (a.chunk_pb) { if (cb == 4) go(k_seed_rows<4, true>); else if (cb == 3) go(k_seed_rows<3, true>); else go(k_seed_rows<2, true>); }
      else { if (cb == 4) go(k_seed_rows<4, false>); else if (cb == 3) go(k_seed_rows<3, false>); else go(k_seed_rows<2, false>); }
      a.pair_skip = a.row_skip;
    }
  }
  if (seed_needs_deep_counters(a)) {  // 65 536+ matches on one diagonal are possible: 32-bit counters, global workspaces
    if (!a.ws || !a.ws_slots || a.ws_words * 4 < seed_lds_bytes(a.max_nd, mem, true)) return -1;
    const uint32_t grid = n_pairs < a.ws_slots ? n_pairs : a.ws_slots;
    if (mem) hipLaunchKernelGGL((k_seed_global<true, true>), dim3(grid), dim3(256), 0, s, a, n_pairs);
    else hipLaunchKernelGGL((k_seed_global<false, true>), dim3(grid), dim3(256), 0, s, a, n_pairs);
    return 0;
  }
  if (!mem && a.sparse && a.threshold >= 0 && !a.force_block_kernel) {
    // one wavefront per pair; LDS per wave: coarse counters + bitmap + fine counters
    const int cb = seed_coarse_bits(a);
    bool wide;
    const uint32_t words = seed_wave_words(a, cb, wide);
    const size_t lds = (size_t)words * 4 * 4;
    const bool lds_index_ok = !a.ref_skeys && a.nbuckets && a.max_ref_len && a.max_ref_len + 4 < 65536 && !a.dump_cover && !a.no_lds_index;
    const uint32_t idx_words = (((a.nbuckets + 2) & ~1u) + ((a.max_ref_len + 5) & ~1u)) / 2;
    const size_t lds2 = ((size_t)idx_words + (size_t)words * 8) * 4;
    const bool lds_index_fits = (size_t)idx_words * 4 <= 48 * 1024 && lds2 <= 80 * 1024;
    // LDS-resident index: bucket-indexed (k <= 8), 16-bit positions, <= 48 KB; implicit read x ref pair order ...
    if (!a.pair_x && lds_index_ok && lds_index_fits && n_pairs % a.n_refs == 0) {
      const uint32_t n_reads = n_pairs / a.n_refs;
      const uint32_t blocks = ((n_reads + kSeedReadsPerBlock - 1) / kSeedReadsPerBlock) * a.n_refs;
      with_seed_variant(wide, cb, [&](auto w, auto c) {
        auto fn = k_seed_wave_lds<decltype(w)::value, decltype(c)::value>;
        if (lds2 > 48 * 1024) (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(512), lds2, s, a, n_reads, words, idx_words);
      });
      return 0;
    }
    // ... or an explicit pair list (overlap): the shared x's index in LDS
    if (a.pair_x && lds_index_ok && lds_index_fits) {
      const uint32_t blocks = (n_pairs + kSeedPairsPerBlock - 1) / kSeedPairsPerBlock;
      with_seed_variant(wide, cb, [&](auto w, auto c) {
        auto fn = k_seed_wave_lds_pairs<decltype(w)::value, decltype(c)::value>;
        if (lds2 > 48 * 1024) (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        hipLaunchKernelGGL(fn, dim3(blocks), dim3(512), lds2, s, a, n_pairs, words, idx_words);
      });
      return 0;
    }
    if (lds <= 150 * 1024) {   // four pairs per workgroup, index gathered from global memory
      with_seed_variant(wide, cb, [&](auto w, auto c) {
        auto fn = k_seed_wave<decltype(w)::value, decltype(c)::value>;
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(fn, dim3((n_pairs + 3) / 4), dim3(256), lds, s, a, n_pairs, words);
      });
      return 0;
    }
  }
  const size_t lds = seed_lds_bytes(a.max_nd, mem, false);
  if (lds > 150 * 1024) {  // too long for LDS: global-memory workspaces
    if (!a.ws || !a.ws_slots || a.ws_words * 4 < lds) return -1;
    const uint32_t grid = n_pairs < a.ws_slots ? n_pairs : a.ws_slots;
    if (mem) hipLaunchKernelGGL((k_seed_global<true, false>), dim3(grid), dim3(256), 0, s, a, n_pairs);
    else hipLaunchKernelGGL((k_seed_global<false, false>), dim3(grid), dim3(256), 0, s, a, n_pairs);
    return 0;
  }
  if (mem) {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_seed<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_seed<true>, dim3(n_pairs), dim3(256), lds, s, a);
  } else {
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_seed<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_seed<false>, dim3(n_pairs), dim3(256), lds, s, a);
  }
  return 0;
}
void launch_bin_units(const SeedArgs& a, uint32_t n_pairs, uint32_t n_ovf, hipStream_t s) {
  const uint64_t threads = n_ovf ? (uint64_t)n_ovf : (uint64_t)n_pairs * kMaxBandsPerPair;
  if (!threads) return;
  hipLaunchKernelGGL(k_bin_units, dim3((uint32_t)((threads + kBinThreads - 1) / kBinThreads)), dim3(kBinThreads), 0, s, a, n_pairs, n_ovf);
}
void launch_finalize(const FinalArgs& a, hipStream_t s) {
  if (a.n_pairs) hipLaunchKernelGGL(k_finalize_pairs, dim3((a.n_pairs + 255) / 256), dim3(256), 0, s, a);
}
void launch_select(const FinalArgs& a, hipStream_t s) {
  if (a.n_reads) hipLaunchKernelGGL(k_select, dim3((a.n_reads + 255) / 256), dim3(256), 0, s, a);
}
void launch_traceback(const FinalArgs& a, hipStream_t s) {
  if (!a.n_recs) return;
  hipLaunchKernelGGL(k_traceback<true>, dim3((a.n_recs + kTbLanes - 1) / kTbLanes), dim3(kTbLanes), 0, s, a);
  hipLaunchKernelGGL(k_traceback<false>, dim3((a.n_recs + kTbLanes - 1) / kTbLanes), dim3(kTbLanes), 0, s, a);
}

}  // namespace qf
