// qf_device.hpp — device-side data layout shared by the HIP kernels and the C-ABI host code.
// gfx950 only: 64-wide wavefronts, fp64 VALU, LDS histograms; no MFMA (the recurrence is
// max-plus / log-sum-exp, not a contraction).
#pragma once
#include <cstdint>

namespace qf {

constexpr int kMaxRefK = 8;        // direct-addressed k-mer index up to 4^8 buckets per sequence
constexpr uint32_t kNoUnit = 0xFFFFFFFFu;
constexpr int kMaxBandsPerPair = 2;  // diagonal runs recorded in per-pair slots; further runs of a pair go to the overflow list
constexpr int kCtxPad = 128;       // junk words before/after the per-column context array

// Packed per-column read context word (one per read base), built by the prep kernel:
//   bits  0..14  erow   = matchKmer*95 + q      row of the match-emission table (q = 94 without quals)
//   bits 15..23  insrow = token*95 + q          row of the insert-emission table
//   bits 24..31  gk     = indel-context k-mer of the context ending at this base
__host__ __device__ inline uint32_t ctx_pack(uint32_t erow, uint32_t insrow, uint32_t gk) {
  return erow | (insrow << 15) | (gk << 24);
}

// One contiguous run of envelope diagonals of one (read, ref) pair.  Runs separated by a
// missing diagonal never exchange probability mass (every transition moves to the same or an
// adjacent diagonal), so each is an independent DP problem.
struct Unit {
  uint32_t pair;      // read * n_refs + ref
  int32_t dlo, dhi;   // diagonals i - j, inclusive
  uint32_t next;      // next unit of the same pair (kNoUnit terminates)
  uint64_t tb_off;    // first traceback word
  double end_val;     // max over the last read column of mat + m2e  (-inf if none)
  uint32_t end_i;     // its 1-based reference row (largest row on ties)
  uint32_t cls;       // fill-kernel class
  double end2_val;    // overlap only: max of mat over the band's cells in the last REFERENCE row (i == xLen)
  uint32_t end2_j;    // its read column (largest column on ties)
  uint32_t staged;    // E-step: k_backward_fill ran on this band and left its column sums for k_count_flush
};

// Fill-kernel classes: class 0 = single diagonal (one lane per unit); class c>0 = G lanes x B
// diagonals per lane.  Kept in one table so host and device agree.
struct FillClass { int G, B; };
constexpr int kNumClasses = 15;
constexpr int kRowClass = 13;   // row-space kernel: bands wider than the diagonal-space kernels take (full DP)
// Overlap only, and only under QF_DEBUG_OV32 (never returned by classify_width): bands of 65 .. 96 diagonals on 32 lanes x 3
// diagonals instead of 16 x 5 / 6.  Three diagonals per lane take 168 registers (five: 212), i.e. three wavefronts per SIMD
// instead of two with twelve wavefronts sharing the packed table -- and the kernel got SLOWER (dense overlaps: 158 vs 141 ms):
// it is not the chained look-ups' latency that bounds it but the CU's one LDS pipe, which every look-up crosses six times, and
// fewer diagonals per lane mean more per-step work per cell.  Kept as the A/B it was.
constexpr int kOv32Class = 14;
__host__ __device__ constexpr FillClass fill_class(int c) {
  constexpr FillClass t[kNumClasses] = {{1, 1},  {16, 2}, {16, 3}, {16, 4}, {16, 5},  {16, 6},  {16, 8},
                                        {64, 3}, {64, 4}, {64, 6}, {64, 8}, {64, 12}, {64, 16},
                                        {64, 8}, {32, 3}};
  return t[c];
}
constexpr int kMaxBandDiagSpace = 64 * 16;  // widest band the diagonal-space kernels take
__host__ __device__ inline int classify_width(int W) {
  if (W <= 1) return 0;
  for (int c = 1; c < kRowClass; ++c)
    if (fill_class(c).G * fill_class(c).B >= W) return c;
  return kRowClass;
}

// Row-space geometry (class kRowClass): the band's rows [ilo, ihi] are cut into stripes of S rows; stripe s sweeps the
// columns [jlo, jhi] that its rows' band segments cover.  Forward/Backward and overlap run one wavefront per unit
// (S = kRowStripe = 64 lanes x 8 rows); Viterbi runs a workgroup of kVitWaves wavefronts (S = kVitStripe).
constexpr int kRowStripe = 64 * 8;
constexpr int kVitWaves = 4;
constexpr int kVitLanes = kVitWaves * 64;
constexpr int kVitStripe = kVitLanes * 8;
constexpr int kVitLag = 8;        // extra column lag per wavefront: neighbouring wavefronts synchronise every kVitLag steps
__host__ __device__ inline int vit_skew(int lane) { return lane + kVitLag * (lane >> 6); }   // lane's column lag
constexpr int kVitSkewMax = kVitLanes - 1 + kVitLag * (kVitWaves - 1);
struct RowGeom { int ilo, ihi, nStripes; };
__host__ __device__ inline RowGeom row_geom(int dlo, int dhi, int xLen, int yLen, int S = kRowStripe) {
  RowGeom g;
  g.ilo = 1 + dlo > 1 ? 1 + dlo : 1;
  g.ihi = yLen + dhi < xLen ? yLen + dhi : xLen;
  g.nStripes = g.ihi >= g.ilo ? (g.ihi - g.ilo + S) / S : 0;
  return g;
}
__host__ __device__ inline void row_stripe_cols(const RowGeom& g, int s, int dlo, int dhi, int yLen, int& jlo, int& jhi,
                                                int S = kRowStripe) {
  const int i0 = g.ilo + s * S, i1 = i0 + S - 1 < g.ihi ? i0 + S - 1 : g.ihi;
  jlo = i0 - dhi > 1 ? i0 - dhi : 1;
  jhi = i1 - dlo < yLen ? i1 - dlo : yLen;
}
// storage of a row-space unit, in 4-byte words: [ (nStripes+1) u64 stripe offsets | 2 boundary rows x 3 states x
// (yLen+2) doubles | per stripe: steps x lanes traceback words ]
__host__ __device__ inline uint64_t row_header_words(const RowGeom& g, int yLen) {
  return 2ull * (g.nStripes + 1) + 2ull * 2 * 3 * (uint64_t)(yLen + 2);
}
__host__ __device__ inline uint64_t row_unit_words(int dlo, int dhi, int xLen, int yLen) {
  const RowGeom g = row_geom(dlo, dhi, xLen, yLen, kVitStripe);
  uint64_t w = row_header_words(g, yLen);
  for (int s = 0; s < g.nStripes; ++s) {
    int jlo, jhi;
    row_stripe_cols(g, s, dlo, dhi, yLen, jlo, jhi, kVitStripe);
    if (jhi >= jlo) w += (uint64_t)(jhi - jlo + 1 + kVitSkewMax) * kVitLanes;
  }
  return w;
}
// Forward storage of a row-space unit, in doubles: [ (nStripes+1) u64 stripe offsets | 2 boundary rows x 3 states x
// (yLen+2) | per stripe: steps x 8 rows x 3 states x 64 lanes ]
__host__ __device__ inline uint64_t row_fw_header(const RowGeom& g, int yLen) {
  return (uint64_t)(g.nStripes + 1) + 2ull * 3 * (uint64_t)(yLen + 2);
}
__host__ __device__ inline uint64_t row_fw_doubles(int dlo, int dhi, int xLen, int yLen) {
  const RowGeom g = row_geom(dlo, dhi, xLen, yLen);
  uint64_t w = row_fw_header(g, yLen);
  for (int s = 0; s < g.nStripes; ++s) {
    int jlo, jhi;
    row_stripe_cols(g, s, dlo, dhi, yLen, jlo, jhi);
    if (jhi >= jlo) w += (uint64_t)(jhi - jlo + 1 + 63) * 64 * 8 * 3;
  }
  return w;
}
// overlap traceback of a row-space unit, in 4-byte words: the Viterbi header, then per stripe steps x 64 lanes x 2 words
// (one byte per cell, 8 rows per lane)
__host__ __device__ inline uint64_t row_ov_words(int dlo, int dhi, int xLen, int yLen) {
  const RowGeom g = row_geom(dlo, dhi, xLen, yLen);
  uint64_t w = row_header_words(g, yLen);
  for (int s = 0; s < g.nStripes; ++s) {
    int jlo, jhi;
    row_stripe_cols(g, s, dlo, dhi, yLen, jlo, jhi);
    if (jhi >= jlo) w += (uint64_t)(jhi - jlo + 1 + 63) * 64 * 2;
  }
  return w;
}
// 16-bit chunk index (k_seed_rows<., true>): first element of chunk c's entries -- behind the sequences before it plus room for one
// pad entry per bucket of the chunks before (buckets are padded to even length), on an even element (4-byte aligned)
__host__ __device__ inline uint64_t chunk_base16(const uint64_t* off, uint32_t c, int cl, uint32_t nbuckets) {
  return (off[(uint64_t)c << cl] + (uint64_t)c * nbuckets + 1) & ~1ull;
}
// traceback words a unit occupies
// One-word classes (B <= 8) keep the words of eight consecutive steps of a fill lane together ([step/8][lane][step%8]): the
// fill stores 16 bytes per lane every four steps (one half of the lane's 32-byte slot) and a traceback, which follows one
// lane for many steps, reads one 128-byte line per eight moves instead of one per move.
__host__ __device__ inline uint64_t tb_word_index(int t, int l, int G) {
  return ((uint64_t)(t >> 3) * G + l) * 8 + (t & 7);
}
// Columns of y a band of diagonals [dlo, dhi] (d = i - j) meets inside the xLen x yLen rectangle: j0 + 1 ... band_last_col.
// Two reads that overlap by a third of their length have bands that cross a third of the columns: the overlap fills step
// over these columns only.
__host__ __device__ inline int band_col0(int dhi) { return dhi < 0 ? -dhi : 0; }
__host__ __device__ inline int band_last_col(int dlo, int xLen, int yLen) { return xLen - dlo < yLen ? xLen - dlo : yLen; }
__host__ __device__ inline int band_cols(int dlo, int dhi, int xLen, int yLen) {
  const int n = band_last_col(dlo, xLen, yLen) - band_col0(dhi);
  return n > 0 ? n : 0;
}
__host__ __device__ inline uint64_t unit_tb_words(int cls, uint32_t yLen) {
  if (cls == 0) return (yLen + 7) / 8;
  const FillClass fc = fill_class(cls);
  if (fc.B > 8) return (uint64_t)(yLen + fc.G - 1) * fc.G * 2;
  return (uint64_t)((yLen + fc.G - 1 + 7) & ~7u) * fc.G;
}

// Forward storage of a diagonal-space unit (single-diagonal bands use the (16,2) geometry).  Per step and lane one row of
// fw_row_floats(B) fp32 values: [0] an anchor (the largest of the row's values, rounded to fp32), then for slot B-1 down
// to 0 the offsets (mat, ins, del) - anchor.  A row is written / read as 16-byte chunks, chunk c of step t and lane l at
// float index ((t * chunks + c) * G + l) * 4 (a wavefront's store covers 256 contiguous bytes per band), which is what the
// store path of the device wants: 4 wide stores per step instead of 15 narrow ones, and half the bytes.  The offsets'
// rounding (2^-24 relative to their distance from the row's best value) is 6e-8 x that distance in the exponent of a count:
// < 1e-5 for anything within e^-150 of the row's best state, against the 1e-4 tolerance of the counts.  After the steps
// come G x B fp64 values: mat(i, yLen) of every diagonal of the band, exactly, for the pair's Forward result
// (k_pair_forward).
__host__ __device__ inline int fb_class(int cls) { return cls == 0 ? 1 : cls; }
__host__ __device__ constexpr int fw_row_floats(int B) { return (3 * B + 1 + 3) / 4 * 4; }
__host__ __device__ constexpr int fw_float_index(int B, int b, int state) { return 1 + 3 * (B - 1 - b) + state; }
__host__ __device__ inline uint64_t unit_fw_steps_doubles(int cls, uint32_t yLen) {   // the per-step rows, in doubles
  const FillClass fc = fill_class(fb_class(cls));
  return (uint64_t)(yLen + fc.G - 1) * fc.G * fw_row_floats(fc.B) / 2;
}
__host__ __device__ inline uint64_t unit_fw_doubles(int cls, uint32_t yLen) {
  const FillClass fc = fill_class(fb_class(cls));
  return unit_fw_steps_doubles(cls, yLen) + (uint64_t)fc.G * fc.B;
}

struct BatchCounters {
  uint32_t n_units;
  uint32_t n_ovf;           // bands spilled past kMaxBandsPerPair
  uint32_t cls_count[kNumClasses];
  uint32_t error;          // bit 0: unit overflow, 1: band too wide, 2: bad symbol, 3: > kMaxBandsPerPair bands, 4: two bands claimed one slot of the slotted list
  uint32_t error_detail;
  unsigned long long tb_words;
  unsigned long long total_cells;
  unsigned long long cls_cells[kNumClasses];
  uint32_t n_align;
  unsigned long long n_runs;          // scratch run capacity reserved by the select kernel
  unsigned long long total_runs_out;  // compacted CIGAR runs written by the traceback kernel
  // overlap totals (k_overlap_finalize): what qf_overlap_rows reports instead of per-pair arrays
  unsigned long long n_finite;        // pairs with a finite result
  unsigned long long sum_ndiag;       // envelope diagonals over all pairs
  unsigned long long result_sum;      // sum of the finite results' bit patterns, mod 2^64
};

constexpr int kLsePieces = 1281;   // quadratic pieces of log(1 + exp(-x)) on a 1/128 grid over [0, 10) + the all-zero piece of the cut-off
struct LsePiece { double c0; float c1, c2; };   // c0 + t (c1 + t c2), t = 128 x - n in [0, 1): 16 bytes, one ds_read_b128 per lookup
// The exact log-sum-exp table (src/logsumexp.cpp:20-28: 100 001 doubles, 800 KB) in a form that fits a CU's LDS: the table is
// a smooth function plus the rounding noise of log(1 + exp(-x)) -- about 1.1e-16 absolute, i.e. 2 units in the last place at
// x = 0 and 16 000 at x = 10 -- so entry n = one fifth-degree polynomial per 256 entries, evaluated with a fixed sequence of
// fused multiply-adds (the same bits on the host and on the device), plus a small signed correction to the result's bit
// pattern, stored in as many bits as the piece needs (2 ... 15).  Piece p serves entries 256 p ... 256 p + 256 (its own
// corrections for all 257), so the two entries of an interpolation come from one piece.  128 KB instead of 800.
// Layout (bytes): coefficient pairs (c0, c1)[391] | (c2, c3)[391] | (c4, c5)[391] of the polynomial in u = (n mod 256) - 128,
// 16 bytes per piece each (a wavefront's 64 pieces spread over all LDS banks; 64-byte piece records would put them on four
// bank groups) | (bit_base, width)[391]: the piece's corrections are `width`-bit two's complement fields from bit `bit_base`
// of the stream | the correction bit stream.
constexpr int kLseEntriesDev = 100001;
constexpr int kLsePackPieces = 391, kLsePackSpan = 256, kLsePackDegree = 5;
constexpr uint32_t kLsePackC01 = 0, kLsePackC23 = kLsePackPieces * 16, kLsePackC45 = 2 * kLsePackPieces * 16,
                   kLsePackMeta = 3 * kLsePackPieces * 16, kLsePackStream = (kLsePackMeta + kLsePackPieces * 8 + 15) & ~15u;
constexpr uint32_t kInsRows = 4 * 95;  // insert-emission table: [token * 95 + quality]

struct DpParams {  // kernel argument block for the fills
  const double* ematch;   // [(matchKmer*95 + q)*4 + refTok]
  const double* eins;     // [tok*95 + q]
  const double* trans;    // m2m[Kg] m2i[Kg] m2d[Kg] m2e[Kg]
  double d2d, d2m, i2i, i2m;
  uint32_t Kg;
  int32_t local;
  uint32_t ematch_ninf_off;  // byte offset of a -inf entry placed after the last row of ematch
  uint32_t pad_;
};

}  // namespace qf
