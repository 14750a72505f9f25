// qf_kernels.hpp — argument blocks and launch entry points of the HIP kernels (qf_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "qf_device.hpp"

namespace qf {

constexpr int kNQualDev = 94;

struct PrepArgs {
  const char* seq;
  const char* qual;        // NULL: no quality scores
  const uint64_t* off;     // [n_reads+1]
  uint32_t match_len, gap_len, seed_k;
  uint8_t* tok;            // [total]
  uint32_t* ctx;           // [total] (already offset past the front pad)
  uint32_t* skmer;         // [total] seeding k-mer starting at each base (k <= 16)
  unsigned long long* skmer64;  // same as 64-bit values, filled instead of skmer when the sorted index is used (k > 8)
  double* nll;             // [n_reads]
  int has_null;
  double null_logEmit, null_log1mEmit;
  double null_logSym[4];
  const double* null_logQual;  // [4][94]
  // overlap only (qf_overlap.hip)
  uint32_t* ctxc;          // [total] context words of the reverse-complement strand, in this sequence's orientation
  const double* eins;      // insert-emission table
  double* ins_sum;         // [n] sum of insert scores (xInsertScore / yInsertScore, src/qoverlap.cpp:105-113)
  double* ins_sum_c;       // [n] same with complemented tokens
  double* nll_c;           // [n] null log-likelihood of the reverse complement
  // E-step only: match-emission rows numbered QUALITY-major, (q - em_qmin) * em_qmajor_Km + context k-mer, instead of
  // k-mer * 95 + q: the rows of the qualities the resident reads actually use are then one contiguous slice of the table,
  // small enough for LDS also with -order 2 contexts (qf_api.hip: count path).  0 = the usual numbering.
  uint32_t em_qmajor_Km, em_qmin;
  BatchCounters* bc;
};

constexpr size_t kSeedRowLdsBig = 160 * 1024;  // LDS of one k_seed_rows_lds workgroup (counters + the chunk's k-mer index): one per CU
struct RowItemL { uint32_t x, ylo, yhi, pbase, xlen, xb_lo, xb_hi, chunk; };   // RowItem + x's offset and length (no dependent loads in the kernel)
size_t seed_rows_lds_stride_bytes(const struct SeedArgs& a);
size_t seed_rows_lds_stride_bytes(const struct SeedArgs& a);
size_t seed_rows_lds_fit(const struct SeedArgs& a, int cl, uint64_t max_entries, bool* e16);
size_t seed_rows_lds_bytes(const struct SeedArgs& a, size_t stride, int cl, uint64_t max_entries, bool e16);
constexpr size_t kSeedRowLdsMax = 76 * 1024;   // LDS of one row-prefilter workgroup (per-y coarse counters of a chunk): two fit a CU
struct RowItem { uint32_t x, chunk, ylo, yhi, pbase; };   // pairs (x, y) for y in [ylo, yhi), all inside one chunk; pair index of ylo
constexpr uint32_t kRowSeg = 32, kRowXcd = 8;   // the item list is dealt out in segments of kRowSeg to the XCDs (overlap_chunk)
// triangle rows X0 .. X0 + R - 1: the dealt-out item list formed on the device; returns its length (items == nullptr: only that)
uint32_t launch_row_items_tri(uint32_t X0, uint32_t R, uint32_t n_seqs, int cl, RowItem* items, size_t capacity, hipStream_t s);

struct SeedArgs {
  uint32_t pair_base, n_refs;
  const uint32_t* pair_x;      // optional explicit pair list (overlap): x / y sequence of each pair
  const uint32_t* pair_y;
  const uint64_t* ref_off;
  const uint64_t* read_off;
  const uint32_t* skmer;
  const unsigned long long* skmer64;   // non-NULL: sorted-index mode (k > 8)
  const unsigned long long* ref_skeys; // sorted k-mers of every x-sequence, aligned with ref_off
  const uint32_t* ref_bucket;  // [n_refs][nbuckets+1] exclusive starts
  const uint32_t* ref_pos;     // [total ref bases] positions grouped by k-mer
  uint32_t nbuckets;
  int sparse, kmer_len, threshold, band;
  unsigned long long cell_size, max_size;
  int max_nd;
  Unit* units;
  uint32_t max_units;
  uint32_t* cls_list;          // [kNumClasses][max_units]
  int2* pair_bands;            // [n_pairs][kMaxBandsPerPair] (dlo, dhi) written by the seeding kernels
  uint32_t* pair_nbands;       // [n_pairs] (zero-initialised)
  int4* ovf_bands;             // [ovf_cap] (pair, dlo, dhi, -) bands beyond kMaxBandsPerPair
  uint32_t ovf_cap;
  uint32_t* pair_head;         // [n_pairs] (kNoUnit-initialised)
  uint32_t* pair_ndiag;        // [n_pairs]
  unsigned long long* pair_cells;  // [n_pairs] (zero-initialised)
  uint8_t* dump_cover;         // optional [nd] membership of a single pair
  uint32_t* cls_key;           // optional [kNumClasses][max_units]: read length of each class-list entry (sort key)
  uint32_t* ws;                // global-memory seeding workspaces (sequences too long for the LDS histogram)
  uint64_t ws_words;           // 4-byte words per workspace
  uint32_t ws_slots;
  uint32_t max_ref_len;        // longest reference (0: unknown / not a reference set)
  uint32_t max_read_len;       // longest y-sequence of the batch (selects 32-bit coarse seeding counters)
  int few_hits;                // a read k-mer is expected less than once in an x-sequence (longest x / 4^k < 1): seeding prefetches two bucket entries, not four
  int no_lds_index;            // 1: never copy the reference index to LDS (debug / A-B)
  const uint8_t* pair_skip;    // optional [n_pairs]: 1 = do not seed this pair (train: pruned references)
  // row prefilter (explicit pair lists made of long runs x, y0, y0 + 1, ...: qf_kernels.hip, k_seed_rows)
  const RowItem* row_items;    // optional
  uint32_t n_row_items;
  const uint32_t* chunk_start; // [n_chunks][nbuckets + 1] bucket starts of each chunk of 2^chunk_log2 consecutive sequences
  const uint32_t* chunk_entries; // per chunk, from position read_off[first sequence of the chunk]: (sequence in chunk) x seed_row_entry_span + (len - 1 - j); with chunk_estride, from chunk x chunk_estride: (sequence in chunk) << 26 | (len - 1 - j)
  int chunk_pb;                // > 0: 16-bit index entries, (sequence in chunk) << chunk_pb | (len - 1 - j) (k_seed_rows<., true>), a chunk's from element chunk_base16() of chunk_entries
  const uint2* chunk_bounds;   // 16-bit index: per chunk and bucket (first entry: even, entries)
  uint64_t chunk_estride;      // > 0: the padded index of k_seed_rows_lds (buckets of even length, pad entries 0xFFFFFFFF)
  int chunk_log2;
  uint8_t* row_skip;           // [n_pairs], zero-initialised: set for the pairs the prefilter settled
  // the prefilter with the chunk's index in LDS (k_seed_rows_lds): the items chunk-major with x's offset / length (RowItemL),
  // cut into pieces (first item, count) of one chunk each; LDS sizing: most k-mer positions of a chunk, 16-bit entries or not
  const void* row_sorted;      // null: the scheduler's triangle, pieces = (first row, rows, chunk, -) and the items are formed on the device
  const uint4* row_pieces4;    // (first item, items, -, -) of row_sorted, or (first row, rows, chunk, -)
  uint32_t n_row_pieces, row_n_seqs, tri_x0;
  uint64_t row_max_entries;
  int row_e16;
  // slotted single-diagonal list (overlap, x-major lists with consecutive x): unit of pair (x, y) at
  // slot_list[((y >> 8) * slot_rows + (x - slot_x0)) * 256 + (y & 255)], holes = ~0u (kNoUnit)
  uint32_t* slot_list;         // optional, ~0u-initialised
  uint32_t slot_x0, slot_rows;
  int ov_use_32x3;             // overlap: bands of 65 .. 96 diagonals go to class kOv32Class (32 lanes x 3) instead of (16, 5) / (16, 6)
  int fb_use_32x3;             // E-step A/B: also bands of 65 .. 80 diagonals go to class kOv32Class (32 lanes x 3 diagonals), not only 81 .. 96
  int ov_wide_on_16x8;         // A/B: overlap bands of 97 .. 128 diagonals stay on class (16, 8) instead of (64, 3)
  int storage_mode;            // 0: packed traceback words (Viterbi); 1: Forward matrix doubles
  int force_block_kernel;      // use the workgroup-per-pair kernel even in threshold mode (tests run both)
  BatchCounters* bc;
};

struct FillArgs {
  uint32_t n_cls_units, n_refs;
  const uint32_t* cls_list;    // this class's unit ids
  Unit* units;
  const uint64_t* ref_off;
  const uint64_t* ref_woff;
  const uint8_t* ref_tok;
  const uint32_t* ref_packed;
  const uint64_t* read_off;
  const uint32_t* ctx;         // offset past the front pad
  uint32_t* tb;
  DpParams dp;
  int no_lds_tables;           // 1: keep the emission tables in global memory even when they fit LDS (debug / A-B)
};

struct FbArgs {  // Forward / Backward fills (qf_fb.hip)
  uint32_t n_cls_units, n_refs;
  const uint32_t* cls_list;
  Unit* units;
  const uint64_t* ref_off;
  const uint64_t* ref_woff;
  const uint8_t* ref_tok;
  const uint32_t* ref_packed;
  const uint64_t* read_off;
  const uint32_t* ctx;
  double* fw;                  // Forward matrix, unit u at fw + units[u].tb_off
  const double* lse;           // 100001-entry log(1+exp(-x)) table
  const double* lse_h;         // the same function as kLsePieces cubic pieces on a 1/32 grid (qf_fb.hip: lseh)
  DpParams dp;
  const double* pair_fwd;      // [n_pairs] Forward result (Backward only)
  const double* pair_weight;   // [n_pairs] posterior weight, 0 = no Backward
  unsigned long long* counts;  // flattened weighted QuaffCounts accumulators as 128-bit fixed point ((low, high) words per entry: qf_fb.hip, fx_add), kCountReplicas copies `counts_stride` entries apart
  int no_band_shortcuts;   // A/B: lone diagonals through the (16,2) Forward kernel, Backward skips no band
  uint64_t counts_stride;
  uint32_t Km;
  // quality-major emission rows (PrepArgs::em_qmajor_Km): row = (q - em_qmin) << em_kshift | k-mer; em_qmajor = 0: row = k-mer * 95 + q
  uint32_t em_qmajor, em_kshift, em_qmin;
  uint32_t lds_limit;          // bytes of LDS one workgroup of the fill kernels may take (tables go to LDS while they fit)
  uint32_t flush_lds;          // k_count_flush: bytes of LDS for its table (0 = default; 1 = none: straight to the global accumulators; tests)
};

struct CountPlanArgs {
  uint32_t n_reads, n_refs;
  int use_null;
  const double* nll;
  const double* lse;
  const double* pair_fwd;
  double* pair_fwd_out;
  double* weight;
  const uint32_t* order_in;    // optional [n_reads][n_refs]
  const uint32_t* order_n_in;  // optional [n_reads]
  uint32_t* order_out;
  uint32_t* order_n_out;
  double* read_loglike;
};

struct AlignRec;
struct OvArgs {  // overlap Viterbi fill / finalize / traceback (qf_overlap.hip)
  uint32_t n_cls_units, n_pairs;
  const uint32_t* cls_list;
  Unit* units;
  const uint32_t* pair_x;
  const uint32_t* pair_y;
  const uint8_t* pair_comp;
  const uint64_t* seq_off;
  const uint32_t* ctx;      // plain context words
  const uint32_t* ctxc;     // complemented-strand context words
  uint32_t* tb;
  const double* mmi[2];     // pair-emission tables [plain, yComplemented]
  double min_score;         // alignments scoring below it are neither traced back nor returned (-inf: keep all)
  int no_fast_steps;        // A/B: the banded overlap fill takes the general step everywhere
  int no_lds_rows;          // A/B: single-diagonal bands gather their emissions from global memory (k_overlap_single)
  int per_pair;             // 1: pair_result / pair_score / pair_end_unit / pair_end_ij for every pair (qf_overlap_resident); 0: only what the kept alignments need
  const double* gap[2];
  const uint32_t* slot_list;  // single-diagonal bands by (y chunk, x row, y) instead of cls_list (SeedArgs::slot_list)
  uint32_t slot_rows, slot_ychunks;
  // k_overlap_single_rows (qf_overlap.hip): compact pair-emission tables (rows mmic_pitch doubles apart, mmic_cpr 16-byte chunks
  // used per row), per-base row offsets of every sequence, per-base 16-bit column offsets transposed per 64 sequences
  uint32_t slot_x0;
  uint32_t mmic_pitch;        // 0: not available (the kernels above run)
  uint32_t mmic_cpr;
  const double* mmic[2];
  const uint32_t* xrowoff;
  const uint4* ycolT[2];
  const uint64_t* ygoff;      // [groups + 1] first chunk of each group of 64 sequences
  const double* lse;
  const uint8_t* lse_pack;  // the same table packed for LDS (qf_device.hpp: kLsePack*) (null: not available)
  uint32_t lse_pack_bytes;
  uint32_t Km, Kg;
  // finalize / traceback
  const uint32_t* pair_head;
  const uint32_t* pair_ndiag;   // envelope diagonals of each pair (summed into BatchCounters::sum_ndiag)
  const double* ins_sum;
  const double* ins_sum_c;
  const double* nll;
  const double* nll_c;
  double* pair_result;      // end + xInsertScore + yInsertScore
  double* pair_score;       // result - null(x) - null(y)
  uint32_t* pair_end_unit;
  uint32_t* pair_end_ij;    // [n_pairs][2] end cell (i, j)
  AlignRec* recs;
  uint32_t n_recs;
  uint32_t* runs_tmp;
  uint32_t* runs_out;
  BatchCounters* bc;
};

struct AlignRec {
  uint32_t read, ref, unit, ok;
  double viterbi, score;
  unsigned long long tmp_off, run_off;
  uint32_t x_start, x_end, n_columns, n_runs;
  uint32_t y_start, y_end;
};

// Device image of qf_alignment (include/quaff_hip.h): best-alignment records are written in their final form and order
struct AlignOut {
  uint32_t read, ref;
  double viterbi, score;
  uint32_t x_start, x_end, n_columns, n_runs;
  unsigned long long run_offset;
};
constexpr uint32_t kAlignHole = 0xFFFFFFFFu;   // n_runs of a read without any alignment

struct FinalArgs {
  uint32_t n_pairs, n_reads, n_refs, n_recs;
  int all;
  double min_score;     // alignments scoring below it are neither traced back nor returned (-inf: keep all)
  int dense;            // best-per-read mode: record r belongs to read r (holes marked), results go to out_align
  uint32_t read_base;   // index of the chunk's first read in the batch
  AlignOut* out_align;
  Unit* units;
  const uint32_t* pair_head;
  double* pair_score;
  uint32_t* pair_end_unit;
  const double* nll;
  const uint64_t* read_off;
  const uint64_t* ref_off;
  AlignRec* recs;
  const uint32_t* tb;
  uint32_t* runs_tmp;
  uint32_t* runs_out;
  BatchCounters* bc;
  // k_pair_forward: the Forward matrices and what it takes to form the end terms of a band's last column
  const double* fw;
  const uint32_t* ctx;
  const double* trans;
  uint32_t Kg;
  int local;
};

void launch_prep_ref(const char* seq, uint64_t total, uint8_t* tok, BatchCounters* bc, hipStream_t s);
void launch_pack_ref(const uint8_t* tok, const uint64_t* off, const uint64_t* woff, uint32_t n_refs, uint64_t max_len,
                     uint32_t* packed, hipStream_t s);
void launch_ref_index(const uint8_t* tok, const uint64_t* off, uint32_t n_refs, uint64_t max_len, uint32_t k,
                      uint32_t nbuckets, uint32_t* starts, uint32_t* cursor, uint32_t* pos, hipStream_t s);
void launch_prep_reads(const PrepArgs& a, uint32_t n_reads, hipStream_t s);
// smallest and largest quality value (as the prep kernels clamp them: 0 .. 93) of `total` quality characters -> out[0], out[1]
void launch_qual_range(const char* qual, uint64_t total, uint32_t* out, hipStream_t s);
void launch_null_ll(const PrepArgs& a, uint32_t n_reads, hipStream_t s);   // needs the token bytes of launch_prep_reads
// Emission counts of one read column go to the accumulator table with one global fp64 atomic each; workgroups spread over
// this many copies of the table (summed at the end) so that popular (context, quality) entries are not serialised in L2.
constexpr int kCountReplicas = 16;
void launch_sum_count_replicas(const unsigned long long* fx, uint32_t n, uint64_t stride, unsigned long long* out_fx, double* out, hipStream_t s);
// Sorts one class list by descending key (read length): the bands a wavefront takes together then have similar lengths
// (a wavefront runs for its longest band) and the longest start first.  Returns 0 or a hipError_t.
int sort_class_list(uint32_t* keys, uint32_t* list, uint32_t n, uint32_t* keys_tmp, uint32_t* list_tmp, void** temp,
                    size_t* temp_cap, hipStream_t s);
size_t seed_lds_bytes(int max_nd, bool mem, bool deep);
bool seed_needs_deep_counters(const SeedArgs& a);
bool seed_needs_workspace(const SeedArgs& a, bool mem);
int launch_seed(const SeedArgs& a, uint32_t n_pairs, bool mem, hipStream_t s);
void launch_bin_units(const SeedArgs& a, uint32_t n_pairs, uint32_t n_ovf, hipStream_t s);
void launch_viterbi_fill(int cls, const FillArgs& a, bool gapctx, hipStream_t s);
uint32_t viterbi_rows_resident_workgroups(const FillArgs& a);   // 0: unknown
double measure_f64_add_rate(double* d_out, hipStream_t s);     // fp64 add lane-operations per second the chip sustains; 0 on error
void launch_finalize(const FinalArgs& a, hipStream_t s);
void launch_forward_fill(int cls, const FbArgs& a, hipStream_t s);
void launch_backward_fill(int cls, const FbArgs& a, hipStream_t s);
void launch_pair_forward(const FinalArgs& a, const double* lse, hipStream_t s);
void launch_count_plan(const CountPlanArgs& a, hipStream_t s);
int sort_kmer_index(const uint8_t* tok, const uint64_t* d_off, const int* d_off32, uint32_t n_seqs, uint64_t total,
                    uint64_t max_len, uint32_t k, unsigned long long* keys_tmp, uint32_t* vals_tmp,
                    unsigned long long* keys_out, uint32_t* pos_out, void** temp, size_t* temp_cap, hipStream_t s);
void launch_prep_overlap(const PrepArgs& a, uint32_t n, hipStream_t s);
// k-mer index of chunks of 2^chunk_log2 consecutive sequences (row prefilter of the overlap seeding)
void launch_chunk_index(const uint8_t* tok, const uint64_t* off, uint32_t n_seqs, uint64_t max_len, uint32_t k, uint32_t nbuckets,
                        int chunk_log2, uint32_t* starts, uint32_t* cursor, uint32_t* entries, uint64_t estride, uint32_t wd, int pb, uint2* bounds, hipStream_t s);
uint32_t seed_row_entry_span(const struct SeedArgs& a);
size_t seed_row_stride_bytes_e16(int pb, int cb);
int seed_row_bits_of(const struct SeedArgs& a);
// LDS bytes per sequence of a chunk in the row prefilter (coarse counters of one pair), or 0 if the prefilter does not apply
size_t seed_row_stride_bytes(const SeedArgs& a);
void launch_overlap_fill(int cls, const OvArgs& a, hipStream_t s);
bool overlap_single_stages_rows(uint32_t Km);
// compact tables / offsets of k_overlap_single_rows: row pitch in doubles for Km x nq entries per row (0: does not apply)
uint32_t overlap_compact_pitch(uint32_t Km, uint32_t nq);
void launch_mmi_compact(const double* mmi, uint32_t Km, uint32_t qmin, uint32_t nq, uint32_t pitch, double* out, hipStream_t s);
void launch_overlap_cols(const uint32_t* ctx, const uint32_t* ctxc, const uint64_t* off, uint32_t n_seqs, uint64_t total, const uint64_t* goff,
                         uint32_t max_blocks, uint32_t Km, uint32_t qmin, uint32_t pitch, uint32_t* xrowoff, uint4* col0, uint4* col1, hipStream_t s);   // the single-diagonal kernel that stages emission rows through LDS applies
// entries of the exact log-sum-exp table that its packed form (qf_device.hpp) does not reproduce on this device; ~0u on a HIP error
uint32_t lse_pack_mismatches(const uint8_t* pack, uint32_t pack_bytes, const double* tab, uint32_t* d_bad, hipStream_t s);
void launch_overlap_finalize(const OvArgs& a, hipStream_t s);
// (x, y, yComplemented) of rows [x0, x0 + rows) of QuaffOverlapScheduler's enumeration over n_seqs resident sequences, row-major
void launch_overlap_row_pairs(uint32_t x0, uint32_t rows, uint32_t n_seqs, uint32_t n_orig, uint32_t* px, uint32_t* py, uint8_t* pc,
                              hipStream_t s);
void launch_overlap_traceback(const OvArgs& a, hipStream_t s);
void launch_select(const FinalArgs& a, hipStream_t s);
void launch_traceback(const FinalArgs& a, hipStream_t s);

}  // namespace qf
