/* qf_internal.h — entry points of libquaffhip that are NOT part of the public ABI (include/quaff_hip.h): switches the
 * tests and the A/B benchmarks use to force a kernel variant the library would not pick by itself for that input.
 * Results never depend on them. */
#ifndef QF_INTERNAL_H
#define QF_INTERNAL_H
#include <stdint.h>

#include "../../include/quaff_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum qf_debug_flag {
  QF_DEBUG_BLOCK_SEED = 1u,           /* workgroup-per-pair seeding kernel (k_seed) even in threshold mode */
  QF_DEBUG_SERIAL_CLASSES = 4u,       /* fill classes one after another on one stream (isolated kernel timings) */
  QF_DEBUG_GLOBAL_TABLES = 8u,        /* emission tables stay in global memory even when they fit LDS */
  QF_DEBUG_GLOBAL_INDEX = 16u,        /* reference k-mer index stays in global memory */
  QF_DEBUG_GLOBAL_OVERLAP_ROWS = 32u, /* overlap single-diagonal bands gather their emissions from global memory */
  QF_DEBUG_NO_BAND_SHORTCUTS = 64u,   /* E-step without the single-diagonal Forward kernel / negligible-band skip */
  QF_DEBUG_GLOBAL_LSE = 128u,         /* overlap fills gather the exact log-sum-exp table from global memory, not its packed form in LDS */
  QF_DEBUG_NO_ROW_PREFILTER = 256u,   /* overlap seeding: every pair through the per-pair kernel (no chunk-of-y prefilter) */
  QF_DEBUG_COUNT_SETTLED = 512u,      /* overlap: count the pairs the prefilter settles (qf_debug_rows_settled; costs a read-back) */
  QF_DEBUG_PAIR_ORDER_SINGLES = 1024u,/* overlap: single-diagonal bands as a plain list in pair order, not by (y chunk, x row) */
  QF_DEBUG_OLD_SINGLE_ROWS = 8192u,   /* overlap: the slotted single-diagonal bands through k_overlap_single_lds (round 3) instead of k_overlap_single_rows */
  QF_DEBUG_LDS_ROW_INDEX = 16384u,    /* overlap seeding: the row prefilter keeps the chunk's k-mer index in LDS (k_seed_rows_lds: fewer instructions, but LDS-bound and slower; A/B) */
  QF_DEBUG_FB32 = 32768u,             /* E-step: bands of 65 .. 96 diagonals through the (32, 3) Forward / Backward kernels (A/B) */
  QF_DEBUG_FLUSH_GLOBAL = 1048576u,   /* E-step: k_count_flush adds straight to the global accumulators (the path of tables too large to slice; tests) */
  QF_DEBUG_FLUSH_SLICES = 2097152u,   /* E-step: k_count_flush with a 12 KB table: many slices of the match-emission rows (tests) */
  QF_DEBUG_NO_BACKWARD_SPLIT = 524288u, /* E-step: the dominant class's Backward as one launch (A/B) */
  QF_DEBUG_HOST_ROW_ITEMS = 262144u,  /* row prefilter: the item list of a triangle block built by the host like any other list's (A/B, tests) */
  QF_DEBUG_ROW_INDEX_32 = 131072u,    /* row prefilter: 32-bit index entries even where 16-bit ones fit (A/B) */
  QF_DEBUG_OV_NO_FAST = 65536u,       /* banded overlap fill: the general step everywhere (no FAST steps; A/B) */
  QF_DEBUG_OV32 = 4096u,              /* overlap: bands of 65 .. 96 diagonals through the (32, 3) kernel at three wavefronts per SIMD (measured slower than (16, 5) / (16, 6) at two) */
  QF_DEBUG_BIG_FORWARD_LDS = 2048u    /* E-step: Forward may take 78 KB of LDS like Backward (two workgroups per CU, the emission slice of long contexts in LDS) */
};
int qf_debug_set_flags(qf_ctx *ctx, uint32_t flags);
/* The exact log-sum-exp table packed for LDS (qf_device.hpp: kLsePack*), built on the host:
 * copies up to `cap` bytes into `out`, returns the size (0: the host's libm does not fit the scheme).  No GPU needed. */
uint32_t qf_debug_pack_lse_table(uint8_t *out, uint32_t cap);
/* Bytes of the packed table this context's overlap fills keep in LDS; 0 if the device did not rebuild the table from it bit for
 * bit (the fills then gather the table from global memory). */
uint32_t qf_debug_lse_pack_bytes(qf_ctx *ctx);
/* Pairs of the last qf_overlap_resident call that the seeding's row prefilter settled (given their single forced diagonal without
 * the per-pair kernel); counted only under QF_DEBUG_COUNT_SETTLED. */
int qf_debug_fail_chunk_reserve(int nth);    /* the nth per-chunk device reserve from now (0 = the next) fails once, as if out of memory; -1 = off; returns the countdown it replaces (< 0: the failure happened) -- tests of the split-and-retry path */
double qf_debug_alloc_ms(void);   /* milliseconds spent growing device buffers (hipFree + hipMalloc) in this process */
uint64_t qf_debug_rows_settled(const qf_ctx *ctx);

/* fp64 vector add lane-operations per second this device sustains (a 5 ms microbenchmark: 8 independent v_add_f64 chains per wavefront,
 * 4 wavefronts per SIMD, all CUs): the attainable issue roof beside the 39.3 T op/s the spec sheet implies. */
int qf_debug_measure_f64_rate(qf_ctx *ctx, double *lane_ops_per_s);
/* qf_overlap_rows cuts its rows into blocks of about this many pairs (0 = the default, 2^24); tests use small values to push a
 * small read set through many blocks. */
int qf_debug_set_overlap_block_pairs(qf_ctx *ctx, uint64_t pairs);

#ifdef __cplusplus
}
#endif
#endif
