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
  QF_DEBUG_NO_BAND_SHORTCUTS = 64u    /* E-step without the single-diagonal Forward kernel / negligible-band skip */
};
int qf_debug_set_flags(qf_ctx *ctx, uint32_t flags);

#ifdef __cplusplus
}
#endif
#endif
