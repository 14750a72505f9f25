/*
 * quaff_hip.h — C ABI of libquaffhip: the MI355X (gfx950) implementation of quaff's
 * k-mer-seeded banded pair-HMM DP hot path.
 *
 * The reference (ihh/quaff) has no FFI seam: the hot path is reached through C++
 * constructors called from three per-task drivers.  This header is the seam a quaff
 * maintainer would bind instead, at the Task level and batched:
 *
 *   reference (file:line under /root/reference)              replaced by
 *   -------------------------------------------------------  ---------------------------
 *   QuaffParams::readJson            src/qmodel.cpp:230-271   qf_set_params_json
 *   defaultQuaffParams               src/defaultparams.cpp:4  qf_set_params_json(ctx, NULL)
 *   QuaffScores::QuaffScores         src/qmodel.cpp:296-325   (built inside qf_set_params_json)
 *   QuaffNullParams::readJson        src/qmodel.cpp:1856-1866 qf_set_null_json
 *   QuaffNullParams::logLikelihood   src/qmodel.cpp:1875-1890 qf_align_result.null_loglike
 *   FastSeq::tokens/kmers/qualScores src/fastseq.cpp:71-109   (device prep kernel)
 *   KmerIndex::KmerIndex             src/fastseq.cpp:240-256  qf_set_refs (device k-mer index)
 *   QuaffDPConfig::makeEnvelope      src/qmodel.cpp:1049-1056 \
 *   DiagonalEnvelope::initSparse     src/diagenv.cpp:20-106    > qf_align_batch (seed kernel)
 *   DiagonalEnvelope::initFull       src/diagenv.cpp:11-18    /
 *   QuaffViterbiMatrix ctor          src/qmodel.cpp:1512-1560 qf_align_batch (fill kernel)
 *   QuaffViterbiMatrix::alignment    src/qmodel.cpp:1562-1646 qf_align_batch (traceback kernel)
 *   QuaffAlignmentTask::run          src/qmodel.cpp:2764-2778 qf_align_batch (best ref per read)
 *   Alignment::cigarString           src/qmodel.cpp:625-653   qf_cigar_string
 *   QuaffForwardMatrix ctor          src/qmodel.cpp:1343-1391 qf_count_resident (forward kernels)
 *   QuaffBackwardMatrix ctor         src/qmodel.cpp:1393-1510 qf_count_resident (backward kernels + counts)
 *   QuaffCountingTask::run           src/qmodel.cpp:2238-2271 qf_count_resident (pruning, weights, new order)
 *   QuaffParamCounts(QuaffCounts)    src/qmodel.cpp:407-417   qf_count_result.counts layout
 *   QuaffCountingScheduler::finalCounts/finalLogLike src/qmodel.cpp:2416-2422 sum over the batch (order-free: counts_exact), then qf_allreduce_counts[_exact] across GPUs
 *   QuaffOverlapScores ctor          src/qoverlap.cpp:9-75    (built inside qf_overlap_resident, once per strand flag)
 *   QuaffOverlapViterbiMatrix ctor   src/qoverlap.cpp:77-160  qf_overlap_resident (overlap fill kernel)
 *   QuaffOverlapViterbiMatrix::alignment :162-290, scoreAdjustedAlignment :292-302   qf_overlap_resident
 *   QuaffOverlapTask::run            src/qoverlap.cpp:457-464 qf_overlap_resident (one task per pair)
 *   QuaffOverlapScheduler::advance / nextOverlapTask src/qoverlap.cpp:475-480,528-547 qf_overlap_rows (pairs enumerated on the device)
 *   QuaffAlignmentPrinter threshold  src/qmodel.cpp:2566-2569 qf_set_score_threshold (applied before the traceback)
 *   runQuaff*Tasks worker threads    src/qmodel.cpp:2870-2882 qf_device_count + one qf_ctx per device
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success and a
 * negative qf_status otherwise (message via qf_last_error); nothing throws or exits across
 * the boundary (the reference prints and exit(1)s: src/util.cpp:80-98).  A context is
 * single-threaded; use one context per GPU / host thread.  Result views point into
 * context-owned memory and stay valid until the next *_batch call on that context.
 * There is no CPU fallback: without a usable HIP device qf_ctx_create fails.
 */
#ifndef QUAFF_HIP_H
#define QUAFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qf_ctx qf_ctx;

enum qf_status {
  QF_OK = 0,
  QF_ERR_DEVICE = -1,      /* no HIP device / HIP runtime error */
  QF_ERR_ARG = -2,         /* bad argument */
  QF_ERR_PARSE = -3,       /* JSON parse / missing parameter */
  QF_ERR_SYMBOL = -4,      /* non-ACGT symbol (reference: "Unknown symbol", src/fastseq.cpp:76-79) */
  QF_ERR_UNSUPPORTED = -5, /* valid for the reference, not built yet (documented in DESIGN.md) */
  QF_ERR_MEMORY = -6,
  QF_ERR_STATE = -7        /* call order (e.g. no params / refs set) */
};

/* QuaffDPConfig, src/qmodel.h:280-335: the DP-relevant fields only. */
typedef struct qf_dp_config {
  int32_t local;           /* 1 = local in the reference sequence (default); 0 = -global */
  int32_t sparse;          /* 1 = k-mer seeded envelope (default); 0 = -kmatchoff (full DP) */
  int32_t kmer_len;        /* -kmatch, default 6 */
  int32_t kmer_threshold;  /* -kmatchn; default 20 (align/train) or 14 (overlap); < 0 = memory mode */
  int32_t band_size;       /* -kmatchband, default 64 */
  int32_t reserved;        /* must be 0 */
  uint64_t max_size;       /* memory mode: effectiveMaxSize() in bytes (-kmatchmb M => M<<20) */
} qf_dp_config;

#define QF_MAX_FILL_CLASSES 16

/* flags for qf_align_batch */
#define QF_ALIGN_BEST 0u      /* traceback only the best reference per read (default; QuaffAlignmentTask) */
#define QF_ALIGN_ALL 1u       /* -printall: traceback every pair with a finite Viterbi score */
#define QF_ALIGN_NO_TRACEBACK 2u /* scores only */

typedef struct qf_alignment {
  uint32_t read;           /* index into the batch */
  uint32_t ref;            /* index into the reference set */
  double viterbi;          /* raw Viterbi log-likelihood (QuaffViterbiMatrix::result) */
  double score;            /* viterbi - null log-likelihood of the read (scoreAdjustedAlignment) */
  uint32_t x_start, x_end; /* 1-based closed interval of the reference covered */
  uint32_t n_columns;      /* alignment columns */
  uint32_t n_runs;         /* CIGAR runs */
  uint64_t run_offset;     /* first run in qf_align_result.cigar_runs */
} qf_alignment;

typedef struct qf_align_result {
  uint32_t n_reads, n_refs;
  const double *viterbi;        /* [n_reads * n_refs], -inf where no path exists */
  const uint64_t *cells;        /* [n_reads * n_refs] DP cells visited (SURVEY 8d definition) */
  const uint32_t *n_diagonals;  /* [n_reads * n_refs] envelope diagonals */
  const double *null_loglike;   /* [n_reads] (0 when no null model is set) */
  uint64_t total_cells;
  uint32_t n_alignments;
  const qf_alignment *alignments;  /* ordered by read, then (QF_ALIGN_ALL) by descending score */
  const uint32_t *cigar_runs;      /* run = (length << 2) | op, op 0=M 1=I 2=D, alignment order */
  /* device-side timings of the last call, milliseconds (HIP events on the context's stream) */
  float ms_prep, ms_seed, ms_fill, ms_traceback, ms_total;
  uint64_t n_units;             /* independent diagonal bands filled */
  uint64_t traceback_bytes;     /* packed traceback bytes written */
  /* per fill-kernel class (qf_fill_class_name): launch duration, cells and bands it processed */
  float ms_fill_class[QF_MAX_FILL_CLASSES];
  uint64_t cells_class[QF_MAX_FILL_CLASSES];
  uint32_t units_class[QF_MAX_FILL_CLASSES];
  uint32_t n_fill_classes;
} qf_align_result;

/* ---- context ---------------------------------------------------------------------- */
/* Number of HIP devices visible to this process (0 if none or on error).  The reference spreads tasks over `-threads`
 * host threads (src/qmodel.cpp:2870-2882); a caller of this library spreads read batches over one context per device. */
int qf_device_count(void);
int qf_ctx_create(int device_id, qf_ctx **ctx);
void qf_ctx_destroy(qf_ctx *ctx);
const char *qf_last_error(const qf_ctx *ctx);   /* ctx may be NULL: last creation error */
int qf_device_name(const qf_ctx *ctx, char *buf, size_t cap);
/* PCI bus id of the context's device ("0000:c1:00.0"): what tells two ranks' GPUs apart when a job reports how many distinct
 * devices its ranks really ran on (bench.py's `rccl` object; the reference has no counterpart: its workers are host threads). */
int qf_device_bus_id(const qf_ctx *ctx, char *buf, size_t cap);
/* Alignments whose null-adjusted score is below `min_score` are not traced back and not returned by qf_align_* /
 * qf_overlap_resident (the per-pair score arrays are still complete).  This is the reference printer's `-threshold`
 * (QuaffAlignmentPrinter, src/qmodel.cpp:2480-2600; default there 0, `-nothreshold` = -inf) applied before the traceback
 * instead of after it: most read pairs of an all-vs-all overlap run do not overlap and score below 0.  In best-per-read
 * mode the read's best alignment is chosen first and then tested, as the reference does.  Default: -inf (keep all). */
int qf_set_score_threshold(qf_ctx *ctx, double min_score);
/* Device bytes one internal chunk may use for traceback / Forward storage.  Default (and 0): what hipMemGetInfo reports
 * free at the time of the call.  Larger batches are processed in halves transparently, and so is a chunk whose allocation
 * fails (another context or process on the same GPU).  (The reference bounds DP memory per thread through -kmatchmb /
 * -kmatchmax, src/qmodel.cpp:788-813,1058-1060, and runs reads one at a time; here whole batches are resident, so the
 * bound is on the batch.) */
int qf_set_memory_budget(qf_ctx *ctx, uint64_t bytes);
/* qf_align_* cuts a batch into this many pieces and keeps two in flight (one piece's seeding and traceback overlap the
 * other's fill).  0 = default (1: the whole batch at once).  Results do not depend on it. */
int qf_set_pipeline_chunks(qf_ctx *ctx, uint32_t n_chunks);

/* ---- model ------------------------------------------------------------------------ */
/* Parse a quaff params JSON document (numbers go through the same non-correctly-rounded
 * decimal algorithm as the reference's gason, src/gason.cpp:73-117), build the log-space
 * score tables and upload them.  text == NULL selects the built-in defaults. */
int qf_set_params_json(qf_ctx *ctx, const char *text);
/* Same from in-memory values (EM iterations keep full doubles; quaff's JSON writer prints 6 significant figures):
 *   begin_insert/begin_delete[4^gap_len]; insert_pqr[4][3], match_pqr[4][4^match_len][3] = (p, q, r) of each SymQualDist,
 *   match indexed [reference base][read context k-mer]; ref_base[4] (only the overlap model reads it). */
int qf_set_params_raw(qf_ctx *ctx, uint32_t match_len, uint32_t gap_len, const double *ref_base,
                      const double *begin_insert, const double *begin_delete, double extend_insert, double extend_delete,
                      const double *insert_pqr, const double *match_pqr);
int qf_set_null_raw(qf_ctx *ctx, double null_emit, const double *null_pqr /* [4][3] */);
/* Score tables as built (for inspection / parity tests).  Any pointer may be NULL.
 *   ins[4][95], mat[4][Km][95] (index 94 = quality-marginalised), trans[4*Kg+4] =
 *   m2m[Kg] m2i[Kg] m2d[Kg] m2e[Kg] d2d d2m i2i i2m. */
int qf_get_scores(const qf_ctx *ctx, int *match_len, int *gap_len, double *ins, double *mat, double *trans);
/* Host-only variant of the two calls above (no context, no device): parse + build the tables. */
int qf_scores_from_json(const char *text, int *match_len, int *gap_len, double *ins, double *mat, double *trans,
                        char *err, size_t err_cap);
/* Kernel name of a fill class, e.g. "k_viterbi_fill<16,5>"; NULL past the last class. */
const char *qf_fill_class_name(uint32_t cls);
/* Null model: JSON as written by quaff -savenull; text == NULL clears it (null_loglike = 0). */
int qf_set_null_json(qf_ctx *ctx, const char *text);
/* The 100001-entry log(1+exp(-x)) table of src/logsumexp.cpp:20-28 as built on the host. */
int qf_get_lse_table(const qf_ctx *ctx, const double **table, int *n);

/* ---- sequences -------------------------------------------------------------------- */
/* Reference set (already including reverse complements if wanted, as SeqList::loadSequences
 * does, t/quaff.cpp:610-636).  seq = concatenated characters, offsets[n_refs+1]. */
int qf_set_refs(qf_ctx *ctx, const char *seq, const uint64_t *offsets, uint32_t n_refs);

/* Reads resident in HBM.  qual == NULL => no quality scores (-noquals). */
int qf_upload_reads(qf_ctx *ctx, const char *seq, const char *qual, const uint64_t *offsets, uint32_t n_reads);

/* ---- the hot path ------------------------------------------------------------------ */
/* Viterbi-align every uploaded read against every reference. */
int qf_align_resident(qf_ctx *ctx, const qf_dp_config *cfg, uint32_t flags, qf_align_result *out);
/* Convenience: qf_upload_reads + qf_align_resident. */
int qf_align_batch(qf_ctx *ctx, const qf_dp_config *cfg, const char *seq, const char *qual,
                   const uint64_t *offsets, uint32_t n_reads, uint32_t flags, qf_align_result *out);

/* ---- Forward-Backward E-step (quaff train / count) -------------------------------------- */
#define QF_COUNT_FORCE 1u     /* -force: no null model in the per-read normalisation (yLogLike starts at -inf) */

typedef struct qf_count_result {
  uint32_t n_reads, n_refs;
  const double *forward;        /* [n_reads * n_refs] Forward log-likelihood; -inf for references not in the order */
  const double *weight;         /* [n_reads * n_refs] posterior weight exp(LL - yLogLike) of pairs that got a Backward pass, else 0 */
  const double *read_loglike;   /* [n_reads] yLogLike = lse(null, LL_x ...) */
  const uint32_t *sort_order;   /* [n_reads * n_refs] next iteration's reference order per read ... */
  const uint32_t *sort_count;   /* [n_reads]          ... and how many entries of it are valid */
  /* flattened QuaffParamCounts summed over the batch (qf_counts_size doubles):
   *   insert[4][94] | match[4][Km][94] | beginInsertNo[Kg] beginInsertYes[Kg] beginDeleteNo[Kg] beginDeleteYes[Kg]
   *   | extendInsertNo extendInsertYes extendDeleteNo extendDeleteYes */
  const double *counts;
  uint32_t counts_size;
  double loglike;               /* sum over reads of yLogLike (QuaffCountingScheduler::finalLogLike) */
  uint64_t total_cells;         /* envelope cells of the pairs that were seeded (each visited by Forward, and by Backward if weighted) */
  uint64_t backward_cells;
  uint64_t forward_bytes;       /* Forward matrix bytes materialised */
  float ms_prep, ms_seed, ms_forward, ms_plan, ms_backward, ms_total;
  /* per fill-kernel class (qf_fill_class_name gives the Viterbi name of the same (lanes, diagonals-per-lane) geometry):
   * launch durations of the Forward and the Backward kernel, cells and bands the class holds */
  float ms_forward_class[QF_MAX_FILL_CLASSES], ms_backward_class[QF_MAX_FILL_CLASSES];
  uint64_t cells_class[QF_MAX_FILL_CLASSES];
  uint32_t units_class[QF_MAX_FILL_CLASSES];
  uint32_t n_fill_classes;
  /* The same sums as 128-bit fixed point (64 fractional bits, two's complement), (low, high) 64-bit words per value: value =
   * high + low / 2^64.  The device adds the count terms as integers, so these words -- and `counts`, which is their
   * conversion -- are the same whatever order the terms were added in: run to run, and however the batch was cut into pieces
   * inside the call.  The words of several calls add exactly (qf_exact_add) and convert once (qf_exact_to_double): totals
   * then do not depend on how the reads were split over calls, contexts or GPUs either (qf_allreduce_counts_exact).  The
   * reference adds per-read counts in read order (src/qmodel.cpp:2416-2422), i.e. with one fixed rounding sequence; this is
   * another fixed one. */
  const uint64_t *counts_exact;   /* [counts_size][2] */
  uint64_t loglike_exact[2];      /* the sum of read_loglike, same format; the "not finite" marker if a read has no finite term */
} qf_count_result;

/* One E-step over the resident reads.  sort_in / sort_n_in (optional, [n_reads*n_refs] / [n_reads]) give each
 * read's reference order from the previous EM iteration (defaultSortOrder = all references in index order). */
int qf_count_resident(qf_ctx *ctx, const qf_dp_config *cfg, uint32_t flags, const uint32_t *sort_in,
                      const uint32_t *sort_n_in, qf_count_result *out);
uint32_t qf_counts_size(const qf_ctx *ctx);

/* ---- E-step reduction across GPUs (quaff train on several devices) ------------------------- */
/* The reference sums the counts and log-likelihoods of all worker threads after every E-step
 * (QuaffCountingScheduler::finalCounts / finalLogLike, src/qmodel.cpp:2416-2422).  With the reads sharded over GPUs that sum
 * is one RCCL all-reduce(sum, fp64) over xGMI of the flattened counts (qf_count_result.counts) and the log-likelihood.  One
 * rank per GPU.  RCCL is loaded at the first qf_comm_* call (an RCCL already in the process, e.g. PyTorch's, is shared).
 *   several processes: rank 0 calls qf_comm_unique_id, hands the 128 bytes to the other ranks by whatever channel launched
 *                      them, and every rank calls qf_comm_init_rank (collective: returns when all n_ranks have called it);
 *   one process:       qf_comm_init_all over one context per device (the `quaff train -gpus N` shell), then one host thread
 *                      per context calls qf_allreduce_counts. */
#define QF_COMM_ID_BYTES 128
int qf_comm_unique_id(uint8_t *id /* [QF_COMM_ID_BYTES] */);
int qf_comm_init_rank(qf_ctx *ctx, const uint8_t *id, int rank, int n_ranks);
int qf_comm_init_all(qf_ctx *const *ctxs, int n);
int qf_comm_size(const qf_ctx *ctx);   /* ranks of the context's communicator; 0 without one */
void qf_comm_destroy(qf_ctx *ctx);     /* also done by qf_ctx_destroy */
/* In place over host arrays: counts[n] (and *loglike, if not NULL) become the sums over all ranks.  Collective: every rank
 * calls it with the same n.  Summation order inside RCCL depends on the rank count, so sums agree between runs on
 * different numbers of GPUs to rounding (1e-15 relative), not bit for bit. */
int qf_allreduce_counts(qf_ctx *ctx, double *counts, uint32_t n, double *loglike);

/* 128-bit fixed-point sums (qf_count_result.counts_exact / loglike_exact): acc[k] += add[k] for n values of two words each;
 * conversion to / from double (values beyond +-9.2e18 and non-finite ones become a marker that absorbs every sum and converts to
 * -inf).  Host arithmetic, no context. */
void qf_exact_add(uint64_t *acc, const uint64_t *add, uint32_t n);
void qf_exact_to_double(const uint64_t *fx, uint32_t n, double *out);
void qf_exact_from_double(const double *v, uint32_t n, uint64_t *fx);
/* The E-step reduction on the exact words (n values of two words, e.g. counts_exact followed by loglike_exact), in place:
 * every rank ends with the same 128-bit totals, identical for any number of ranks.  Collective, like qf_allreduce_counts. */
int qf_allreduce_counts_exact(qf_ctx *ctx, uint64_t *fx, uint32_t n);

/* ---- read-vs-read overlap (quaff overlap) -------------------------------------------------- */
typedef struct qf_overlap_alignment {
  uint32_t pair;             /* index into the pair list */
  double viterbi;            /* QuaffOverlapViterbiMatrix::result (end + both insert scores) */
  double score;              /* result - null(x) - null(y), scoreAdjustedAlignment */
  uint32_t x_start, x_end, y_start, y_end;  /* 1-based closed intervals */
  uint32_t n_columns, n_runs;
  uint64_t run_offset;       /* first run in qf_overlap_result.state_runs */
} qf_overlap_alignment;

typedef struct qf_overlap_result {
  uint32_t n_pairs;
  const double *viterbi;         /* [n_pairs] -inf where no path exists */
  const double *score;           /* [n_pairs] */
  const uint64_t *cells;         /* [n_pairs] */
  const uint32_t *n_diagonals;   /* [n_pairs] */
  uint64_t total_cells;
  uint32_t n_alignments;         /* pairs with a finite result, ordered by pair index */
  const qf_overlap_alignment *alignments;
  /* traceback STATE runs (length << 2 | state, 0=M 1=I 2=D) in alignment order, before the indel squashing of
   * src/qoverlap.cpp:231-267 (a host-side re-pairing of adjacent I/D runs when the rows are written) */
  const uint32_t *state_runs;
  float ms_prep, ms_seed, ms_fill, ms_traceback, ms_total;
  uint64_t traceback_bytes;
  float ms_fill_class[QF_MAX_FILL_CLASSES];   /* per fill-kernel class, as in qf_align_result */
  uint64_t cells_class[QF_MAX_FILL_CLASSES];
  uint32_t units_class[QF_MAX_FILL_CLASSES];
  uint32_t n_fill_classes;
} qf_overlap_result;

/* Align pairs of the resident sequences (qf_upload_reads: originals followed, if wanted, by their reverse complements,
 * as SeqList::loadSequences builds them).  pair_x / pair_y index the resident set; y_complemented[p] != 0 tells the
 * scorer that y is a reverse-complemented read (QuaffOverlapScheduler: ny >= nOriginals).  At most 2^28 pairs per call
 * (as for reads x references in qf_align_* / qf_count_resident); callers feed longer lists in blocks. */
int qf_overlap_resident(qf_ctx *ctx, const qf_dp_config *cfg, const uint32_t *pair_x, const uint32_t *pair_y,
                        const uint8_t *y_complemented, uint32_t n_pairs, qf_overlap_result *out);

/* ---- quaff overlap, the scheduler's own enumeration (rows of the pair triangle) ------------------------------------
 * QuaffOverlapScheduler (src/qoverlap.cpp:457-480,528-547) walks nx = 0 ... nOriginals - 2 and, for each nx,
 * ny = nx + 1 ... y.size() - 1 over the originals followed by their reverse complements (yComplemented = ny >= nOriginals),
 * one QuaffOverlapTask per (nx, ny); the printer keeps what scores at least -threshold (src/qmodel.cpp:2566-2569).
 * qf_overlap_rows does rows [x0, x1) of that enumeration on the device: the (nx, ny, yComplemented) triples are generated
 * there, the score threshold (qf_set_score_threshold) is applied there, and only the alignments that pass come back, in the
 * scheduler's order, with totals over all the row block's pairs.  Nothing per pair crosses PCIe in either direction: an
 * all-vs-all run of N reads has ~N^2 pairs, of which a fraction ~coverage/N overlap.  Row blocks are independent, so
 * several GPUs take disjoint [x0, x1) (rows are not equally long: row nx has n_seqs - 1 - nx pairs). */
typedef struct qf_overlap_hit {
  uint32_t x, y;             /* nx, ny: indices into the resident set (y >= n_originals: a reverse complement) */
  double viterbi;            /* QuaffOverlapViterbiMatrix::result */
  double score;              /* result - null(x) - null(y) */
  uint32_t x_start, x_end, y_start, y_end;  /* 1-based closed intervals */
  uint32_t n_columns, n_runs;
  uint64_t run_offset;       /* first run in qf_overlap_rows_result.state_runs */
} qf_overlap_hit;

typedef struct qf_overlap_rows_result {
  uint32_t x0, x1;
  uint64_t n_pairs;            /* pairs enumerated: sum over nx in [x0, x1) of n_seqs - 1 - nx */
  uint64_t n_finite;           /* pairs with a finite Viterbi result (every one of them is a candidate for the printer) */
  uint64_t total_cells;        /* DP cells visited (SURVEY 8d definition) */
  uint64_t total_diagonals;    /* envelope diagonals, summed over the pairs */
  uint64_t result_checksum;    /* sum over the finite pairs of the bit pattern of `viterbi`, mod 2^64: independent of order
                                  and blocking, equal between this entry point and qf_overlap_resident on the same pairs */
  uint32_t n_hits;             /* alignments scoring >= the threshold, ordered by (x, y) */
  const qf_overlap_hit *hits;
  const uint32_t *state_runs;  /* as qf_overlap_result.state_runs */
  uint32_t n_blocks;           /* row blocks the call was cut into (device memory, 2^28-pair tables) */
  float ms_prep, ms_seed, ms_fill, ms_traceback, ms_total;   /* device-side, summed over the blocks */
  uint64_t traceback_bytes;
  float ms_fill_class[QF_MAX_FILL_CLASSES];
  uint64_t cells_class[QF_MAX_FILL_CLASSES];
  uint32_t units_class[QF_MAX_FILL_CLASSES];
  uint32_t n_fill_classes;
} qf_overlap_rows_result;

/* The resident set (qf_upload_reads) is n_originals reads, optionally followed by their reverse complements (n_seqs =
 * n_originals or 2 n_originals, as SeqList::loadSequences builds it).  Rows: 0 <= x0 <= x1 <= n_originals - 1 (the scheduler
 * stops when nx + 1 == nOriginals, src/qoverlap.cpp:520-522).  Results stay valid until the next overlap call on ctx. */
int qf_overlap_rows(qf_ctx *ctx, const qf_dp_config *cfg, uint32_t n_originals, uint32_t x0, uint32_t x1,
                    qf_overlap_rows_result *out);
/* Pairs in rows [x0, x1) of that enumeration over n_seqs resident sequences (host arithmetic; for splitting the triangle). */
uint64_t qf_overlap_rows_pairs(uint32_t n_seqs, uint32_t x0, uint32_t x1);

/* Envelope only (DiagonalEnvelope::diagonals for pair (read, ref)); returns the number of
 * diagonals, writes at most cap of them.  For tests and debugging. */
int64_t qf_envelope(qf_ctx *ctx, const qf_dp_config *cfg, uint32_t read, uint32_t ref, int32_t *diags, uint64_t cap);

/* Alignment::cigarString (letter BEFORE count, src/qmodel.cpp:625-653); returns length needed. */
size_t qf_cigar_string(const uint32_t *runs, uint32_t n_runs, char *buf, size_t cap);

/* ---- synthetic workloads (SURVEY 8d generator; deterministic, host side) ----------- */
/* i.i.d. uniform ACGT reference of ref_len bases (xoshiro256**, seed). */
int qf_synth_ref(uint64_t seed, uint64_t ref_len, char *seq);
/* n_reads reads of read_len SOURCE bases sampled from ref, odd reads reverse-complemented,
 * 3% del / 3% ins / 5% sub, Phred 5..25.  seq/qual need capacity n_reads*(2*read_len);
 * offsets[n_reads+1]. */
int qf_synth_reads(uint64_t seed, const char *ref, uint64_t ref_len, uint32_t n_reads, uint32_t read_len,
                   char *seq, char *qual, uint64_t *offsets);

#ifdef __cplusplus
}
#endif
#endif /* QUAFF_HIP_H */
