// qf_api.hip — the C ABI of include/quaff_hip.h: context, device memory, batch orchestration.
// No kernels here (qf_kernels.hip) and no model arithmetic (qf_model.cpp).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/quaff_hip.h"
#include "qf_internal.h"
#include "qf_kernels.hpp"
#include "qf_model.hpp"

using namespace qf;

namespace {
std::string g_create_error;

// microseconds this process has spent in hipFree / hipMalloc growing device buffers (qf_debug_alloc_ms: the CLI's timing line)
static std::atomic<uint64_t> g_alloc_us{0};
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  template <class T> T* as() const { return (T*)p; }
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    const auto t0 = std::chrono::steady_clock::now();
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {  // retry exact
      want = bytes;
      e = hipMalloc(&p, want);
    }
    if (e == hipSuccess) cap = want; else p = nullptr;
    g_alloc_us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Host-side result array in pinned memory (device-to-host copies into pageable memory run at a fraction of the link rate).
// The little of std::vector the result code uses; contents survive growth.
template <typename T>
struct HostBuf {
  T* p = nullptr;
  size_t n = 0, cap = 0;
  bool pinned = false;
  HostBuf() = default;
  HostBuf(const HostBuf&) = delete;
  HostBuf& operator=(const HostBuf&) = delete;
  ~HostBuf() { release(); }
  void release() {
    if (p) { if (pinned) (void)hipHostFree(p); else free(p); }
    p = nullptr; n = cap = 0;
  }
  void reserve(size_t want) {
    if (want <= cap) return;
    const size_t ncap = std::max(want, cap + cap / 2 + 64);
    T* q = nullptr;
    bool pin = hipHostMalloc((void**)&q, ncap * sizeof(T), hipHostMallocDefault) == hipSuccess;
    if (!pin) q = (T*)malloc(ncap * sizeof(T));
    if (n) memcpy(q, p, n * sizeof(T));
    const size_t keep = n;
    release();
    p = q; n = keep; cap = ncap; pinned = pin;
  }
  void resize(size_t m) { reserve(m); n = m; }
  void clear() { n = 0; }
  size_t size() const { return n; }
  T* data() { return p; }
  const T* data() const { return p; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  T* begin() { return p; }
  T* end() { return p + n; }
  void append(const T* src, size_t m) { reserve(n + m); if (m) memcpy(p + n, src, m * sizeof(T)); n += m; }
};
}  // namespace

// One in-flight chunk of a batch: its streams, events, per-chunk device buffers and result staging.  The context is
// itself slot 0 (every single-chunk path uses it); align batches run two slots from two host threads so that one
// chunk's seeding and traceback (latency-bound) overlap the other's fill (VALU-bound).
struct Slot {
  hipStream_t stream = nullptr;
  hipEvent_t ev[8] = {};   // [6]: pair results final (their copy to the host overlaps selection and traceback); [7]: class lists final
  hipEvent_t cls_ev[kNumClasses] = {}, cls_end[kNumClasses] = {};  // per fill class, on the stream the class runs on
  hipEvent_t cls_ev2[kNumClasses] = {}, cls_end2[kNumClasses] = {};   // E-step: the Backward kernels (cls_ev / cls_end time Forward)
  hipEvent_t ev_split[2] = {};                                         // E-step: the dominant class's Backward in two parts (count_chunk)
  hipStream_t aux[3] = {};                                         // side streams: fill classes run concurrently
  hipStream_t hi[3] = {};                                          // high priority: classes too small to fill the chip (latency-bound chains)
  DevBuf d_units, d_cls_list, d_pair_head, d_pair_bands, d_pair_nbands, d_ovf, d_pair_ndiag, d_pair_cells, d_pair_score,
      d_pair_end_unit, d_bc, d_tb, d_recs, d_runs_tmp, d_runs_out, d_seed_ws, d_align_out, d_cls_key, d_sort_k, d_sort_v,
      d_fw, d_weight, d_fwd_out, d_order_out, d_order_n_out, d_rll;   // E-step: Forward storage and per-pair / per-read results of the chunk
  void* sort_tmp = nullptr;
  size_t sort_tmp_cap = 0;
  HostBuf<AlignRec> h_recs;
  HostBuf<uint32_t> h_runs;
  std::string err;

  int create() {
    if (hipStreamCreate(&stream) != hipSuccess) return 1;
    for (auto& e : ev) (void)hipEventCreate(&e);
    for (auto& e : cls_ev) (void)hipEventCreate(&e);
    for (auto& e : cls_end) (void)hipEventCreate(&e);
    for (auto& e : cls_ev2) (void)hipEventCreate(&e);
    for (auto& e : cls_end2) (void)hipEventCreate(&e);
    for (auto& e : ev_split) (void)hipEventCreate(&e);
    // side streams at the lowest priority: the small fill classes they carry should fill the gaps of the dominant class
    // (main stream), not compete with it
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    for (auto& s : aux) (void)hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least);
    for (auto& s : hi) (void)hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest);
    return 0;
  }
  void destroy() {
    for (DevBuf* b : {&d_units, &d_cls_list, &d_pair_head, &d_pair_bands, &d_pair_nbands, &d_ovf, &d_pair_ndiag, &d_pair_cells,
                      &d_pair_score, &d_pair_end_unit, &d_bc, &d_tb, &d_recs, &d_runs_tmp, &d_runs_out, &d_seed_ws, &d_align_out, &d_cls_key, &d_sort_k, &d_sort_v,
                      &d_fw, &d_weight, &d_fwd_out, &d_order_out, &d_order_n_out, &d_rll})
      b->release();
    if (sort_tmp) (void)hipFree(sort_tmp);
    sort_tmp = nullptr;
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : cls_ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : cls_end) if (e) (void)hipEventDestroy(e);
    for (auto& e : cls_ev2) if (e) (void)hipEventDestroy(e);
    for (auto& e : cls_end2) if (e) (void)hipEventDestroy(e);
    for (auto& e : ev_split) if (e) (void)hipEventDestroy(e);
    for (auto& s : aux) if (s) (void)hipStreamDestroy(s);
    for (auto& s : hi) if (s) (void)hipStreamDestroy(s);
    if (stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
  }
};

struct qf_ctx : Slot {
  int device = 0;
  Slot second;             // created on first use
  bool second_ready = false;
  hipEvent_t ev_tok = nullptr, ev_nll = nullptr;   // read tokens ready / null log-likelihoods ready
  uint32_t pipeline_chunks = 0;  // 0 = automatic
  bool ragged_reads = false;     // resident read lengths differ by more than 25 %: class lists are sorted by length
  uint32_t debug = 0;            // qf_debug_set_flags (qf_internal.h): test / A-B switches, never set by the product
  std::string devname;
  // model
  Params params;
  Scores scores;
  bool have_params = false;
  NullParams null;
  bool have_null = false;
  DevBuf d_ematch, d_eins, d_trans, d_nullq;
  // references
  uint32_t n_refs = 0;
  std::vector<uint64_t> ref_off, ref_woff;
  uint64_t ref_total = 0, ref_maxlen = 0;
  DevBuf d_ref_seq, d_ref_tok, d_ref_off, d_ref_woff, d_ref_packed, d_bucket, d_cursor, d_pos;
  int index_k = 0;
  // reads
  uint32_t n_reads = 0;
  std::vector<uint64_t> read_off;
  uint64_t read_total = 0, read_maxlen = 0;
  bool reads_have_qual = false;
  uint32_t read_qmin = 0, read_qmax = 93;   // quality values the resident reads use (device scan at upload)
  DevBuf d_seq, d_qual, d_roff, d_tok, d_ctx, d_skmer, d_nll, d_qrange;
  // E-step: the match-emission rows of qualities read_qmin .. read_qmax, quality-major (PrepArgs::em_qmajor_Km), + a -inf row
  DevBuf d_ematch_q;
  uint64_t ematch_q_params_epoch = 0;
  uint32_t ematch_q_lo = 1, ematch_q_hi = 0;
  uint64_t params_epoch = 0;
  uint32_t prep_qmajor_Km = 0;     // layout the next prep_reads gives the context words' emission rows (0 = k-mer major)
  bool count_qmajor = false;       // the running E-step uses the quality-major slice
  // batch state
  DevBuf d_cover, d_lse, d_counts, d_order_in, d_order_n_in,
      d_skip, d_ctxc, d_ins_sum, d_ins_sum_c, d_nll_c, d_rbucket, d_rcursor, d_rpos,
      d_px, d_py, d_pc, d_mmi0, d_mmi1, d_gap0, d_gap1, d_pair_result, d_pair_ij, d_skmer64, d_skeys, d_keys_tmp, d_vals_tmp,
      d_off32, d_rskeys, d_roff32;
  void* sort_temp = nullptr;
  size_t sort_temp_cap = 0;
  // host results
  HostBuf<double> h_viterbi, h_nll;
  HostBuf<uint64_t> h_cells;
  HostBuf<uint32_t> h_ndiag;
  HostBuf<qf_alignment> h_align;
  struct DenseChunk { uint32_t lo, hi; size_t runs0; bool second; };
  std::vector<DenseChunk> dense_chunks;   // best-per-read mode: where each chunk's records and runs went
  std::vector<double> h_fwd, h_weight, h_rll, h_pcounts;
  std::vector<uint64_t> h_counts_fx, h_pcounts_fx;   // E-step counts as 128-bit fixed point, (low, high) per entry: raw QuaffCounts layout / QuaffParamCounts layout
  std::vector<uint32_t> h_order, h_order_n;
  bool lse_uploaded = false;
  DevBuf d_lse_pack;            // the exact table packed for LDS (qf_device.hpp: kLsePack*); 0 bytes: not usable on this device
  uint32_t lse_pack_bytes = 0;
  double min_score = -INFINITY;   // qf_set_score_threshold
  uint64_t tb_budget = 0;   // qf_set_memory_budget: per-chunk device storage budget (traceback / Forward matrices); 0 = what the device has free
  bool ov_scores[2] = {false, false};
  int read_index_k = 0;
  // the reads' derived arrays (tokens, context words, insert sums, null log-likelihoods) as the overlap path leaves them
  // stay valid while reads, parameters, null model and k are unchanged and no other entry point re-derives them
  uint64_t prep_epoch = 1, ov_prep_epoch = 0;
  int ov_prep_k = -1;
  // row prefilter of the overlap seeding (qf_kernels.hip: k_seed_rows): runs (x, y0), (x, y0 + 1), ... of the current pair list
  // and the k-mer index of chunks of 2^chunk_log2 consecutive sequences
  struct PairRow { uint32_t x, y0, n, p0; };
  std::vector<PairRow> ov_rows;
  bool ov_use_rows = false;
  DevBuf d_cstart, d_ccursor, d_centries, d_cbounds, d_row_items, d_row_skip, d_slot_list;
  std::vector<PairRow> row_items_rows;   // the runs / block / chunk size d_row_items was built for
  uint32_t row_items_lo = 0, row_items_hi = 0, row_items_n = 0;
  int row_items_cl = -1;
  bool row_items_lds = false;
  uint64_t row_items_epoch = 0;
  uint64_t rows_settled = 0;    // QF_DEBUG_COUNT_SETTLED: pairs the prefilter settled during the last qf_overlap_resident
  uint64_t chunk_epoch = 0;
  int chunk_k = -1, chunk_log2 = 0;
  uint32_t chunk_span = 0;       // seed_row_entry_span the index entries were built with
  uint64_t chunk_estride = 0;    // > 0: the chunk index is the padded one of k_seed_rows_lds
  int chunk_pb = 0;              // > 0: 16-bit index entries, sequence << chunk_pb | position
  // k_seed_rows_lds (the prefilter with the chunk's index in LDS): whether the current chunk size was chosen for it, 16-bit entries,
  // the most k-mer positions a chunk holds; its chunk-major items (RowItemL) and pieces
  bool row_lds = false, row_e16 = false;
  uint64_t row_max_entries = 0;
  DevBuf d_row_sorted, d_row_pieces;
  uint32_t row_pieces_n = 0, row_tri_x0 = 0;
  bool row_tri = false;          // the pieces are (first row, rows, chunk) of the scheduler's triangle, not pieces of d_row_sorted
  HostBuf<double> h_ov_result, h_ov_score;
  std::vector<uint32_t> h_ov_slot;
  std::vector<qf_overlap_alignment> h_ov_align;
  // qf_overlap_rows: no per-pair arrays come back; totals from the device's counters, hits accumulated over the row blocks
  bool ov_per_pair = true;
  bool ov_slot_collision = false;   // set by a chunk whose slotted single-diagonal list had a collision: plain lists for the rest of the call
  struct OvTotals { uint64_t n_finite = 0, sum_ndiag = 0, result_sum = 0; } ov_tot;
  // k_overlap_single_rows: compact pair-emission tables per strand flag, per-base row offsets, transposed column offsets
  DevBuf d_mmic0, d_mmic1, d_xrowoff, d_ycol0, d_ycol1, d_ygoff;
  uint32_t ov_pitch = 0, ov_cpr = 0;      // row pitch (doubles) / 16-byte chunks per row; 0: the compact form does not apply
  uint64_t ov_cols_epoch = 0, ov_mmic_epoch[2] = {0, 0};
  uint64_t ov_block_pairs = 0;   // qf_debug_set_overlap_block_pairs: pairs per internal row block (0 = default)
  std::vector<qf_overlap_hit> h_hits;
  std::vector<uint32_t> h_hit_runs;
  // E-step reduction over RCCL (qf_comm_*): communicator, this context's rank, a device staging buffer
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_size = 1;
  DevBuf d_comm;
  HostBuf<uint64_t> h_comm;       // pinned staging of the all-reduce (doubles or limbs)
  hipEvent_t ev_comm = nullptr;   // recorded behind the collective: the deadline wait polls it
};

#define HIPCHK(ctx, call)                                                                       \
  do {                                                                                          \
    hipError_t _e = (call);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(_e);                           \
      return QF_ERR_DEVICE;                                                                     \
    }                                                                                           \
  } while (0)

// A per-chunk device buffer (unit tables, alignment records, run lists, sort keys ...).  When it cannot be had the chunk function
// returns kSplitChunk: its wrapper waits for what the abandoned chunk still has in flight, releases the big buffers of the entry
// points that are not running, and has the caller cut the chunk in two -- the way the traceback / Forward budget does -- unless it
// is one read or pair already (QF_ERR_MEMORY).  qf_debug_fail_chunk_reserve(n): the n-th such reserve from now fails once (tests).
constexpr int kSplitChunk = -4242;
static std::atomic<int> g_fail_chunk_reserve{-1};
static hipError_t chunk_reserve(DevBuf& buf, size_t bytes) {
  if (g_fail_chunk_reserve.load() >= 0 && g_fail_chunk_reserve.fetch_sub(1) == 0) return hipErrorOutOfMemory;
  const hipError_t e = buf.reserve(bytes);
  if (e != hipSuccess) (void)hipGetLastError();
  return e;
}
#define CHUNKRES(ctx, buf, bytes)                                      \
  do {                                                                 \
    if (chunk_reserve((buf), (bytes)) != hipSuccess) {                 \
      (ctx)->err = "out of device memory (" #buf ")";                  \
      return kSplitChunk;                                              \
    }                                                                  \
  } while (0)

constexpr uint64_t kMaxPairsPerCall = 1ull << 28;   // unit tables are sized 4 x pairs + slack in 32 bits
constexpr size_t kLseHermiteOffset = 100002;  // doubles: the exact table (100001) padded to even, then the quadratic pieces

// One Viterbi fill at a time per device, whatever context or slot it comes from: two fills side by side only share the
// fp64 issue slots (each takes twice as long), while a fill next to another batch's seeding, selection, traceback and result
// copies (latency-bound, little arithmetic) costs neither much.  Callers keep several batches in flight with one context
// and host thread each (bench.py --inflight, the CLI's -gpus sharding); the token orders their fills.
static std::mutex& device_fill_token(int device) {
  static std::mutex tokens[64];
  return tokens[device & 63];
}

static uint32_t sc_Km(const qf_ctx* c) { return c->scores.Km; }

static int fail(Slot* c, int code, const std::string& msg) {
  c->err = msg;
  return code;
}

// Calls of the batch entry points in progress on each device: callers keep a few contexts per GPU, one host thread each.
static std::atomic<int>& device_calls(int device) {
  static std::atomic<int> n[64];
  return n[device & 63];
}
struct CallInProgress {
  int device;
  explicit CallInProgress(int d) : device(d) { ++device_calls(d); }
  ~CallInProgress() { --device_calls(device); }
};

// Bytes one chunk may spend on its traceback / Forward storage: the caller's figure (qf_set_memory_budget), else this call's
// share of what the device has free right now -- the calls in progress on a device (bench --inflight, QUAFF_HIP_DEVICES=0,0) read
// that figure at the same moment, and each claiming all of it would send all but one into the allocation-failure path -- plus what
// `own` already holds (it is reused), less 1/16 for the per-pair arrays of the batch.
static uint64_t chunk_budget(const qf_ctx* c, const DevBuf& own, int slots_in_flight) {
  if (c->tb_budget) return c->tb_budget / (uint64_t)slots_in_flight;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return (16ull << 30) / (uint64_t)slots_in_flight;
  const uint64_t share = (uint64_t)free_b / (uint64_t)std::max(1, device_calls(c->device).load());
  const uint64_t avail = share + own.cap;
  return (avail - avail / 16) / (uint64_t)slots_in_flight;
}

// The big per-chunk buffer (traceback words or Forward matrices).  Buffers never shrink by themselves, so the other entry
// points' big buffers may still hold most of the device: on failure they are released and the allocation is tried again.
// (`idle`: the big buffers no chunk in flight is using.)
static hipError_t reserve_big(DevBuf& buf, size_t bytes, std::initializer_list<DevBuf*> idle) {
  hipError_t e = buf.reserve(bytes);
  if (e == hipSuccess) return e;
  static std::mutex mu;   // two slots can run out of memory at the same moment
  std::lock_guard<std::mutex> lk(mu);
  (void)hipGetLastError();
  for (DevBuf* other : idle)
    if (other != &buf) other->release();
  e = buf.reserve(bytes);
  if (e != hipSuccess) (void)hipGetLastError();
  return e;
}

extern "C" {

int qf_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
}

int qf_ctx_create(int device_id, qf_ctx** out) {
  if (!out) return QF_ERR_ARG;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_error = "no HIP device available (libquaffhip has no CPU fallback)";
    return QF_ERR_DEVICE;
  }
  if (device_id < 0 || device_id >= n) {
    g_create_error = "device id out of range";
    return QF_ERR_ARG;
  }
  if ((e = hipSetDevice(device_id)) != hipSuccess) {
    g_create_error = hipGetErrorString(e);
    return QF_ERR_DEVICE;
  }
  qf_ctx* c = new qf_ctx();
  c->device = device_id;
  if (const char* e = getenv("QUAFF_HIP_CHUNKS")) c->pipeline_chunks = (uint32_t)atoi(e);  // tuning aid; 0 = automatic
  if (const char* e = getenv("QUAFF_HIP_DEBUG_FLAGS")) c->debug = (uint32_t)strtoul(e, nullptr, 0);   // developer A/B (qf_internal.h)
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->devname = std::string(prop.name) + " (" + prop.gcnArchName + ")";
  if (c->create()) {
    g_create_error = "hipStreamCreate failed";
    delete c;
    return QF_ERR_DEVICE;
  }
  (void)hipEventCreateWithFlags(&c->ev_tok, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&c->ev_nll, hipEventDisableTiming);
  *out = c;
  return QF_OK;
}

void qf_ctx_destroy(qf_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (DevBuf* b : {&c->d_ematch, &c->d_eins, &c->d_trans, &c->d_nullq, &c->d_ref_seq, &c->d_ref_tok, &c->d_ref_off,
                    &c->d_ref_woff, &c->d_ref_packed, &c->d_bucket, &c->d_cursor, &c->d_pos, &c->d_seq, &c->d_qual,
                    &c->d_roff, &c->d_tok, &c->d_ctx, &c->d_skmer, &c->d_nll, &c->d_qrange, &c->d_ematch_q, &c->d_cover, &c->d_lse, &c->d_lse_pack, &c->d_cstart, &c->d_ccursor, &c->d_centries, &c->d_cbounds, &c->d_row_items, &c->d_row_skip, &c->d_slot_list, &c->d_row_sorted, &c->d_row_pieces,
                    &c->d_mmic0, &c->d_mmic1, &c->d_xrowoff, &c->d_ycol0, &c->d_ycol1, &c->d_ygoff,
                    &c->d_counts, &c->d_order_in, &c->d_order_n_in,
                    &c->d_skip, &c->d_ctxc, &c->d_ins_sum, &c->d_ins_sum_c, &c->d_nll_c,
                    &c->d_rbucket, &c->d_rcursor, &c->d_rpos, &c->d_px, &c->d_py, &c->d_pc, &c->d_mmi0, &c->d_mmi1,
                    &c->d_gap0, &c->d_gap1, &c->d_pair_result, &c->d_pair_ij, &c->d_skmer64, &c->d_skeys, &c->d_keys_tmp,
                    &c->d_vals_tmp, &c->d_off32, &c->d_rskeys, &c->d_roff32})
    b->release();
  qf_comm_destroy(c);
  if (c->sort_temp) (void)hipFree(c->sort_temp);
  if (c->ev_tok) (void)hipEventDestroy(c->ev_tok);
  if (c->ev_nll) (void)hipEventDestroy(c->ev_nll);
  if (c->ev_comm) (void)hipEventDestroy(c->ev_comm);
  if (c->second_ready) {
    (void)hipStreamSynchronize(c->second.stream);
    c->second.destroy();
  }
  c->destroy();
  delete c;
}

const char* qf_last_error(const qf_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int qf_device_name(const qf_ctx* c, char* buf, size_t cap) {
  if (!c || !buf || !cap) return QF_ERR_ARG;
  snprintf(buf, cap, "%s", c->devname.c_str());
  return QF_OK;
}

int qf_device_bus_id(const qf_ctx* c, char* buf, size_t cap) {
  if (!c || !buf || cap < 16) return QF_ERR_ARG;
  return hipDeviceGetPCIBusId(buf, (int)cap, c->device) == hipSuccess ? QF_OK : QF_ERR_DEVICE;
}

// ---------------------------------------------------------------------------------- model
static int install_params(qf_ctx* c, const Params& p);

int qf_set_params_json(qf_ctx* c, const char* text) {
  if (!c) return QF_ERR_ARG;
  Json j;
  std::string err;
  if (!parse_json(text ? text : kDefaultParamsJson, j, err)) return fail(c, QF_ERR_PARSE, err);
  Params p;
  if (!p.read_json(j, err)) return fail(c, QF_ERR_PARSE, err);
  return install_params(c, p);
}

int qf_set_params_raw(qf_ctx* c, uint32_t match_len, uint32_t gap_len, const double* ref_base, const double* begin_insert,
                      const double* begin_delete, double extend_insert, double extend_delete, const double* insert_pqr,
                      const double* match_pqr) {
  if (!c) return QF_ERR_ARG;
  if (!begin_insert || !begin_delete || !insert_pqr || !match_pqr) return fail(c, QF_ERR_ARG, "null parameter array");
  if (match_len < 1 || match_len > 4 || gap_len > 4) return fail(c, QF_ERR_UNSUPPORTED, "unsupported matchOrder/gapOrder (need 1..4 / 0..4)");
  Params p;
  p.match_len = match_len;
  p.gap_len = gap_len;
  p.resize();
  if (ref_base) for (int i = 0; i < 4; ++i) p.refBase[i] = ref_base[i];
  for (uint32_t g = 0; g < p.Kg(); ++g) { p.beginInsert[g] = begin_insert[g]; p.beginDelete[g] = begin_delete[g]; }
  p.extendInsert = extend_insert;
  p.extendDelete = extend_delete;
  for (int i = 0; i < 4; ++i) p.insert[i] = SymQualDist{insert_pqr[i * 3], insert_pqr[i * 3 + 1], insert_pqr[i * 3 + 2]};
  for (size_t m = 0; m < (size_t)4 * p.Km(); ++m) p.match[m] = SymQualDist{match_pqr[m * 3], match_pqr[m * 3 + 1], match_pqr[m * 3 + 2]};
  return install_params(c, p);
}

static int install_params(qf_ctx* c, const Params& p) {
  HIPCHK(c, hipSetDevice(c->device));
  c->params = p;
  c->scores.build(p);
  const Scores& s = c->scores;
  // device layout of the match table: [(kmer*95 + q)*4 + refTok] so the four reference-token variants
  // of one read column are adjacent
  std::vector<double> em((size_t)s.Km * kNQ1 * 4 + 4, -INFINITY);  // + one row of -inf (slots outside a band read it)
  for (uint32_t t = 0; t < 4; ++t)
    for (uint32_t k = 0; k < s.Km; ++k)
      for (int q = 0; q < kNQ1; ++q) em[((size_t)k * kNQ1 + q) * 4 + t] = s.mat[((size_t)t * s.Km + k) * kNQ1 + q];
  HIPCHK(c, c->d_ematch.reserve(em.size() * 8));
  HIPCHK(c, c->d_eins.reserve(s.ins.size() * 8));
  HIPCHK(c, c->d_trans.reserve(s.trans.size() * 8));
  HIPCHK(c, hipMemcpy(c->d_ematch.p, em.data(), em.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_eins.p, s.ins.data(), s.ins.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_trans.p, s.trans.data(), s.trans.size() * 8, hipMemcpyHostToDevice));
  c->have_params = true;
  c->ov_scores[0] = c->ov_scores[1] = false;
  ++c->prep_epoch;
  ++c->params_epoch;
  return QF_OK;
}

int qf_get_scores(const qf_ctx* c, int* match_len, int* gap_len, double* ins, double* mat, double* trans) {
  if (!c || !c->have_params) return QF_ERR_STATE;
  const Scores& s = c->scores;
  if (match_len) *match_len = (int)s.match_len;
  if (gap_len) *gap_len = (int)s.gap_len;
  if (ins) memcpy(ins, s.ins.data(), s.ins.size() * 8);
  if (mat) memcpy(mat, s.mat.data(), s.mat.size() * 8);
  if (trans) memcpy(trans, s.trans.data(), s.trans.size() * 8);
  return QF_OK;
}

int qf_scores_from_json(const char* text, int* match_len, int* gap_len, double* ins, double* mat, double* trans,
                        char* errbuf, size_t err_cap) {
  Json j;
  std::string err;
  Params p;
  if (!parse_json(text ? text : kDefaultParamsJson, j, err) || !p.read_json(j, err)) {
    if (errbuf && err_cap) snprintf(errbuf, err_cap, "%s", err.c_str());
    return QF_ERR_PARSE;
  }
  Scores s;
  s.build(p);
  if (match_len) *match_len = (int)s.match_len;
  if (gap_len) *gap_len = (int)s.gap_len;
  if (ins) memcpy(ins, s.ins.data(), s.ins.size() * 8);
  if (mat) memcpy(mat, s.mat.data(), s.mat.size() * 8);
  if (trans) memcpy(trans, s.trans.data(), s.trans.size() * 8);
  return QF_OK;
}

const char* qf_fill_class_name(uint32_t cls) {
  static std::string names[kNumClasses];
  if (cls >= (uint32_t)kNumClasses) return nullptr;
  if (names[cls].empty())
    names[cls] = cls == 0 ? std::string("k_viterbi_single") : cls == (uint32_t)kRowClass ? std::string("k_viterbi_rows")
                          : "k_viterbi_fill<" + std::to_string(fill_class((int)cls).G) + "," + std::to_string(fill_class((int)cls).B) + ">";
  return names[cls].c_str();
}

static int install_null(qf_ctx* c, const NullParams& n);

int qf_set_null_json(qf_ctx* c, const char* text) {
  if (!c) return QF_ERR_ARG;
  if (!text) {
    c->have_null = false;
    return QF_OK;
  }
  Json j;
  std::string err;
  if (!parse_json(text, j, err)) return fail(c, QF_ERR_PARSE, err);
  NullParams n;
  if (!n.read_json(j, err)) return fail(c, QF_ERR_PARSE, err);
  return install_null(c, n);
}

int qf_set_null_raw(qf_ctx* c, double null_emit, const double* pqr) {
  if (!c || !pqr) return QF_ERR_ARG;
  NullParams n;
  n.nullEmit = null_emit;
  for (int i = 0; i < 4; ++i) n.null[i] = SymQualDist{pqr[i * 3], pqr[i * 3 + 1], pqr[i * 3 + 2]};
  return install_null(c, n);
}

static int install_null(qf_ctx* c, const NullParams& n) {
  HIPCHK(c, hipSetDevice(c->device));
  c->null = n;
  double le, l1, ls[4];
  std::vector<double> lq(4 * kNQual);
  n.tables(le, l1, ls, lq.data());
  HIPCHK(c, c->d_nullq.reserve(lq.size() * 8));
  HIPCHK(c, hipMemcpy(c->d_nullq.p, lq.data(), lq.size() * 8, hipMemcpyHostToDevice));
  c->have_null = true;
  ++c->prep_epoch;
  return QF_OK;
}

int qf_get_lse_table(const qf_ctx*, const double** table, int* n) {
  const std::vector<double>& t = lse_table();
  if (table) *table = t.data();
  if (n) *n = kLseEntries;
  return QF_OK;
}

// ------------------------------------------------------------------------------ sequences
static int read_counters(Slot* c, BatchCounters& bc) {
  HIPCHK(c, hipMemcpyAsync(&bc, c->d_bc.p, sizeof bc, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return QF_OK;
}

int qf_set_refs(qf_ctx* c, const char* seq, const uint64_t* offsets, uint32_t n_refs) {
  if (!c || !seq || !offsets || !n_refs) return QF_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  c->n_refs = 0;
  c->index_k = 0;
  c->ref_off.assign(offsets, offsets + n_refs + 1);
  c->ref_total = offsets[n_refs] - offsets[0];
  if (offsets[0] != 0) return fail(c, QF_ERR_ARG, "offsets[0] must be 0");
  c->ref_maxlen = 0;
  c->ref_woff.assign(n_refs + 1, 0);
  for (uint32_t x = 0; x < n_refs; ++x) {
    if (offsets[x + 1] <= offsets[x]) return fail(c, QF_ERR_ARG, "empty reference sequence");
    const uint64_t len = offsets[x + 1] - offsets[x];
    if (len > 0x7FFFFFF0ull) return fail(c, QF_ERR_ARG, "reference longer than 2^31");
    c->ref_maxlen = std::max(c->ref_maxlen, len);
    c->ref_woff[x + 1] = c->ref_woff[x] + (len + 15) / 16 + 2;
  }
  HIPCHK(c, c->d_ref_seq.reserve(c->ref_total));
  HIPCHK(c, c->d_ref_tok.reserve(c->ref_total));
  HIPCHK(c, c->d_ref_off.reserve((n_refs + 1) * 8));
  HIPCHK(c, c->d_ref_woff.reserve((n_refs + 1) * 8));
  HIPCHK(c, c->d_ref_packed.reserve(c->ref_woff[n_refs] * 4));
  HIPCHK(c, c->d_bc.reserve(sizeof(BatchCounters)));
  HIPCHK(c, hipMemcpyAsync(c->d_ref_seq.p, seq, c->ref_total, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_ref_off.p, c->ref_off.data(), (n_refs + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_ref_woff.p, c->ref_woff.data(), (n_refs + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_bc.p, 0, sizeof(BatchCounters), c->stream));
  launch_prep_ref(c->d_ref_seq.as<char>(), c->ref_total, c->d_ref_tok.as<uint8_t>(), c->d_bc.as<BatchCounters>(), c->stream);
  launch_pack_ref(c->d_ref_tok.as<uint8_t>(), c->d_ref_off.as<uint64_t>(), c->d_ref_woff.as<uint64_t>(), n_refs,
                  c->ref_maxlen, c->d_ref_packed.as<uint32_t>(), c->stream);
  HIPCHK(c, hipGetLastError());
  BatchCounters bc;
  if (int rc = read_counters(c, bc)) return rc;
  if (bc.error & 4u) {
    char m[96];
    snprintf(m, sizeof m, "Unknown symbol %c in reference sequence (offset %u)", seq[bc.error_detail], bc.error_detail);
    return fail(c, QF_ERR_SYMBOL, m);
  }
  c->n_refs = n_refs;
  return QF_OK;
}

// sorted (k-mer, position) index of a sequence set, for k > kMaxRefK
static int build_sorted_index(qf_ctx* c, const uint8_t* d_tok, const uint64_t* d_off, const std::vector<uint64_t>& off,
                              uint64_t max_len, int k, DevBuf& d_off32, DevBuf& d_keys_out, DevBuf& d_pos_out) {
  const uint32_t n = (uint32_t)off.size() - 1;
  const uint64_t total = off[n];
  if (total > 0x7FFFFFF0ull) return fail(c, QF_ERR_UNSUPPORTED, "sorted k-mer index: more than 2^31 bases in one sequence set");
  std::vector<int> off32(off.begin(), off.end());
  HIPCHK(c, d_off32.reserve((size_t)(n + 1) * 4));
  HIPCHK(c, hipMemcpyAsync(d_off32.p, off32.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, c->d_keys_tmp.reserve((total + 16) * 8));
  HIPCHK(c, c->d_vals_tmp.reserve((total + 16) * 4));
  HIPCHK(c, d_keys_out.reserve((total + 16) * 8));
  HIPCHK(c, d_pos_out.reserve((total + 16) * 4));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // off32 is a stack-lifetime host buffer
  const int rc = sort_kmer_index(d_tok, d_off, d_off32.as<int>(), n, total, max_len, (uint32_t)k, c->d_keys_tmp.as<unsigned long long>(),
                                 c->d_vals_tmp.as<uint32_t>(), d_keys_out.as<unsigned long long>(), d_pos_out.as<uint32_t>(),
                                 &c->sort_temp, &c->sort_temp_cap, c->stream);
  if (rc != 0) return fail(c, QF_ERR_DEVICE, std::string("k-mer index sort: ") + hipGetErrorString((hipError_t)rc));
  HIPCHK(c, hipGetLastError());
  return QF_OK;
}

static int ensure_ref_index(qf_ctx* c, int k) {
  if (c->index_k == k) return QF_OK;
  if (k < 1 || k > 32) return fail(c, QF_ERR_ARG, "kmer_len out of range");
  if (k > kMaxRefK) {
    if (int rc = build_sorted_index(c, c->d_ref_tok.as<uint8_t>(), c->d_ref_off.as<uint64_t>(), c->ref_off, c->ref_maxlen, k,
                                    c->d_off32, c->d_skeys, c->d_pos))
      return rc;
    c->index_k = k;
    return QF_OK;
  }
  const uint32_t nb = 1u << (2 * k);
  const size_t bytes = (size_t)c->n_refs * (nb + 1) * 4;
  HIPCHK(c, c->d_bucket.reserve(bytes));
  HIPCHK(c, c->d_cursor.reserve(bytes));
  HIPCHK(c, c->d_pos.reserve((c->ref_total + 16) * 4));  // + slack: the seeding kernel reads bucket entries four at a time
  HIPCHK(c, hipMemsetAsync(c->d_bucket.p, 0, bytes, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_cursor.p, 0, bytes, c->stream));
  launch_ref_index(c->d_ref_tok.as<uint8_t>(), c->d_ref_off.as<uint64_t>(), c->n_refs, c->ref_maxlen, (uint32_t)k, nb,
                   c->d_bucket.as<uint32_t>(), c->d_cursor.as<uint32_t>(), c->d_pos.as<uint32_t>(), c->stream);
  HIPCHK(c, hipGetLastError());
  c->index_k = k;
  return QF_OK;
}

int qf_upload_reads(qf_ctx* c, const char* seq, const char* qual, const uint64_t* offsets, uint32_t n_reads) {
  if (!c || !seq || !offsets) return QF_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  c->n_reads = 0;
  if (n_reads && offsets[0] != 0) return fail(c, QF_ERR_ARG, "offsets[0] must be 0");
  c->read_off.assign(offsets, offsets + n_reads + 1);
  c->read_total = n_reads ? offsets[n_reads] : 0;
  c->read_maxlen = 0;
  uint64_t minlen = ~0ull;
  for (uint32_t r = 0; r < n_reads; ++r) {
    if (offsets[r + 1] <= offsets[r]) return fail(c, QF_ERR_ARG, "empty read (the reference drops zero-length sequences on load)");
    c->read_maxlen = std::max(c->read_maxlen, offsets[r + 1] - offsets[r]);
    minlen = std::min(minlen, offsets[r + 1] - offsets[r]);
  }
  c->ragged_reads = n_reads > 64 && c->read_maxlen * 4 > minlen * 5 && !getenv("QUAFF_HIP_NO_LENGTH_SORT");   // (A/B switch)
  if (c->read_maxlen > 0xFFFFu * 16ull) return fail(c, QF_ERR_UNSUPPORTED, "read longer than 1M bases");
  const uint64_t tot = c->read_total;
  HIPCHK(c, c->d_seq.reserve(tot + 16));
  if (qual) HIPCHK(c, c->d_qual.reserve(tot + 16));
  HIPCHK(c, c->d_roff.reserve((n_reads + 1) * 8));
  HIPCHK(c, c->d_tok.reserve(tot + 16));
  HIPCHK(c, c->d_ctx.reserve((tot + 2 * kCtxPad) * 4));
  HIPCHK(c, c->d_skmer.reserve((tot + 16) * 4));
  HIPCHK(c, c->d_nll.reserve((n_reads + 1) * 8));
  HIPCHK(c, c->d_bc.reserve(sizeof(BatchCounters)));
  HIPCHK(c, hipMemcpyAsync(c->d_seq.p, seq, tot, hipMemcpyHostToDevice, c->stream));
  if (qual) HIPCHK(c, hipMemcpyAsync(c->d_qual.p, qual, tot, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_roff.p, c->read_off.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_ctx.p, 0, (tot + 2 * kCtxPad) * 4, c->stream));
  uint32_t qr[2] = {0, (uint32_t)kNQual - 1};
  if (qual && tot) {   // which quality values occur (the E-step keeps only their emission rows in LDS)
    const uint32_t init[2] = {0xFFFFFFFFu, 0};
    HIPCHK(c, c->d_qrange.reserve(8));
    HIPCHK(c, hipMemcpyAsync(c->d_qrange.p, init, 8, hipMemcpyHostToDevice, c->stream));
    launch_qual_range(c->d_qual.as<char>(), tot, c->d_qrange.as<uint32_t>(), c->stream);
    HIPCHK(c, hipMemcpyAsync(qr, c->d_qrange.p, 8, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->read_qmin = std::min(qr[0], (uint32_t)kNQual - 1);
  c->read_qmax = std::max(std::min(qr[1], (uint32_t)kNQual - 1), c->read_qmin);
  c->reads_have_qual = qual != nullptr;
  c->read_index_k = 0;
  ++c->prep_epoch;
  c->n_reads = n_reads;
  return QF_OK;
}

// tokens / context words / seeding k-mers / null log-likelihoods for the resident reads
// `side`: run the null log-likelihoods (a serial sum per read, needed only when pairs are finalised) on that stream, beside
// whatever the caller launches next on the main stream; c->ev_nll is recorded behind them.
static int prep_reads(qf_ctx* c, int seed_k, hipStream_t side = nullptr) {
  ++c->prep_epoch;
  PrepArgs a{};
  a.seq = c->d_seq.as<char>();
  a.qual = c->reads_have_qual ? c->d_qual.as<char>() : nullptr;
  a.off = c->d_roff.as<uint64_t>();
  a.match_len = c->scores.match_len;
  a.gap_len = c->scores.gap_len;
  a.seed_k = (uint32_t)seed_k;
  a.tok = c->d_tok.as<uint8_t>();
  a.ctx = c->d_ctx.as<uint32_t>() + kCtxPad;
  a.skmer = c->d_skmer.as<uint32_t>();
  a.skmer64 = nullptr;
  if (seed_k > kMaxRefK) {
    HIPCHK(c, c->d_skmer64.reserve((c->read_total + 16) * 8));
    a.skmer64 = c->d_skmer64.as<unsigned long long>();
  }
  a.nll = c->d_nll.as<double>();
  a.has_null = c->have_null;
  if (c->have_null) {
    std::vector<double> lq(4 * kNQual);
    c->null.tables(a.null_logEmit, a.null_log1mEmit, a.null_logSym, lq.data());
    a.null_logQual = c->d_nullq.as<double>();
  }
  a.em_qmajor_Km = c->prep_qmajor_Km;
  a.em_qmin = c->read_qmin;
  a.bc = c->d_bc.as<BatchCounters>();
  launch_prep_reads(a, c->n_reads, c->stream);
  if (side) {
    HIPCHK(c, hipEventRecord(c->ev_tok, c->stream));
    HIPCHK(c, hipStreamWaitEvent(side, c->ev_tok, 0));
  }
  launch_null_ll(a, c->n_reads, side ? side : c->stream);
  HIPCHK(c, hipEventRecord(c->ev_nll, side ? side : c->stream));
  HIPCHK(c, hipGetLastError());
  return QF_OK;
}

static int check_cfg(qf_ctx* c, const qf_dp_config* cfg) {
  if (!c) return QF_ERR_ARG;
  if (!cfg) return fail(c, QF_ERR_ARG, "null config");
  if (cfg->reserved) return fail(c, QF_ERR_ARG, "qf_dp_config.reserved must be 0 (kernel-variant switches: qf_debug_set_flags)");
  if (!c->have_params) return fail(c, QF_ERR_STATE, "no parameters set (qf_set_params_json)");
  if (!c->n_refs) return fail(c, QF_ERR_STATE, "no references set (qf_set_refs)");
  if (cfg->band_size < 0) return fail(c, QF_ERR_ARG, "negative band size");
  if (cfg->sparse && (cfg->kmer_len < 1 || cfg->kmer_len > 32)) return fail(c, QF_ERR_ARG, "kmer_len out of range");
  return QF_OK;
}

static void fill_seed_args(qf_ctx* c, Slot& S, const qf_dp_config* cfg, SeedArgs& s, uint32_t max_units, int max_nd) {
  s = SeedArgs{};
  s.n_refs = c->n_refs;
  s.ref_off = c->d_ref_off.as<uint64_t>();
  s.read_off = c->d_roff.as<uint64_t>();
  s.skmer = c->d_skmer.as<uint32_t>();
  const bool sorted = cfg->sparse && cfg->kmer_len > kMaxRefK;
  s.skmer64 = sorted ? c->d_skmer64.as<unsigned long long>() : nullptr;
  s.ref_skeys = sorted ? c->d_skeys.as<unsigned long long>() : nullptr;
  s.ref_bucket = c->d_bucket.as<uint32_t>();
  s.ref_pos = c->d_pos.as<uint32_t>();
  s.nbuckets = (cfg->sparse && !sorted) ? 1u << (2 * cfg->kmer_len) : 0;
  s.sparse = cfg->sparse;
  s.kmer_len = cfg->kmer_len;
  s.threshold = cfg->kmer_threshold;
  s.band = cfg->band_size;
  s.cell_size = 24;  // QuaffDPMatrixContainer::cellSize(), src/qmodel.h:384 (align / overlap)
  s.max_size = cfg->max_size;
  s.max_nd = max_nd;
  s.units = S.d_units.as<Unit>();
  s.max_units = max_units;
  s.cls_list = S.d_cls_list.as<uint32_t>();
  s.cls_key = c->ragged_reads ? S.d_cls_key.as<uint32_t>() : nullptr;
  s.pair_head = S.d_pair_head.as<uint32_t>();
  s.pair_bands = S.d_pair_bands.as<int2>();
  s.pair_nbands = S.d_pair_nbands.as<uint32_t>();
  s.ovf_bands = S.d_ovf.as<int4>();
  s.ovf_cap = max_units;
  s.pair_ndiag = S.d_pair_ndiag.as<uint32_t>();
  s.pair_cells = S.d_pair_cells.as<unsigned long long>();
  s.force_block_kernel = c->debug & QF_DEBUG_BLOCK_SEED;
  s.max_ref_len = (uint32_t)std::min<uint64_t>(c->ref_maxlen, 0xFFFFFFFFull);
  s.max_read_len = (uint32_t)std::min<uint64_t>(c->read_maxlen, 0xFFFFFFFFull);
  s.no_lds_index = (c->debug & QF_DEBUG_GLOBAL_INDEX) != 0;
  s.few_hits = 0;   // set by seed_pairs once the x side is known (references, or reads for overlap)
  s.bc = S.d_bc.as<BatchCounters>();
}

// Sequences too long for the LDS diagonal histogram are seeded through global-memory workspaces, one per resident
// workgroup (at most 1024, at most ~4 GiB in all).
static int reserve_seed_workspace(Slot* c, SeedArgs& sa, bool mem, uint64_t n_pairs) {
  sa.ws = nullptr;
  sa.ws_slots = 0;
  sa.ws_words = 0;
  const size_t per = (seed_lds_bytes(sa.max_nd, mem, seed_needs_deep_counters(sa)) + 15) & ~(size_t)15;
  if (!seed_needs_workspace(sa, mem)) return QF_OK;
  const uint64_t slots = std::max<uint64_t>(1, std::min<uint64_t>({n_pairs, 1024, (4ull << 30) / per}));
  HIPCHK(c, c->d_seed_ws.reserve(per * slots));
  sa.ws = c->d_seed_ws.as<uint32_t>();
  sa.ws_words = per / 4;
  sa.ws_slots = (uint32_t)slots;
  return QF_OK;
}

// Ragged read lengths: a wavefront's bands run in lockstep for as long as the longest of them, so each class list is sorted
// by read length (longest first) before the fills.
static int sort_class_lists(qf_ctx* c, Slot* S, const BatchCounters& bc, uint32_t max_units, bool pair_order_cls0 = false,
                            bool slotted_cls0 = false) {
  if (!c->ragged_reads && !pair_order_cls0) return QF_OK;
  CHUNKRES(S, S->d_sort_k, (size_t)max_units * 4);
  CHUNKRES(S, S->d_sort_v, (size_t)max_units * 4);
  for (int cls = 0; cls < kNumClasses; ++cls) {
    // (overlap sorts every list: bands by the columns they cross, the single-diagonal list into pair order)
    if (cls == kRowClass || bc.cls_count[cls] <= 64 || (cls == 0 && slotted_cls0)) continue;
    const int rc = sort_class_list(S->d_cls_key.as<uint32_t>() + (size_t)cls * max_units, S->d_cls_list.as<uint32_t>() + (size_t)cls * max_units,
                                   bc.cls_count[cls], S->d_sort_k.as<uint32_t>(), S->d_sort_v.as<uint32_t>(), &S->sort_tmp, &S->sort_tmp_cap,
                                   S->stream);
    if (rc) return fail(S, QF_ERR_DEVICE, "class-list sort failed");
  }
  return QF_OK;
}

static int reserve_pair_buffers(Slot* c, uint64_t n_pairs, uint32_t max_units) {
  CHUNKRES(c, c->d_bc, sizeof(BatchCounters));
  CHUNKRES(c, c->d_cls_key, (size_t)kNumClasses * max_units * 4);
  CHUNKRES(c, c->d_units, (size_t)max_units * sizeof(Unit));
  CHUNKRES(c, c->d_cls_list, (size_t)kNumClasses * max_units * 4);
  CHUNKRES(c, c->d_pair_head, n_pairs * 4);
  CHUNKRES(c, c->d_pair_bands, n_pairs * kMaxBandsPerPair * sizeof(int2));
  CHUNKRES(c, c->d_pair_nbands, n_pairs * 4);
  CHUNKRES(c, c->d_ovf, (size_t)max_units * sizeof(int4));
  HIPCHK(c, hipMemsetAsync(c->d_pair_nbands.p, 0, n_pairs * 4, c->stream));
  CHUNKRES(c, c->d_pair_ndiag, n_pairs * 4);
  CHUNKRES(c, c->d_pair_cells, n_pairs * 8);
  CHUNKRES(c, c->d_pair_score, n_pairs * 8);
  CHUNKRES(c, c->d_pair_end_unit, n_pairs * 4);
  HIPCHK(c, hipMemsetAsync(c->d_pair_head.p, 0xFF, n_pairs * 4, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_pair_ndiag.p, 0, n_pairs * 4, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_pair_cells.p, 0, n_pairs * 8, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_bc.p, 0, sizeof(BatchCounters), c->stream));
  return QF_OK;
}

// Seeds the slot's pairs and bins their bands.  The unit table and the band overflow list are provisioned for four bands per
// pair; a batch with more (low thresholds, narrow bands, repeats) gets them grown to what it asked for and is seeded again.
static int seed_pairs(qf_ctx* c, Slot* S, const qf_dp_config* cfg, uint32_t n_pairs, bool mem, int max_nd,
                      const std::function<void(SeedArgs&)>& customize, uint32_t& max_units, SeedArgs& sa, BatchCounters& bc) {
  for (int attempt = 0;; ++attempt) {
    if (int rc = reserve_pair_buffers(S, n_pairs, max_units)) return rc;
    fill_seed_args(c, *S, cfg, sa, max_units, max_nd);
    customize(sa);
    sa.few_hits = sa.sparse && (sa.kmer_len >= 16 || (uint64_t)(sa.max_ref_len ? sa.max_ref_len : sa.max_read_len) < (1ull << (2 * sa.kmer_len)));
    if (int rc = reserve_seed_workspace(S, sa, mem, n_pairs)) return rc;
    if (launch_seed(sa, n_pairs, mem, S->stream) != 0)
      return fail(S, QF_ERR_UNSUPPORTED, "sequence pair of " + std::to_string(max_nd) + " diagonals: no seeding workspace");
    launch_bin_units(sa, n_pairs, 0, S->stream);
    HIPCHK(S, hipGetLastError());
    if (int rc = read_counters(S, bc)) return rc;
    if (bc.n_ovf && !(bc.error & 8u)) {  // pairs with more than kMaxBandsPerPair bands: second binning pass
      launch_bin_units(sa, n_pairs, bc.n_ovf, S->stream);
      HIPCHK(S, hipGetLastError());
      if (int rc = read_counters(S, bc)) return rc;
    }
    if (!(bc.error & 9u)) return QF_OK;
    const uint64_t want = std::max<uint64_t>({2ull * max_units, (uint64_t)bc.n_ovf + 1024, (uint64_t)bc.n_units + bc.n_ovf + 1024});
    if (attempt == 5 || want > 0x3FFFFFFFull) return fail(S, QF_ERR_MEMORY, "more envelope bands than the unit table can hold");
    max_units = (uint32_t)want;
  }
}

// Reads [lo, hi) of the resident set against every reference.  Results go to the context's host arrays at the chunk's
// offsets and are accumulated into *out.  If the chunk's traceback would exceed the memory budget nothing is filled and
// *too_big is set (the caller splits the range).
static int align_chunk_impl(qf_ctx* c, Slot* S, const qf_dp_config* cfg, uint32_t flags, uint32_t lo, uint32_t hi,
                       uint64_t budget, qf_align_result* out, std::mutex& out_mu, bool two_in_flight, bool* too_big,
                       uint64_t* need_bytes, uint32_t* row_granule) {
  *too_big = false;
  *need_bytes = 0;
  *row_granule = 0;
  const uint32_t n_reads = hi - lo, n_refs = c->n_refs;
  const uint32_t n_pairs = n_reads * n_refs;
  const bool sparse = cfg->sparse != 0;
  const bool mem = sparse && cfg->kmer_threshold < 0;
  uint32_t max_units = n_pairs * 4 + 1024;
  const uint64_t* d_roff = c->d_roff.as<uint64_t>() + lo;   // the chunk's reads: offsets stay absolute, indices local
  const double* d_nll = c->d_nll.as<double>() + lo;
  HIPCHK(S, hipEventRecord(S->ev[1], S->stream));

  // ---- seeding
  const int max_nd = (int)(c->ref_maxlen + c->read_maxlen - 1);
  SeedArgs sa;
  BatchCounters bc;
  if (int rc = seed_pairs(c, S, cfg, n_pairs, mem, sparse ? max_nd : 2, [&](SeedArgs& s) { s.read_off = d_roff; }, max_units, sa, bc)) return rc;
  HIPCHK(S, hipEventRecord(S->ev[2], S->stream));
  if (bc.error & 4u) return fail(S, QF_ERR_SYMBOL, "Unknown symbol in read " + std::to_string(bc.error_detail));
  if (bc.error & 2u) return fail(S, QF_ERR_UNSUPPORTED, "unsupported band of " + std::to_string(bc.error_detail) + " diagonals");

  // ---- fill
  const uint64_t tb_bytes = (uint64_t)bc.tb_words * 4;
  if (tb_bytes > budget) {
    if (n_reads == 1) return fail(S, QF_ERR_MEMORY, "one read needs " + std::to_string(tb_bytes >> 20) + " MiB of traceback, over the memory budget");
    *too_big = true;
    *need_bytes = tb_bytes;
    // mostly row-space bands (-kmatchoff): the pieces should be whole rounds of the kernel's resident workgroups
    if (bc.cls_count[kRowClass] && bc.cls_cells[kRowClass] * 2 > bc.total_cells) {
      FillArgs probe{};
      probe.dp.ematch_ninf_off = (uint32_t)((size_t)c->scores.Km * kNQ1 * 4 * 8);
      probe.dp.Kg = c->scores.Kg;
      probe.no_lds_tables = (c->debug & QF_DEBUG_GLOBAL_TABLES) != 0;
      *row_granule = viterbi_rows_resident_workgroups(probe);
    }
    return QF_OK;
  }
  if (reserve_big(S->d_tb, tb_bytes + 64, {&c->d_fw, &c->second.d_fw}) != hipSuccess) {   // less memory than the budget assumed: halve the chunk
    if (n_reads == 1) return fail(S, QF_ERR_MEMORY, "cannot allocate " + std::to_string(tb_bytes >> 20) + " MiB of traceback for one read");
    *too_big = true;
    return QF_OK;
  }
  if (int rc = sort_class_lists(c, S, bc, max_units)) return rc;
  FillArgs fa{};
  fa.n_refs = n_refs;
  fa.units = S->d_units.as<Unit>();
  fa.ref_off = c->d_ref_off.as<uint64_t>();
  fa.ref_woff = c->d_ref_woff.as<uint64_t>();
  fa.ref_tok = c->d_ref_tok.as<uint8_t>();
  fa.ref_packed = c->d_ref_packed.as<uint32_t>();
  fa.read_off = d_roff;
  fa.ctx = c->d_ctx.as<uint32_t>() + kCtxPad;
  fa.tb = S->d_tb.as<uint32_t>();
  const Scores& sc = c->scores;
  fa.dp.ematch = c->d_ematch.as<double>();
  fa.dp.ematch_ninf_off = (uint32_t)((size_t)c->scores.Km * kNQ1 * 4 * 8);
  fa.dp.eins = c->d_eins.as<double>();
  fa.dp.trans = c->d_trans.as<double>();
  fa.dp.d2d = sc.trans[4 * sc.Kg + 0];
  fa.dp.d2m = sc.trans[4 * sc.Kg + 1];
  fa.dp.i2i = sc.trans[4 * sc.Kg + 2];
  fa.dp.i2m = sc.trans[4 * sc.Kg + 3];
  fa.dp.Kg = sc.Kg;
  fa.dp.local = cfg->local;
  fa.no_lds_tables = (c->debug & QF_DEBUG_GLOBAL_TABLES) != 0;
  // One kernel per class, the class with the most cells first, spread over the main and the side streams: the small
  // classes (and the single-diagonal chains, which are latency-bound) fill the SIMDs the big class's last wavefronts
  // leave idle.  Everything was seeded on the main stream, which the host has already waited for.
  // With two chunks in flight only one of them fills at a time: the token keeps the other on its seeding or traceback,
  // which is what overlaps well with a fill (two fills side by side just share the VALUs).
  std::unique_lock<std::mutex> token;
  token = std::unique_lock<std::mutex>(device_fill_token(c->device));
  {
    int order[kNumClasses], n_used = 0;
    for (int cls = 0; cls < kNumClasses; ++cls) if (bc.cls_count[cls]) order[n_used++] = cls;
    std::sort(order, order + n_used, [&](int p, int q) { return bc.cls_cells[p] > bc.cls_cells[q]; });
    const bool concurrent = !(c->debug & QF_DEBUG_SERIAL_CLASSES);
    const int n_lanes = two_in_flight ? 2 : 4;  // two chunks in flight: one side stream each (hardware queues are few)
    // the side streams are non-blocking: they must not start before what the main stream still has queued for this chunk
    // (the in-place sorts of the class lists above, and the buffer clears of reserve_pair_buffers)
    if (concurrent && n_used > 1) {
      HIPCHK(S, hipEventRecord(S->ev[7], S->stream));
      for (int k = 0; k < n_lanes - 1; ++k) HIPCHK(S, hipStreamWaitEvent(S->aux[k], S->ev[7], 0));
    }
    for (int k = 0; k < n_used; ++k) {
      const int cls = order[k], lane = concurrent ? (k < n_lanes ? k : 1 + (k - 1) % (n_lanes - 1)) : 0;
      hipStream_t s = lane == 0 ? S->stream : S->aux[lane - 1];
      HIPCHK(S, hipEventRecord(S->cls_ev[cls], s));
      fa.n_cls_units = bc.cls_count[cls];
      fa.cls_list = S->d_cls_list.as<uint32_t>() + (size_t)cls * max_units;
      launch_viterbi_fill(cls, fa, sc.Kg > 1, s);
      HIPCHK(S, hipEventRecord(S->cls_end[cls], s));
      if (lane) HIPCHK(S, hipStreamWaitEvent(S->stream, S->cls_end[cls], 0));
    }
  }
  const BatchCounters seed_bc = bc;
  HIPCHK(S, hipGetLastError());
  HIPCHK(S, hipEventRecord(S->ev[3], S->stream));

  // ---- pair results, selection, traceback
  FinalArgs fin{};
  fin.n_pairs = n_pairs;
  fin.n_reads = n_reads;
  fin.n_refs = n_refs;
  fin.all = (flags & QF_ALIGN_ALL) != 0;
  fin.min_score = c->min_score;
  fin.units = S->d_units.as<Unit>();
  fin.pair_head = S->d_pair_head.as<uint32_t>();
  fin.pair_score = S->d_pair_score.as<double>();
  fin.pair_end_unit = S->d_pair_end_unit.as<uint32_t>();
  fin.nll = d_nll;
  fin.read_off = d_roff;
  fin.ref_off = c->d_ref_off.as<uint64_t>();
  fin.tb = S->d_tb.as<uint32_t>();
  fin.bc = S->d_bc.as<BatchCounters>();
  HIPCHK(S, hipStreamWaitEvent(S->stream, c->ev_nll, 0));   // null log-likelihoods (side stream) before the scores
  launch_finalize(fin, S->stream);
  // per-pair results are final: copy them out on a side stream while the alignments are selected and traced
  const size_t p0 = (size_t)lo * n_refs;
  HIPCHK(S, hipEventRecord(S->ev[6], S->stream));
  HIPCHK(S, hipStreamWaitEvent(S->aux[2], S->ev[6], 0));
  HIPCHK(S, hipMemcpyAsync(c->h_viterbi.data() + p0, S->d_pair_score.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, S->aux[2]));
  HIPCHK(S, hipMemcpyAsync(c->h_cells.data() + p0, S->d_pair_cells.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, S->aux[2]));
  HIPCHK(S, hipMemcpyAsync(c->h_ndiag.data() + p0, S->d_pair_ndiag.p, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, S->aux[2]));
  uint32_t n_recs = 0, n_valid = 0;
  uint64_t total_runs = 0;
  const bool dense = !(flags & QF_ALIGN_ALL) && !(flags & QF_ALIGN_NO_TRACEBACK);
  if (!(flags & QF_ALIGN_NO_TRACEBACK)) {
    const size_t max_recs = fin.all ? n_pairs : n_reads;
    CHUNKRES(S, S->d_recs, max_recs * sizeof(AlignRec));
    fin.recs = S->d_recs.as<AlignRec>();
    fin.dense = !fin.all;
    fin.read_base = lo;
    if (fin.dense) {
      CHUNKRES(S, S->d_align_out, (size_t)n_reads * sizeof(AlignOut));
      fin.out_align = S->d_align_out.as<AlignOut>();
    }
    launch_select(fin, S->stream);
    HIPCHK(S, hipGetLastError());
    if (int rc = read_counters(S, bc)) return rc;
    if (token.owns_lock()) token.unlock();  // the fill has drained
    n_recs = fin.dense ? n_reads : bc.n_align;
    n_valid = bc.n_align;
    CHUNKRES(S, S->d_runs_tmp, (size_t)bc.n_runs * 4 + 64);
    CHUNKRES(S, S->d_runs_out, (size_t)bc.n_runs * 4 + 64);
    fin.n_recs = n_recs;
    fin.runs_tmp = S->d_runs_tmp.as<uint32_t>();
    fin.runs_out = S->d_runs_out.as<uint32_t>();
    launch_traceback(fin, S->stream);
    HIPCHK(S, hipGetLastError());
    // the final records do not wait for the run count: their copy is queued before the counters are read back
    static_assert(sizeof(AlignOut) == sizeof(qf_alignment), "AlignOut mirrors qf_alignment");
    if (dense && n_recs)
      HIPCHK(S, hipMemcpyAsync(c->h_align.data() + lo, S->d_align_out.p, (size_t)n_recs * sizeof(AlignOut), hipMemcpyDeviceToHost, S->stream));
    if (int rc = read_counters(S, bc)) return rc;
    total_runs = bc.total_runs_out;
    if (bc.error & 16u) return fail(S, QF_ERR_DEVICE, "traceback did not reach the start state");
  }
  HIPCHK(S, hipEventRecord(S->ev[4], S->stream));

  // ---- results to the host, at the chunk's offsets
  const size_t recs0 = S->h_recs.size(), runs0 = S->h_runs.size();
  if (!dense) S->h_recs.resize(recs0 + n_recs);
  S->h_runs.resize(runs0 + total_runs);
  if (!dense && n_recs) HIPCHK(S, hipMemcpyAsync(S->h_recs.data() + recs0, S->d_recs.p, (size_t)n_recs * sizeof(AlignRec), hipMemcpyDeviceToHost, S->stream));
  if (total_runs) HIPCHK(S, hipMemcpyAsync(S->h_runs.data() + runs0, S->d_runs_out.p, (size_t)total_runs * 4, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(S, hipEventRecord(S->ev[5], S->stream));
  HIPCHK(S, hipStreamSynchronize(S->stream));
  HIPCHK(S, hipStreamSynchronize(S->aux[2]));
  if (!dense)
    for (size_t a = recs0; a < recs0 + n_recs; ++a) {
      S->h_recs[a].read += lo;
      S->h_recs[a].run_off += runs0;
    }
  {
    std::lock_guard<std::mutex> lk(out_mu);
    if (dense) {
      c->dense_chunks.push_back({lo, hi, runs0, S != static_cast<Slot*>(c)});
      out->n_alignments += n_valid;
    }
    out->total_cells += seed_bc.total_cells;
    out->n_units += seed_bc.n_units;
    out->traceback_bytes += tb_bytes;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, S->ev[1], S->ev[2]); out->ms_seed += ms;
    (void)hipEventElapsedTime(&ms, S->ev[2], S->ev[3]); out->ms_fill += ms;
    (void)hipEventElapsedTime(&ms, S->ev[3], S->ev[4]); out->ms_traceback += ms;
    for (int cls = 0; cls < kNumClasses; ++cls) {
      ms = 0;
      if (seed_bc.cls_count[cls]) (void)hipEventElapsedTime(&ms, S->cls_ev[cls], S->cls_end[cls]);
      out->ms_fill_class[cls] += ms;
      out->cells_class[cls] += seed_bc.cls_cells[cls];
      out->units_class[cls] += seed_bc.cls_count[cls];
    }
  }
  return QF_OK;
}


// (see CHUNKRES)
static int split_instead(qf_ctx* c, Slot* S, int rc, bool single, bool* too_big, std::initializer_list<DevBuf*> idle) {
  if (rc != kSplitChunk) return rc;
  (void)hipDeviceSynchronize();          // what the abandoned chunk still has in flight
  for (DevBuf* b : idle) b->release();   // the big buffers of the entry points that are not running
  if (single) return fail(S, QF_ERR_MEMORY, S->err + ": one read / pair does not fit");
  *too_big = true;
  (void)c;
  return QF_OK;
}
static int align_chunk(qf_ctx* c, Slot* S, const qf_dp_config* cfg, uint32_t flags, uint32_t lo, uint32_t hi,
                       uint64_t budget, qf_align_result* out, std::mutex& out_mu, bool two_in_flight, bool* too_big,
                       uint64_t* need_bytes, uint32_t* row_granule) {
  const int rc = align_chunk_impl(c, S, cfg, flags, lo, hi, budget, out, out_mu, two_in_flight, too_big, need_bytes, row_granule);
  if (rc == kSplitChunk) { *need_bytes = 0; *row_granule = 0; }
  return split_instead(c, S, rc, hi - lo <= 1, too_big, {&c->d_fw, &c->second.d_fw});
}

int qf_align_resident(qf_ctx* c, const qf_dp_config* cfg, uint32_t flags, qf_align_result* out) {
  if (int rc = check_cfg(c, cfg)) return rc;
  if (!out) return fail(c, QF_ERR_ARG, "null result");
  HIPCHK(c, hipSetDevice(c->device));
  const CallInProgress in_progress(c->device);
  memset(out, 0, sizeof *out);
  const uint32_t n_reads = c->n_reads, n_refs = c->n_refs;
  const uint64_t n_pairs64 = (uint64_t)n_reads * n_refs;
  if (n_pairs64 > kMaxPairsPerCall) return fail(c, QF_ERR_UNSUPPORTED, "more than 2^28 read x reference pairs in one batch");
  const uint32_t n_pairs = (uint32_t)n_pairs64;
  out->n_reads = n_reads;
  out->n_refs = n_refs;
  if (!n_pairs) return QF_OK;
  const bool sparse = cfg->sparse != 0;
  if (sparse) if (int rc = ensure_ref_index(c, cfg->kmer_len)) return rc;
  const auto t_begin = std::chrono::steady_clock::now();
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  HIPCHK(c, c->d_bc.reserve(sizeof(BatchCounters)));
  HIPCHK(c, hipMemsetAsync(c->d_bc.p, 0, sizeof(BatchCounters), c->stream));
  if (int rc = prep_reads(c, sparse ? cfg->kmer_len : 0, c->aux[1])) return rc;
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
  c->h_viterbi.resize(n_pairs);
  c->h_cells.resize(n_pairs);
  c->h_ndiag.resize(n_pairs);
  c->h_nll.resize(n_reads);
  c->h_recs.clear();
  c->h_runs.clear();
  c->dense_chunks.clear();
  const bool dense = !(flags & QF_ALIGN_ALL) && !(flags & QF_ALIGN_NO_TRACEBACK);
  if (dense) c->h_align.resize(n_reads);
  HIPCHK(c, hipMemcpyAsync(c->h_nll.data(), c->d_nll.p, (size_t)n_reads * 8, hipMemcpyDeviceToHost, c->aux[1]));   // behind k_null_ll
  {
    BatchCounters pb;
    if (int rc = read_counters(c, pb)) return rc;
    if (pb.error & 4u) return fail(c, QF_ERR_SYMBOL, "Unknown symbol in read " + std::to_string(pb.error_detail));
  }
  (void)hipEventElapsedTime(&out->ms_prep, c->ev[0], c->ev[1]);
  // Work list: the batch in `n_chunks` pieces (one when it is small); a piece whose traceback exceeds the memory budget
  // is halved.  Two host threads, each with its own slot (stream, buffers), take pieces off the list, so the device
  // always has one chunk's fill to run under the other's seeding / selection / traceback.
  // Measured on config 2 (MI355X): the overlap gained is paid back by the smaller grids, so the default is one piece;
  // pieces bound the peak traceback memory (two in flight) and overlap the result copies with compute.
  uint32_t n_chunks = c->pipeline_chunks ? c->pipeline_chunks : 1u;
  if (c->debug & QF_DEBUG_SERIAL_CLASSES) n_chunks = 1;
  n_chunks = std::min(n_chunks, n_reads);
  std::vector<std::pair<uint32_t, uint32_t>> todo;
  for (uint32_t k = n_chunks; k-- > 0;)
    todo.push_back({(uint32_t)((uint64_t)n_reads * k / n_chunks), (uint32_t)((uint64_t)n_reads * (k + 1) / n_chunks)});
  std::mutex mu, out_mu;
  int in_flight = 0, rc_all = QF_OK;
  std::condition_variable cv;
  auto worker = [&](Slot* S) {
    (void)hipSetDevice(c->device);
    for (;;) {
      std::pair<uint32_t, uint32_t> job;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !todo.empty() || in_flight == 0 || rc_all != QF_OK; });
        if (rc_all != QF_OK || todo.empty()) return;
        job = todo.back();
        todo.pop_back();
        ++in_flight;
      }
      bool too_big = false;
      uint64_t need = 0;
      uint32_t granule = 0;
      const uint64_t budget = chunk_budget(c, S->d_tb, n_chunks > 1 ? 2 : 1);
      const int rc = align_chunk(c, S, cfg, flags, job.first, job.second, budget, out, out_mu, n_chunks > 1, &too_big, &need, &granule);
      {
        std::lock_guard<std::mutex> lk(mu);
        --in_flight;
        if (rc != QF_OK && rc_all == QF_OK) {
          rc_all = rc;
          if (S != static_cast<Slot*>(c)) c->err = S->err;
        }
        if (too_big) {
          // Pieces of as many reads as the budget holds (by the piece's own bytes per read; a piece that still comes out too
          // big is cut again), not blind halves: fewer wasted seeding passes and no piece much smaller than it has to be.  A
          // full-DP batch is all row-space bands, one workgroup each: there a piece is a whole number of rounds of the
          // kernel's resident workgroups (624 bands on 512 slots take as long as 1024).  An allocation that failed inside the
          // budget (another context on the device) halves.
          const uint32_t n = job.second - job.first;
          uint32_t fit = n / 2;
          if (need) {
            const double per_read = (double)need / n;
            fit = (uint32_t)std::min<double>(n - 1, std::max(1.0, std::floor(0.97 * (double)budget / per_read)));
            const uint32_t g = granule / std::max(1u, c->n_refs);
            if (g && fit >= g) fit = fit / g * g;
          }
          fit = std::max(1u, std::min(fit, n - 1));
          std::vector<std::pair<uint32_t, uint32_t>> pieces;
          for (uint32_t p = job.first; p < job.second; p += fit) pieces.push_back({p, std::min(job.second, p + fit)});
          for (size_t k = pieces.size(); k-- > 0;) todo.push_back(pieces[k]);
        }
      }
      cv.notify_all();
    }
  };
  c->second.h_recs.clear();
  c->second.h_runs.clear();
  if (n_chunks > 1) {
    if (!c->second_ready) {
      if (c->second.create()) return fail(c, QF_ERR_DEVICE, "cannot create the second stream");
      c->second_ready = true;
    }
    std::thread t(worker, &c->second);
    worker(c);
    t.join();
  } else {
    worker(c);
  }
  HIPCHK(c, hipStreamSynchronize(c->aux[1]));   // null log-likelihoods on the host
  if (rc_all != QF_OK) return rc_all;
  // Output order: by read; within a read by descending score, earlier reference first on ties (the reference's multiset
  // order, src/qmodel.cpp:2773-2775).  Records arrive in device order from two slots: bucket them by read (one pass),
  // then order the few reads that have several (QF_ALIGN_ALL).
  const size_t runs_first = c->h_runs.size();
  c->h_runs.append(c->second.h_runs.data(), c->second.h_runs.size());
  if (dense) {
    // best alignment per read: the device wrote final records at their read's index.  One piece: nothing to do.  Several
    // pieces: rebase the run offsets.  Reads without any alignment (holes) are squeezed out.
    if (c->dense_chunks.size() > 1 || runs_first != c->h_runs.size())
      for (const auto& dc : c->dense_chunks) {
        const size_t add = dc.runs0 + (dc.second ? runs_first : 0);
        if (add) for (uint32_t r = dc.lo; r < dc.hi; ++r) c->h_align[r].run_offset += add;
      }
    if (out->n_alignments != n_reads) {
      uint32_t w = 0;
      for (uint32_t r = 0; r < n_reads; ++r)
        if (c->h_align[r].n_runs != kAlignHole) c->h_align[w++] = c->h_align[r];
    }
    out->viterbi = c->h_viterbi.data();
    out->cells = c->h_cells.data();
    out->n_diagonals = c->h_ndiag.data();
    out->null_loglike = c->h_nll.data();
    out->alignments = c->h_align.data();
    out->cigar_runs = c->h_runs.data();
    out->n_fill_classes = kNumClasses;
    out->ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return QF_OK;
  }
  const uint32_t n_first = (uint32_t)c->h_recs.size(), n_recs = n_first + (uint32_t)c->second.h_recs.size();
  auto rec_at = [&](uint32_t k) -> const AlignRec& { return k < n_first ? c->h_recs[k] : c->second.h_recs[k - n_first]; };
  std::vector<uint32_t> start(n_reads + 1, 0), order(n_recs);
  for (uint32_t k = 0; k < n_recs; ++k) ++start[rec_at(k).read + 1];
  for (uint32_t r = 0; r < n_reads; ++r) start[r + 1] += start[r];
  {
    std::vector<uint32_t> cursor(start.begin(), start.end() - 1);
    for (uint32_t k = 0; k < n_recs; ++k) order[cursor[rec_at(k).read]++] = k;
  }
  for (uint32_t r = 0; r < n_reads; ++r)
    if (start[r + 1] - start[r] > 1)
      std::sort(order.begin() + start[r], order.begin() + start[r + 1], [&](uint32_t x, uint32_t y) {
        const AlignRec &p = rec_at(x), &q = rec_at(y);
        if (p.score != q.score) return p.score > q.score;
        return p.ref < q.ref;
      });
  c->h_align.resize(n_recs);
  for (uint32_t a = 0; a < n_recs; ++a) {
    const AlignRec& r = rec_at(order[a]);
    if (!r.ok) return fail(c, QF_ERR_DEVICE, "traceback did not reach the start state (read " + std::to_string(r.read) + ")");
    qf_alignment& o = c->h_align[a];
    o.read = r.read;
    o.ref = r.ref;
    o.viterbi = r.viterbi;
    o.score = r.score;
    o.x_start = r.x_start;
    o.x_end = r.x_end;
    o.n_columns = r.n_columns;
    o.n_runs = r.n_runs;
    o.run_offset = r.run_off + (order[a] >= n_first ? runs_first : 0);
  }
  out->viterbi = c->h_viterbi.data();
  out->cells = c->h_cells.data();
  out->n_diagonals = c->h_ndiag.data();
  out->null_loglike = c->h_nll.data();
  out->n_alignments = n_recs;
  out->alignments = c->h_align.data();
  out->cigar_runs = c->h_runs.data();
  out->n_fill_classes = kNumClasses;
  out->ms_total = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  return QF_OK;
}

// QUAFF_HIP_DEBUG_FLAGS: switches OR-ed into every context's (developer A/B: a whole test run on a kernel variant)
static uint32_t env_debug_flags() {
  const char* e = getenv("QUAFF_HIP_DEBUG_FLAGS");
  return e ? (uint32_t)strtoul(e, nullptr, 0) : 0u;
}
int qf_debug_set_flags(qf_ctx* c, uint32_t flags) {
  if (!c) return QF_ERR_ARG;
  c->debug = flags | env_debug_flags();
  return QF_OK;
}

static int ensure_lse(qf_ctx* c);
static std::vector<uint8_t> pack_lse_table(const std::vector<double>& tab);
uint32_t qf_debug_pack_lse_table(uint8_t* out, uint32_t cap) {
  const std::vector<uint8_t> pack = pack_lse_table(lse_table());
  if (out) memcpy(out, pack.data(), std::min<size_t>(cap, pack.size()));
  return (uint32_t)pack.size();
}
uint64_t qf_debug_rows_settled(const qf_ctx* c) { return c ? c->rows_settled : 0; }
double qf_debug_alloc_ms(void) { return (double)g_alloc_us.load() * 1e-3; }
int qf_debug_fail_chunk_reserve(int nth) { return g_fail_chunk_reserve.exchange(nth); }
uint32_t qf_debug_lse_pack_bytes(qf_ctx* c) {
  if (!c || hipSetDevice(c->device) != hipSuccess || ensure_lse(c) != QF_OK) return 0;
  return c->lse_pack_bytes;
}

int qf_set_pipeline_chunks(qf_ctx* c, uint32_t n_chunks) {
  if (!c) return QF_ERR_ARG;
  c->pipeline_chunks = n_chunks;
  return QF_OK;
}

int qf_set_score_threshold(qf_ctx* c, double min_score) {
  if (!c) return QF_ERR_ARG;
  if (min_score != min_score) return fail(c, QF_ERR_ARG, "score threshold is NaN");
  c->min_score = min_score;
  return QF_OK;
}

int qf_set_memory_budget(qf_ctx* c, uint64_t bytes) {
  if (!c) return QF_ERR_ARG;
  c->tb_budget = bytes;
  return QF_OK;
}

int qf_align_batch(qf_ctx* c, const qf_dp_config* cfg, const char* seq, const char* qual, const uint64_t* offsets,
                   uint32_t n_reads, uint32_t flags, qf_align_result* out) {
  if (int rc = qf_upload_reads(c, seq, qual, offsets, n_reads)) return rc;
  return qf_align_resident(c, cfg, flags, out);
}

// ------------------------------------------------------------------------- Forward-Backward E-step
// Device copy of the reference's 100 001-entry log(1 + exp(-x)) table (src/logsumexp.cpp:20-28; exact-bits paths: overlap
// gap states, per-read / per-pair sums) followed by the same function as kLsePieces quadratic pieces on a 1/128 grid for the
// 1e-4-tolerance Forward / Backward fills (qf_fb.hip: lseh).  Piece n covers x in [n/128, (n+1)/128): the quadratic through
// the function at the three Chebyshev nodes of the piece, in t = 128 x - n, with the linear and quadratic coefficients
// rounded to fp32 (16 bytes per piece): within 3.3e-10 of log1p(exp(-x)) everywhere, the accuracy of the reference's own
// 1e-4-step linear table.  The last piece is all zero: x >= 10 is the reference's cut-off (:84-90).
// The exact table packed for LDS (qf_device.hpp).  Pieces: Chebyshev interpolants of log1p(exp(-x)) in extended precision,
// coefficients rounded to fp64; corrections: bit pattern of the table entry minus bit pattern of the piece evaluated with the
// device's sequence of fused multiply-adds.  Returns an empty vector if a piece would need fields wider than 16 bits (a libm
// whose exp/log are off by more than the 1 + exp(-x) rounding this scheme is sized for).
static std::vector<uint8_t> pack_lse_table(const std::vector<double>& tab) {
  std::vector<uint8_t> head(kLsePackStream, 0);
  std::vector<uint32_t> words;
  uint64_t bit = 0;
  constexpr int K = kLsePackDegree + 1, W = kLsePackSpan;
  auto pattern = [](double d) { int64_t b; memcpy(&b, &d, 8); return b; };
  for (int p = 0; p < kLsePackPieces; ++p) {
    long double node[K], dd[K];
    for (int k = 0; k < K; ++k) {
      node[k] = (W / 2) * cosl(M_PIl * (2 * k + 1) / (2 * K));
      dd[k] = log1pl(expl(-((long double)(W * p + W / 2) + node[k]) * 1e-4L));
    }
    for (int j = 1; j < K; ++j)
      for (int k = K - 1; k >= j; --k) dd[k] = (dd[k] - dd[k - 1]) / (node[k] - node[k - j]);
    std::vector<long double> poly(1, dd[K - 1]);   // Newton form -> monomials in u
    for (int k = K - 2; k >= 0; --k) {
      std::vector<long double> np(poly.size() + 1, 0.0L);
      for (size_t i = 0; i < poly.size(); ++i) { np[i + 1] += poly[i]; np[i] -= node[k] * poly[i]; }
      np[0] += dd[k];
      poly.swap(np);
    }
    struct { double c[K]; uint32_t bit_base, width; } pc;
    for (int i = 0; i < K; ++i) pc.c[i] = (double)poly[i];
    int64_t corr[W + 1], lo = 0, hi = 0;
    for (int t = 0; t <= W; ++t) {
      const int n = std::min(W * p + t, kLseEntries - 1);   // the last piece ends at entry 100 000
      const double u = (double)(t - W / 2);
      double v = pc.c[kLsePackDegree];
      for (int i = kLsePackDegree - 1; i >= 0; --i) v = std::fma(v, u, pc.c[i]);
      corr[t] = W * p + t < kLseEntries ? pattern(tab[n]) - pattern(v) : 0;
      lo = std::min(lo, corr[t]); hi = std::max(hi, corr[t]);
    }
    uint32_t w = 1;
    while (lo < -(1ll << (w - 1)) || hi > (1ll << (w - 1)) - 1) ++w;
    if (w > 16) return {};
    pc.bit_base = (uint32_t)bit;
    pc.width = w;
    for (int t = 0; t <= W; ++t, bit += w) {
      const uint64_t field = (uint64_t)corr[t] & ((1ull << w) - 1);
      words.resize((bit + w + 63) / 32 + 1, 0u);
      words[bit >> 5] |= (uint32_t)(field << (bit & 31));
      if ((bit & 31) + w > 32) words[(bit >> 5) + 1] |= (uint32_t)(field >> (32 - (bit & 31)));
    }
    memcpy(&head[kLsePackC01 + p * 16], &pc.c[0], 16);
    memcpy(&head[kLsePackC23 + p * 16], &pc.c[2], 16);
    memcpy(&head[kLsePackC45 + p * 16], &pc.c[4], 16);
    memcpy(&head[kLsePackMeta + p * 8], &pc.bit_base, 8);
  }
  words.resize((words.size() + 5) & ~(size_t)3, 0u);   // a look-up reads two words; whole 16-byte blocks for the copy to LDS
  std::vector<uint8_t> out(kLsePackStream + words.size() * 4);
  memcpy(out.data(), head.data(), kLsePackStream);
  memcpy(out.data() + kLsePackStream, words.data(), words.size() * 4);
  return out;
}

static int ensure_lse(qf_ctx* c) {
  if (c->lse_uploaded) return QF_OK;
  std::vector<double> t = lse_table();
  t.resize(kLseHermiteOffset, 0.0);
  static_assert(sizeof(LsePiece) == 16, "one 16-byte LDS read per lookup");
  std::vector<LsePiece> pieces(kLsePieces);
  const double h = 1.0 / 128.0;
  auto g = [](double x) { return std::log1p(std::exp(-x)); };
  const double tn[3] = {0.5 - 0.5 * std::cos(M_PI / 6), 0.5, 0.5 - 0.5 * std::cos(5 * M_PI / 6)};   // Chebyshev nodes on [0, 1]
  for (int n = 0; n < kLsePieces - 1; ++n) {
    const double y0 = g((n + tn[0]) * h), y1 = g((n + tn[1]) * h), y2 = g((n + tn[2]) * h);
    // Newton form through (tn[k], yk), expanded to monomials in t
    const double d01 = (y1 - y0) / (tn[1] - tn[0]), d12 = (y2 - y1) / (tn[2] - tn[1]), d012 = (d12 - d01) / (tn[2] - tn[0]);
    const double c2 = d012, c1 = d01 - d012 * (tn[0] + tn[1]), c0 = y0 - tn[0] * d01 + d012 * tn[0] * tn[1];
    pieces[n] = LsePiece{c0, (float)c1, (float)c2};
  }
  pieces[kLsePieces - 1] = LsePiece{0.0, 0.f, 0.f};
  const size_t bytes = kLseHermiteOffset * 8 + pieces.size() * sizeof(LsePiece);
  HIPCHK(c, c->d_lse.reserve(bytes));
  HIPCHK(c, hipMemcpy(c->d_lse.p, t.data(), kLseHermiteOffset * 8, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_lse.as<char>() + kLseHermiteOffset * 8, pieces.data(), pieces.size() * sizeof(LsePiece), hipMemcpyHostToDevice));
  // packed form for the overlap fills: used only if this device rebuilds every entry from it bit for bit
  {
    const std::vector<uint8_t> pack = pack_lse_table(lse_table());
    int lds_max = 0;
    (void)hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device);
    if (!pack.empty() && pack.size() <= (size_t)lds_max) {
      DevBuf bad;
      HIPCHK(c, c->d_lse_pack.reserve(pack.size()));
      HIPCHK(c, bad.reserve(4));
      HIPCHK(c, hipMemcpy(c->d_lse_pack.p, pack.data(), pack.size(), hipMemcpyHostToDevice));
      if (lse_pack_mismatches(c->d_lse_pack.as<uint8_t>(), (uint32_t)pack.size(), c->d_lse.as<double>(), bad.as<uint32_t>(), c->stream) == 0)
        c->lse_pack_bytes = (uint32_t)pack.size();
      bad.release();
      (void)hipGetLastError();
    }
  }
  c->lse_uploaded = true;
  return QF_OK;
}

uint32_t qf_counts_size(const qf_ctx* c) {
  if (!c || !c->have_params) return 0;
  return (uint32_t)((4 + 4 * c->scores.Km) * kNQual + 4 * c->scores.Kg + 4);
}

// Classes of one phase on concurrent streams (the class with the most cells on the main stream, the others on the low-
// priority side streams), joined back into the main stream: small classes fill the tail of the big one.
static int launch_classes_concurrently(Slot* c, const BatchCounters& bc, bool serial,
                                       const std::function<void(int, hipStream_t)>& launch, int first_cls = 1,
                                       bool small_first = false, hipEvent_t* ev_begin = nullptr, hipEvent_t* ev_end = nullptr) {
  if (!ev_begin) { ev_begin = c->cls_ev; ev_end = c->cls_end; }
  int order[kNumClasses], n_used = 0;
  for (int cls = first_cls; cls < kNumClasses; ++cls) if (bc.cls_count[cls]) order[n_used++] = cls;
  std::sort(order, order + n_used, [&](int p, int q) { return bc.cls_cells[p] > bc.cls_cells[q]; });
  if (n_used > 1 && !serial) {   // side streams start after everything queued on the main stream so far
    HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
    for (auto& s : c->aux) HIPCHK(c, hipStreamWaitEvent(s, c->ev[6], 0));
    if (small_first) for (auto& s : c->hi) HIPCHK(c, hipStreamWaitEvent(s, c->ev[6], 0));
  }
  // small_first (overlap): a class with fewer wavefronts than the chip holds cannot hide its own latency — every step of a
  // band is a chain of dependent table lookups — and behind millions of single-diagonal bands at equal or lower priority it
  // crawls and ends up setting the length of the phase.  Such classes go to high-priority streams and start first.
  auto waves_of = [&](int cls) -> uint64_t {
    const uint64_t n = bc.cls_count[cls];
    if (cls == 0) return (n + 63) / 64;
    if (cls == kRowClass) return n;
    return (n * (uint64_t)fill_class(cls).G + 63) / 64;
  };
  int k_big = 0, k_small = 0, joins[kNumClasses], n_join = 0;
  if (small_first && !serial && n_used > 1)
    for (int k = 0; k < n_used; ++k) {
      const int cls = order[k];
      if (waves_of(cls) > 4096) continue;
      hipStream_t s = c->hi[k_small++ % 3];
      HIPCHK(c, hipEventRecord(ev_begin[cls], s));
      launch(cls, s);
      HIPCHK(c, hipEventRecord(ev_end[cls], s));
      joins[n_join++] = cls;      // the main stream joins them after its own class has been queued
      order[k] = -1;
    }
  for (int k = 0; k < n_used; ++k) {
    const int cls = order[k];
    if (cls < 0) continue;
    const int lane = serial ? 0 : (k_big < 4 ? k_big : 1 + (k_big - 1) % 3);
    ++k_big;
    hipStream_t s = lane == 0 ? c->stream : c->aux[lane - 1];
    HIPCHK(c, hipEventRecord(ev_begin[cls], s));
    launch(cls, s);
    HIPCHK(c, hipEventRecord(ev_end[cls], s));
    if (lane) HIPCHK(c, hipStreamWaitEvent(c->stream, ev_end[cls], 0));
  }
  for (int k = 0; k < n_join; ++k) HIPCHK(c, hipStreamWaitEvent(c->stream, ev_end[joins[k]], 0));
  return QF_OK;
}

// Forward-Backward over reads [lo, hi) of the resident set; counts accumulate in d_counts, per-read / per-pair results go to
// the host arrays at the chunk's offsets.  Sets *too_big (and does nothing) when the Forward matrices exceed the budget.
static int count_chunk_impl(qf_ctx* c, Slot* S, const qf_dp_config* cfg, bool use_null, bool have_sort, uint32_t lo, uint32_t hi,
                       int slots_in_flight, qf_count_result* out, std::mutex& out_mu, bool* too_big) {
  *too_big = false;
  const uint32_t n_reads = hi - lo, n_refs = c->n_refs;
  const uint32_t n_pairs = n_reads * n_refs;
  const size_t p0 = (size_t)lo * n_refs;
  const bool sparse = cfg->sparse != 0;
  const bool mem = sparse && cfg->kmer_threshold < 0;
  uint32_t max_units = n_pairs * 4 + 1024;
  const uint32_t csize = qf_counts_size(c);
  const uint64_t* d_roff = c->d_roff.as<uint64_t>() + lo;
  const uint8_t* d_skip = have_sort ? c->d_skip.as<uint8_t>() + p0 : nullptr;
  HIPCHK(S, hipEventRecord(S->ev[1], S->stream));

  // ---- seeding (cellSize = 2 * 24 for counting, qmodel.cpp:2249; only matters in memory mode)
  const int max_nd = (int)(c->ref_maxlen + c->read_maxlen - 1);
  SeedArgs sa;
  BatchCounters bc;
  if (int rc = seed_pairs(c, S, cfg, n_pairs, mem, sparse ? max_nd : 2, [&](SeedArgs& s) {
        s.read_off = d_roff;
        s.cell_size = 48;
        s.storage_mode = 1;
        s.fb_use_32x3 = (c->debug & QF_DEBUG_FB32) != 0;
        s.pair_skip = d_skip;
      }, max_units, sa, bc))
    return rc;
  HIPCHK(S, hipEventRecord(S->ev[2], S->stream));
  if (bc.error & 4u) return fail(S, QF_ERR_SYMBOL, "Unknown symbol in read " + std::to_string(bc.error_detail));
  if (bc.error & 2u)
    return fail(S, QF_ERR_UNSUPPORTED, "envelope band of " + std::to_string(bc.error_detail) + " diagonals exceeds the diagonal-space kernels");

  // ---- Forward
  const uint64_t fw_bytes = (uint64_t)bc.tb_words * 8;
  if (fw_bytes > chunk_budget(c, S->d_fw, slots_in_flight)) {
    if (n_reads == 1) return fail(S, QF_ERR_MEMORY, "one read needs " + std::to_string(fw_bytes >> 20) + " MiB of Forward matrices, over the memory budget");
    *too_big = true;
    return QF_OK;
  }
  if (reserve_big(S->d_fw, fw_bytes + 64, {&c->d_tb, &c->second.d_tb}) != hipSuccess) {
    if (n_reads == 1) return fail(S, QF_ERR_MEMORY, "cannot allocate " + std::to_string(fw_bytes >> 20) + " MiB of Forward matrices for one read");
    *too_big = true;
    return QF_OK;
  }
  if (int rc = sort_class_lists(c, S, bc, max_units)) return rc;
  CHUNKRES(S, S->d_weight, (size_t)n_pairs * 8);
  CHUNKRES(S, S->d_fwd_out, (size_t)n_pairs * 8);
  CHUNKRES(S, S->d_order_out, (size_t)n_pairs * 4);
  CHUNKRES(S, S->d_order_n_out, (size_t)n_reads * 4);
  CHUNKRES(S, S->d_rll, (size_t)n_reads * 8);
  const Scores& sc = c->scores;
  FbArgs fa{};
  fa.n_refs = n_refs;
  fa.units = S->d_units.as<Unit>();
  fa.ref_off = c->d_ref_off.as<uint64_t>();
  fa.ref_woff = c->d_ref_woff.as<uint64_t>();
  fa.ref_tok = c->d_ref_tok.as<uint8_t>();
  fa.ref_packed = c->d_ref_packed.as<uint32_t>();
  fa.read_off = d_roff;
  fa.ctx = c->d_ctx.as<uint32_t>() + kCtxPad;
  fa.fw = S->d_fw.as<double>();
  fa.lse = c->d_lse.as<double>();
  fa.lse_h = c->d_lse.as<double>() + kLseHermiteOffset;
  fa.dp.ematch = c->d_ematch.as<double>();
  fa.dp.ematch_ninf_off = (uint32_t)((size_t)c->scores.Km * kNQ1 * 4 * 8);
  if (c->count_qmajor) {   // the slice of the qualities in use, quality-major (qf_count_resident)
    fa.dp.ematch = c->d_ematch_q.as<double>();
    fa.dp.ematch_ninf_off = (uint32_t)((size_t)(c->read_qmax - c->read_qmin + 1) * sc.Km * 32);
    fa.em_qmajor = 1;
    fa.em_kshift = 2 * sc.match_len;
    fa.em_qmin = c->read_qmin;
    // Backward runs two workgroups per CU (its registers decide that), so each may take half the CU's 160 KB: 16.9 -> 15.1 ms
    // on config 4.  Forward runs three on 53 KB each, which the order-2 slice does not fit beside the log-sum-exp pieces: it
    // keeps the table in global memory unless the slice is small (QF_DEBUG_BIG_FORWARD_LDS: two workgroups with the table)
    fa.lds_limit = 78 * 1024;
  fa.flush_lds = (c->debug & QF_DEBUG_FLUSH_GLOBAL) ? 1u : (c->debug & QF_DEBUG_FLUSH_SLICES) ? 12u * 1024u : 0u;
  }
  fa.dp.eins = c->d_eins.as<double>();
  fa.dp.trans = c->d_trans.as<double>();
  fa.dp.d2d = sc.trans[4 * sc.Kg + 0];
  fa.dp.d2m = sc.trans[4 * sc.Kg + 1];
  fa.dp.i2i = sc.trans[4 * sc.Kg + 2];
  fa.dp.i2m = sc.trans[4 * sc.Kg + 3];
  fa.dp.Kg = sc.Kg;
  fa.dp.local = cfg->local;
  fa.pair_fwd = S->d_pair_score.as<double>();
  fa.pair_weight = S->d_weight.as<double>();
  fa.counts = c->d_counts.as<unsigned long long>();
  fa.counts_stride = (csize + 31) & ~31ull;
  fa.Km = sc.Km;
  fa.no_band_shortcuts = (c->debug & QF_DEBUG_NO_BAND_SHORTCUTS) != 0;
  const bool serial_classes = (c->debug & QF_DEBUG_SERIAL_CLASSES) != 0;
  if (int rc = launch_classes_concurrently(S, bc, serial_classes, [&](int cls, hipStream_t s) {
        FbArgs f2 = fa;
        f2.n_cls_units = bc.cls_count[cls];
        f2.cls_list = S->d_cls_list.as<uint32_t>() + (size_t)cls * max_units;
        // Forward keeps the quality-major slice in global memory (lds_limit 1: nothing fits): its LDS pipe is already busy with the
        // log-sum-exp pieces.  Measured: order 2, 11.9 ms from global memory at three workgroups per CU against 12.1 ms from LDS
        // at two; order 1 (an 11 KB slice, three workgroups either way), 11.9 against 12.6 ms.
        if (c->count_qmajor && !(c->debug & QF_DEBUG_BIG_FORWARD_LDS)) f2.lds_limit = 1;
        launch_forward_fill(cls, f2, s);
      }, 0, true))   // classes with few wavefronts first, on the high-priority streams: each of their wavefronts still runs its ~1 000 dependent steps
    return rc;
  FinalArgs fin{};
  fin.n_pairs = n_pairs;
  fin.n_reads = n_reads;
  fin.n_refs = n_refs;
  fin.units = S->d_units.as<Unit>();
  fin.pair_head = S->d_pair_head.as<uint32_t>();
  fin.pair_score = S->d_pair_score.as<double>();
  fin.read_off = d_roff;
  fin.ref_off = c->d_ref_off.as<uint64_t>();
  fin.fw = S->d_fw.as<double>();
  fin.ctx = c->d_ctx.as<uint32_t>() + kCtxPad;
  fin.trans = fa.dp.trans;
  fin.Kg = fa.dp.Kg;
  fin.local = fa.dp.local;
  launch_pair_forward(fin, c->d_lse.as<double>(), S->stream);
  HIPCHK(S, hipGetLastError());
  HIPCHK(S, hipEventRecord(S->ev[3], S->stream));

  // ---- per-read plan: running log-likelihood, Backward flags, weights, next order
  CountPlanArgs pa{};
  pa.n_reads = n_reads;
  pa.n_refs = n_refs;
  pa.use_null = use_null;
  pa.nll = c->d_nll.as<double>() + lo;
  pa.lse = c->d_lse.as<double>();
  pa.pair_fwd = S->d_pair_score.as<double>();
  pa.pair_fwd_out = S->d_fwd_out.as<double>();
  pa.weight = S->d_weight.as<double>();
  pa.order_in = have_sort ? c->d_order_in.as<uint32_t>() + p0 : nullptr;
  pa.order_n_in = have_sort ? c->d_order_n_in.as<uint32_t>() + lo : nullptr;
  pa.order_out = S->d_order_out.as<uint32_t>();
  pa.order_n_out = S->d_order_n_out.as<uint32_t>();
  pa.read_loglike = S->d_rll.as<double>();
  launch_count_plan(pa, S->stream);
  HIPCHK(S, hipGetLastError());
  HIPCHK(S, hipEventRecord(S->ev[4], S->stream));

  // ---- Backward + counts
  // The class with the most cells sets the length of the phase, and what follows its last wavefront is exposed: k_count_flush
  // over all its bands (1 ms of the bench's 26).  Its first three quarters go to a high-priority stream as a launch of their own:
  // they are dispatched first and done first, and their flush runs beside the last quarter's Backward, whose own flush is short.
  int big_cls = -1;
  for (int cls = 0; cls < kNumClasses; ++cls)
    if (bc.cls_count[cls] && cls != kRowClass && (big_cls < 0 || bc.cls_cells[cls] > bc.cls_cells[big_cls])) big_cls = cls;
  if (int rc = launch_classes_concurrently(S, bc, serial_classes, [&](int cls, hipStream_t s) {
        FbArgs f2 = fa;
        f2.n_cls_units = bc.cls_count[cls];
        f2.cls_list = S->d_cls_list.as<uint32_t>() + (size_t)cls * max_units;
        const uint32_t tail = cls == big_cls && !serial_classes && f2.n_cls_units >= 8192 && !(c->debug & QF_DEBUG_NO_BACKWARD_SPLIT) ? f2.n_cls_units / 4 : 0;
        if (tail) {
          const uint32_t head = f2.n_cls_units - tail;
          (void)hipEventRecord(S->ev_split[0], s);
          (void)hipStreamWaitEvent(S->hi[2], S->ev_split[0], 0);
          f2.n_cls_units = head;
          launch_backward_fill(cls, f2, S->hi[2]);
          (void)hipEventRecord(S->ev_split[1], S->hi[2]);
          f2.cls_list += head;
          f2.n_cls_units = tail;
          launch_backward_fill(cls, f2, s);
          (void)hipStreamWaitEvent(s, S->ev_split[1], 0);
        } else launch_backward_fill(cls, f2, s);
      }, 0, true, S->cls_ev2, S->cls_end2))
    return rc;
  HIPCHK(S, hipGetLastError());
  HIPCHK(S, hipEventRecord(S->ev[5], S->stream));

  HIPCHK(S, hipMemcpyAsync(c->h_fwd.data() + p0, S->d_fwd_out.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(S, hipMemcpyAsync(c->h_weight.data() + p0, S->d_weight.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(S, hipMemcpyAsync(c->h_rll.data() + lo, S->d_rll.p, (size_t)n_reads * 8, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(S, hipMemcpyAsync(c->h_order.data() + p0, S->d_order_out.p, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(S, hipMemcpyAsync(c->h_order_n.data() + lo, S->d_order_n_out.p, (size_t)n_reads * 4, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(S, hipMemcpyAsync(c->h_cells.data() + p0, S->d_pair_cells.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, S->stream));
  HIPCHK(S, hipStreamSynchronize(S->stream));
  std::lock_guard<std::mutex> lk(out_mu);
  out->total_cells += bc.total_cells;
  out->forward_bytes += fw_bytes;
  float ms = 0;
  (void)hipEventElapsedTime(&ms, S->ev[1], S->ev[2]); out->ms_seed += ms;
  (void)hipEventElapsedTime(&ms, S->ev[2], S->ev[3]); out->ms_forward += ms;
  (void)hipEventElapsedTime(&ms, S->ev[3], S->ev[4]); out->ms_plan += ms;
  (void)hipEventElapsedTime(&ms, S->ev[4], S->ev[5]); out->ms_backward += ms;
  (void)hipEventElapsedTime(&ms, S->ev[1], S->ev[5]); out->ms_total += ms;
  for (int cls = 0; cls < kNumClasses; ++cls) {
    if (!bc.cls_count[cls]) continue;
    ms = 0; (void)hipEventElapsedTime(&ms, S->cls_ev[cls], S->cls_end[cls]); out->ms_forward_class[cls] += ms;
    ms = 0; (void)hipEventElapsedTime(&ms, S->cls_ev2[cls], S->cls_end2[cls]); out->ms_backward_class[cls] += ms;
    out->cells_class[cls] += bc.cls_cells[cls];
    out->units_class[cls] += bc.cls_count[cls];
  }
  out->n_fill_classes = kNumClasses;
  (void)csize;
  return QF_OK;
}

static int count_chunk(qf_ctx* c, Slot* S, const qf_dp_config* cfg, bool use_null, bool have_sort, uint32_t lo, uint32_t hi,
                       int slots_in_flight, qf_count_result* out, std::mutex& out_mu, bool* too_big) {
  const int rc = count_chunk_impl(c, S, cfg, use_null, have_sort, lo, hi, slots_in_flight, out, out_mu, too_big);
  return split_instead(c, S, rc, hi - lo <= 1, too_big, {&c->d_tb, &c->second.d_tb});
}

int qf_count_resident(qf_ctx* c, const qf_dp_config* cfg, uint32_t flags, const uint32_t* sort_in,
                      const uint32_t* sort_n_in, qf_count_result* out) {
  if (int rc = check_cfg(c, cfg)) return rc;
  if (!out) return fail(c, QF_ERR_ARG, "null result");
  if ((sort_in == nullptr) != (sort_n_in == nullptr)) return fail(c, QF_ERR_ARG, "sort_in and sort_n_in go together");
  HIPCHK(c, hipSetDevice(c->device));
  const CallInProgress in_progress(c->device);
  memset(out, 0, sizeof *out);
  const uint32_t n_reads = c->n_reads, n_refs = c->n_refs;
  const uint64_t n_pairs64 = (uint64_t)n_reads * n_refs;
  if (n_pairs64 > kMaxPairsPerCall) return fail(c, QF_ERR_UNSUPPORTED, "more than 2^28 read x reference pairs in one batch");
  const uint32_t n_pairs = (uint32_t)n_pairs64;
  out->n_reads = n_reads;
  out->n_refs = n_refs;
  const uint32_t csize = qf_counts_size(c);
  out->counts_size = csize;
  c->h_pcounts.assign(csize, 0.0);
  c->h_pcounts_fx.assign((size_t)csize * 2, 0);
  out->counts = c->h_pcounts.data();
  out->counts_exact = c->h_pcounts_fx.data();
  if (!n_pairs) return QF_OK;
  if (!c->reads_have_qual)  // QuaffBackwardMatrix ctor, src/qmodel.cpp:1398
    return fail(c, QF_ERR_ARG, "Forward-Backward algorithm requires quality scores to fit model");
  const bool use_null = !(flags & QF_COUNT_FORCE);
  if (use_null && !c->have_null) return fail(c, QF_ERR_STATE, "no null model set (qf_set_null_json) and QF_COUNT_FORCE not given");
  const bool sparse = cfg->sparse != 0;
  if (sparse) if (int rc = ensure_ref_index(c, cfg->kmer_len)) return rc;
  if (int rc = ensure_lse(c)) return rc;

  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  HIPCHK(c, c->d_bc.reserve(sizeof(BatchCounters)));
  HIPCHK(c, hipMemsetAsync(c->d_bc.p, 0, sizeof(BatchCounters), c->stream));
  // Match-emission table of the fills.  Every cell gathers one 8-byte entry of it, 64 different rows per wavefront: from LDS
  // that is cheap, from L2 it is a large part of the fills' time.  The whole table -- (context k-mer, quality 0 .. 94) rows of 32
  // bytes -- fits LDS beside the log-sum-exp pieces only for one-base contexts (12 KB); at -order 1 it is 49 KB, at -order 2
  // 195 KB.  But a read set uses few quality values (Phred 5 .. 25 in the synthetic sets, ~40 in real ones), so the E-step
  // numbers the rows QUALITY-major -- (q - qmin) Km + k-mer, in the reads' context words (prep) and in a re-ordered copy of
  // the table -- and the rows of the qualities in use are one contiguous slice: 43 KB at -order 2 for 21 qualities.
  c->count_qmajor = false;
  {
    const uint32_t qspan = c->read_qmax - c->read_qmin + 1;
    const size_t full = (size_t)sc_Km(c) * kNQ1 * 32, slice = (size_t)qspan * sc_Km(c) * 32;
    if (full > 24 * 1024 && slice <= 48 * 1024 && !(c->debug & QF_DEBUG_GLOBAL_TABLES)) {
      const Scores& s = c->scores;
      if (c->ematch_q_params_epoch != c->params_epoch || c->ematch_q_lo != c->read_qmin || c->ematch_q_hi != c->read_qmax) {
        std::vector<double> em((size_t)qspan * s.Km * 4 + 4, -INFINITY);   // + the -inf row slots outside a band read
        for (uint32_t t = 0; t < 4; ++t)
          for (uint32_t k = 0; k < s.Km; ++k)
            for (uint32_t q = c->read_qmin; q <= c->read_qmax; ++q)
              em[((size_t)(q - c->read_qmin) * s.Km + k) * 4 + t] = s.mat[((size_t)t * s.Km + k) * kNQ1 + q];
        HIPCHK(c, c->d_ematch_q.reserve(em.size() * 8));
        HIPCHK(c, hipMemcpy(c->d_ematch_q.p, em.data(), em.size() * 8, hipMemcpyHostToDevice));
        c->ematch_q_params_epoch = c->params_epoch;
        c->ematch_q_lo = c->read_qmin;
        c->ematch_q_hi = c->read_qmax;
      }
      c->count_qmajor = true;
    }
  }
  c->prep_qmajor_Km = c->count_qmajor ? c->scores.Km : 0;
  const int prep_rc = prep_reads(c, sparse ? cfg->kmer_len : 0);
  c->prep_qmajor_Km = 0;     // (the other entry points number the rows k-mer major; every call derives its own context words)
  if (prep_rc) return prep_rc;
  // pairs outside the read's reference order are not even seeded (their LL stays -inf, qmodel.cpp:2245)
  HIPCHK(c, c->d_order_in.reserve((size_t)n_pairs * 4));
  HIPCHK(c, c->d_order_n_in.reserve((size_t)n_reads * 4));
  if (sort_in) {
    std::vector<uint8_t> skip(n_pairs, 1);
    for (uint32_t r = 0; r < n_reads; ++r) {
      if (sort_n_in[r] > n_refs) return fail(c, QF_ERR_ARG, "sort_n_in out of range");
      for (uint32_t k = 0; k < sort_n_in[r]; ++k) {
        const uint32_t x = sort_in[(size_t)r * n_refs + k];
        if (x >= n_refs) return fail(c, QF_ERR_ARG, "sort_in out of range");
        skip[(size_t)r * n_refs + x] = 0;
      }
    }
    HIPCHK(c, c->d_skip.reserve(n_pairs));
    HIPCHK(c, hipMemcpyAsync(c->d_skip.p, skip.data(), n_pairs, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_order_in.p, sort_in, (size_t)n_pairs * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_order_n_in.p, sort_n_in, (size_t)n_reads * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // `skip` is a stack-lifetime host buffer
  }
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));

  {
    BatchCounters pb;
    if (int rc = read_counters(c, pb)) return rc;
    if (pb.error & 4u) return fail(c, QF_ERR_SYMBOL, "Unknown symbol in read " + std::to_string(pb.error_detail));
  }
  const size_t counts_stride = (csize + 31) & ~(size_t)31;
  // accumulators: 128-bit fixed point (two words per entry), kCountReplicas copies; behind them the summed words and doubles
  HIPCHK(c, c->d_counts.reserve(counts_stride * kCountReplicas * 16 + (size_t)csize * 24));
  HIPCHK(c, hipMemsetAsync(c->d_counts.p, 0, counts_stride * kCountReplicas * 16, c->stream));
  c->h_fwd.resize(n_pairs);
  c->h_weight.resize(n_pairs);
  c->h_rll.resize(n_reads);
  c->h_counts_fx.resize((size_t)csize * 2);
  c->h_order.resize(n_pairs);
  c->h_order_n.resize(n_reads);
  c->h_cells.resize(n_pairs);
  (void)hipEventElapsedTime(&out->ms_prep, c->ev[0], c->ev[1]);
  out->ms_total = out->ms_prep;
  // Work list: the batch in `n_chunks` pieces (qf_set_pipeline_chunks; default one), two in flight (one slot = stream +
  // buffers each, one host thread per slot); a piece whose Forward storage exceeds the memory budget is halved.  Measured on
  // config 4 (MI355X): both fills are bound by fp64 issue and LDS lookups, not by the tails of their grids, so pieces in
  // flight buy nothing there (31.1 / 30.4 / 37.1 ms per E-step with 1 / 2 / 4 pieces); what they do is bound the memory.
  uint32_t n_chunks = c->pipeline_chunks ? c->pipeline_chunks : 1u;
  if (c->debug & QF_DEBUG_SERIAL_CLASSES) n_chunks = 1;
  n_chunks = std::min(n_chunks, n_reads);
  const auto t_begin = std::chrono::steady_clock::now();
  {
    std::vector<std::pair<uint32_t, uint32_t>> todo;
    for (uint32_t k = n_chunks; k-- > 0;)
      todo.push_back({(uint32_t)((uint64_t)n_reads * k / n_chunks), (uint32_t)((uint64_t)n_reads * (k + 1) / n_chunks)});
    std::mutex mu, out_mu;
    int in_flight = 0, rc_all = QF_OK;
    std::condition_variable cv;
    const int slots = n_chunks > 1 ? 2 : 1;
    auto worker = [&](Slot* S) {
      (void)hipSetDevice(c->device);
      for (;;) {
        std::pair<uint32_t, uint32_t> job;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return !todo.empty() || in_flight == 0 || rc_all != QF_OK; });
          if (rc_all != QF_OK || todo.empty()) return;
          job = todo.back();
          todo.pop_back();
          ++in_flight;
        }
        bool too_big = false;
        const int rc = count_chunk(c, S, cfg, use_null, sort_in != nullptr, job.first, job.second, slots, out, out_mu, &too_big);
        {
          std::lock_guard<std::mutex> lk(mu);
          --in_flight;
          if (rc != QF_OK && rc_all == QF_OK) {
            rc_all = rc;
            if (S != static_cast<Slot*>(c)) c->err = S->err;
          }
          if (too_big) {
            const uint32_t mid = job.first + (job.second - job.first) / 2;
            todo.push_back({mid, job.second});
            todo.push_back({job.first, mid});
          }
        }
        cv.notify_all();
      }
    };
    HIPCHK(c, hipStreamSynchronize(c->stream));   // prep, orders and the cleared accumulators are in place before either slot starts
    if (slots > 1) {
      if (!c->second_ready) {
        if (c->second.create()) return fail(c, QF_ERR_DEVICE, "cannot create the second stream");
        c->second_ready = true;
      }
      std::thread t(worker, &c->second);
      worker(c);
      t.join();
    } else {
      worker(c);
    }
    if (rc_all != QF_OK) return rc_all;
  }
  out->ms_total = out->ms_prep + (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  unsigned long long* d_sum_fx = c->d_counts.as<unsigned long long>() + counts_stride * kCountReplicas * 2;
  launch_sum_count_replicas(c->d_counts.as<unsigned long long>(), csize, counts_stride, d_sum_fx, (double*)(d_sum_fx + (size_t)csize * 2), c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_counts_fx.data(), d_sum_fx, (size_t)csize * 16, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const Scores& sc = c->scores;

  // QuaffParamCounts(const QuaffCounts&), src/qmodel.cpp:407-417 (linear, so it commutes with the weighted sum), in the exact
  // words; `counts` is their conversion
  const size_t ne = (size_t)(4 + 4 * sc.Km) * kNQual, Kg = sc.Kg;
  typedef unsigned __int128 u128;
  auto rd = [&](size_t k) -> u128 { return ((u128)c->h_counts_fx[2 * k + 1] << 64) | c->h_counts_fx[2 * k]; };
  c->h_pcounts_fx.assign((size_t)csize * 2, 0);
  auto wr = [&](size_t k, u128 v) { c->h_pcounts_fx[2 * k] = (uint64_t)v; c->h_pcounts_fx[2 * k + 1] = (uint64_t)(v >> 64); };
  for (size_t k = 0; k < ne; ++k) wr(k, rd(k));
  for (size_t g = 0; g < Kg; ++g) {
    const u128 m2m = rd(ne + g), m2i = rd(ne + Kg + g), m2d = rd(ne + 2 * Kg + g), m2e = rd(ne + 3 * Kg + g);
    wr(ne + g, m2m + m2d);            // beginInsertNo
    wr(ne + Kg + g, m2i + m2e);       // beginInsertYes
    wr(ne + 2 * Kg + g, m2m);         // beginDeleteNo
    wr(ne + 3 * Kg + g, m2d);         // beginDeleteYes
  }
  wr(ne + 4 * Kg + 0, rd(ne + 4 * Kg + 3));  // extendInsertNo  = i2m
  wr(ne + 4 * Kg + 1, rd(ne + 4 * Kg + 2));  // extendInsertYes = i2i
  wr(ne + 4 * Kg + 2, rd(ne + 4 * Kg + 1));  // extendDeleteNo  = d2m
  wr(ne + 4 * Kg + 3, rd(ne + 4 * Kg + 0));  // extendDeleteYes = d2d
  qf_exact_to_double(c->h_pcounts_fx.data(), csize, c->h_pcounts.data());
  double ll = 0;
  for (uint32_t r = 0; r < n_reads; ++r) ll += c->h_rll[r];  // serial read-order sum, qmodel.cpp:2420-2422
  {  // ... and the same sum exactly (order-free), for callers that combine several calls
    uint64_t acc[2] = {0, 0}, term[2];
    for (uint32_t r = 0; r < n_reads; ++r) {
      qf_exact_from_double(&c->h_rll[r], 1, term);
      qf_exact_add(acc, term, 1);
    }
    out->loglike_exact[0] = acc[0];
    out->loglike_exact[1] = acc[1];
  }
  out->counts_exact = c->h_pcounts_fx.data();
  uint64_t bcells = 0;
  for (uint32_t p = 0; p < n_pairs; ++p) if (c->h_weight[p] > 0) bcells += c->h_cells[p];
  out->forward = c->h_fwd.data();
  out->weight = c->h_weight.data();
  out->read_loglike = c->h_rll.data();
  out->sort_order = c->h_order.data();
  out->sort_count = c->h_order_n.data();
  out->loglike = ll;
  out->backward_cells = bcells;
  return QF_OK;
}

// ------------------------------------------------------------------------------ read-vs-read overlap
// Pairs [lo, hi) of the uploaded pair list.  Results go to the host arrays at the chunk's offsets; *too_big (nothing done)
// when the chunk's traceback exceeds the memory budget.
static int overlap_chunk(qf_ctx* c, const qf_dp_config* cfg, const bool need[2], uint32_t lo, uint32_t hi,
                         qf_overlap_result* out, bool* too_big);
static int overlap_chunk_impl(qf_ctx* c, const qf_dp_config* cfg, const bool need[2], uint32_t lo, uint32_t hi,
                         qf_overlap_result* out, bool* too_big) {
  *too_big = false;
  const uint32_t n_pairs = hi - lo;
  const bool sparse = cfg->sparse != 0;
  const bool mem = sparse && cfg->kmer_threshold < 0;
  const Scores& sc = c->scores;
  // Unit table: four bands per pair is the provision for arbitrary lists; a row block of the all-vs-all enumeration has barely
  // more bands than pairs (nearly all of them the forced single diagonal), and seed_pairs grows the table if a block needs more.
  uint32_t max_units = c->ov_per_pair ? n_pairs * 4 + 1024 : n_pairs + n_pairs / 4 + 65536;
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));

  // ---- seeding: x = pair_x's k-mer index, y = pair_y's k-mers (as stored)
  const int max_nd = (int)(2 * c->read_maxlen - 1);
  uint32_t n_row_items = 0;
  if (c->ov_use_rows) {
    // This block's runs, cut at the chunk boundaries of the y index.  Order: every x of one chunk before the next chunk, and
    // the list dealt out so that the workgroups of one chunk land on one XCD (workgroup ids go round the eight XCDs): the
    // chunk's index (its bucket starts and entries, ~150 KB) is then fetched into that XCD's L2 once for all the x that walk
    // it, instead of once per x from HBM (the index of a 100 k-sequence set is 900 MB).  The list depends only on the runs
    // and the block, so a later call with the same pair list reuses the device copy.
    const int cl = c->chunk_log2;
    const bool cached = c->row_items_epoch == c->prep_epoch && c->row_items_lo == lo && c->row_items_hi == hi && c->row_items_cl == cl && c->row_items_lds == c->row_lds && c->row_items_rows.size() == c->ov_rows.size() &&
                        !memcmp(c->row_items_rows.data(), c->ov_rows.data(), c->ov_rows.size() * sizeof(qf_ctx::PairRow));
    // The scheduler's triangle (whole rows x0, x0 + 1, ..., each x against x + 1 ... n_seqs - 1) with the LDS prefilter: no item
    // list at all -- a piece is (first row, rows, chunk) and k_seed_rows_lds forms its items itself.
    bool tri = !c->ov_rows.empty() && lo == 0 && !(c->debug & QF_DEBUG_HOST_ROW_ITEMS);
    if (tri) {
      uint64_t p = 0;
      for (size_t q = 0; q < c->ov_rows.size() && tri; ++q) {
        const auto& r = c->ov_rows[q];
        tri = r.x == c->ov_rows[0].x + q && r.y0 == r.x + 1 && r.n == c->n_reads - 1 - r.x && r.p0 == p;
        p += r.n;
      }
      tri = tri && p == hi;
    }
    if (tri && !c->row_lds) {
      // ... and with the plain prefilter the list is formed by a kernel (k_row_items_tri), not built and copied by the host
      const uint32_t X0 = c->ov_rows[0].x, R = (uint32_t)c->ov_rows.size();
      const uint32_t n = launch_row_items_tri(X0, R, c->n_reads, cl, nullptr, 0, c->stream);
      CHUNKRES(c, c->d_row_items, (size_t)n * sizeof(RowItem));
      c->row_items_n = launch_row_items_tri(X0, R, c->n_reads, cl, c->d_row_items.as<RowItem>(), n, c->stream);
      HIPCHK(c, hipGetLastError());
      c->row_items_epoch = 0;   // (d_row_items no longer holds a cached host-built list)
    } else if (tri && !cached) {
      constexpr uint32_t kPiece = 64;
      const uint32_t X0 = c->ov_rows[0].x, X1 = X0 + (uint32_t)c->ov_rows.size(), n_chunks = (c->n_reads + (1u << cl) - 1) >> cl;
      std::vector<uint4> pieces;
      for (uint32_t ch = (X0 + 1) >> cl; ch < n_chunks; ++ch) {
        const uint32_t xend = std::min<uint64_t>(X1, (((uint64_t)ch + 1) << cl) - 1);   // rows x with x + 1 < (ch + 1) << cl have pairs in the chunk
        for (uint32_t x = X0; x < xend; x += kPiece) pieces.push_back(make_uint4(x, std::min(kPiece, xend - x), ch, 0u));
      }
      c->row_pieces_n = (uint32_t)pieces.size();
      c->row_tri = true;
      c->row_tri_x0 = X0;
      c->row_items_n = 1;   // (nothing of the plain list is used)
      if (!pieces.empty()) {
        CHUNKRES(c, c->d_row_pieces, pieces.size() * sizeof(uint4));
        HIPCHK(c, hipMemcpyAsync(c->d_row_pieces.p, pieces.data(), pieces.size() * sizeof(uint4), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));   // stack-lifetime host buffer
      }
      c->row_items_rows = c->ov_rows;
      c->row_items_lo = lo; c->row_items_hi = hi; c->row_items_cl = cl; c->row_items_lds = c->row_lds; c->row_items_epoch = c->prep_epoch;
    } else if (!cached) {
      const uint32_t n_chunks = (c->n_reads + (1u << cl) - 1) >> cl;
      std::vector<uint32_t> first(n_chunks + 1, 0);
      auto clip = [&](const qf_ctx::PairRow& r, uint32_t& y, uint32_t& yend, uint32_t& p) {
        const uint64_t a0 = std::max<uint64_t>(r.p0, lo), a1 = std::min<uint64_t>((uint64_t)r.p0 + r.n, hi);
        if (a0 >= a1) return false;
        y = r.y0 + (uint32_t)(a0 - r.p0); yend = r.y0 + (uint32_t)(a1 - r.p0); p = (uint32_t)(a0 - lo);
        return true;
      };
      uint32_t y, yend, p;
      for (const auto& r : c->ov_rows)
        if (clip(r, y, yend, p)) for (uint32_t ch = y >> cl; ch <= (yend - 1) >> cl; ++ch) ++first[ch + 1];
      for (uint32_t ch = 0; ch < n_chunks; ++ch) first[ch + 1] += first[ch];
      std::vector<RowItem> sorted(first[n_chunks]);
      for (const auto& r : c->ov_rows)
        if (clip(r, y, yend, p))
          for (uint32_t yy = y; yy < yend;) {
            const uint32_t ch = yy >> cl, stop = std::min<uint32_t>(yend, (ch + 1) << cl);
            sorted[first[ch]++] = {r.x, ch, yy, stop, p + (yy - y)};
            yy = stop;
          }
      if (c->row_lds) {   // k_seed_rows_lds: the chunk-major list itself, with x's offset and length, in pieces of one chunk each
        constexpr uint32_t kPiece = 64;
        std::vector<RowItemL> sl(sorted.size());
        std::vector<uint4> pieces;
        for (size_t q = 0; q < sorted.size(); ++q) {
          const RowItem& r = sorted[q];
          const uint64_t xb = c->read_off[r.x];
          sl[q] = {r.x, r.ylo, r.yhi, r.pbase, (uint32_t)(c->read_off[r.x + 1] - xb), (uint32_t)xb, (uint32_t)(xb >> 32), r.chunk};
          if (pieces.empty() || sorted[pieces.back().x].chunk != r.chunk || pieces.back().y == kPiece) pieces.push_back(make_uint4((uint32_t)q, 0u, 0u, 0u));
          ++pieces.back().y;
        }
        c->row_pieces_n = (uint32_t)pieces.size();
        c->row_tri = false;
        if (!sl.empty()) {
          CHUNKRES(c, c->d_row_sorted, sl.size() * sizeof(RowItemL));
          CHUNKRES(c, c->d_row_pieces, pieces.size() * sizeof(uint4));
          HIPCHK(c, hipMemcpyAsync(c->d_row_sorted.p, sl.data(), sl.size() * sizeof(RowItemL), hipMemcpyHostToDevice, c->stream));
          HIPCHK(c, hipMemcpyAsync(c->d_row_pieces.p, pieces.data(), pieces.size() * sizeof(uint4), hipMemcpyHostToDevice, c->stream));
          HIPCHK(c, hipStreamSynchronize(c->stream));   // stack-lifetime host buffers
        }
      }
      // deal: segments of kSeg consecutive items go to XCD 0, 1, ... 7, 0, ...; workgroup b sits on XCD b % 8
      constexpr uint32_t kSeg = kRowSeg, kXcd = kRowXcd;
      const uint32_t n_items = (uint32_t)sorted.size(), n_seg = (n_items + kSeg - 1) / kSeg, seg_rounds = (n_seg + kXcd - 1) / kXcd;
      std::vector<RowItem> items((size_t)seg_rounds * kXcd * kSeg, RowItem{~0u, 0, 0, 0, 0});
      for (uint32_t k = 0; k < n_items; ++k) {
        const uint32_t seg = k / kSeg, pos = k % kSeg;
        items[((size_t)(seg / kXcd) * kSeg + pos) * kXcd + seg % kXcd] = sorted[k];
      }
      c->row_items_n = (uint32_t)items.size();
      if (!items.empty()) {
        CHUNKRES(c, c->d_row_items, items.size() * sizeof(RowItem));
        HIPCHK(c, hipMemcpyAsync(c->d_row_items.p, items.data(), items.size() * sizeof(RowItem), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));   // `items` is a stack-lifetime host buffer
      }
      c->row_items_rows = c->ov_rows;
      c->row_items_lo = lo; c->row_items_hi = hi; c->row_items_cl = cl; c->row_items_lds = c->row_lds; c->row_items_epoch = c->prep_epoch;
    }
    n_row_items = c->row_items_n;
    if (n_row_items) CHUNKRES(c, c->d_row_skip, n_pairs);
  }
  // Slotted single-diagonal list: x rows x0, x0 + 1, ... (the scheduler's order), at most one single-diagonal band per pair
  // (bands are at least 2 wide otherwise) and the staging kernel applies.
  uint32_t slot_rows = 0, slot_x0 = 0, slot_ychunks = 0;
  // (k_overlap_single_rows: compact rows of up to 512 entries, context-free gap scores; else k_overlap_single_lds while whole rows fit)
  const bool single_rows = c->ov_pitch && c->ov_cols_epoch == c->prep_epoch && sc.Kg == 1 && !(c->debug & QF_DEBUG_OLD_SINGLE_ROWS) &&
                           (!need[0] || c->ov_mmic_epoch[0] == c->prep_epoch) && (!need[1] || c->ov_mmic_epoch[1] == c->prep_epoch);
  if (c->ov_use_rows && cfg->band_size >= 2 && (single_rows || overlap_single_stages_rows(sc.Km)) && !c->ov_slot_collision &&
      !(c->debug & (QF_DEBUG_GLOBAL_OVERLAP_ROWS | QF_DEBUG_PAIR_ORDER_SINGLES))) {
    bool ok = true;
    uint32_t xa = 0, xb = 0, first = 1;
    for (const auto& r : c->ov_rows) {
      if ((uint64_t)r.p0 + r.n <= lo || r.p0 >= hi) continue;
      if (first) { xa = xb = r.x; first = 0; }
      else if (r.x == xb + 1) xb = r.x;
      else { ok = false; break; }
    }
    slot_ychunks = (c->n_reads + 255) / 256;
    if (ok && !first && (uint64_t)(xb - xa + 1) * slot_ychunks * 256 <= (1ull << 28)) {
      slot_rows = xb - xa + 1;
      slot_x0 = xa;
      CHUNKRES(c, c->d_slot_list, (size_t)slot_rows * slot_ychunks * 256 * 4);
    }
  }
  SeedArgs sa;
  BatchCounters bc;
  if (int rc = seed_pairs(c, c, cfg, n_pairs, mem, sparse ? max_nd : 2, [&](SeedArgs& s) {
        // (cleared here, not once before: a batch that outgrows its unit table is seeded again from scratch)
        if (slot_rows) {
          (void)hipMemsetAsync(c->d_slot_list.p, 0xFF, (size_t)slot_rows * slot_ychunks * 256 * 4, c->stream);
          s.slot_list = c->d_slot_list.as<uint32_t>();
          s.slot_x0 = slot_x0;
          s.slot_rows = slot_rows;
        }
        if (n_row_items) {
          (void)hipMemsetAsync(c->d_row_skip.p, 0, n_pairs, c->stream);
          s.row_items = c->d_row_items.as<RowItem>();
          s.n_row_items = n_row_items;
          s.chunk_start = c->d_cstart.as<uint32_t>();
          s.chunk_entries = c->d_centries.as<uint32_t>();
          s.chunk_log2 = c->chunk_log2;
          s.chunk_estride = c->chunk_estride;
          s.chunk_pb = c->chunk_pb;
          s.chunk_bounds = c->d_cbounds.as<uint2>();
          s.row_skip = c->d_row_skip.as<uint8_t>();
          if (c->row_lds && c->row_pieces_n) {
            s.row_sorted = c->row_tri ? nullptr : c->d_row_sorted.p;
            s.tri_x0 = c->row_tri_x0;
            s.row_pieces4 = c->d_row_pieces.as<uint4>();
            s.n_row_pieces = c->row_pieces_n;
            s.row_n_seqs = c->n_reads;
            s.row_max_entries = c->row_max_entries;
            s.row_e16 = c->row_e16 ? 1 : 0;
          }
        }
        s.pair_x = c->d_px.as<uint32_t>() + lo;
        s.pair_y = c->d_py.as<uint32_t>() + lo;
        s.ref_off = c->d_roff.as<uint64_t>();
        s.ref_bucket = c->d_rbucket.as<uint32_t>();
        s.ref_pos = c->d_rpos.as<uint32_t>();
        s.ref_skeys = (sparse && cfg->kmer_len > kMaxRefK) ? c->d_rskeys.as<unsigned long long>() : nullptr;
        s.storage_mode = 2;
        s.ov_use_32x3 = (c->debug & QF_DEBUG_OV32) != 0;   // measured slower (dense overlaps: fill 158 vs 141 ms): off unless asked for
        s.max_ref_len = s.max_read_len;   // the x side is a read too
        s.cls_key = c->d_cls_key.as<uint32_t>();   // (the single-diagonal list is sorted back into pair order)
      }, max_units, sa, bc))
    return rc;
  HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  if ((c->debug & QF_DEBUG_COUNT_SETTLED) && n_row_items) {   // tests: how many pairs the row prefilter settled
    std::vector<uint8_t> sk(n_pairs);
    HIPCHK(c, hipMemcpy(sk.data(), c->d_row_skip.p, n_pairs, hipMemcpyDeviceToHost));
    for (uint8_t v : sk) c->rows_settled += v;
  }
  if (bc.error & 4u) return fail(c, QF_ERR_SYMBOL, "Unknown symbol in read " + std::to_string(bc.error_detail));
  if ((bc.error & 16u) && slot_rows) {   // two bands claimed one slot (k_bin_units): this chunk again, single-diagonal bands as a plain list
    c->ov_slot_collision = true;
    return overlap_chunk(c, cfg, need, lo, hi, out, too_big);
  }
  if (bc.error & 2u)
    return fail(c, QF_ERR_UNSUPPORTED, "unsupported overlap band of " + std::to_string(bc.error_detail) + " diagonals");

  // ---- fill
  const uint64_t tb_bytes = (uint64_t)bc.tb_words * 4;
  if (tb_bytes > chunk_budget(c, c->d_tb, 1)) {
    if (n_pairs == 1) return fail(c, QF_ERR_MEMORY, "one pair needs " + std::to_string(tb_bytes >> 20) + " MiB of traceback, over the memory budget");
    *too_big = true;
    return QF_OK;
  }
  if (reserve_big(c->d_tb, tb_bytes + 64, {&c->d_fw, &c->second.d_fw, &c->second.d_tb}) != hipSuccess) {
    if (n_pairs == 1) return fail(c, QF_ERR_MEMORY, "cannot allocate " + std::to_string(tb_bytes >> 20) + " MiB of traceback for one pair");
    *too_big = true;
    return QF_OK;
  }
  if (int rc = sort_class_lists(c, c, bc, max_units, true, slot_rows != 0)) return rc;
  CHUNKRES(c, c->d_pair_result, (size_t)n_pairs * 8);
  CHUNKRES(c, c->d_pair_ij, (size_t)n_pairs * 8);
  CHUNKRES(c, c->d_recs, (size_t)n_pairs * sizeof(AlignRec));
  OvArgs oa{};
  oa.n_pairs = n_pairs;
  oa.units = c->d_units.as<Unit>();
  oa.pair_x = c->d_px.as<uint32_t>() + lo;
  oa.pair_y = c->d_py.as<uint32_t>() + lo;
  oa.pair_comp = c->d_pc.as<uint8_t>() + lo;
  oa.seq_off = c->d_roff.as<uint64_t>();
  oa.ctx = c->d_ctx.as<uint32_t>() + kCtxPad;
  oa.ctxc = c->d_ctxc.as<uint32_t>() + kCtxPad;
  oa.tb = c->d_tb.as<uint32_t>();
  oa.mmi[0] = c->d_mmi0.as<double>();
  oa.mmi[1] = c->d_mmi1.as<double>();
  oa.gap[0] = c->d_gap0.as<double>();
  oa.gap[1] = c->d_gap1.as<double>();
  if (!need[0]) { oa.mmi[0] = oa.mmi[1]; oa.gap[0] = oa.gap[1]; }
  if (!need[1]) { oa.mmi[1] = oa.mmi[0]; oa.gap[1] = oa.gap[0]; }
  if (slot_rows) {
    oa.slot_list = c->d_slot_list.as<uint32_t>(); oa.slot_rows = slot_rows; oa.slot_ychunks = slot_ychunks; oa.slot_x0 = slot_x0;
    if (single_rows) {
      oa.mmic_pitch = c->ov_pitch;
      oa.mmic_cpr = c->ov_cpr;
      oa.mmic[0] = c->d_mmic0.as<double>();
      oa.mmic[1] = c->d_mmic1.as<double>();
      if (!need[0]) oa.mmic[0] = oa.mmic[1];
      if (!need[1]) oa.mmic[1] = oa.mmic[0];
      oa.xrowoff = c->d_xrowoff.as<uint32_t>();
      oa.ycolT[0] = c->d_ycol0.as<uint4>();
      oa.ycolT[1] = c->d_ycol1.as<uint4>();
      oa.ygoff = c->d_ygoff.as<uint64_t>();
    }
  }
  oa.lse = c->d_lse.as<double>();
  if (c->lse_pack_bytes && !(c->debug & QF_DEBUG_GLOBAL_LSE)) { oa.lse_pack = c->d_lse_pack.as<uint8_t>(); oa.lse_pack_bytes = c->lse_pack_bytes; }
  oa.min_score = c->min_score;
  oa.per_pair = c->ov_per_pair ? 1 : 0;
  oa.no_lds_rows = (c->debug & QF_DEBUG_GLOBAL_OVERLAP_ROWS) != 0;
  oa.no_fast_steps = (c->debug & QF_DEBUG_OV_NO_FAST) != 0;
  oa.Km = sc.Km;
  oa.Kg = sc.Kg;
  oa.pair_head = c->d_pair_head.as<uint32_t>();
  oa.pair_ndiag = c->d_pair_ndiag.as<uint32_t>();
  oa.ins_sum = c->d_ins_sum.as<double>();
  oa.ins_sum_c = c->d_ins_sum_c.as<double>();
  oa.nll = c->d_nll.as<double>();
  oa.nll_c = c->d_nll_c.as<double>();
  oa.pair_result = c->d_pair_result.as<double>();
  oa.pair_score = c->d_pair_score.as<double>();
  oa.pair_end_unit = c->d_pair_end_unit.as<uint32_t>();
  oa.pair_end_ij = c->d_pair_ij.as<uint32_t>();
  oa.recs = c->d_recs.as<AlignRec>();
  oa.bc = c->d_bc.as<BatchCounters>();
  if (int rc = launch_classes_concurrently(c, bc, (c->debug & QF_DEBUG_SERIAL_CLASSES) != 0, [&](int cls, hipStream_t s) {
        if (cls > 10 && cls != kRowClass && cls != kOv32Class) return;
        OvArgs o2 = oa;
        o2.n_cls_units = bc.cls_count[cls];
        o2.cls_list = c->d_cls_list.as<uint32_t>() + (size_t)cls * max_units;
        launch_overlap_fill(cls, o2, s);
      }, 0, true))
    return rc;
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
  launch_overlap_finalize(oa, c->stream);
  HIPCHK(c, hipGetLastError());
  const BatchCounters seed_bc = bc;
  if (int rc = read_counters(c, bc)) return rc;
  const uint32_t n_recs = bc.n_align;
  CHUNKRES(c, c->d_runs_tmp, (size_t)bc.n_runs * 4 + 64);
  CHUNKRES(c, c->d_runs_out, (size_t)bc.n_runs * 4 + 64);
  oa.n_recs = n_recs;
  oa.runs_tmp = c->d_runs_tmp.as<uint32_t>();
  oa.runs_out = c->d_runs_out.as<uint32_t>();
  launch_overlap_traceback(oa, c->stream);
  HIPCHK(c, hipGetLastError());
  if (int rc = read_counters(c, bc)) return rc;
  const uint64_t total_runs = bc.total_runs_out;
  c->ov_tot.n_finite += bc.n_finite;
  c->ov_tot.sum_ndiag += bc.sum_ndiag;
  c->ov_tot.result_sum += bc.result_sum;
  HIPCHK(c, hipEventRecord(c->ev[4], c->stream));

  const size_t recs0 = c->h_recs.size(), runs0 = c->h_runs.size();
  c->h_recs.resize(recs0 + n_recs);
  c->h_runs.resize(runs0 + total_runs);
  if (c->ov_per_pair) {
    HIPCHK(c, hipMemcpyAsync(c->h_ov_result.data() + lo, c->d_pair_result.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_ov_score.data() + lo, c->d_pair_score.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_cells.data() + lo, c->d_pair_cells.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_ndiag.data() + lo, c->d_pair_ndiag.p, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, c->stream));
  }
  if (n_recs) HIPCHK(c, hipMemcpyAsync(c->h_recs.data() + recs0, c->d_recs.p, (size_t)n_recs * sizeof(AlignRec), hipMemcpyDeviceToHost, c->stream));
  if (total_runs) HIPCHK(c, hipMemcpyAsync(c->h_runs.data() + runs0, c->d_runs_out.p, (size_t)total_runs * 4, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (size_t a = recs0; a < recs0 + n_recs; ++a) {
    c->h_recs[a].read += lo;
    c->h_recs[a].run_off += runs0;
  }
  out->total_cells += seed_bc.total_cells;
  out->traceback_bytes += tb_bytes;
  float ms = 0;
  (void)hipEventElapsedTime(&ms, c->ev[1], c->ev[2]); out->ms_seed += ms;
  (void)hipEventElapsedTime(&ms, c->ev[2], c->ev[3]); out->ms_fill += ms;
  (void)hipEventElapsedTime(&ms, c->ev[3], c->ev[4]); out->ms_traceback += ms;
  (void)hipEventElapsedTime(&ms, c->ev[1], c->ev[5]); out->ms_total += ms;
  for (int cls = 0; cls < kNumClasses; ++cls) {
    if (!seed_bc.cls_count[cls] || (cls > 10 && cls != kRowClass && cls != kOv32Class)) continue;
    ms = 0; (void)hipEventElapsedTime(&ms, c->cls_ev[cls], c->cls_end[cls]); out->ms_fill_class[cls] += ms;
    out->cells_class[cls] += seed_bc.cls_cells[cls];
    out->units_class[cls] += seed_bc.cls_count[cls];
  }
  out->n_fill_classes = kNumClasses;
  return QF_OK;
}
static int overlap_chunk(qf_ctx* c, const qf_dp_config* cfg, const bool need[2], uint32_t lo, uint32_t hi,
                         qf_overlap_result* out, bool* too_big) {
  const int rc = overlap_chunk_impl(c, cfg, need, lo, hi, out, too_big);
  return split_instead(c, c, rc, hi - lo <= 1, too_big, {&c->d_fw, &c->second.d_fw, &c->second.d_tb});
}

// The resident sequences' derived arrays for the overlap path: tokens / context words / null log-likelihoods (prep_reads), the
// k-mer index over the sequences themselves (every one of them can be an x), the reverse-strand context words, insert-score
// sums and reverse-strand null log-likelihoods.  Kept across the blocks of a pair list (qf_ctx::ov_prep_epoch).
static int prep_overlap_reads(qf_ctx* c, const qf_dp_config* cfg, int prep_k) {
  const bool sparse = cfg->sparse != 0;
  const uint32_t n_seqs = c->n_reads;
  const Scores& sc = c->scores;
  if (int rc = prep_reads(c, prep_k)) return rc;
  if (sparse && c->read_index_k != cfg->kmer_len && cfg->kmer_len > kMaxRefK) {
    if (int rc = build_sorted_index(c, c->d_tok.as<uint8_t>(), c->d_roff.as<uint64_t>(), c->read_off, c->read_maxlen, cfg->kmer_len,
                                    c->d_roff32, c->d_rskeys, c->d_rpos))
      return rc;
    c->read_index_k = cfg->kmer_len;
  }
  if (sparse && c->read_index_k != cfg->kmer_len) {  // needs the token bytes prep_reads just wrote
    launch_ref_index(c->d_tok.as<uint8_t>(), c->d_roff.as<uint64_t>(), n_seqs, c->read_maxlen, (uint32_t)cfg->kmer_len,
                     1u << (2 * cfg->kmer_len), c->d_rbucket.as<uint32_t>(), c->d_rcursor.as<uint32_t>(),
                     c->d_rpos.as<uint32_t>(), c->stream);
    HIPCHK(c, hipGetLastError());
    c->read_index_k = cfg->kmer_len;
  }
  HIPCHK(c, c->d_ctxc.reserve((c->read_total + 2 * kCtxPad) * 4));
  HIPCHK(c, c->d_ins_sum.reserve((size_t)n_seqs * 8));
  HIPCHK(c, c->d_ins_sum_c.reserve((size_t)n_seqs * 8));
  HIPCHK(c, c->d_nll_c.reserve((size_t)n_seqs * 8));
  HIPCHK(c, hipMemsetAsync(c->d_ctxc.p, 0, (c->read_total + 2 * kCtxPad) * 4, c->stream));
  {
    PrepArgs pa{};
    pa.seq = c->d_seq.as<char>();
    pa.qual = c->reads_have_qual ? c->d_qual.as<char>() : nullptr;
    pa.off = c->d_roff.as<uint64_t>();
    pa.match_len = sc.match_len;
    pa.gap_len = sc.gap_len;
    pa.tok = c->d_tok.as<uint8_t>();
    pa.ctx = c->d_ctx.as<uint32_t>() + kCtxPad;
    pa.ctxc = c->d_ctxc.as<uint32_t>() + kCtxPad;
    pa.eins = c->d_eins.as<double>();
    pa.ins_sum = c->d_ins_sum.as<double>();
    pa.ins_sum_c = c->d_ins_sum_c.as<double>();
    pa.nll_c = c->d_nll_c.as<double>();
    pa.has_null = c->have_null;
    if (c->have_null) {
      std::vector<double> lq(4 * kNQual);
      c->null.tables(pa.null_logEmit, pa.null_log1mEmit, pa.null_logSym, lq.data());
      pa.null_logQual = c->d_nullq.as<double>();
    }
    launch_prep_overlap(pa, n_seqs, c->stream);
    HIPCHK(c, hipGetLastError());
  }
  // k_overlap_single_rows: row / column offsets into the compact pair-emission table (the quality values the reads use)
  const uint32_t qmin = c->reads_have_qual ? c->read_qmin : (uint32_t)kNQualDev, nq = c->reads_have_qual ? c->read_qmax - c->read_qmin + 1 : 1u;
  c->ov_pitch = overlap_compact_pitch(sc.Km, nq);
  c->ov_cpr = sc.Km * nq / 2;
  c->ov_cols_epoch = 0;
  if (c->ov_pitch && n_seqs) {
    const uint32_t groups = (n_seqs + 63) / 64;
    std::vector<uint64_t> goff(groups + 1, 0);
    uint32_t max_blocks = 1;
    for (uint32_t g = 0; g < groups; ++g) {
      uint64_t longest = 1;
      for (uint32_t y = g * 64; y < std::min(n_seqs, g * 64 + 64); ++y) longest = std::max(longest, c->read_off[y + 1] - c->read_off[y]);
      const uint32_t blocks = (uint32_t)((longest + 7) / 8);
      max_blocks = std::max(max_blocks, blocks);
      goff[g + 1] = goff[g] + (uint64_t)blocks * 64;
    }
    HIPCHK(c, c->d_ygoff.reserve((size_t)(groups + 1) * 8));
    HIPCHK(c, c->d_xrowoff.reserve((c->read_total + 16) * 4));
    HIPCHK(c, c->d_ycol0.reserve((size_t)goff[groups] * 16));
    HIPCHK(c, c->d_ycol1.reserve((size_t)goff[groups] * 16));
    HIPCHK(c, hipMemcpyAsync(c->d_ygoff.p, goff.data(), (size_t)(groups + 1) * 8, hipMemcpyHostToDevice, c->stream));
    launch_overlap_cols(c->d_ctx.as<uint32_t>() + kCtxPad, c->d_ctxc.as<uint32_t>() + kCtxPad, c->d_roff.as<uint64_t>(), n_seqs, c->read_total,
                        c->d_ygoff.as<uint64_t>(), max_blocks, sc.Km, qmin, c->ov_pitch, c->d_xrowoff.as<uint32_t>(), c->d_ycol0.as<uint4>(),
                        c->d_ycol1.as<uint4>(), c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));   // `goff` is a stack-lifetime host buffer
    c->ov_cols_epoch = c->prep_epoch;
  }
  return QF_OK;
}

// Everything from the pair list to the host-side records.  `put_pairs` queues what leaves the list on the device (d_px / d_py /
// d_pc: copies of the caller's arrays, or the enumeration kernel); c->ov_rows describes its runs (x, y0, y0 + 1, ...), or is
// empty for an arbitrary list; c->ov_per_pair says whether the per-pair result arrays go back to the host.  Leaves the
// alignment records in c->h_recs (AlignRec::read = index into the list) and their runs in c->h_runs.
static int overlap_run(qf_ctx* c, const qf_dp_config* cfg, const bool need[2], uint32_t n_pairs, qf_overlap_result* out,
                       const std::function<int()>& put_pairs) {
  const CallInProgress in_progress(c->device);
  const uint32_t n_seqs = c->n_reads;
  const bool sparse = cfg->sparse != 0;
  if (int rc = ensure_lse(c)) return rc;
  const Scores& sc = c->scores;
  // pair-emission tables: once per (parameters, strand flag), not once per pair (src/qoverlap.cpp:79)
  for (int v = 0; v < 2; ++v) {
    if (!need[v] || c->ov_scores[v]) continue;
    OverlapScores os;
    os.build(c->params, sc, v == 1);
    DevBuf& dm = v ? c->d_mmi1 : c->d_mmi0;
    DevBuf& dg = v ? c->d_gap1 : c->d_gap0;
    HIPCHK(c, dm.reserve(os.mmi.size() * 8));
    HIPCHK(c, dg.reserve(os.gap.size() * 8));
    HIPCHK(c, hipMemcpy(dm.p, os.mmi.data(), os.mmi.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(dg.p, os.gap.data(), os.gap.size() * 8, hipMemcpyHostToDevice));
    c->ov_scores[v] = true;
  }
  // k-mer index over the resident sequences themselves (every sequence can be an x)
  if (sparse && c->read_index_k != cfg->kmer_len && cfg->kmer_len <= kMaxRefK) {
    const int k = cfg->kmer_len;
    const uint32_t nb = 1u << (2 * k);
    const size_t bytes = (size_t)n_seqs * (nb + 1) * 4;
    HIPCHK(c, c->d_rbucket.reserve(bytes));
    HIPCHK(c, c->d_rcursor.reserve(bytes));
    HIPCHK(c, c->d_rpos.reserve((c->read_total + 16) * 4));
    HIPCHK(c, hipMemsetAsync(c->d_rbucket.p, 0, bytes, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_rcursor.p, 0, bytes, c->stream));
  }
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  HIPCHK(c, c->d_bc.reserve(sizeof(BatchCounters)));
  HIPCHK(c, hipMemsetAsync(c->d_bc.p, 0, sizeof(BatchCounters), c->stream));
  const int prep_k = sparse ? cfg->kmer_len : 0;
  const bool prepped = c->ov_prep_epoch == c->prep_epoch && c->ov_prep_k == prep_k;   // a later block of the same pair list
  if (!prepped)
    if (int rc = prep_overlap_reads(c, cfg, prep_k)) return rc;
  if (c->ov_pitch && c->ov_cols_epoch == c->prep_epoch) {   // compact pair-emission tables for the quality values in use
    const uint32_t qmin = c->reads_have_qual ? c->read_qmin : (uint32_t)kNQualDev, nq = c->reads_have_qual ? c->read_qmax - c->read_qmin + 1 : 1u;
    for (int v = 0; v < 2; ++v) {
      if (!need[v] || c->ov_mmic_epoch[v] == c->prep_epoch) continue;
      DevBuf& dc = v ? c->d_mmic1 : c->d_mmic0;
      HIPCHK(c, dc.reserve((size_t)sc.Km * nq * c->ov_pitch * 8 + 4096));   // (+ slack: a staged row is fetched in whole 16-byte chunks)
      launch_mmi_compact((v ? c->d_mmi1 : c->d_mmi0).as<double>(), sc.Km, qmin, nq, c->ov_pitch, dc.as<double>(), c->stream);
      HIPCHK(c, hipGetLastError());
      c->ov_mmic_epoch[v] = c->prep_epoch;
    }
  }
  HIPCHK(c, c->d_px.reserve((size_t)n_pairs * 4));
  HIPCHK(c, c->d_py.reserve((size_t)n_pairs * 4));
  HIPCHK(c, c->d_pc.reserve((size_t)n_pairs));
  if (int rc = put_pairs()) return rc;
  HIPCHK(c, hipEventRecord(c->ev[1], c->stream));

  if (!prepped) {
    BatchCounters pb;
    if (int rc = read_counters(c, pb)) return rc;
    if (pb.error & 4u) return fail(c, QF_ERR_SYMBOL, "Unknown symbol in read " + std::to_string(pb.error_detail));
  }
  c->ov_prep_epoch = c->prep_epoch;   // (only once the symbols have been checked)
  c->ov_prep_k = prep_k;
  // Row prefilter: worth it when the list is the scheduler's (x-major, runs of consecutive y: src/qoverlap.cpp:528-547)
  c->ov_use_rows = false;
  if (sparse && cfg->kmer_threshold >= 0 && cfg->kmer_len <= kMaxRefK && !(c->debug & QF_DEBUG_NO_ROW_PREFILTER) &&
      !c->ov_rows.empty() && (uint64_t)c->ov_rows.size() * 32 <= n_pairs) {
    SeedArgs t;
    fill_seed_args(c, *c, cfg, t, 0, (int)(2 * c->read_maxlen - 1));
    t.max_ref_len = t.max_read_len;
    t.ref_skeys = nullptr;
    const size_t stride = seed_row_stride_bytes(t);
    int cl = 6;
    // First choice: chunks small enough that counters AND the chunk's index fit one CU's LDS (k_seed_rows_lds); entries take 16
    // bits when every position fits beside the sequence-in-chunk bits.
    c->row_lds = false;
    if (stride && (c->debug & QF_DEBUG_LDS_ROW_INDEX)) {   // (measured: 16.3 ms against k_seed_rows' 13.9 per row block, and its 156 KB of LDS shut every other kernel out of the CU: off unless asked for)
      for (int tcl = 6; tcl >= 2 && !c->row_lds; --tcl) {
        uint64_t most = 0;
        for (uint64_t y = 0; y < n_seqs; y += 1ull << tcl) most = std::max(most, c->read_off[std::min<uint64_t>(n_seqs, y + (1ull << tcl))] - c->read_off[y]);
        bool e16 = false;
        const size_t need = seed_rows_lds_fit(t, tcl, most, &e16);
        if (need && need <= kSeedRowLdsBig) {
          c->row_lds = true; c->row_e16 = e16; c->row_max_entries = most; cl = tcl;
        }
      }
    }
    // Second: 16-bit index entries, sequence << pb | position with 2^pb > the longest sequence -- the prefilter is bound by its
    // gathers and these halve them -- when at least 8 sequences' counters (2^(pb + 1) diagonals each) fit the workgroup's LDS.
    int pb = 0;
    if (stride && !c->row_lds && !(c->debug & QF_DEBUG_ROW_INDEX_32)) {
      int bits = 1;
      while ((1ull << bits) < c->read_maxlen) ++bits;
      int tcl = std::min(6, 16 - bits);
      while (tcl >= 3 && (seed_row_stride_bytes_e16(bits, seed_row_bits_of(t)) << tcl) > kSeedRowLdsMax) --tcl;
      if (tcl >= 3) { pb = bits; cl = tcl; }
    }
    if (!c->row_lds && !pb)
      while (stride && cl > 3 && (stride << cl) > kSeedRowLdsMax) --cl;   // two workgroups' counters per CU
    if (stride && (c->row_lds || pb || (stride << cl) <= kSeedRowLdsMax)) {
      const uint64_t estride = c->row_lds ? ((c->row_max_entries + (1ull << (2 * cfg->kmer_len)) + 16 + 15) & ~15ull) : 0;   // padded index: entries per chunk
      if (c->chunk_epoch != c->prep_epoch || c->chunk_k != cfg->kmer_len || c->chunk_log2 != cl || c->chunk_estride != estride || c->chunk_span != seed_row_entry_span(t) || c->chunk_pb != pb) {
        const uint32_t nb = 1u << (2 * cfg->kmer_len), n_chunks = (n_seqs + (1u << cl) - 1) >> cl;
        const size_t bytes = (size_t)n_chunks * (nb + 1) * 4;
        // (16-bit entries: a chunk's buckets are padded to even length, at most one pad entry per bucket: chunk_base16)
        const size_t ebytes = estride ? (size_t)n_chunks * estride * 4 : std::max<size_t>((c->read_total + 16) * 4, pb ? (c->read_total + (size_t)n_chunks * nb + 64) * 2 : 0);
        HIPCHK(c, c->d_cstart.reserve(bytes));
        HIPCHK(c, c->d_ccursor.reserve(bytes));
        HIPCHK(c, c->d_centries.reserve(ebytes));
        if (pb) HIPCHK(c, c->d_cbounds.reserve((size_t)n_chunks * nb * sizeof(uint2)));
        HIPCHK(c, hipMemsetAsync(c->d_cstart.p, 0, bytes, c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_ccursor.p, 0, bytes, c->stream));
        if (estride) HIPCHK(c, hipMemsetAsync(c->d_centries.p, 0xFF, ebytes, c->stream));   // pad entries
        launch_chunk_index(c->d_tok.as<uint8_t>(), c->d_roff.as<uint64_t>(), n_seqs, c->read_maxlen, (uint32_t)cfg->kmer_len, nb, cl,
                           c->d_cstart.as<uint32_t>(), c->d_ccursor.as<uint32_t>(), c->d_centries.as<uint32_t>(), estride, seed_row_entry_span(t), pb, c->d_cbounds.as<uint2>(), c->stream);
        HIPCHK(c, hipGetLastError());
        c->chunk_epoch = c->prep_epoch;
        c->chunk_k = cfg->kmer_len;
        c->chunk_log2 = cl;
        c->chunk_estride = estride;
        c->chunk_span = seed_row_entry_span(t);
        c->chunk_pb = pb;
      }
      c->ov_use_rows = true;
    }
  }
  float ms_prep = 0;
  HIPCHK(c, hipEventSynchronize(c->ev[1]));
  (void)hipEventElapsedTime(&ms_prep, c->ev[0], c->ev[1]);
  out->ms_prep += ms_prep;
  out->ms_total += ms_prep;
  if (c->ov_per_pair) {
    c->h_ov_result.resize(n_pairs);
    c->h_ov_score.resize(n_pairs);
    c->h_cells.resize(n_pairs);
    c->h_ndiag.resize(n_pairs);
  }
  c->h_recs.clear();
  c->h_runs.clear();
  c->ov_slot_collision = false;
  std::vector<std::pair<uint32_t, uint32_t>> todo{{0u, n_pairs}};
  while (!todo.empty()) {
    const auto [lo, hi] = todo.back();
    todo.pop_back();
    bool too_big = false;
    if (int rc = overlap_chunk(c, cfg, need, lo, hi, out, &too_big)) return rc;
    if (too_big) {
      const uint32_t mid = lo + (hi - lo) / 2;
      todo.push_back({mid, hi});
      todo.push_back({lo, mid});
    }
  }
  for (size_t a = 0; a < c->h_recs.size(); ++a)
    if (!c->h_recs[a].ok)
      return fail(c, QF_ERR_DEVICE, "overlap traceback did not reach the start state (pair " + std::to_string(c->h_recs[a].read) + ")");
  return QF_OK;
}

static int check_overlap_call(qf_ctx* c, const qf_dp_config* cfg, const void* out) {
  if (!c) return QF_ERR_ARG;
  if (!cfg || !out) return fail(c, QF_ERR_ARG, "null argument");
  if (cfg->reserved) return fail(c, QF_ERR_ARG, "qf_dp_config.reserved must be 0");
  if (!c->have_params) return fail(c, QF_ERR_STATE, "no parameters set (qf_set_params_json)");
  if (cfg->band_size < 0) return fail(c, QF_ERR_ARG, "negative band size");
  if (cfg->sparse && (cfg->kmer_len < 1 || cfg->kmer_len > 32)) return fail(c, QF_ERR_ARG, "kmer_len out of range");
  return QF_OK;
}

int qf_overlap_resident(qf_ctx* c, const qf_dp_config* cfg, const uint32_t* pair_x, const uint32_t* pair_y,
                        const uint8_t* y_comp, uint32_t n_pairs, qf_overlap_result* out) {
  if (int rc = check_overlap_call(c, cfg, out)) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  memset(out, 0, sizeof *out);
  out->n_pairs = n_pairs;
  if (!n_pairs) return QF_OK;
  if (!pair_x || !pair_y || !y_comp) return fail(c, QF_ERR_ARG, "null pair list");
  if (n_pairs > kMaxPairsPerCall) return fail(c, QF_ERR_UNSUPPORTED, "more than 2^28 pairs in one call");
  const uint32_t n_seqs = c->n_reads;
  bool need[2] = {false, false};
  c->ov_rows.clear();
  c->rows_settled = 0;
  c->ov_per_pair = true;
  c->ov_tot = qf_ctx::OvTotals();
  bool runs = true;   // still looks like the scheduler's order: runs (x, y0), (x, y0 + 1), ... of 32+ pairs on average
  for (uint32_t p = 0; p < n_pairs; ++p) {
    if (pair_x[p] >= n_seqs || pair_y[p] >= n_seqs) return fail(c, QF_ERR_ARG, "pair index out of range");
    need[y_comp[p] ? 1 : 0] = true;
    if (!runs) continue;
    if (!c->ov_rows.empty() && c->ov_rows.back().x == pair_x[p] && c->ov_rows.back().y0 + c->ov_rows.back().n == pair_y[p]) ++c->ov_rows.back().n;
    else {
      c->ov_rows.push_back({pair_x[p], pair_y[p], 1u, p});
      if (c->ov_rows.size() > 4096 && (uint64_t)c->ov_rows.size() * 32 > p) { runs = false; c->ov_rows.clear(); }   // (an arbitrary list: do not keep a row per pair)
    }
  }
  if (int rc = overlap_run(c, cfg, need, n_pairs, out, [&]() -> int {
        HIPCHK(c, hipMemcpyAsync(c->d_px.p, pair_x, (size_t)n_pairs * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_py.p, pair_y, (size_t)n_pairs * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_pc.p, y_comp, (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
        return QF_OK;
      }))
    return rc;
  const uint32_t n_recs = (uint32_t)c->h_recs.size();
  // records arrive in the device's completion order; a pair has at most one, so pair order is one scatter + one sweep
  c->h_ov_slot.assign(n_pairs, ~0u);
  for (uint32_t a = 0; a < n_recs; ++a) c->h_ov_slot[c->h_recs[a].read] = a;
  c->h_ov_align.resize(n_recs);
  uint32_t n_out = 0;
  for (uint32_t p = 0; p < n_pairs; ++p) {
    if (c->h_ov_slot[p] == ~0u) continue;
    const AlignRec& r = c->h_recs[c->h_ov_slot[p]];
    qf_overlap_alignment& o = c->h_ov_align[n_out++];
    o.pair = r.read;
    o.viterbi = r.viterbi;
    o.score = r.score;
    o.x_start = r.x_start; o.x_end = r.x_end; o.y_start = r.y_start; o.y_end = r.y_end;
    o.n_columns = r.n_columns;
    o.n_runs = r.n_runs;
    o.run_offset = r.run_off;
  }
  if (n_out != n_recs) return fail(c, QF_ERR_DEVICE, "overlap records do not map one-to-one onto pairs");
  out->viterbi = c->h_ov_result.data();
  out->score = c->h_ov_score.data();
  out->cells = c->h_cells.data();
  out->n_diagonals = c->h_ndiag.data();
  out->n_alignments = n_recs;
  out->alignments = c->h_ov_align.data();
  out->state_runs = c->h_runs.data();
  return QF_OK;
}

uint64_t qf_overlap_rows_pairs(uint32_t n_seqs, uint32_t x0, uint32_t x1) {
  if (x1 <= x0 || !n_seqs) return 0;
  if (x1 > n_seqs - 1) x1 = n_seqs - 1;
  if (x1 <= x0) return 0;
  const uint64_t r = x1 - x0;
  return r * (uint64_t)(n_seqs - 1 - x0) - r * (r - 1) / 2;
}

int qf_debug_measure_f64_rate(qf_ctx* c, double* lane_ops_per_s) {
  if (!c || !lane_ops_per_s) return QF_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  DevBuf tmp;
  HIPCHK(c, tmp.reserve((size_t)1024 * 256 * 8));
  *lane_ops_per_s = measure_f64_add_rate(tmp.as<double>(), c->stream);
  tmp.release();
  return *lane_ops_per_s > 0 ? QF_OK : fail(c, QF_ERR_DEVICE, "fp64 issue-rate measurement failed");
}

int qf_debug_set_overlap_block_pairs(qf_ctx* c, uint64_t pairs) {
  if (!c) return QF_ERR_ARG;
  c->ov_block_pairs = pairs;
  return QF_OK;
}

// Rows [x0, x1) of QuaffOverlapScheduler's enumeration (src/qoverlap.cpp:475-480,528-547), in row blocks of about 2^24 pairs
// (a block's per-pair tables and unit lists take ~400 bytes a pair; with a few million bands per class the latency-bound
// classes of a block -- a hundred wide bands per 34 rows -- run beside enough single-diagonal work to vanish).
int qf_overlap_rows(qf_ctx* c, const qf_dp_config* cfg, uint32_t n_originals, uint32_t x0, uint32_t x1,
                    qf_overlap_rows_result* out) {
  if (int rc = check_overlap_call(c, cfg, out)) return rc;
  HIPCHK(c, hipSetDevice(c->device));
  memset(out, 0, sizeof *out);
  out->x0 = x0;
  out->x1 = x1;
  const uint32_t n_seqs = c->n_reads;
  if (!n_originals || (n_seqs != n_originals && n_seqs != 2 * (uint64_t)n_originals))
    return fail(c, QF_ERR_ARG, "the resident set must be n_originals reads, optionally followed by their reverse complements");
  if (x0 > x1 || x1 > n_originals - 1) return fail(c, QF_ERR_ARG, "rows out of range: 0 <= x0 <= x1 <= n_originals - 1");
  c->rows_settled = 0;
  c->ov_per_pair = false;
  c->ov_tot = qf_ctx::OvTotals();
  c->h_hits.clear();
  c->h_hit_runs.clear();
  // Blocks of about 2^24 pairs, of EQUAL size: a call's rows cut at exactly 2^24 leave a small last block, and a block costs
  // ~10 ms however few pairs it has (one banded unit's 2000 dependent steps, the traceback's, the host's turnarounds).
  uint64_t want = c->ov_block_pairs ? c->ov_block_pairs : (1ull << 24);
  if (!c->ov_block_pairs) {
    uint64_t total = 0;
    for (uint32_t x = x0; x < x1; ++x) total += n_seqs - 1 - x;
    // (a block may pass 2^24 pairs by a sixteenth if that saves one: 4.2 x 2^24 pairs are four blocks, not five; never by more --
    // a block's traceback has to fit the memory budget, and a block cut in two for that loses the triangle's fast paths)
    uint64_t nb = total / want;
    if (!nb || (total + nb - 1) / nb > want + want / 16) nb = std::max<uint64_t>(1, (total + want - 1) / want);   // (an empty range of rows: one empty block)
    want = std::min<uint64_t>((total + nb - 1) / nb + n_seqs, want + want / 16);   // (+ a row: the cut falls on a row boundary)
  }
  std::vector<uint64_t> row_start;
  for (uint32_t b0 = x0; b0 < x1;) {
    // rows of this block: as many as stay under the pair target (at least one; a row has fewer than 2^28 pairs by the
    // library's own limit on resident sequences), at most what one launch of the enumeration kernel takes
    uint32_t b1 = b0;
    uint64_t np = 0;
    row_start.clear();
    while (b1 < x1 && b1 - b0 < 32768u) {
      const uint64_t len = n_seqs - 1 - b1;
      if (b1 > b0 && np + len > want) break;
      if (np + len > kMaxPairsPerCall) break;
      row_start.push_back(np);
      np += len;
      ++b1;
    }
    if (b1 == b0) return fail(c, QF_ERR_UNSUPPORTED, "a row of more than 2^28 pairs");
    const uint32_t n_pairs = (uint32_t)np;
    bool need[2] = {b0 + 1 < n_originals, n_seqs > n_originals};   // (a row's ny run from nx + 1 to n_seqs - 1)
    c->ov_rows.clear();
    for (uint32_t x = b0; x < b1; ++x)
      if (n_seqs - 1 - x) c->ov_rows.push_back({x, x + 1, n_seqs - 1 - x, (uint32_t)row_start[x - b0]});
    qf_overlap_result blk;
    memset(&blk, 0, sizeof blk);
    if (n_pairs) {
      if (int rc = overlap_run(c, cfg, need, n_pairs, &blk, [&]() -> int {
            launch_overlap_row_pairs(b0, b1 - b0, n_seqs, n_originals, c->d_px.as<uint32_t>(), c->d_py.as<uint32_t>(), c->d_pc.as<uint8_t>(),
                                     c->stream);
            HIPCHK(c, hipGetLastError());
            return QF_OK;
          }))
        return rc;
    }
    // this block's records -> hits in (x, y) order
    const size_t n_recs = c->h_recs.size();
    std::vector<uint32_t> idx(n_recs);
    for (size_t a = 0; a < n_recs; ++a) idx[a] = (uint32_t)a;
    std::sort(idx.begin(), idx.end(), [&](uint32_t p, uint32_t q) { return c->h_recs[p].read < c->h_recs[q].read; });
    for (size_t k = 0; k < n_recs; ++k) {
      const AlignRec& r = c->h_recs[idx[k]];
      const size_t row = std::upper_bound(row_start.begin(), row_start.end(), (uint64_t)r.read) - row_start.begin() - 1;
      qf_overlap_hit h;
      h.x = b0 + (uint32_t)row;
      h.y = h.x + 1 + (uint32_t)(r.read - row_start[row]);
      h.viterbi = r.viterbi;
      h.score = r.score;
      h.x_start = r.x_start; h.x_end = r.x_end; h.y_start = r.y_start; h.y_end = r.y_end;
      h.n_columns = r.n_columns;
      h.n_runs = r.n_runs;
      h.run_offset = c->h_hit_runs.size();
      c->h_hit_runs.insert(c->h_hit_runs.end(), c->h_runs.data() + r.run_off, c->h_runs.data() + r.run_off + r.n_runs);
      c->h_hits.push_back(h);
    }
    out->n_pairs += n_pairs;
    out->total_cells += blk.total_cells;
    out->traceback_bytes += blk.traceback_bytes;
    out->ms_prep += blk.ms_prep; out->ms_seed += blk.ms_seed; out->ms_fill += blk.ms_fill;
    out->ms_traceback += blk.ms_traceback; out->ms_total += blk.ms_total;
    for (int k = 0; k < kNumClasses; ++k) {
      out->ms_fill_class[k] += blk.ms_fill_class[k];
      out->cells_class[k] += blk.cells_class[k];
      out->units_class[k] += blk.units_class[k];
    }
    ++out->n_blocks;
    b0 = b1;
  }
  if (c->h_hits.size() > 0xFFFFFFFFull) return fail(c, QF_ERR_UNSUPPORTED, "more than 2^32 alignments in one call");
  out->n_fill_classes = kNumClasses;
  out->n_finite = c->ov_tot.n_finite;
  out->total_diagonals = c->ov_tot.sum_ndiag;
  out->result_checksum = c->ov_tot.result_sum;
  out->n_hits = (uint32_t)c->h_hits.size();
  out->hits = c->h_hits.data();
  out->state_runs = c->h_hit_runs.data();
  return QF_OK;
}

// ------------------------------------------------------------------------------ E-step reduction (RCCL)
// QuaffCountingScheduler::finalCounts / finalLogLike (src/qmodel.cpp:2416-2422) sum the per-read counts and log-likelihoods
// of all worker threads; with reads sharded over GPUs that sum is one all-reduce(sum, fp64) of the flattened counts and the
// log-likelihood per EM iteration.  RCCL is bound at run time (dlopen): a process that already carries an RCCL (PyTorch's)
// shares it, and single-GPU users never load it.
namespace {
struct Rccl {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;     // optional: an error a rank hit inside a collective
  decltype(&ncclCommAbort) CommAbort = nullptr;                     // optional
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err, path;
  bool load() {
    if (h) return true;
    // An RCCL already in the process (PyTorch ships its own librccl.so and loads it with local scope) is reused whatever its
    // path: a second copy would have its own topology state and its own idea of the devices in use.  RTLD_NOLOAD finds a
    // resident library by soname without loading anything.
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
      if (h) { path = std::string(name) + " (already resident)"; break; }
    }
    if (!h)
      for (const char* name : {"librccl.so.1", "librccl.so"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) { path = name; break; }
      }
    if (!h) { err = std::string("cannot load RCCL: ") + dlerror(); return false; }
    auto sym = [&](const char* n, bool required) { void* p = dlsym(h, n); if (!p && required) err = std::string("RCCL lacks ") + n; return p; };
    GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId", true);
    CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank", true);
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll", true);
    AllReduce = (decltype(AllReduce))sym("ncclAllReduce", true);
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy", true);
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString", true);
    CommGetAsyncError = (decltype(CommGetAsyncError))sym("ncclCommGetAsyncError", false);
    CommAbort = (decltype(CommAbort))sym("ncclCommAbort", false);
    if (!GetUniqueId || !CommInitRank || !CommInitAll || !AllReduce || !CommDestroy || !GetErrorString) { h = nullptr; return false; }
    return true;
  }
};
Rccl g_rccl;
std::mutex g_rccl_mu;

// Seconds a rank waits for its peers (communicator set-up, a collective) before it gives up with an error status instead of
// hanging: a peer that died or never started must not leave the others blocked for ever.  QUAFF_HIP_COMM_TIMEOUT overrides.
double comm_timeout_s() {
  if (const char* e = getenv("QUAFF_HIP_COMM_TIMEOUT")) { const double v = atof(e); if (v > 0) return v; }
  return 180.0;
}

bool comm_trace() { static const bool on = getenv("QUAFF_HIP_COMM_TRACE") != nullptr; return on; }
#define COMM_TRACE(...) do { if (comm_trace()) { fprintf(stderr, "[quaffhip comm] " __VA_ARGS__); fputc('\n', stderr); } } while (0)

// Tear a communicator down without waiting for its peers.  ncclCommAbort itself can block -- measured: on a communicator whose
// set-up is still in its bootstrap, waiting for a rank that never arrives, it waits for that set-up -- so it runs on a thread
// of its own, which is given two seconds and then left behind (the caller reports an error and normally exits).
void comm_abort(qf_ctx* c) {
  if (!c->comm) return;
  ncclComm_t comm = c->comm;
  c->comm = nullptr;
  c->comm_rank = 0;
  c->comm_size = 1;
  auto done = std::make_shared<std::promise<void>>();
  std::future<void> fut = done->get_future();
  std::thread([comm, done] {
    COMM_TRACE("abort: start");
    if (g_rccl.CommAbort) (void)g_rccl.CommAbort(comm);
    else (void)g_rccl.CommDestroy(comm);
    COMM_TRACE("abort: returned");
    done->set_value();
  }).detach();
  if (fut.wait_for(std::chrono::seconds(2)) != std::future_status::ready) COMM_TRACE("abort: still running after 2 s, left behind");
}

}  // namespace

int qf_comm_unique_id(uint8_t* id) {
  static_assert(QF_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "QF_COMM_ID_BYTES mirrors ncclUniqueId");
  if (!id) return QF_ERR_ARG;
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (!g_rccl.load()) { g_create_error = g_rccl.err; return QF_ERR_DEVICE; }
  ncclUniqueId u;
  const ncclResult_t r = g_rccl.GetUniqueId(&u);
  if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return QF_ERR_DEVICE; }
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return QF_OK;
}

int qf_comm_init_rank(qf_ctx* c, const uint8_t* id, int rank, int n_ranks) {
  if (!c || !id) return QF_ERR_ARG;
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(c, QF_ERR_ARG, "rank out of range");
  {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!g_rccl.load()) return fail(c, QF_ERR_DEVICE, g_rccl.err);
  }
  qf_comm_destroy(c);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->d_comm.reserve((((size_t)qf_counts_size(c) + 1) * 5 + 64) * 8));   // the staging buffers now (five words per value on the exact path): nothing to allocate between the ranks' collectives
  c->h_comm.reserve(((size_t)qf_counts_size(c) + 1) * 5 + 64);
  ncclUniqueId u;
  memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  // ncclCommInitRank returns only when all n_ranks have called it, and a rank whose peer died before getting here would wait
  // for ever.  (RCCL 2.27's non-blocking form, ncclCommInitRankConfig with blocking = 0, does not return either while the
  // bootstrap waits for a peer -- measured on this pool.)  So the call runs on a thread of its own and this one waits for it
  // with a deadline; a set-up that is still stuck then is left behind with its thread, and the caller gets a status.
  const double limit = comm_timeout_s();
  struct InitState { std::promise<ncclResult_t> done; ncclComm_t comm = nullptr; };
  auto st = std::make_shared<InitState>();
  std::future<ncclResult_t> fut = st->done.get_future();
  const int device = c->device;
  COMM_TRACE("init: ncclCommInitRank on a helper thread, rank %d of %d, limit %.0f s", rank, n_ranks, limit);
  std::thread([st, u, rank, n_ranks, device] {
    (void)hipSetDevice(device);
    const ncclResult_t r = g_rccl.CommInitRank(&st->comm, n_ranks, u, rank);
    COMM_TRACE("init: ncclCommInitRank returned %d", (int)r);
    st->done.set_value(r);
  }).detach();
  if (fut.wait_for(std::chrono::duration<double>(limit)) != std::future_status::ready)
    return fail(c, QF_ERR_DEVICE, "RCCL communicator set-up: rank " + std::to_string(rank) + " of " + std::to_string(n_ranks) + " gave up after " +
                                      std::to_string((int)limit) + " s waiting for the other ranks");
  const ncclResult_t r = fut.get();
  if (r != ncclSuccess) return fail(c, QF_ERR_DEVICE, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r));
  c->comm = st->comm;
  c->comm_rank = rank;
  c->comm_size = n_ranks;
  return QF_OK;
}

int qf_comm_init_all(qf_ctx* const* ctxs, int n) {
  if (!ctxs || n < 1) return QF_ERR_ARG;
  for (int k = 0; k < n; ++k) if (!ctxs[k]) return QF_ERR_ARG;
  qf_ctx* c0 = ctxs[0];
  std::vector<int> dev(n);
  for (int k = 0; k < n; ++k) {
    dev[k] = ctxs[k]->device;
    for (int q = 0; q < k; ++q)
      if (dev[q] == dev[k]) return fail(c0, QF_ERR_UNSUPPORTED, "two contexts on one device cannot share an RCCL communicator (one rank per GPU)");
  }
  {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!g_rccl.load()) return fail(c0, QF_ERR_DEVICE, g_rccl.err);
  }
  for (int k = 0; k < n; ++k) qf_comm_destroy(ctxs[k]);
  for (int k = 0; k < n; ++k) {     // staging buffers before the communicators exist: a failed allocation fails the call, not one rank of a collective
    HIPCHK(ctxs[k], hipSetDevice(ctxs[k]->device));
    HIPCHK(ctxs[k], ctxs[k]->d_comm.reserve((((size_t)qf_counts_size(ctxs[k]) + 1) * 5 + 64) * 8));
    ctxs[k]->h_comm.reserve(((size_t)qf_counts_size(ctxs[k]) + 1) * 5 + 64);
  }
  std::vector<ncclComm_t> comms(n, nullptr);
  const ncclResult_t r = g_rccl.CommInitAll(comms.data(), n, dev.data());
  if (r != ncclSuccess) return fail(c0, QF_ERR_DEVICE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
  for (int k = 0; k < n; ++k) { ctxs[k]->comm = comms[k]; ctxs[k]->comm_rank = k; ctxs[k]->comm_size = n; }
  return QF_OK;
}

int qf_comm_size(const qf_ctx* c) { return c && c->comm ? c->comm_size : 0; }

void qf_comm_destroy(qf_ctx* c) {
  if (!c || !c->comm) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)g_rccl.CommDestroy(c->comm);
  c->comm = nullptr;
  c->comm_rank = 0;
  c->comm_size = 1;
  c->d_comm.release();
}

// Every exit that is not success aborts the communicator: a rank that fails before or inside the collective must not leave its
// peers blocked in it (they then time out of their own wait below, or see RCCL's error), and the communicator is unusable after a
// failed collective anyway.
// One all-reduce of `words` 8-byte values held in c->h_comm (pinned, owned by the context), in place: up, ncclAllReduce, wait
// with a deadline, down.  The wait polls an event recorded behind the collective -- a collective finishes only when every rank
// has entered it -- and the download is queued only after the event has fired: a rank whose peer never arrives returns a status
// with nothing of its own still in flight into host memory (the staging buffer outlives the call anyway).
static int comm_allreduce_staged(qf_ctx* c, size_t words, ncclDataType_t type) {
  auto bail = [&](const std::string& what) {
    (void)hipGetLastError();
    comm_abort(c);
    return fail(c, QF_ERR_DEVICE, what + " (communicator aborted)");
  };
  if (hipSetDevice(c->device) != hipSuccess) return bail("hipSetDevice");
  if (c->d_comm.reserve(words * 8) != hipSuccess) return bail("cannot allocate the all-reduce staging buffer");
  if (!c->ev_comm && hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming) != hipSuccess) return bail("hipEventCreate");
  uint64_t* d = c->d_comm.as<uint64_t>();
  if (hipMemcpyAsync(d, c->h_comm.data(), words * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) return bail("upload of the counts");
  const ncclResult_t r = g_rccl.AllReduce(d, d, words, type, ncclSum, c->comm, c->stream);
  if (r != ncclSuccess) return bail(std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
  if (hipEventRecord(c->ev_comm, c->stream) != hipSuccess) return bail("hipEventRecord");
  const double limit = comm_timeout_s();
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipEventQuery(c->ev_comm);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) return bail(std::string("all-reduce stream: ") + hipGetErrorString(q));
    if (g_rccl.CommGetAsyncError) {
      ncclResult_t st = ncclSuccess;
      if (g_rccl.CommGetAsyncError(c->comm, &st) == ncclSuccess && st != ncclSuccess && st != ncclInProgress)
        return bail(std::string("ncclAllReduce: ") + g_rccl.GetErrorString(st));
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit)
      return bail("the all-reduce did not complete within " + std::to_string((int)limit) + " s: a peer rank is missing");
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  if (hipMemcpyAsync(c->h_comm.data(), d, words * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return bail("download of the counts");
  if (hipStreamSynchronize(c->stream) != hipSuccess) return bail("download of the counts");
  return QF_OK;
}

int qf_allreduce_counts(qf_ctx* c, double* counts, uint32_t n, double* loglike) {
  if (!c || (n && !counts)) return QF_ERR_ARG;
  if (!c->comm) return fail(c, QF_ERR_STATE, "no communicator (qf_comm_init_rank / qf_comm_init_all)");
  const size_t m = (size_t)n + (loglike ? 1 : 0);
  if (!m) return QF_OK;
  c->h_comm.resize(m);
  if (n) memcpy(c->h_comm.data(), counts, (size_t)n * 8);
  if (loglike) memcpy(c->h_comm.data() + n, loglike, 8);
  if (int rc = comm_allreduce_staged(c, m, ncclDouble)) return rc;
  if (n) memcpy(counts, c->h_comm.data(), (size_t)n * 8);
  if (loglike) memcpy(loglike, c->h_comm.data() + n, 8);
  return QF_OK;
}

// ---- 128-bit fixed point (64 fractional bits, two's complement), (low, high) words per value
static const uint64_t kExactInfHigh = 0x8000000000000000ull;   // high word of the "not finite" marker (low word 0)
static bool exact_is_marker(const uint64_t* v) { return v[1] == kExactInfHigh && v[0] == 0; }

void qf_exact_add(uint64_t* acc, const uint64_t* add, uint32_t n) {
  for (uint32_t k = 0; k < n; ++k) {
    uint64_t* a = acc + 2 * (size_t)k;
    const uint64_t* b = add + 2 * (size_t)k;
    if (exact_is_marker(a) || exact_is_marker(b)) { a[0] = 0; a[1] = kExactInfHigh; continue; }
    const unsigned __int128 s = (((unsigned __int128)a[1] << 64) | a[0]) + (((unsigned __int128)b[1] << 64) | b[0]);
    a[0] = (uint64_t)s;
    a[1] = (uint64_t)(s >> 64);
  }
}

void qf_exact_from_double(const double* v, uint32_t n, uint64_t* fx) {
  for (uint32_t k = 0; k < n; ++k) {
    const double x = v[k];
    uint64_t* o = fx + 2 * (size_t)k;
    if (!std::isfinite(x) || std::fabs(x) >= 9.2e18) { o[0] = 0; o[1] = kExactInfHigh; continue; }
    const double ax = std::fabs(x), ip = std::floor(ax);
    unsigned __int128 m = ((unsigned __int128)(uint64_t)ip << 64) | (uint64_t)std::ldexp(ax - ip, 64);   // (ax - ip) 2^64 < 2^64, truncated
    if (x < 0) m = (unsigned __int128)0 - m;
    o[0] = (uint64_t)m;
    o[1] = (uint64_t)(m >> 64);
  }
}

void qf_exact_to_double(const uint64_t* fx, uint32_t n, double* out) {
  for (uint32_t k = 0; k < n; ++k) {
    const uint64_t* v = fx + 2 * (size_t)k;
    if (exact_is_marker(v)) { out[k] = -INFINITY; continue; }   // (a sum that met a non-finite term: log-likelihoods of reads without any path)
    unsigned __int128 m = ((unsigned __int128)v[1] << 64) | v[0];
    const bool neg = (v[1] >> 63) != 0;
    if (neg) m = (unsigned __int128)0 - m;
    // high + low 2^-64 in double (two roundings, always the same two: |high| < 2^53 for every sum this library forms)
    const uint64_t hi = (uint64_t)(m >> 64), lo = (uint64_t)m;
    const double d = (double)hi + std::ldexp((double)lo, -64);
    out[k] = neg ? -d : d;
  }
}

// The exact E-step reduction: every value as four 32-bit limbs in 64-bit words, one ncclAllReduce(sum, uint64) -- limb sums
// cannot overflow below 2^32 ranks -- and the carries resolved afterwards: the same 128-bit totals on every rank, whatever the
// number of ranks and the order RCCL adds in (qf_allreduce_counts adds doubles: equal to rounding only).
int qf_allreduce_counts_exact(qf_ctx* c, uint64_t* fx, uint32_t n) {
  if (!c || (n && !fx)) return QF_ERR_ARG;
  if (!c->comm) return fail(c, QF_ERR_STATE, "no communicator (qf_comm_init_rank / qf_comm_init_all)");
  if (!n) return QF_OK;
  // a marker (non-finite sum) travels as a count in a fifth word per value
  c->h_comm.resize((size_t)n * 5);
  uint64_t* limbs = c->h_comm.data();
  for (uint32_t k = 0; k < n; ++k) {
    const uint64_t lo = fx[2 * (size_t)k], hi = fx[2 * (size_t)k + 1];
    const bool mark = exact_is_marker(fx + 2 * (size_t)k);
    uint64_t* L = &limbs[(size_t)k * 5];
    L[0] = mark ? 0 : (lo & 0xFFFFFFFFull); L[1] = mark ? 0 : (lo >> 32); L[2] = mark ? 0 : (hi & 0xFFFFFFFFull); L[3] = mark ? 0 : (hi >> 32);
    L[4] = mark ? 1 : 0;
  }
  if (int rc = comm_allreduce_staged(c, (size_t)n * 5, ncclUint64)) return rc;
  limbs = c->h_comm.data();
  for (uint32_t k = 0; k < n; ++k) {
    const uint64_t* L = &limbs[(size_t)k * 5];
    if (L[4]) { fx[2 * (size_t)k] = 0; fx[2 * (size_t)k + 1] = kExactInfHigh; continue; }
    unsigned __int128 v = (unsigned __int128)L[0] + ((unsigned __int128)L[1] << 32) + ((unsigned __int128)L[2] << 64) + ((unsigned __int128)L[3] << 96);
    fx[2 * (size_t)k] = (uint64_t)v;
    fx[2 * (size_t)k + 1] = (uint64_t)(v >> 64);
  }
  return QF_OK;
}

int64_t qf_envelope(qf_ctx* c, const qf_dp_config* cfg, uint32_t read, uint32_t ref, int32_t* diags, uint64_t cap) {
  if (int rc = check_cfg(c, cfg)) return rc;
  if (read >= c->n_reads || ref >= c->n_refs) return fail(c, QF_ERR_ARG, "pair out of range");
  if (hipSetDevice(c->device) != hipSuccess) return QF_ERR_DEVICE;
  const bool sparse = cfg->sparse != 0;
  if (sparse) if (int rc = ensure_ref_index(c, cfg->kmer_len)) return rc;
  const uint64_t n_pairs = (uint64_t)c->n_reads * c->n_refs;
  const uint32_t max_units = 4096;
  if (int rc = reserve_pair_buffers(c, n_pairs, max_units)) return rc == kSplitChunk ? QF_ERR_MEMORY : rc;
  if (int rc = prep_reads(c, sparse ? cfg->kmer_len : 0)) return rc;
  const int xLen = (int)(c->ref_off[ref + 1] - c->ref_off[ref]), yLen = (int)(c->read_off[read + 1] - c->read_off[read]);
  const int nd = xLen + yLen - 1;
  if (c->d_cover.reserve((size_t)nd + 16) != hipSuccess) return fail(c, QF_ERR_MEMORY, "out of device memory");
  SeedArgs sa;
  fill_seed_args(c, *c, cfg, sa, max_units, sparse ? (int)(c->ref_maxlen + c->read_maxlen - 1) : 2);
  sa.pair_base = read * c->n_refs + ref;
  sa.dump_cover = c->d_cover.as<uint8_t>();
  if (int rc = reserve_seed_workspace(c, sa, sparse && cfg->kmer_threshold < 0, 1)) return rc;
  if (launch_seed(sa, 1, sparse && cfg->kmer_threshold < 0, c->stream) != 0) return fail(c, QF_ERR_UNSUPPORTED, "sequence too long for the LDS histogram");
  std::vector<uint8_t> cover(nd);
  if (hipMemcpyAsync(cover.data(), c->d_cover.p, nd, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess)
    return fail(c, QF_ERR_DEVICE, "copy failed");
  int64_t n = 0;
  for (int b = 0; b < nd; ++b)
    if (cover[b]) {
      if ((uint64_t)n < cap && diags) diags[n] = b + (1 - yLen);
      ++n;
    }
  return n;
}

size_t qf_cigar_string(const uint32_t* runs, uint32_t n_runs, char* buf, size_t cap) {
  std::string s;
  for (uint32_t a = 0; a < n_runs; ++a) {
    s += "MID"[runs[a] & 3u];
    s += std::to_string(runs[a] >> 2);
  }
  if (buf && cap) snprintf(buf, cap, "%s", s.c_str());
  return s.size();
}

int qf_synth_ref(uint64_t seed, uint64_t ref_len, char* seq) {
  if (!seq) return QF_ERR_ARG;
  synth_ref(seed, ref_len, seq);
  return QF_OK;
}

int qf_synth_reads(uint64_t seed, const char* ref, uint64_t ref_len, uint32_t n_reads, uint32_t read_len, char* seq,
                   char* qual, uint64_t* offsets) {
  if (!ref || !seq || !qual || !offsets || !ref_len || !read_len) return QF_ERR_ARG;
  synth_reads(seed, ref, ref_len, n_reads, read_len, seq, qual, offsets);
  return QF_OK;
}

}  // extern "C"
