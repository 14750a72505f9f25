"""ctypes binding of include/quaff_hip.h (one function per C entry point, same names minus qf_)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
_LIB = None


class QuaffHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libquaffhip error %d: %s" % (code, msg))
        self.code = code


def library_path():
    # QUAFF_HIP_LIBRARY: developer A/B builds of the same library (tools/dev/variant.sh); never a different implementation
    return os.environ.get("QUAFF_HIP_LIBRARY") or os.path.join(HERE, "libquaffhip.so")


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of csrc/ (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j4"]
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    subprocess.check_call(args)
    return library_path()


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the device code's sources (csrc/*.hip, *.hpp, *.h, Makefile): profiles/r*_pmc_*.json
    record it at capture time, and bench.py replays a capture's counters only while it still matches."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".hpp", ".h")) or name == "Makefile":
            h.update(name.encode())
            h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()[:16]


class DPConfig(C.Structure):
    """qf_dp_config; defaults are `quaff align`'s (t/quaff.cpp:128, src/qmodel.h:303-335)."""
    _fields_ = [("local", C.c_int32), ("sparse", C.c_int32), ("kmer_len", C.c_int32), ("kmer_threshold", C.c_int32),
                ("band_size", C.c_int32), ("reserved", C.c_int32), ("max_size", C.c_uint64)]

    def __init__(self, local=True, sparse=True, kmer_len=6, kmer_threshold=20, band_size=64, max_size=0):
        super().__init__(int(local), int(sparse), kmer_len, kmer_threshold, band_size, 0, max_size)


class _Alignment(C.Structure):
    _fields_ = [("read", C.c_uint32), ("ref", C.c_uint32), ("viterbi", C.c_double), ("score", C.c_double),
                ("x_start", C.c_uint32), ("x_end", C.c_uint32), ("n_columns", C.c_uint32), ("n_runs", C.c_uint32),
                ("run_offset", C.c_uint64)]


class _AlignResult(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("n_refs", C.c_uint32), ("viterbi", C.POINTER(C.c_double)),
                ("cells", C.POINTER(C.c_uint64)), ("n_diagonals", C.POINTER(C.c_uint32)),
                ("null_loglike", C.POINTER(C.c_double)), ("total_cells", C.c_uint64), ("n_alignments", C.c_uint32),
                ("alignments", C.POINTER(_Alignment)), ("cigar_runs", C.POINTER(C.c_uint32)),
                ("ms_prep", C.c_float), ("ms_seed", C.c_float), ("ms_fill", C.c_float), ("ms_traceback", C.c_float),
                ("ms_total", C.c_float), ("n_units", C.c_uint64), ("traceback_bytes", C.c_uint64),
                ("ms_fill_class", C.c_float * 16), ("cells_class", C.c_uint64 * 16), ("units_class", C.c_uint32 * 16),
                ("n_fill_classes", C.c_uint32)]


class _CountResult(C.Structure):
    _fields_ = [("n_reads", C.c_uint32), ("n_refs", C.c_uint32), ("forward", C.POINTER(C.c_double)),
                ("weight", C.POINTER(C.c_double)), ("read_loglike", C.POINTER(C.c_double)),
                ("sort_order", C.POINTER(C.c_uint32)), ("sort_count", C.POINTER(C.c_uint32)),
                ("counts", C.POINTER(C.c_double)), ("counts_size", C.c_uint32), ("loglike", C.c_double),
                ("total_cells", C.c_uint64), ("backward_cells", C.c_uint64), ("forward_bytes", C.c_uint64),
                ("ms_prep", C.c_float), ("ms_seed", C.c_float), ("ms_forward", C.c_float), ("ms_plan", C.c_float),
                ("ms_backward", C.c_float), ("ms_total", C.c_float),
                ("ms_forward_class", C.c_float * 16), ("ms_backward_class", C.c_float * 16), ("cells_class", C.c_uint64 * 16),
                ("units_class", C.c_uint32 * 16), ("n_fill_classes", C.c_uint32),
                ("counts_exact", C.POINTER(C.c_uint64)), ("loglike_exact", C.c_uint64 * 2)]


class _OverlapAlignment(C.Structure):
    _fields_ = [("pair", C.c_uint32), ("viterbi", C.c_double), ("score", C.c_double), ("x_start", C.c_uint32),
                ("x_end", C.c_uint32), ("y_start", C.c_uint32), ("y_end", C.c_uint32), ("n_columns", C.c_uint32),
                ("n_runs", C.c_uint32), ("run_offset", C.c_uint64)]


class _OverlapResult(C.Structure):
    _fields_ = [("n_pairs", C.c_uint32), ("viterbi", C.POINTER(C.c_double)), ("score", C.POINTER(C.c_double)),
                ("cells", C.POINTER(C.c_uint64)), ("n_diagonals", C.POINTER(C.c_uint32)), ("total_cells", C.c_uint64),
                ("n_alignments", C.c_uint32), ("alignments", C.POINTER(_OverlapAlignment)),
                ("state_runs", C.POINTER(C.c_uint32)), ("ms_prep", C.c_float), ("ms_seed", C.c_float),
                ("ms_fill", C.c_float), ("ms_traceback", C.c_float), ("ms_total", C.c_float), ("traceback_bytes", C.c_uint64),
                ("ms_fill_class", C.c_float * 16), ("cells_class", C.c_uint64 * 16), ("units_class", C.c_uint32 * 16),
                ("n_fill_classes", C.c_uint32)]


class _OverlapHit(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32), ("viterbi", C.c_double), ("score", C.c_double), ("x_start", C.c_uint32),
                ("x_end", C.c_uint32), ("y_start", C.c_uint32), ("y_end", C.c_uint32), ("n_columns", C.c_uint32),
                ("n_runs", C.c_uint32), ("run_offset", C.c_uint64)]


class _OverlapRowsResult(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("x1", C.c_uint32), ("n_pairs", C.c_uint64), ("n_finite", C.c_uint64),
                ("total_cells", C.c_uint64), ("total_diagonals", C.c_uint64), ("result_checksum", C.c_uint64),
                ("n_hits", C.c_uint32), ("hits", C.POINTER(_OverlapHit)), ("state_runs", C.POINTER(C.c_uint32)),
                ("n_blocks", C.c_uint32), ("ms_prep", C.c_float), ("ms_seed", C.c_float), ("ms_fill", C.c_float),
                ("ms_traceback", C.c_float), ("ms_total", C.c_float), ("traceback_bytes", C.c_uint64),
                ("ms_fill_class", C.c_float * 16), ("cells_class", C.c_uint64 * 16), ("units_class", C.c_uint32 * 16),
                ("n_fill_classes", C.c_uint32)]


# numpy view of the hit records (same layout as _OverlapHit): a whole triangle returns ~10^6 of them
HIT_DTYPE = np.dtype([("x", "<u4"), ("y", "<u4"), ("viterbi", "<f8"), ("score", "<f8"), ("x_start", "<u4"), ("x_end", "<u4"),
                      ("y_start", "<u4"), ("y_end", "<u4"), ("n_columns", "<u4"), ("n_runs", "<u4"), ("run_offset", "<u8")])
assert HIT_DTYPE.itemsize == C.sizeof(_OverlapHit)


EXPORTS = ["qf_ctx_create", "qf_ctx_destroy", "qf_last_error", "qf_device_name", "qf_set_params_json", "qf_get_scores",
           "qf_set_null_json", "qf_get_lse_table", "qf_set_refs", "qf_upload_reads", "qf_align_resident",
           "qf_align_batch", "qf_envelope", "qf_cigar_string", "qf_synth_ref", "qf_synth_reads", "qf_scores_from_json",
           "qf_fill_class_name", "qf_count_resident", "qf_counts_size", "qf_overlap_resident", "qf_set_params_raw", "qf_set_null_raw", "qf_set_memory_budget", "qf_set_pipeline_chunks",
           "qf_device_count", "qf_set_score_threshold", "qf_comm_unique_id", "qf_comm_init_rank", "qf_comm_init_all",
           "qf_comm_size", "qf_comm_destroy", "qf_allreduce_counts", "qf_overlap_rows", "qf_overlap_rows_pairs",
           "qf_exact_add", "qf_exact_to_double", "qf_exact_from_double", "qf_allreduce_counts_exact", "qf_device_bus_id"]


def load_library():
    """dlopen libquaffhip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise QuaffHipError(-1, "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        L = C.CDLL(path)
        L.qf_last_error.restype = C.c_char_p
        L.qf_last_error.argtypes = [C.c_void_p]
        L.qf_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.qf_set_score_threshold.argtypes = [C.c_void_p, C.c_double]
        L.qf_device_count.argtypes = []
        L.qf_device_count.restype = C.c_int
        L.qf_ctx_destroy.argtypes = [C.c_void_p]
        L.qf_ctx_destroy.restype = None
        L.qf_set_params_json.argtypes = [C.c_void_p, C.c_char_p]
        L.qf_set_null_json.argtypes = [C.c_void_p, C.c_char_p]
        L.qf_get_scores.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p]
        L.qf_get_lse_table.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int)]
        L.qf_set_refs.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32]
        L.qf_upload_reads.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint32]
        L.qf_align_resident.argtypes = [C.c_void_p, C.POINTER(DPConfig), C.c_uint32, C.POINTER(_AlignResult)]
        L.qf_align_batch.argtypes = [C.c_void_p, C.POINTER(DPConfig), C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint32,
                                     C.c_uint32, C.POINTER(_AlignResult)]
        L.qf_envelope.restype = C.c_int64
        L.qf_envelope.argtypes = [C.c_void_p, C.POINTER(DPConfig), C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]
        L.qf_cigar_string.restype = C.c_size_t
        L.qf_cigar_string.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]
        L.qf_synth_ref.argtypes = [C.c_uint64, C.c_uint64, C.c_char_p]
        L.qf_synth_reads.argtypes = [C.c_uint64, C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_char_p, C.c_char_p,
                                     C.c_void_p]
        L.qf_device_name.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.qf_scores_from_json.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_char_p, C.c_size_t]
        L.qf_count_resident.argtypes = [C.c_void_p, C.POINTER(DPConfig), C.c_uint32, C.c_void_p, C.c_void_p,
                                        C.POINTER(_CountResult)]
        L.qf_overlap_resident.argtypes = [C.c_void_p, C.POINTER(DPConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                          C.POINTER(_OverlapResult)]
        L.qf_overlap_rows.argtypes = [C.c_void_p, C.POINTER(DPConfig), C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.POINTER(_OverlapRowsResult)]
        L.qf_overlap_rows_pairs.restype = C.c_uint64
        L.qf_overlap_rows_pairs.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.qf_counts_size.restype = C.c_uint32
        L.qf_counts_size.argtypes = [C.c_void_p]
        L.qf_fill_class_name.restype = C.c_char_p
        L.qf_fill_class_name.argtypes = [C.c_uint32]
        L.qf_comm_unique_id.argtypes = [C.c_char_p]
        L.qf_comm_init_rank.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        L.qf_comm_size.argtypes = [C.c_void_p]
        L.qf_comm_destroy.argtypes = [C.c_void_p]
        L.qf_comm_destroy.restype = None
        L.qf_allreduce_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.qf_allreduce_counts_exact.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        for f in (L.qf_exact_add, L.qf_exact_to_double, L.qf_exact_from_double):
            f.restype = None
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32] if f is not L.qf_exact_to_double else [C.c_void_p, C.c_uint32, C.c_void_p]
        L.qf_exact_from_double.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        _LIB = L
    return _LIB


ALIGN_BEST, ALIGN_ALL, ALIGN_NO_TRACEBACK = 0, 1, 2


def scores_from_json(text=None):
    """Host-only: (match_len, gap_len, ins[4][95], mat[4][Km][95], trans[4Kg+4]) for a params JSON (None = defaults)."""
    L = load_library()
    ml, gl = C.c_int(), C.c_int()
    err = C.create_string_buffer(256)
    t = text.encode() if text is not None else None
    rc = L.qf_scores_from_json(t, C.byref(ml), C.byref(gl), None, None, None, err, 256)
    if rc:
        raise QuaffHipError(rc, err.value.decode())
    Km, Kg = 4 ** ml.value, 4 ** gl.value
    ins, mat, trans = np.zeros((4, 95)), np.zeros((4, Km, 95)), np.zeros(4 * Kg + 4)
    rc = L.qf_scores_from_json(t, None, None, ins.ctypes.data, mat.ctypes.data, trans.ctypes.data, err, 256)
    if rc:
        raise QuaffHipError(rc, err.value.decode())
    return ml.value, gl.value, ins, mat, trans


def exact_add(acc, add):
    """acc += add for arrays of 128-bit fixed-point values ((low, high) uint64 words per value: qf_count_result.counts_exact);
    returns a new array."""
    a = np.ascontiguousarray(acc, np.uint64).copy().reshape(-1, 2)
    b = np.ascontiguousarray(add, np.uint64).reshape(-1, 2)
    assert a.shape == b.shape
    load_library().qf_exact_add(a.ctypes.data, b.ctypes.data, len(a))
    return a


def exact_to_double(fx):
    a = np.ascontiguousarray(fx, np.uint64).reshape(-1, 2)
    out = np.zeros(len(a))
    load_library().qf_exact_to_double(a.ctypes.data, len(a), out.ctypes.data)
    return out


def exact_from_double(v):
    a = np.ascontiguousarray(v, np.float64).reshape(-1)
    out = np.zeros((len(a), 2), np.uint64)
    load_library().qf_exact_from_double(a.ctypes.data, len(a), out.ctypes.data)
    return out


def synth_ref(seed, length):
    buf = C.create_string_buffer(length)
    load_library().qf_synth_ref(seed, length, buf)
    return buf.raw


def synth_reads(seed, ref, n_reads, read_len):
    """Returns (seq bytes, qual bytes, offsets uint64[n_reads+1])."""
    cap = n_reads * 2 * read_len + 16
    seq, qual = C.create_string_buffer(cap), C.create_string_buffer(cap)
    off = np.zeros(n_reads + 1, np.uint64)
    rc = load_library().qf_synth_reads(seed, ref, len(ref), n_reads, read_len, seq, qual, off.ctypes.data_as(C.c_void_p))
    if rc:
        raise QuaffHipError(rc, "qf_synth_reads")
    tot = int(off[-1])
    return seq.raw[:tot], qual.raw[:tot], off


_COMP = bytes.maketrans(b"ACGTacgt", b"TGCATGCA")


def revcomp(seq):
    """Reverse complement (revcomp(), src/fastseq.cpp:209-216) of bytes/str."""
    b = seq.encode() if isinstance(seq, str) else seq
    return b.translate(_COMP)[::-1]


def pack(seqs):
    """list of str/bytes -> (concatenated bytes, offsets)."""
    bs = [s.encode() if isinstance(s, str) else s for s in seqs]
    off = np.zeros(len(bs) + 1, np.uint64)
    np.cumsum([len(b) for b in bs], out=off[1:])
    return b"".join(bs), off


class Context:
    """qf_ctx: one per GPU.  Mirrors the call order of a quaff run: params -> null -> refs -> reads -> align."""

    def __init__(self, device=0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.qf_ctx_create(device, C.byref(h))
        if rc:
            raise QuaffHipError(rc, self.L.qf_last_error(None).decode())
        self.h = h
        self.n_reads = self.n_refs = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.qf_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc):
        if rc:
            raise QuaffHipError(rc, self.L.qf_last_error(self.h).decode())

    def set_memory_budget(self, nbytes):
        self.L.qf_set_memory_budget.argtypes = [C.c_void_p, C.c_uint64]
        self._chk(self.L.qf_set_memory_budget(self.h, nbytes))

    def set_debug_flags(self, flags):
        """Tests / A-B only (csrc/qf_internal.h, not part of the public ABI): force a kernel variant; results never change."""
        self.L.qf_debug_set_flags.argtypes = [C.c_void_p, C.c_uint32]
        self._chk(self.L.qf_debug_set_flags(self.h, flags))

    def fail_chunk_reserve(self, nth):
        """Tests only (csrc/qf_internal.h): the nth per-chunk device reserve from now (0 = the next one, counted over the whole
        process) fails once as if the device were out of memory: the chunk is split and retried.  -1 switches it off.  Returns
        the countdown it replaces: negative once the failure has happened."""
        self.L.qf_debug_fail_chunk_reserve.argtypes = [C.c_int]
        self.L.qf_debug_fail_chunk_reserve.restype = C.c_int
        return int(self.L.qf_debug_fail_chunk_reserve(int(nth)))

    def rows_settled(self):
        """Tests only (csrc/qf_internal.h): pairs the overlap seeding's row prefilter settled in the last overlap call
        (counted under debug flag 512)."""
        self.L.qf_debug_rows_settled.argtypes = [C.c_void_p]
        self.L.qf_debug_rows_settled.restype = C.c_uint64
        return int(self.L.qf_debug_rows_settled(self.h))

    def lse_pack_bytes(self):
        """Tests only (csrc/qf_internal.h): bytes of the packed log-sum-exp table the overlap fills keep in LDS (0: not used)."""
        self.L.qf_debug_lse_pack_bytes.argtypes = [C.c_void_p]
        self.L.qf_debug_lse_pack_bytes.restype = C.c_uint32
        return int(self.L.qf_debug_lse_pack_bytes(self.h))

    # ---- E-step reduction across GPUs (RCCL)
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes from rank 0 (ncclGetUniqueId) for every rank's comm_init_rank."""
        L = load_library()
        buf = C.create_string_buffer(128)
        rc = L.qf_comm_unique_id(buf)
        if rc:
            raise QuaffHipError(rc, L.qf_last_error(None).decode())
        return buf.raw

    def comm_init_rank(self, unique_id, rank, n_ranks):
        """Collective: returns when all n_ranks contexts (one per GPU, any processes) have called it."""
        assert len(unique_id) == 128
        self._chk(self.L.qf_comm_init_rank(self.h, unique_id, rank, n_ranks))

    def comm_size(self):
        return int(self.L.qf_comm_size(self.h))

    def allreduce_counts(self, counts, loglike):
        """Sum the flattened E-step counts and the log-likelihood over all ranks (RCCL all-reduce, fp64); returns (counts, loglike)."""
        v = np.ascontiguousarray(counts, dtype=np.float64).copy()
        ll = C.c_double(float(loglike))
        self._chk(self.L.qf_allreduce_counts(self.h, v.ctypes.data, len(v), C.byref(ll)))
        return v, ll.value

    def allreduce_counts_exact(self, fx):
        """Sum 128-bit fixed-point values ((n, 2) uint64: counts_exact, loglike_exact ...) over all ranks, exactly: every rank gets
        the same words for any number of ranks."""
        v = np.ascontiguousarray(fx, np.uint64).copy().reshape(-1, 2)
        self._chk(self.L.qf_allreduce_counts_exact(self.h, v.ctypes.data, len(v)))
        return v

    def set_score_threshold(self, min_score):
        """Alignments scoring below min_score are not traced back / returned (-inf = all; the CLI's -threshold)."""
        self._chk(self.L.qf_set_score_threshold(self.h, float(min_score)))

    def set_pipeline_chunks(self, n):
        self.L.qf_set_pipeline_chunks.argtypes = [C.c_void_p, C.c_uint32]
        self._chk(self.L.qf_set_pipeline_chunks(self.h, n))

    def device_name(self):
        b = C.create_string_buffer(256)
        self._chk(self.L.qf_device_name(self.h, b, 256))
        return b.value.decode()

    def device_bus_id(self):
        b = C.create_string_buffer(64)
        self.L.qf_device_bus_id.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        self._chk(self.L.qf_device_bus_id(self.h, b, 64))
        return b.value.decode()

    def set_params_json(self, text=None):
        self._chk(self.L.qf_set_params_json(self.h, text.encode() if text is not None else None))

    def set_null_json(self, text=None):
        self._chk(self.L.qf_set_null_json(self.h, text.encode() if text is not None else None))

    def get_scores(self):
        ml, gl = C.c_int(), C.c_int()
        self._chk(self.L.qf_get_scores(self.h, C.byref(ml), C.byref(gl), None, None, None))
        Km, Kg = 4 ** ml.value, 4 ** gl.value
        ins, mat, trans = np.zeros((4, 95)), np.zeros((4, Km, 95)), np.zeros(4 * Kg + 4)
        self._chk(self.L.qf_get_scores(self.h, None, None, ins.ctypes.data, mat.ctypes.data, trans.ctypes.data))
        return ml.value, gl.value, ins, mat, trans

    def lse_table(self):
        p, n = C.POINTER(C.c_double)(), C.c_int()
        self._chk(self.L.qf_get_lse_table(self.h, C.byref(p), C.byref(n)))
        return np.ctypeslib.as_array(p, (n.value,)).copy()

    def set_refs(self, seqs):
        data, off = pack(seqs)
        self._chk(self.L.qf_set_refs(self.h, data, off.ctypes.data, len(seqs)))
        self.n_refs = len(seqs)

    def upload_reads_packed(self, seq, qual, off):
        self._chk(self.L.qf_upload_reads(self.h, seq, qual, off.ctypes.data, len(off) - 1))
        self.n_reads = len(off) - 1

    def upload_reads(self, seqs, quals=None):
        data, off = pack(seqs)
        q = pack(quals)[0] if quals is not None else None
        self.upload_reads_packed(data, q, off)

    def align_resident(self, cfg=None, flags=ALIGN_BEST, raw=False, reads_below=None):
        cfg = cfg or DPConfig()
        res = _AlignResult()
        self._chk(self.L.qf_align_resident(self.h, C.byref(cfg), flags, C.byref(res)))
        return res if raw else self._unpack(res, reads_below)

    def _unpack(self, res, reads_below=None):
        n = res.n_reads * res.n_refs
        shape = (res.n_reads, res.n_refs)
        out = {
            "viterbi": np.ctypeslib.as_array(res.viterbi, (n,)).reshape(shape).copy() if n else np.zeros(shape),
            "cells": np.ctypeslib.as_array(res.cells, (n,)).reshape(shape).copy() if n else np.zeros(shape, np.uint64),
            "n_diagonals": np.ctypeslib.as_array(res.n_diagonals, (n,)).reshape(shape).copy() if n else np.zeros(shape, np.uint32),
            "null_loglike": np.ctypeslib.as_array(res.null_loglike, (res.n_reads,)).copy() if res.n_reads else np.zeros(0),
            "total_cells": int(res.total_cells), "n_units": int(res.n_units), "traceback_bytes": int(res.traceback_bytes),
            "ms": {k: getattr(res, "ms_" + k) for k in ("prep", "seed", "fill", "traceback", "total")},
            "alignments": [],
            "classes": [{"name": self.L.qf_fill_class_name(k).decode(), "ms": res.ms_fill_class[k],
                         "cells": int(res.cells_class[k]), "units": int(res.units_class[k])}
                        for k in range(res.n_fill_classes) if res.units_class[k]],
        }
        for a in range(res.n_alignments):
            al = res.alignments[a]
            if reads_below is not None and al.read >= reads_below:
                continue
            runs = np.ctypeslib.as_array(C.cast(C.addressof(res.cigar_runs.contents) + 4 * al.run_offset,
                                                C.POINTER(C.c_uint32)), (al.n_runs,)).copy() if al.n_runs else np.zeros(0, np.uint32)
            out["alignments"].append({
                "read": al.read, "ref": al.ref, "viterbi": al.viterbi, "score": al.score, "xStart": al.x_start,
                "xEnd": al.x_end, "n_columns": al.n_columns, "runs": runs,
                "cigar": "".join("MID"[int(r) & 3] + str(int(r) >> 2) for r in runs),
                "ops": "".join("MID"[int(r) & 3] * (int(r) >> 2) for r in runs)})
        return out

    def count_resident(self, cfg=None, force=False, sort_order=None, packed_order=False):
        """One Forward-Backward E-step over the resident reads.  sort_order: optional list (per read) of reference
        indices from the previous iteration, or the packed form (uint32 [n_reads, n_refs] order, uint32 [n_reads] counts) that
        packed_order=True returns (no per-read Python lists: 20 k reads cost ~20 ms of interpreter time otherwise).
        Returns dict(forward, weight, read_loglike, sort_order, counts, ...)."""
        cfg = cfg or DPConfig()
        res = _CountResult()
        si = sn = None
        if isinstance(sort_order, tuple):
            si, sn = np.ascontiguousarray(sort_order[0], np.uint32), np.ascontiguousarray(sort_order[1], np.uint32)
            assert si.shape == (self.n_reads, self.n_refs) and sn.shape == (self.n_reads,)
        elif sort_order is not None:
            si = np.zeros((self.n_reads, self.n_refs), np.uint32)
            sn = np.zeros(self.n_reads, np.uint32)
            for r, o in enumerate(sort_order):
                si[r, :len(o)] = o
                sn[r] = len(o)
        self._chk(self.L.qf_count_resident(self.h, C.byref(cfg), 1 if force else 0,
                                           si.ctypes.data if si is not None else None,
                                           sn.ctypes.data if sn is not None else None, C.byref(res)))
        n = res.n_reads * res.n_refs
        shape = (res.n_reads, res.n_refs)
        arr = lambda ptr, cnt, shp=None: (np.ctypeslib.as_array(ptr, (cnt,)).copy().reshape(shp or (cnt,)) if cnt else np.zeros(shp or (0,)))
        order = arr(res.sort_order, n, shape)
        cnt = arr(res.sort_count, res.n_reads)
        return {"forward": arr(res.forward, n, shape), "weight": arr(res.weight, n, shape),
                "read_loglike": arr(res.read_loglike, res.n_reads),
                "sort_order": (order.astype(np.uint32), cnt.astype(np.uint32)) if packed_order
                              else [list(map(int, order[r, :int(cnt[r])])) for r in range(res.n_reads)],
                "counts": arr(res.counts, res.counts_size), "loglike": res.loglike, "total_cells": int(res.total_cells),
                # the same sums as 128-bit fixed point, (low, high) words per value: order-free, they add exactly across calls
                "counts_exact": (np.ctypeslib.as_array(res.counts_exact, (2 * res.counts_size,)).copy().reshape(-1, 2)
                                 if res.counts_size else np.zeros((0, 2), np.uint64)),
                "loglike_exact": np.array([res.loglike_exact[0], res.loglike_exact[1]], np.uint64),
                "backward_cells": int(res.backward_cells), "forward_bytes": int(res.forward_bytes),
                "ms": {k: getattr(res, "ms_" + k) for k in ("prep", "seed", "forward", "plan", "backward", "total")},
                "classes": [{"cls": k, "geometry": self.L.qf_fill_class_name(k).decode().replace("k_viterbi_", ""),
                             "ms_forward": res.ms_forward_class[k], "ms_backward": res.ms_backward_class[k],
                             "cells": int(res.cells_class[k]), "units": int(res.units_class[k])}
                            for k in range(res.n_fill_classes) if res.units_class[k]]}

    def overlap_resident(self, pairs, cfg=None, raw=False):
        """pairs: list of (x index, y index, y_complemented) over the resident sequences, or a tuple of three numpy arrays
        (uint32 x, uint32 y, uint8 flag).  quaff overlap's defaults are kmer_threshold=14 (DEFAULT_KMER_THRESHOLD,
        src/diagenv.h:15).  raw=True returns the ctypes result struct without unpacking the alignments."""
        cfg = cfg or DPConfig(kmer_threshold=14)
        if isinstance(pairs, tuple):
            px, py, pc = (np.ascontiguousarray(pairs[0], np.uint32), np.ascontiguousarray(pairs[1], np.uint32),
                          np.ascontiguousarray(pairs[2], np.uint8))
        else:
            px = np.array([q[0] for q in pairs], np.uint32)
            py = np.array([q[1] for q in pairs], np.uint32)
            pc = np.array([1 if q[2] else 0 for q in pairs], np.uint8)
        res = _OverlapResult()
        self._chk(self.L.qf_overlap_resident(self.h, C.byref(cfg), px.ctypes.data, py.ctypes.data, pc.ctypes.data, len(px),
                                             C.byref(res)))
        if raw:
            return res
        n = res.n_pairs
        arr = lambda ptr, dt=None: np.ctypeslib.as_array(ptr, (n,)).copy() if n else np.zeros(0)
        out = {"viterbi": arr(res.viterbi), "score": arr(res.score), "cells": arr(res.cells), "n_diagonals": arr(res.n_diagonals),
               "total_cells": int(res.total_cells), "traceback_bytes": int(res.traceback_bytes),
               "ms": {k: getattr(res, "ms_" + k) for k in ("prep", "seed", "fill", "traceback", "total")}, "alignments": {}}
        for a in range(res.n_alignments):
            al = res.alignments[a]
            runs = np.ctypeslib.as_array(C.cast(C.addressof(res.state_runs.contents) + 4 * al.run_offset,
                                                C.POINTER(C.c_uint32)), (al.n_runs,)).copy() if al.n_runs else np.zeros(0, np.uint32)
            out["alignments"][al.pair] = {"pair": al.pair, "result": al.viterbi, "score": al.score, "xStart": al.x_start,
                                          "xEnd": al.x_end, "yStart": al.y_start, "yEnd": al.y_end,
                                          "ops": "".join("MID"[int(r) & 3] * (int(r) >> 2) for r in runs)}
        return out

    def measure_f64_rate(self):
        """Tests / bench only (csrc/qf_internal.h): fp64 vector add lane-operations per second this device sustains."""
        v = C.c_double()
        self.L.qf_debug_measure_f64_rate.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self._chk(self.L.qf_debug_measure_f64_rate(self.h, C.byref(v)))
        return v.value

    def set_overlap_block_pairs(self, pairs):
        """Tests only (csrc/qf_internal.h): pairs per internal row block of overlap_rows (0 = default)."""
        self.L.qf_debug_set_overlap_block_pairs.argtypes = [C.c_void_p, C.c_uint64]
        self._chk(self.L.qf_debug_set_overlap_block_pairs(self.h, pairs))

    def overlap_rows(self, n_originals, x0, x1, cfg=None, raw=False):
        """Rows [x0, x1) of QuaffOverlapScheduler's enumeration (src/qoverlap.cpp:475-480,528-547) over the resident set
        (n_originals reads, optionally followed by their reverse complements): pairs enumerated, thresholded
        (set_score_threshold) and reduced on the device.  Returns totals + the hits as a numpy record array (HIT_DTYPE, ordered
        by (x, y)) + the concatenated state runs; raw=True returns the ctypes struct."""
        cfg = cfg or DPConfig(kmer_threshold=14)
        res = _OverlapRowsResult()
        self._chk(self.L.qf_overlap_rows(self.h, C.byref(cfg), n_originals, x0, x1, C.byref(res)))
        if raw:
            return res
        n = res.n_hits
        hits = (np.frombuffer(C.string_at(res.hits, n * HIT_DTYPE.itemsize), dtype=HIT_DTYPE).copy() if n else np.zeros(0, HIT_DTYPE))
        n_runs = int(hits["run_offset"][-1] + hits["n_runs"][-1]) if n else 0
        runs = np.ctypeslib.as_array(res.state_runs, (n_runs,)).copy() if n_runs else np.zeros(0, np.uint32)
        return {"n_pairs": int(res.n_pairs), "n_finite": int(res.n_finite), "total_cells": int(res.total_cells),
                "total_diagonals": int(res.total_diagonals), "result_checksum": int(res.result_checksum), "hits": hits, "runs": runs,
                "n_blocks": int(res.n_blocks), "traceback_bytes": int(res.traceback_bytes),
                "ms": {k: getattr(res, "ms_" + k) for k in ("prep", "seed", "fill", "traceback", "total")},
                "classes": {k: {"ms": res.ms_fill_class[k], "cells": int(res.cells_class[k]), "units": int(res.units_class[k])}
                            for k in range(res.n_fill_classes) if res.units_class[k]}}

    @staticmethod
    def hit_ops(hit, runs):
        """State path of one hit ("M" / "I" / "D" per column) from the concatenated runs."""
        r = runs[int(hit["run_offset"]):int(hit["run_offset"]) + int(hit["n_runs"])]
        return "".join("MID"[int(v) & 3] * (int(v) >> 2) for v in r)

    def envelope(self, read, ref, cfg=None):
        cfg = cfg or DPConfig()
        n = self.L.qf_envelope(self.h, C.byref(cfg), read, ref, None, 0)
        if n < 0:
            self._chk(int(n))
        d = np.zeros(max(int(n), 1), np.int32)
        n = self.L.qf_envelope(self.h, C.byref(cfg), read, ref, d.ctypes.data, len(d))
        if n < 0:
            self._chk(int(n))
        return d[:n]
