"""oracle/oracle.py — TEST INFRASTRUCTURE ONLY (see oracle/quaff_oracle.c header).

Python glue over oracle/liboracle.so (the plain-C CPU restatement) plus the
host-side pieces of the reference that are easier to restate in Python:
quaff's JSON formats read with gason's number parser, the null-model fit, the
per-read align / count task logic and the Stockholm / SAM writers.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
Each function cites the reference file:line it follows (/root/reference/...).
"""
import ctypes as C
import gzip
import json
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
NQUAL = 94
NQ1 = 95
NEG_INF = float("-inf")

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build():
    """Compile the C restatement (gcc, no FMA contraction)."""
    src = os.path.join(HERE, "quaff_oracle.c")
    out = os.path.join(HERE, "liboracle.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", out, src, "-lm"])
    return out


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.qo_gason_number.restype = C.c_double
        L.qo_gason_number.argtypes = [C.c_char_p]
        L.qo_lse.restype = C.c_double
        L.qo_lse.argtypes = [C.c_double, C.c_double]
        L.qo_lse_table.restype = C.POINTER(C.c_double)
        L.qo_log_negbinom.restype = C.c_double
        L.qo_log_negbinom.argtypes = [C.c_int, C.c_double, C.c_double]
        L.qo_null_loglike.restype = C.c_double
        L.qo_envelope_cells.restype = C.c_uint64
        L.qo_viterbi.restype = C.c_double
        L.qo_forward_backward.restype = C.c_double
        _LIB = L
    return _LIB


# ---------------------------------------------------------------- sequences
def tokens(seq):
    """FastSeq::tokens, src/fastseq.cpp:71-83 (raises on non-ACGT)."""
    b = seq.encode() if isinstance(seq, str) else seq
    out = np.empty(len(b), np.uint8)
    rc = lib().qo_tokenize(b, len(b), out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError("Unknown symbol %r at %d" % (chr(b[-rc - 1]), -rc - 1))
    return out


def quals(qual):
    """FastSeq::qualScores, src/fastseq.cpp:101-109."""
    b = qual.encode() if isinstance(qual, str) else qual
    out = np.empty(len(b), np.uint8)
    lib().qo_quals(b, len(b), out.ctypes.data_as(C.c_void_p))
    return out


def kmers(tok, k):
    """FastSeq::kmers, src/fastseq.cpp:85-99."""
    out = np.empty(len(tok), np.uint32)
    lib().qo_kmers(tok.ctypes.data_as(C.c_void_p), len(tok), int(k), out.ctypes.data_as(C.c_void_p))
    return out


_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "a": "T", "c": "G", "g": "C", "t": "A"}


def revcomp_str(s):
    """revcomp(), src/fastseq.cpp:209-216 (dnaComplementChar upper-cases ACGT, keeps others)."""
    return "".join(_COMP.get(c, c) for c in reversed(s))


class FastSeq:
    """name/comment/seq/qual (+ source coords), src/fastseq.h:41-72."""

    def __init__(self, name, seq, qual="", comment="", source=None):
        self.name, self.seq, self.qual, self.comment = name, seq, qual, comment
        self.source = source  # (name, start, end, rev) or None

    def has_qual(self):
        return len(self.qual) == len(self.seq)

    def revcomp(self):
        """FastSeq::revcomp, src/fastseq.cpp:218-230 (+ compose :51-65)."""
        src = compose(("" + self.name, 1, len(self.seq), True), self.source)
        return FastSeq("revcomp(" + self.name + ")", revcomp_str(self.seq), self.qual[::-1], self.comment, src)


def compose(c, src):
    """SeqIntervalCoords::compose, src/fastseq.cpp:51-65."""
    if src is None:
        return c
    name, start, end, rev = c
    sname, sstart, send, srev = src
    if srev:
        return (sname, send - end + 1, send - start + 1, rev != srev)
    return (sname, start + sstart - 1, end + sstart - 1, rev != srev)


def read_fastx(path):
    """readFastSeqs via kseq, src/fastseq.cpp:143-171: FASTA/FASTQ, optionally gzipped;
    quality kept only if as long as the sequence (initFastSeq :133-141)."""
    op = gzip.open if open(path, "rb").read(2) == b"\x1f\x8b" else open
    with op(path, "rt") as f:
        lines = [l.rstrip("\n").rstrip("\r") for l in f]
    seqs, i = [], 0
    while i < len(lines):
        if not lines[i] or lines[i][0] not in ">@":
            i += 1
            continue
        hdr = lines[i][1:]
        name, _, comment = hdr.partition(" ")
        i += 1
        seq = ""
        while i < len(lines) and lines[i][:1] not in (">", "@", "+"):
            seq += lines[i].strip()
            i += 1
        qual = ""
        if i < len(lines) and lines[i][:1] == "+":
            i += 1
            while i < len(lines) and len(qual) < len(seq):
                qual += lines[i]
                i += 1
            if len(qual) != len(seq):
                qual = ""
        seqs.append(FastSeq(name, seq, qual, comment))
    return seqs


# ------------------------------------------------------------------- JSON
def gason_number(s):
    """string2double, src/gason.cpp:73-117 — python floats are IEEE doubles, so the same
    operation sequence gives the same bits as the C restatement qo_gason_number."""
    return lib().qo_gason_number(s.encode())


def gason_loads(text):
    return json.loads(text, parse_float=gason_number, parse_int=gason_number)


def kmer_string(km, k):
    """kmerToString, src/fastseq.cpp:44-49."""
    s = ""
    for _ in range(k):
        s = "ACGT"[km % 4] + s
        km //= 4
    return s


class Params:
    """QuaffParams, src/qmodel.h:147-163; JSON reader src/qmodel.cpp:230-271."""

    def __init__(self, match_len=1, gap_len=0):
        self.match_len, self.gap_len = match_len, gap_len
        self.Km, self.Kg = 4 ** match_len, 4 ** gap_len
        self.refBase = [0.25] * 4
        self.beginInsert = [0.5] * self.Kg
        self.beginDelete = [0.5] * self.Kg
        self.extendInsert = self.extendDelete = 0.5
        self.insert = np.zeros((4, 3))
        self.match = np.zeros((4, self.Km, 3))

    @staticmethod
    def from_json(text):
        jm = gason_loads(text) if isinstance(text, str) else text
        ml = int(jm.get("matchOrder", 1))  # readJsonKmerLen, src/qmodel.cpp:122-128
        gl = int(jm.get("gapOrder", 0))
        p = Params(ml, gl)
        # NB: refBase is written but never read back (src/qmodel.cpp:236-271) -> stays 0.25
        for g in range(p.Kg):
            ks = kmer_string(g, gl)
            p.beginInsert[g] = jm["beginInsert"][ks]
            p.beginDelete[g] = jm["beginDelete"][ks]
        p.extendInsert, p.extendDelete = jm["extendInsert"], jm["extendDelete"]
        for i in range(4):
            d = jm["insert"]["ACGT"[i]]
            p.insert[i] = (d["p"], d["q"], d["r"])
        for jp in range(0, p.Km, 4):
            pref = kmer_string(jp, ml)[: ml - 1]
            for i in range(4):
                for js in range(4):
                    d = jm["match"][pref]["ACGT"[i]]["ACGT"[js]]
                    p.match[i, jp + js] = (d["p"], d["q"], d["r"])
        return p


class Scores:
    """QuaffScores, src/qmodel.cpp:296-325, flattened: ins[4][95], mat[4][Km][95], trans[4Kg+4]."""

    def __init__(self, p):
        self.Km, self.Kg, self.match_len, self.gap_len = p.Km, p.Kg, p.match_len, p.gap_len
        self.ins = np.zeros((4, NQ1))
        self.mat = np.zeros((4, p.Km, NQ1))
        self.trans = np.zeros(4 * p.Kg + 4)
        lib().qo_build_scores(
            p.Km, p.Kg,
            np.ascontiguousarray(p.insert).ctypes.data_as(C.c_void_p),
            np.ascontiguousarray(p.match).ctypes.data_as(C.c_void_p),
            (C.c_double * p.Kg)(*p.beginInsert), (C.c_double * p.Kg)(*p.beginDelete),
            C.c_double(p.extendInsert), C.c_double(p.extendDelete),
            self.ins.ctypes.data_as(C.c_void_p), self.mat.ctypes.data_as(C.c_void_p),
            self.trans.ctypes.data_as(C.c_void_p))


class NullParams:
    """QuaffNullParams, src/qmodel.cpp:1806-1907."""

    def __init__(self):
        self.nullEmit = 0.5
        self.null = np.tile(np.array([0.25, 0.5, 47.0]), (4, 1))

    @staticmethod
    def from_json(text):
        jm = gason_loads(text) if isinstance(text, str) else text
        n = NullParams()
        n.nullEmit = jm["nullEmit"]
        for i in range(4):
            d = jm["null"]["ACGT"[i]]
            n.null[i] = (d["p"], d["q"], d["r"])
        return n

    @staticmethod
    def fit(seqs, pseudocount=1.0):
        """QuaffNullParams(seqs), src/qmodel.cpp:1811-1843."""
        cnt = np.full((4, NQUAL), pseudocount / NQUAL)
        yes = no = pseudocount
        sym = np.full(4, float(pseudocount))
        for s in seqs:
            no += 1
            yes += len(s.seq)
            t = tokens(s.seq)
            for b in range(4):
                sym[b] += np.count_nonzero(t == b)
            if s.has_qual():
                q = quals(s.qual)
                np.add.at(cnt, (t, q), 1.0)
        n = NullParams()
        n.nullEmit = 1 / (1 + no / yes)
        tot = sym.sum()
        for b in range(4):
            p, r = fit_negbinom(cnt[b])
            n.null[b] = (sym[b] / tot, p, r)
        return n

    def loglike(self, fs):
        """QuaffNullParams::logLikelihood, src/qmodel.cpp:1875-1890."""
        t = tokens(fs.seq)
        q = quals(fs.qual) if fs.has_qual() else None
        return lib().qo_null_loglike(
            C.c_double(self.nullEmit), np.ascontiguousarray(self.null).ctypes.data_as(C.c_void_p),
            t.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p) if q is not None else None, len(t))

    def to_json(self):
        """writeJson, src/qmodel.cpp:1892-1901 (+ SymQualDist::writeJson :58-66)."""
        parts = []
        for i in range(4):
            p, q, r = self.null[i]
            parts.append(' "%s": %s' % ("ACGT"[i], sqd_json(p, q, r)))
        return '{\n  "nullEmit": %s,\n  "null": {%s } }' % (fmt(self.nullEmit), ",".join(parts))


def fmt(x):
    """default ostream<<double (precision 6, %g)."""
    s = "%g" % x
    return s


def sqd_json(p, q, r):
    m = r * (1 - q) / q
    sd = math.sqrt(r * (1 - q) / (q * q))
    return '{ "p": %s, "q": %s, "r": %s, "m": %s, "sd": %s }' % (fmt(p), fmt(q), fmt(r), fmt(m), fmt(sd))


# ------------------------------------------------ negative-binomial ML fit
def _psi(x):
    from scipy.special import digamma
    return float(digamma(x))


def _psi1(x):
    from scipy.special import polygamma
    return float(polygamma(1, x))


def fit_negbinom(kfreq):
    """fitNegativeBinomial, src/negbinom.cpp:112-129: moments -> bracketed root of
    dLL/dr (Brent in the reference; GSL is absent so a bisection/secant bracket to the
    same 1e-3 interval test) -> Newton polish (relative 1e-4).  Returns (p, r)."""
    k = np.arange(len(kfreq), dtype=float)
    cnt = kfreq.sum()
    if cnt <= 0:
        return float("nan"), float("nan")
    mean = (kfreq * k).sum() / cnt
    var = (kfreq * k * k).sum() / cnt - mean * mean
    nz = kfreq > 0

    def d1(r):  # logNegativeBinomialSingleDeriv1, :45-58
        fs = kfreq[nz].sum()
        ks = (kfreq[nz] * k[nz]).sum()
        kd = sum(f * _psi(r + kk) for f, kk in zip(kfreq[nz], k[nz]))
        return -fs * math.log(1. + ks / (fs * r)) - fs * _psi(r) + kd

    def d2(r):  # :60-71
        fs = kfreq[nz].sum()
        kt = sum(f * _psi1(r + kk) for f, kk in zip(kfreq[nz], k[nz]))
        return -fs * _psi1(r) + kt

    def popt(r):  # optimalNegativeBinomialSuccessProb, :78-86
        return 1. / (1 + (kfreq * k).sum() / (kfreq.sum() * r))

    def ll(r):
        p = popt(r)
        return sum(f * lib().qo_log_negbinom(int(kk), p, r) for f, kk in zip(kfreq, k))

    lo, hi = 1., max(1., len(kfreq) - 1.)
    if var > 0 and var > mean:  # momentFit :142-162
        p0 = mean / var
        r0 = mean * p0 / (1 - p0)
        lo, hi = max(1., r0 / 2), min(len(kfreq) - 1., r0 * 2)
    flo, fhi = d1(lo), d1(hi)
    if (flo > 0) == (fhi > 0):  # :190-203: same sign -> better endpoint
        r = lo if ll(lo) > ll(hi) else hi
    else:
        a, b, fa = lo, hi, flo
        for _ in range(100):
            mid = 0.5 * (a + b)
            fm = d1(mid)
            if (fm > 0) == (fa > 0):
                a, fa = mid, fm
            else:
                b = mid
            if abs(b - a) < 1e-3 + 1e-3 * min(abs(a), abs(b)):
                break
        r = 0.5 * (a + b)
    # gradientFit :262-322.  GSL's Newton iterate fails on a zero derivative (GSL_EZERODIV) or a non-finite value at the
    # new point (GSL_EBADFUNC); the reference reads the solver's root only after a successful iterate (:288-290), so the
    # last accepted root stands.
    f, df = d1(r), d2(r)
    for _ in range(100):
        if df == 0.0 or not (math.isfinite(f) and math.isfinite(df)):
            break
        rn = r - f / df
        if not math.isfinite(rn) or rn <= 0:
            break
        try:
            fn, dfn = d1(rn), d2(rn)
        except (ValueError, ZeroDivisionError, OverflowError):
            break
        if not (math.isfinite(fn) and math.isfinite(dfn)):
            break
        done = abs(rn - r) < 1e-4 * abs(rn) or rn == r
        r, f, df = rn, fn, dfn
        if done or r > len(kfreq):
            break
    return popt(r), r


# ------------------------------------------------------------ DP wrappers
class DPConfig:
    """QuaffDPConfig defaults, src/qmodel.h:303-335 (kmerThreshold 20 for align/train: t/quaff.cpp:128)."""

    def __init__(self, local=True, sparse=True, kmer_len=6, kmer_threshold=20, band=64, max_size=0):
        self.local, self.sparse, self.kmer_len = local, sparse, kmer_len
        self.kmer_threshold, self.band, self.max_size = kmer_threshold, band, max_size


def envelope(xtok, ytok, cfg, cell_size=24):
    """QuaffDPConfig::makeEnvelope, src/qmodel.cpp:1049-1056 -> sorted diagonal list."""
    diags = np.empty(len(xtok) + len(ytok) - 1, np.int32)
    n = lib().qo_envelope(
        xtok.ctypes.data_as(C.c_void_p), len(xtok), ytok.ctypes.data_as(C.c_void_p), len(ytok),
        int(cfg.sparse), cfg.kmer_len, cfg.band, cfg.kmer_threshold,
        C.c_uint64(cell_size), C.c_uint64(cfg.max_size), diags.ctypes.data_as(C.c_void_p))
    return diags[:n].copy()


def diag_histogram(xtok, ytok, k):
    h = np.empty(len(xtok) + len(ytok) - 1, np.uint32)
    lib().qo_diag_histogram(xtok.ctypes.data_as(C.c_void_p), len(xtok), ytok.ctypes.data_as(C.c_void_p), len(ytok), k,
                            h.ctypes.data_as(C.c_void_p))
    return h


def envelope_cells(diags, xlen, ylen):
    return int(lib().qo_envelope_cells(diags.ctypes.data_as(C.c_void_p), len(diags), xlen, ylen))


class ReadCtx:
    """Per-read arrays of QuaffDPMatrix, src/qmodel.cpp:1308-1324."""

    def __init__(self, fs, sc):
        self.fs = fs
        self.tok = tokens(fs.seq)
        self.qual = quals(fs.qual) if (fs.has_qual() and len(fs.qual)) else None
        self.mk = kmers(self.tok, sc.match_len)
        self.gk = kmers(self.tok, sc.gap_len)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def viterbi(xtok, rc, sc, diags, local=True, want_tb=True, dump=False):
    """QuaffViterbiMatrix (+alignment), src/qmodel.cpp:1512-1646.
    Returns dict(result, xStart, xEnd, ops) ; ops is a str of M/I/D or None."""
    xlen, ylen = len(xtok), len(rc.tok)
    ops = C.create_string_buffer(xlen + ylen + 2)
    xs, xe, nops = C.c_int(0), C.c_int(0), C.c_int(-1)
    mdump = np.empty((len(diags), ylen + 1, 3)) if dump else None
    res = lib().qo_viterbi(
        xlen, ylen, sc.Km, sc.Kg, int(local), _vp(xtok), _vp(rc.tok), _vp(rc.qual), _vp(rc.mk), _vp(rc.gk),
        _vp(sc.ins), _vp(sc.mat), _vp(sc.trans), _vp(diags), len(diags),
        int(want_tb), C.byref(xs), C.byref(xe), ops, xlen + ylen + 2, C.byref(nops), _vp(mdump))
    out = {"result": res, "xStart": None, "xEnd": None, "ops": None}
    if nops.value >= 0:
        out.update(xStart=xs.value, xEnd=xe.value, ops=ops.raw[: nops.value].decode())
    elif nops.value < -1:
        raise RuntimeError("oracle traceback failed (%d)" % nops.value)
    if dump:
        out["matrix"] = mdump
    return out


def rescore_path(xtok, rc, sc, x_start, ops, local=True):
    """Log-probability of one alignment path in the Viterbi recurrence's own association (qo_rescore_path)."""
    lib().qo_rescore_path.restype = C.c_double
    return lib().qo_rescore_path(len(xtok), len(rc.tok), sc.Km, sc.Kg, int(local), _vp(xtok), _vp(rc.tok), _vp(rc.qual),
                                 _vp(rc.mk), _vp(rc.gk), _vp(sc.ins), _vp(sc.mat), _vp(sc.trans), int(x_start),
                                 ops.encode(), len(ops))


def counts_size(Km, Kg):
    return (4 + 4 * Km) * NQUAL + 4 * Kg + 4


def forward_backward(xtok, rc, sc, diags, local=True, want_back=True):
    """QuaffForwardMatrix / QuaffBackwardMatrix, src/qmodel.cpp:1343-1510.
    Returns (forward result, backward result or None, flattened QuaffCounts or None)."""
    xlen, ylen = len(xtok), len(rc.tok)
    cnt = np.zeros(counts_size(sc.Km, sc.Kg)) if want_back else None
    bres = C.c_double(float("nan"))
    f = lib().qo_forward_backward(
        xlen, ylen, sc.Km, sc.Kg, int(local), _vp(xtok), _vp(rc.tok), _vp(rc.qual), _vp(rc.mk), _vp(rc.gk),
        _vp(sc.ins), _vp(sc.mat), _vp(sc.trans), _vp(diags), len(diags),
        int(want_back), _vp(cnt), C.byref(bres))
    return f, (bres.value if want_back else None), cnt


def lse(a, b):
    return lib().qo_lse(a, b)


# ------------------------------------------------------- per-read tasks
def cigar(ops):
    """Alignment::cigarString, src/qmodel.cpp:625-653: letter BEFORE count."""
    out, last, n = "", None, 0
    for c in ops:
        if c == last:
            n += 1
        else:
            if n:
                out += last + str(n)
            last, n = c, 1
    if n:
        out += last + str(n)
    return out


def align_read(refs, read, sc, null, cfg, print_all=False):
    """QuaffAlignmentTask::run, src/qmodel.cpp:2764-2778.  refs: list of FastSeq (originals
    then revcomps).  Returns the kept alignments, best first: dicts with ref index, raw and
    adjusted score, xStart/xEnd, ops."""
    rc = ReadCtx(read, sc)
    nll = null.loglike(read)
    kept = []
    for nx, x in enumerate(refs):
        xt = tokens(x.seq)
        d = envelope(xt, rc.tok, cfg, 24)
        v = viterbi(xt, rc, sc, d, cfg.local)
        if v["result"] > NEG_INF:
            v.update(ref=nx, raw=v["result"], score=v["result"] - nll, ndiag=len(d),
                     cells=envelope_cells(d, len(xt), len(rc.tok)))
            # multiset insert at upper bound of equal scores, then keep first (:2773-2775)
            pos = len(kept)
            for a, k in enumerate(kept):
                if v["score"] > k["score"]:
                    pos = a
                    break
            kept.insert(pos, v)
            if not print_all:
                kept = kept[:1]
    return kept


def gapped_rows(x, read, al):
    """Gapped rows built by the traceback, src/qmodel.cpp:1577-1645."""
    xr, yr, qr = [], [], []
    i, j = al["xStart"] - 1, 0
    for c in al["ops"]:
        if c == "M":
            xr.append(x.seq[i]); yr.append(read.seq[j]); qr.append(read.qual[j] if read.has_qual() else "")
            i += 1; j += 1
        elif c == "I":
            xr.append("-"); yr.append(read.seq[j]); qr.append(read.qual[j] if read.has_qual() else "")
            j += 1
        else:
            xr.append(x.seq[i]); yr.append("-"); qr.append("~" if read.has_qual() else "")
            i += 1
    return "".join(xr), "".join(yr), "".join(qr)


def stockholm(x, read, al, local=True):
    """Alignment::writeStockholm, src/qmodel.cpp:553-606 for the 2-row Ref/Read alignment."""
    xrow, yrow, qrow = gapped_rows(x, read, al)
    names = ["Ref", "Read"]
    data = [xrow, yrow]
    cons = "".join("-" if (a in "-." or b in "-.") else (a.upper() if a.upper() == b.upper() else ":")
                   for a, b in zip(xrow, yrow))
    names.insert(1, "#=GC id")
    data.insert(1, cons)
    if read.has_qual():
        names.append("#=GR Read QS")
        data.append(qrow)
    nw = max(len(n) for n in names)
    dw = max(nw, 79 - nw)
    out = ["# STOCKHOLM 1.0", "#=GF Score " + fmt(al["score"])]
    xc = ("substr(%s,%d..%d)" % (x.name, al["xStart"], al["xEnd"])) if local else x.name
    out.append("#=GS CC Ref " + xc)
    out.append("#=GS CC Read " + read.name)
    ncol = len(xrow)
    for col in range(0, ncol, dw):
        if col > 0:
            out.append("")
        for n, d in zip(names, data):
            out.append(n.ljust(nw) + " " + d[col: col + dw])
    out.append("//")
    return "\n".join(out) + "\n"


def sam_line(x, read, al):
    """Alignment::writeSam, src/qmodel.cpp:608-616 with the revcomp path (:655-660,
    fastseq.cpp:218-230): a reverse-strand ref (x.source.rev) prints the reverse-complemented
    alignment, whose POS is computed from the GAPPED row length."""
    ops = al["ops"]
    xsrc = compose((x.name, al["xStart"], al["xEnd"], False), x.source)
    ysrc = compose((read.name, 1, len(read.seq), False), read.source)
    if xsrc[3]:
        ncol = len(ops)
        # FastSeq::revcomp on the gapped row: source (name,1,ncol,rev) composed with the row's own source
        xsrc = compose((None, 1, ncol, True), xsrc)
        ysrc = compose((None, 1, ncol, True), ysrc)
        ops = ops[::-1]
    flag = 16 if ysrc[3] else 0
    return "%s\t%d\t%s\t%d\t0\t%s\t*\t0\t0\t*\t*\tAS:i:%d\n" % (
        ysrc[0], flag, xsrc[0], xsrc[1], cigar(ops), int(round_half_away(al["score"])))


def round_half_away(x):
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


def count_read(refs, read, sc, null, cfg, sort_order=None, use_null=True, skip_pathless=False, details=None):
    """QuaffCountingTask::run, src/qmodel.cpp:2238-2271.  Returns (yCounts as flattened
    QuaffParamCounts-style QuaffCounts sum, yLogLike, new sort order).  The returned counts are
    still in QuaffCounts layout (m2m,m2i,m2d,m2e,...); param_counts() converts.
    skip_pathless: under -force (use_null False) the running yLogLike starts at -inf, so a reference without any path (Forward
    = -inf) passes the `>= yLogLike - 20` test (:2252), gets a Backward pass whose counts are NaN, and 0 x NaN (:2259-2261) makes
    the read's -- and the E-step's -- totals NaN in the reference.  True leaves such references out (what the library does).
    details: a dict that receives "forward" (the per-reference Forward log-likelihoods) and "counted" (the references that got a
    Backward pass), for tests that pin WHICH pairs carry weight."""
    rc = ReadCtx(read, sc)
    ynull = null.loglike(read) if use_null else NEG_INF
    ylog = ynull
    order = list(range(len(refs))) if sort_order is None else list(sort_order)
    xyll = [NEG_INF] * len(refs)
    xyc = [None] * len(refs)
    for nx in order:
        xt = tokens(refs[nx].seq)
        d = envelope(xt, rc.tok, cfg, 48)
        f, _, _ = forward_backward(xt, rc, sc, d, cfg.local, want_back=False)
        xyll[nx] = f
        if f >= ylog - 20 and not (skip_pathless and f == NEG_INF):
            _, _, cnt = forward_backward(xt, rc, sc, d, cfg.local, want_back=True)
            xyc[nx] = cnt
        ylog = lse(ylog, f)
    if details is not None:
        details["forward"] = list(xyll)
        details["counted"] = [nx for nx in range(len(refs)) if xyc[nx] is not None]
    tot = np.zeros(counts_size(sc.Km, sc.Kg))
    for nx in range(len(refs)):
        # exp(xyLogLike - yLogLike), src/qmodel.cpp:2259: NaN for -inf - -inf (a read without any path under -force), as there
        w = 0.0 if (skip_pathless and xyll[nx] == NEG_INF) else math.exp(xyll[nx] - ylog) if not (math.isinf(xyll[nx]) and math.isinf(ylog)) else float("nan")
        if xyc[nx] is not None:
            tot += w * param_counts(xyc[nx], sc.Km, sc.Kg)
    asc = sorted(range(len(refs)), key=lambda a: xyll[a])
    new_order = []
    for nx in reversed(asc):
        if xyll[nx] < ylog - 20:
            break
        new_order.append(nx)
    return tot, ylog, new_order


def param_counts(cnt, Km, Kg):
    """QuaffParamCounts(const QuaffCounts&), src/qmodel.cpp:407-417, flattened as
    ins[4][94] | mat[4][Km][94] | beginInsertNo[Kg] beginInsertYes[Kg] beginDeleteNo[Kg] beginDeleteYes[Kg]
    | extendInsertNo extendInsertYes extendDeleteNo extendDeleteYes."""
    ne = (4 + 4 * Km) * NQUAL
    out = np.zeros_like(cnt)
    out[:ne] = cnt[:ne]
    m2m, m2i, m2d, m2e = (cnt[ne + a * Kg: ne + (a + 1) * Kg] for a in range(4))
    d2d, d2m, i2i, i2m = cnt[ne + 4 * Kg: ne + 4 * Kg + 4]
    out[ne: ne + Kg] = m2m + m2d
    out[ne + Kg: ne + 2 * Kg] = m2i + m2e
    out[ne + 2 * Kg: ne + 3 * Kg] = m2m
    out[ne + 3 * Kg: ne + 4 * Kg] = m2d
    out[ne + 4 * Kg: ne + 4 * Kg + 4] = (i2m, i2i, d2m, d2d)
    return out


def join6(v):
    return ", ".join(fmt(x) for x in v)


def param_counts_json(pc, match_len, gap_len):
    """QuaffParamCounts::writeJson, src/qmodel.cpp:458-470 (+ QuaffEmitCounts::writeJson :341-362)."""
    Km, Kg = 4 ** match_len, 4 ** gap_len
    ne = (4 + 4 * Km) * NQUAL
    ins = pc[: 4 * NQUAL].reshape(4, NQUAL)
    mat = pc[4 * NQUAL: ne].reshape(4, Km, NQUAL)
    o = "{\n"
    if match_len != 1:
        o += '  "matchOrder": %d,\n' % match_len
    if gap_len != 0:
        o += '  "gapOrder": %d,\n' % gap_len
    o += '  "insert": {\n'
    for i in range(4):
        o += '    "%s": [ %s ]%s\n' % ("ACGT"[i], join6(ins[i]), " }," if i == 3 else ",")
    o += '  "match": {\n'
    for jp in range(0, Km, 4):
        o += '   "%s": {\n' % kmer_string(jp, match_len)[: match_len - 1]
        for i in range(4):
            o += '    "%s": {\n' % "ACGT"[i]
            for js in range(4):
                o += '      "%s": [ %s ]%s' % ("ACGT"[js], join6(mat[i, jp + js]), " }" if js == 3 else ",\n")
            o += " }" if i == 3 else ",\n"
        o += (" }" if jp == Km - 4 else ",") + "\n"
    o += ",\n"

    def kmers_obj(name, v):
        return '  "%s": {%s }' % (name, ",".join(' "%s": %s' % (kmer_string(g, gap_len), fmt(v[g])) for g in range(Kg)))
    tr = pc[ne:]
    o += kmers_obj("beginInsertNo", tr[0:Kg]) + ",\n"
    o += kmers_obj("beginInsertYes", tr[Kg:2 * Kg]) + ",\n"
    o += kmers_obj("beginDeleteNo", tr[2 * Kg:3 * Kg]) + ",\n"
    o += kmers_obj("beginDeleteYes", tr[3 * Kg:4 * Kg]) + ",\n"
    e = tr[4 * Kg:]
    o += '  "extendInsertNo": %s,\n  "extendInsertYes": %s,\n  "extendDeleteNo": %s,\n  "extendDeleteYes": %s }' % tuple(fmt(x) for x in e)
    return o


# ------------------------------------------------------------------- EM (quaff train): prior, M-step, convergence
# Counts are the flattened QuaffParamCounts of param_counts():
#   ins[4][94] | mat[4][Km][94] | beginInsertNo[Kg] beginInsertYes[Kg] beginDeleteNo[Kg] beginDeleteYes[Kg]
#   | extendInsertNo extendInsertYes extendDeleteNo extendDeleteYes
# GSL (un-vendored, un-pinned: doc/manual.tex:78) supplies gsl_ran_beta_pdf / gsl_ran_dirichlet_pdf /
# gsl_ran_negative_binomial_pdf to these functions; their published formulas are restated with libm lgamma.
class CountsView:
    """Named views into a flattened QuaffParamCounts vector (src/qmodel.h:205-233)."""

    def __init__(self, pc, match_len, gap_len):
        self.match_len, self.gap_len = match_len, gap_len
        self.Km, self.Kg = 4 ** match_len, 4 ** gap_len
        ne = (4 + 4 * self.Km) * NQUAL
        assert len(pc) == ne + 4 * self.Kg + 4
        self.v = pc
        self.ins = pc[:4 * NQUAL].reshape(4, NQUAL)
        self.mat = pc[4 * NQUAL:ne].reshape(4, self.Km, NQUAL)
        Kg = self.Kg
        self.beginInsertNo, self.beginInsertYes = pc[ne:ne + Kg], pc[ne + Kg:ne + 2 * Kg]
        self.beginDeleteNo, self.beginDeleteYes = pc[ne + 2 * Kg:ne + 3 * Kg], pc[ne + 3 * Kg:ne + 4 * Kg]
        self.ext = pc[ne + 4 * Kg:]          # extendInsertNo, extendInsertYes, extendDeleteNo, extendDeleteYes


def negbinom_pdf(k, p, r):
    """gsl_ran_negative_binomial_pdf, as published (see qo_log_negbinom in quaff_oracle.c)."""
    return math.exp(math.lgamma(k + r) - math.lgamma(r) - math.lgamma(k + 1.0) + r * math.log(p) + k * math.log1p(-p))


def init_counts(match_len, gap_len, no_begin, yes_extend, match_ident, other, null=None):
    """QuaffParamCounts::initCounts, src/qmodel.cpp:431-456; the auto-prior is initCounts(9, 9, 5, 1, &nullModel)
    (t/quaff.cpp:490-512).  NB the identity test `i == j` (:446, :448) compares the reference base i with the FULL context
    k-mer index j = jPrefix + jSuffix, not with the read base jSuffix: for match contexts longer than one base only the
    contexts whose prefix is all-A ever get matchIdentCount."""
    Km, Kg = 4 ** match_len, 4 ** gap_len
    pc = np.zeros(counts_size(Km, Kg))
    cv = CountsView(pc, match_len, gap_len)
    for j in range(4):
        for k in range(NQUAL):
            if null is not None:
                cv.ins[j, k] = other * null.null[j][0] * 4 * negbinom_pdf(k, null.null[j][1], null.null[j][2])
            else:
                cv.ins[j, k] = other / NQUAL
    for i in range(4):
        for jp in range(0, Km, 4):
            for js in range(4):
                j = jp + js
                for k in range(NQUAL):
                    if null is not None:
                        w = match_ident if i == j else other * null.null[js][0] * 4 / (1 - null.null[i][0])
                        cv.mat[i, j, k] = w * negbinom_pdf(k, null.null[js][1], null.null[js][2])
                    else:
                        cv.mat[i, j, k] = (match_ident if i == j else other) / NQUAL
    cv.beginInsertNo[:] = no_begin
    cv.beginInsertYes[:] = other
    cv.beginDeleteNo[:] = no_begin
    cv.beginDeleteYes[:] = other
    cv.ext[:] = (other, yes_extend, other, yes_extend)
    return pc


def log_beta_pdf(prob, yes_count, no_count):
    """logBetaPdf, src/qmodel.cpp:35-37: log(gsl_ran_beta_pdf(prob, yes + 1, no + 1)).  GSL: 0 outside [0, 1]; at the ends
    Gamma(a+b)/(Gamma(a)Gamma(b)) x^(a-1) (1-x)^(b-1) (0 when a, b > 1); inside exp(lgamma terms + (a-1) log x + (b-1) log1p(-x))."""
    a, b = yes_count + 1.0, no_count + 1.0
    if prob < 0 or prob > 1:
        return NEG_INF
    g = math.lgamma(a + b) - math.lgamma(a) - math.lgamma(b)
    if prob == 0.0 or prob == 1.0:
        pdf = 0.0 if (a > 1.0 and b > 1.0) else math.exp(g) * prob ** (a - 1) * (1 - prob) ** (b - 1)
    else:
        pdf = math.exp(g + math.log(prob) * (a - 1) + math.log1p(-prob) * (b - 1))
    return math.log(pdf) if pdf > 0 else NEG_INF


def log_dirichlet_pdf(alpha, theta):
    """log(gsl_ran_dirichlet_pdf(K, alpha, theta)) = log(exp(sum (alpha_i - 1) log theta_i + lgamma(sum alpha) - sum lgamma(alpha_i)))."""
    lp = sum((a - 1.0) * math.log(t) for a, t in zip(alpha, theta)) + math.lgamma(sum(alpha)) - sum(math.lgamma(a) for a in alpha)
    pdf = math.exp(lp)
    return math.log(pdf) if pdf > 0 else NEG_INF


def log_qual_prob(q, r, kfreq):
    """SymQualDist::logQualProb(kFreq), src/qmodel.cpp:83-85 -> logNegativeBinomial(kFreq, p, n), src/negbinom.cpp:34-39."""
    lp = 0.0
    for k in range(len(kfreq)):
        lp += kfreq[k] * lib().qo_log_negbinom(k, q, r)
    return lp


def log_prior(pc, p):
    """QuaffParamCounts::logPrior, src/qmodel.cpp:1681-1710 (pc = the pseudocounts, p = Params)."""
    cv = CountsView(pc, p.match_len, p.gap_len)
    lp = 0.0
    for j in range(p.Kg):
        lp += log_beta_pdf(p.beginInsert[j], cv.beginInsertYes[j], cv.beginInsertNo[j])
        lp += log_beta_pdf(p.beginDelete[j], cv.beginDeleteYes[j], cv.beginDeleteNo[j])
    lp += log_beta_pdf(p.extendInsert, cv.ext[1], cv.ext[0])
    lp += log_beta_pdf(p.extendDelete, cv.ext[3], cv.ext[2])
    alpha, theta = [0.0] * 4, [0.0] * 4
    for i in range(4):
        lp += log_qual_prob(p.insert[i][1], p.insert[i][2], cv.ins[i])
        theta[i] = p.insert[i][0]
        alpha[i] = 1.0 + float(np.sum(cv.ins[i]))     # accumulate(..., 1.)
    lp += log_dirichlet_pdf(alpha, theta)
    for i in range(4):
        for jp in range(0, p.Km, 4):
            for js in range(4):
                j = jp + js
                lp += log_qual_prob(p.match[i, j][1], p.match[i, j][2], cv.mat[i, j])
                theta[js] = p.match[i, j][0]
                alpha[js] = 1.0 + float(np.sum(cv.mat[i, j]))
            lp += log_dirichlet_pdf(alpha, theta)
    return lp


def m_step(pc, match_len, gap_len):
    """QuaffParamCounts::fit, src/qmodel.cpp:1733-1768."""
    cv = CountsView(pc, match_len, gap_len)
    p = Params(match_len, gap_len)
    for j in range(p.Kg):
        p.beginDelete[j] = 1. / (1. + cv.beginDeleteNo[j] / cv.beginDeleteYes[j])
        p.beginInsert[j] = 1. / (1. + cv.beginInsertNo[j] / cv.beginInsertYes[j])
    p.extendDelete = 1. / (1. + cv.ext[2] / cv.ext[3])
    p.extendInsert = 1. / (1. + cv.ext[0] / cv.ext[1])
    ins_freq = [float(np.sum(cv.ins[i])) for i in range(4)]
    for i in range(4):
        q, r = fit_negbinom(np.array(cv.ins[i], float))
        p.insert[i] = (ins_freq[i] / sum(ins_freq), q, r)
    for i in range(4):
        for jp in range(0, p.Km, 4):
            f = [float(np.sum(cv.mat[i, jp + js])) for js in range(4)]
            for js in range(4):
                q, r = fit_negbinom(np.array(cv.mat[i, jp + js], float))
                p.match[i, jp + js] = (f[js] / sum(f), q, r)
    return p


def fit_ref_seqs(p, refs):
    """QuaffParams::fitRefSeqs, src/qmodel.cpp:284-294, with totalLen initialised (the reference reads it uninitialised;
    DESIGN.md 7)."""
    cnt = np.zeros(4)
    for fs in refs:
        t = tokens(fs.seq)
        for b in range(4):
            cnt[b] += np.count_nonzero(t == b)
    p.refBase = list(cnt / cnt.sum())


def params_json(p):
    """QuaffParams::writeJson, src/qmodel.cpp:187-228."""
    o = "{\n"
    if p.match_len != 1:
        o += '  "matchOrder": %d,\n' % p.match_len
    if p.gap_len != 0:
        o += '  "gapOrder": %d,\n' % p.gap_len
    o += '  "refBase": {' + ",".join(' "%s": %s' % ("ACGT"[i], fmt(p.refBase[i])) for i in range(4)) + " },\n"
    for name in ("beginInsert", "beginDelete"):
        v = getattr(p, name)
        o += '  "%s": {%s },\n' % (name, ",".join(' "%s": %s' % (kmer_string(g, p.gap_len), fmt(v[g])) for g in range(p.Kg)))
    o += '  "extendInsert": %s,\n  "extendDelete": %s,\n' % (fmt(p.extendInsert), fmt(p.extendDelete))
    o += '  "insert": {\n'
    for i in range(4):
        o += '    "%s": %s%s\n' % ("ACGT"[i], sqd_json(*p.insert[i]), " }," if i == 3 else ",")
    o += '  "match": {\n'
    for jp in range(0, p.Km, 4):
        o += '   "%s": {\n' % kmer_string(jp, p.match_len)[: p.match_len - 1]
        for i in range(4):
            o += '    "%s": {\n' % "ACGT"[i]
            for js in range(4):
                o += '      "%s": %s%s' % ("ACGT"[js], sqd_json(*p.match[i, jp + js]), " }" if js == 3 else ",\n")
            o += " }" if i == 3 else ",\n"
        o += " }" if jp == p.Km - 4 else ",\n"
    return o + " }"


def counts_from_json(text):
    """QuaffParamCounts::readJson, src/qmodel.cpp:491-536 (+ QuaffEmitCounts::readJson :364-405): the flattened vector and
    (match_len, gap_len)."""
    jm = gason_loads(text) if isinstance(text, str) else text
    ml, gl = int(jm.get("matchOrder", 1)), int(jm.get("gapOrder", 0))
    Km, Kg = 4 ** ml, 4 ** gl
    pc = np.zeros(counts_size(Km, Kg))
    cv = CountsView(pc, ml, gl)
    for i in range(4):
        cv.ins[i] = jm["insert"]["ACGT"[i]]
    for jp in range(0, Km, 4):
        pref = kmer_string(jp, ml)[: ml - 1]
        for i in range(4):
            for js in range(4):
                cv.mat[i, jp + js] = jm["match"][pref]["ACGT"[i]]["ACGT"[js]]
    for g in range(Kg):
        ks = kmer_string(g, gl)
        cv.beginInsertNo[g], cv.beginInsertYes[g] = jm["beginInsertNo"][ks], jm["beginInsertYes"][ks]
        cv.beginDeleteNo[g], cv.beginDeleteYes[g] = jm["beginDeleteNo"][ks], jm["beginDeleteYes"][ks]
    cv.ext[:] = (jm["extendInsertNo"], jm["extendInsertYes"], jm["extendDeleteNo"], jm["extendDeleteYes"])
    return pc, ml, gl


def em_converged(it, llp, prev, min_inc):
    """The stopping test of QuaffTrainer::fitUnlimited, src/qmodel.cpp:2204-2206 (it = 0-based iteration)."""
    return it > 0 and llp < prev + abs(prev) * min_inc


def train(refs, reads, null, prior, seed, cfg, max_iter=100, min_inc=0.01, use_null=True, e_step=None):
    """QuaffTrainer::fitUnlimited, src/qmodel.cpp:2186-2231: E-step (one QuaffCountingTask per read, counts summed in read
    order :2416-2422), log-prior, the convergence rule (:2204-2206: stop when iter > 0 and logLike + logPrior < previous +
    |previous| * minFractionalLoglikeIncrement -- before the M-step of that iteration), M-step on counts + pseudocounts,
    fitRefSeqs.  Returns (final Params, [per-iteration dict(loglike, logprior, counts, counts_with_prior, params)]); an iteration that
    stopped has no counts_with_prior / params.  `e_step(params, sort_orders) -> (counts, loglike, new sort_orders)` replaces
    the oracle's own E-step (tests drive the device's through the same loop)."""
    p = seed
    orders = [None] * len(reads)
    prev = NEG_INF
    log = []
    for it in range(max_iter):
        if e_step is not None:
            counts, ll, orders = e_step(p, orders)
        else:
            sc = Scores(p)
            counts = np.zeros(counts_size(p.Km, p.Kg))
            ll = 0.0
            for n, rd in enumerate(reads):
                c, yl, orders[n] = count_read(refs, rd, sc, null, cfg, orders[n], use_null)
                counts += c
                ll += yl
        lp = log_prior(prior, p)
        rec = {"loglike": ll, "logprior": lp, "counts": counts}
        log.append(rec)
        if em_converged(it, ll + lp, prev, min_inc):
            break
        prev = ll + lp
        rec["counts_with_prior"] = counts + 1. * prior      # addWeighted(pseudocounts, 1.), :1656-1673
        p = m_step(rec["counts_with_prior"], p.match_len, p.gap_len)
        fit_ref_seqs(p, refs)
        rec["params"] = p
    return p, log


# ------------------------------------------------------------------- overlap
class OverlapScores:
    """QuaffOverlapScores, src/qoverlap.cpp:9-75, for one strand flag."""

    def __init__(self, params, sc, y_complemented):
        self.Km, self.Kg, self.match_len, self.gap_len = sc.Km, sc.Kg, sc.match_len, sc.gap_len
        self.y_complemented = bool(y_complemented)
        self.mmi = np.zeros((sc.Km, sc.Km, NQ1, NQ1))
        self.gap = np.zeros(3 * sc.Kg * sc.Kg + 6)
        self.ins = sc.ins
        lib().qo_overlap_scores(sc.Km, sc.Kg, (C.c_double * 4)(*params.refBase), (C.c_double * sc.Kg)(*params.beginInsert),
                                (C.c_double * sc.Kg)(*params.beginDelete), C.c_double(params.extendInsert),
                                C.c_double(params.extendDelete), _vp(sc.ins), _vp(sc.mat), int(self.y_complemented),
                                _vp(self.mmi), _vp(self.gap))


def overlap_pair(x, y, y_complemented, osc, sc, null, cfg):
    """QuaffOverlapTask::run + QuaffOverlapViterbiMatrix (+alignment, scoreAdjustedAlignment),
    src/qoverlap.cpp:77-302,457-464.  x, y: FastSeq (y is the sequence as stored, i.e. already reverse-complemented
    when y_complemented).  Returns None if no finite path, else dict(result, score, xStart..yEnd, ops (raw states))."""
    xt, yt = tokens(x.seq), tokens(y.seq)
    xmk, xgk = kmers(xt, sc.match_len), kmers(xt, sc.gap_len)
    xq = quals(x.qual) if (x.has_qual() and len(x.qual)) else None
    yq = quals(y.qual) if (y.has_qual() and len(y.qual)) else None
    if y_complemented:                     # :91-98: arrays of revcomp(y), reversed
        yrt = tokens(revcomp_str(y.seq))
        ytok_eff = yrt[::-1].copy()
        ymk = kmers(yrt, sc.match_len)[::-1].copy()
        ygk = kmers(yrt, sc.gap_len)[::-1].copy()
    else:
        ytok_eff, ymk, ygk = yt, kmers(yt, sc.match_len), kmers(yt, sc.gap_len)
    d = envelope(xt, yt, cfg, 24)
    xlen, ylen = len(xt), len(yt)
    ops = C.create_string_buffer(xlen + ylen + 2)
    xs, xe, ys, ye, nops = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(-1)
    lib().qo_overlap_viterbi.restype = C.c_double
    end = lib().qo_overlap_viterbi(xlen, ylen, sc.Km, sc.Kg, _vp(xmk), _vp(xgk), _vp(xq), _vp(ymk), _vp(ygk), _vp(yq),
                                   _vp(osc.mmi), _vp(osc.gap), _vp(d), len(d), 1, C.byref(xs), C.byref(xe), C.byref(ys),
                                   C.byref(ye), ops, xlen + ylen + 2, C.byref(nops))
    # xInsertScore / yInsertScore, :105-113 (sequential sums)
    xins = 0.0
    for i in range(xlen):
        xins += sc.ins[xt[i], xq[i] if xq is not None else NQUAL]
    yins = 0.0
    for j in range(ylen):
        yins += sc.ins[ytok_eff[j], yq[j] if yq is not None else NQUAL]
    result = end + xins + yins
    out = {"end": end, "result": result, "ndiag": len(d), "cells": envelope_cells(d, xlen, ylen)}
    if not (end > NEG_INF):
        return None
    if nops.value < 0:
        raise RuntimeError("oracle overlap traceback failed (%d)" % nops.value)
    ynull = null.loglike(y.revcomp() if y_complemented else y)   # :295
    score = result - null.loglike(x)
    score -= ynull
    out.update(score=score, xStart=xs.value, xEnd=xe.value, yStart=ys.value, yEnd=ye.value, ops=ops.raw[:nops.value].decode())
    return out


def overlap_rows(x, y, al):
    """Gapped rows with the indel squashing of src/qoverlap.cpp:231-267: a run of adjacent insertions and deletions
    (between two matches) is re-paired into aligned columns first, then surplus deletions, then surplus insertions."""
    xr, yr, xq, yq = [], [], [], []
    i, j = al["xStart"] - 1, al["yStart"] - 1
    hx, hy = x.has_qual(), y.has_qual()
    ops = al["ops"]
    k = 0
    while k < len(ops):
        if ops[k] == "M":
            xr.append(x.seq[i]); yr.append(y.seq[j]); xq.append(x.qual[i] if hx else ""); yq.append(y.qual[j] if hy else "")
            i += 1; j += 1; k += 1
            continue
        e = k
        while e < len(ops) and ops[e] != "M":
            e += 1
        nins = ops[k:e].count("I")
        ndel = (e - k) - nins
        shared = min(nins, ndel)
        xs, ys = x.seq[i:i + ndel], y.seq[j:j + nins]
        xqs, yqs = (x.qual[i:i + ndel] if hx else ""), (y.qual[j:j + nins] if hy else "")
        # aligned pairs, then surplus deletions (x only), then surplus insertions (y only)
        xr.append(xs[:shared] + xs[shared:] + "-" * (nins - shared))
        yr.append(ys[:shared] + "-" * (ndel - shared) + ys[shared:])
        if hx:
            xq.append(xqs[:shared] + xqs[shared:] + "~" * (nins - shared))
        if hy:
            yq.append(yqs[:shared] + "~" * (ndel - shared) + yqs[shared:])
        i += ndel; j += nins; k = e
    return "".join(xr), "".join(yr), "".join(xq), "".join(yq)


def overlap_stockholm(x, y, al):
    """Alignment::writeStockholm (src/qmodel.cpp:553-606) for the read_x / read_y alignment of src/qoverlap.cpp:271-289."""
    xrow, yrow, xq, yq = overlap_rows(x, y, al)
    cons = "".join("-" if (a in "-." or b in "-.") else (a.upper() if a.upper() == b.upper() else ":") for a, b in zip(xrow, yrow))
    names, data = ["read_x"], [xrow]
    if x.has_qual():
        names.append("#=GR read_x QS"); data.append(xq)
    idx1 = len(names)
    names.append("read_y"); data.append(yrow)
    if y.has_qual():
        names.append("#=GR read_y QS"); data.append(yq)
    names.insert(idx1, "#=GC id"); data.insert(idx1, cons)
    if x.has_qual():
        names[0], names[1] = names[1], names[0]
        data[0], data[1] = data[1], data[0]
    nw = max(len(n) for n in names)
    dw = max(nw, 79 - nw)
    out = ["# STOCKHOLM 1.0", "#=GF Score " + fmt(al["score"]),
           "#=GS CC read_x substr(%s,%d..%d)" % (x.name, al["xStart"], al["xEnd"]),
           "#=GS CC read_y substr(%s,%d..%d)" % (y.name, al["yStart"], al["yEnd"])]
    for col in range(0, len(xrow), dw):
        if col > 0:
            out.append("")
        for n, d in zip(names, data):
            out.append(n.ljust(nw) + " " + d[col: col + dw])
    out.append("//")
    return "\n".join(out) + "\n"


def overlap_task_pairs(n_originals, n_total):
    """Pair enumeration of QuaffOverlapScheduler (src/qoverlap.cpp:475-480,528-547): (nx, ny, yComplemented) with
    0 <= nx <= N-2, nx < ny < n_total, yComplemented = ny >= N."""
    return [(nx, ny, ny >= n_originals) for nx in range(max(n_originals - 1, 0)) for ny in range(nx + 1, n_total)]
