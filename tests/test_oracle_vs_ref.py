"""Pins oracle/quaff_oracle.c against the REAL reference sources that build here without GSL
(oracle/_ref/libquaffref.so = diagenv.cpp, fastseq.cpp, logsumexp.cpp, gason.cpp compiled from
/root/reference by oracle/Makefile).  Skipped where the reference is absent (GPU box)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import rand_seq, mutate, rand_qual

REFSO = os.path.join(O.HERE, "_ref", "libquaffref.so")
if not os.path.exists(REFSO) and os.path.isdir("/root/reference/src"):
    import subprocess
    subprocess.call(["make", "-C", O.HERE, "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
pytestmark = pytest.mark.skipif(not os.path.exists(REFSO), reason="oracle/_ref not built (no /root/reference)")


@pytest.fixture(scope="module")
def ref():
    L = C.CDLL(REFSO)
    L.ref_lse.restype = C.c_double
    L.ref_lse.argtypes = [C.c_double, C.c_double]
    L.ref_lse_unary.restype = C.c_double
    L.ref_lse_unary.argtypes = [C.c_double]
    L.ref_json_number.restype = C.c_double
    L.ref_envelope_cells.restype = C.c_uint64
    return L


def ref_env(L, x, y, sparse, k, band, thr, cell, maxsize):
    d = np.empty(len(x) + len(y) + 2, np.int32)
    st = C.c_uint64(0)
    n = L.ref_envelope(x.encode(), y.encode(), int(sparse), k, band, thr, C.c_uint64(cell), C.c_uint64(maxsize),
                       d.ctypes.data_as(C.c_void_p), C.byref(st))
    return d[:n].copy()


def test_kmers_quals_revcomp(ref):
    rng = np.random.default_rng(1)
    for n in (1, 2, 7, 50, 300):
        s = rand_seq(rng, n)
        for k in (0, 1, 2, 3, 4, 6):
            out = np.empty(n, np.uint64)
            ref.ref_kmers(s.encode(), k, out.ctypes.data_as(C.c_void_p))
            assert np.array_equal(out.astype(np.uint32), O.kmers(O.tokens(s), k))
        q = "".join(chr(int(c)) for c in rng.integers(33, 127, n))
        out = np.empty(n, np.uint32)
        ref.ref_quals(s.encode(), q.encode(), out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(out.astype(np.uint8), O.quals(q))
        buf = C.create_string_buffer(n + 1)
        ref.ref_revcomp(s.encode(), buf)
        assert buf.value.decode() == O.revcomp_str(s)
    # skewed composition: the padding token is the most frequent one, first max on ties
    for s in ("TTTTACG", "ACGTACGT", "GGCC", "CCGG"):
        out = np.empty(len(s), np.uint64)
        ref.ref_kmers(s.encode(), 3, out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(out.astype(np.uint32), O.kmers(O.tokens(s), 3))


def test_log_sum_exp_bits(ref):
    rng = np.random.default_rng(2)
    vals = list(rng.uniform(-50, 5, 4000)) + [0.0, -0.0, float("-inf"), 1e-5, 9.99995, 10.0, 10.00001]
    for a, b in zip(vals, reversed(vals)):
        r, o = ref.ref_lse(a, b), O.lse(a, b)
        assert (r == o) or (np.isnan(r) and np.isnan(o)), (a, b, r, o)
    assert ref.ref_lse(float("-inf"), float("-inf")) == O.lse(float("-inf"), float("-inf")) == float("-inf")
    # the table itself
    tab = np.ctypeslib.as_array(O.lib().qo_lse_table(), (100001,))
    for n in (0, 1, 17, 50000, 99999):
        assert tab[n] == ref.ref_lse_unary(n * .0001) or n == 99999


def test_gason_numbers(ref):
    rng = np.random.default_rng(3)
    texts = ["0.0277689", "0.641208", "94.1305", "1e-5", "-3.25E+2", "12345678901234567890", "0.1", "1.7976931348623157e308",
             "6.02e23", "4.9e-324", "0.30000000000000004"]
    texts += ["%.*g" % (int(rng.integers(1, 18)), x) for x in rng.uniform(-1e3, 1e3, 300)]
    texts += ["%.6g" % x for x in 10 ** rng.uniform(-8, 8, 300)]
    for t in texts:
        assert ref.ref_json_number(t.encode()) == O.gason_number(t), t


@pytest.mark.parametrize("seed", range(6))
def test_envelope_threshold_mode(ref, seed):
    rng = np.random.default_rng(100 + seed)
    x = rand_seq(rng, int(rng.integers(300, 1500)))
    s = int(rng.integers(0, len(x) - 250))
    y = mutate(rng, x[s: s + int(rng.integers(120, 250))])
    if seed & 1:
        y = O.revcomp_str(y)
    xt, yt = O.tokens(x), O.tokens(y)
    for k, band, thr in ((6, 64, 20), (6, 64, 14), (5, 16, 3), (6, 8, 0), (7, 32, 5), (6, 65, 2), (6, 0, 4)):
        cfg = O.DPConfig(kmer_len=k, kmer_threshold=thr, band=band)
        mine = O.envelope(xt, yt, cfg)
        theirs = ref_env(ref, x, y, True, k, band, thr, 24, 0)
        assert np.array_equal(mine, theirs), (k, band, thr)
        assert O.envelope_cells(mine, len(x), len(y)) == ref.ref_envelope_cells(
            x.encode(), y.encode(), theirs.ctypes.data_as(C.c_void_p), len(theirs))


def test_envelope_two_bands_and_repeats(ref):
    rng = np.random.default_rng(7)
    unit = rand_seq(rng, 150)
    x = rand_seq(rng, 200) + unit + rand_seq(rng, 300) + unit + rand_seq(rng, 100)   # repeat -> two seeded bands
    y = mutate(rng, unit, sub=0.02, ins=0.01, dele=0.01)
    xt, yt = O.tokens(x), O.tokens(y)
    cfg = O.DPConfig(kmer_len=6, kmer_threshold=10, band=32)
    mine = O.envelope(xt, yt, cfg)
    theirs = ref_env(ref, x, y, True, 6, 32, 10, 24, 0)
    assert np.array_equal(mine, theirs)
    runs = 1 + int(np.count_nonzero(np.diff(mine) > 1))
    assert runs >= 3          # {0} plus two bands


def test_envelope_short_and_full(ref):
    rng = np.random.default_rng(8)
    x = rand_seq(rng, 400)
    for ylen in (6, 20, 51, 52, 53):               # 2*(k+thr) = 52 is the sparse/full switch (diagenv.cpp:23-29)
        y = x[100: 100 + ylen]
        mine = O.envelope(O.tokens(x), O.tokens(y), O.DPConfig())
        theirs = ref_env(ref, x, y, True, 6, 64, 20, 24, 0)
        assert np.array_equal(mine, theirs), ylen
    mine = O.envelope(O.tokens(x), O.tokens(x[:80]), O.DPConfig(sparse=False))
    assert np.array_equal(mine, ref_env(ref, x, x[:80], False, 6, 64, 20, 24, 0))
    assert len(mine) == 400 + 80 - 1


@pytest.mark.parametrize("seed", range(4))
def test_envelope_memory_mode(ref, seed):
    rng = np.random.default_rng(200 + seed)
    x = rand_seq(rng, 900)
    y = mutate(rng, x[200:600])
    xt, yt = O.tokens(x), O.tokens(y)
    diag = min(len(x), len(y))
    for cell in (24, 48):
        for nd_budget in (0, 1, 2, 30, 67, 68, 69, 75, 140, 400, 5000):
            maxsize = nd_budget * diag * cell
            cfg = O.DPConfig(kmer_threshold=-1, max_size=maxsize, band=64)
            mine = O.envelope(xt, yt, cfg, cell)
            theirs = ref_env(ref, x, y, True, 6, 64, -1, cell, maxsize)
            assert np.array_equal(mine, theirs), (cell, nd_budget)


def test_c8f30_testdiagenv_case(ref, golden):
    """The reference's own unit test input (Makefile:134-135): c8f30 vs itself, k=6 n=14 band 64."""
    fs = O.read_fastx(os.path.join(golden, "c8f30.fastq.gz"))[0]
    t = O.tokens(fs.seq)
    mine = O.envelope(t, t, O.DPConfig(kmer_len=6, kmer_threshold=14, band=64))
    theirs = ref_env(ref, fs.seq, fs.seq, True, 6, 64, 14, 24, 0)
    assert np.array_equal(mine, theirs)
    assert 0 in mine and len(mine) >= 65
