"""CPU-side checks of the product library: it loads, exports every symbol include/quaff_hip.h declares,
its host-only entry points (score tables from quaff's params JSON, CIGAR text, the synthetic generator)
agree with the oracle, and it refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    import quaff_amd
    from quaff_amd import api as A
    if not os.path.exists(A.library_path()):
        quaff_amd.build_library()
    return A


def test_exports_every_declared_symbol(api):
    hdr = open(os.path.join(ROOT, "include", "quaff_hip.h")).read()
    body = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(qf_[a-z0-9_]+)\s*\(", body)))
    assert len(declared) >= 15
    L = api.load_library()
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(api.EXPORTS) == declared


def test_no_cpu_fallback(api):
    import quaff_amd as Q
    L = api.load_library()
    h = C.c_void_p()
    rc = L.qf_ctx_create(0, C.byref(h))
    if rc == 0:          # a GPU is present: nothing to check here
        L.qf_ctx_destroy(h)
        pytest.skip("GPU present")
    assert rc == -1 and b"no HIP device" in L.qf_last_error(None)
    with pytest.raises(Q.QuaffHipError):
        Q.Context(0)


@pytest.mark.parametrize("name", ["defaultparams.json", "testquaffparams.json"])
def test_score_tables_bit_equal_to_oracle(api, golden, name):
    text = open(os.path.join(golden, name)).read()
    ml, gl, ins, mat, trans = api.scores_from_json(text)
    sc = O.Scores(O.Params.from_json(text))
    assert (ml, gl) == (sc.match_len, sc.gap_len)
    assert np.array_equal(ins, sc.ins) and np.array_equal(mat, sc.mat) and np.array_equal(trans, sc.trans)


def test_builtin_defaults_are_the_reference_defaults(api, golden):
    a = api.scores_from_json(None)
    b = api.scores_from_json(open(os.path.join(golden, "defaultparams.json")).read())
    assert all(np.array_equal(x, y) for x, y in zip(a[2:], b[2:]))


def test_params_errors(api):
    import quaff_amd as Q
    with pytest.raises(Q.QuaffHipError) as e:
        api.scores_from_json('{ "beginInsert": { "": 0.1 } }')
    assert e.value.code == -3 and "Missing parameter" in str(e.value)
    with pytest.raises(Q.QuaffHipError):
        api.scores_from_json("{ not json")


def test_cigar_string_letter_first(api):
    runs = np.array([(14 << 2) | 0, (1 << 2) | 1, (11 << 2) | 0, (2 << 2) | 2], np.uint32)
    buf = C.create_string_buffer(64)
    n = api.load_library().qf_cigar_string(runs.ctypes.data, len(runs), buf, 64)
    assert buf.value == b"M14I1M11D2" and n == 10        # Alignment::cigarString, src/qmodel.cpp:625-653
    assert O.cigar("M" * 14 + "I" + "M" * 11 + "DD") == "M14I1M11D2"


def test_synthetic_generator(api):
    ref = api.synth_ref(1, 5000)
    assert ref == api.synth_ref(1, 5000) and ref != api.synth_ref(2, 5000)
    assert set(ref) <= set(b"ACGT") and min(ref.count(c) for c in b"ACGT") > 1000
    seq, qual, off = api.synth_reads(2, ref, 200, 400)
    seq2, qual2, off2 = api.synth_reads(2, ref, 200, 400)
    assert seq == seq2 and qual == qual2 and np.array_equal(off, off2)
    lens = np.diff(off).astype(int)
    assert len(seq) == len(qual) == int(off[-1]) and 330 < lens.mean() < 430 and lens.min() > 300
    assert min(qual) >= 33 + 5 and max(qual) <= 33 + 25
    # even reads come from the forward strand, odd ones from the reverse: check by seeding
    sc_cfg = O.DPConfig()
    xt, xr = O.tokens(ref.decode()), O.tokens(O.revcomp_str(ref.decode()))
    for n in (0, 1, 2, 3):
        y = O.tokens(seq[int(off[n]):int(off[n + 1])].decode())
        nf, nr = len(O.envelope(xt, y, sc_cfg)), len(O.envelope(xr, y, sc_cfg))
        assert (nf > 60 and nr == 1) if n % 2 == 0 else (nr > 60 and nf == 1)
    assert api.revcomp(b"AACGT") == b"ACGTT"


def test_packed_lse_table_rebuilds_the_table(api):
    """The exact log-sum-exp table in the form the overlap fills keep in LDS (csrc/qf_device.hpp: one fifth-degree piece per
    256 entries + bit-packed corrections): evaluated with exactly rounded fused multiply-adds (rational arithmetic here, the
    device's v_fma_f64 there), every sampled entry comes back bit for bit, reached both as entry n of its piece and as the
    piece's entry n + 1.  (On a GPU the library checks all 100 001 entries itself before it uses the packed form.)"""
    import struct
    from fractions import Fraction
    L = api.load_library()
    L.qf_debug_pack_lse_table.restype = C.c_uint32
    size = L.qf_debug_pack_lse_table(None, 0)
    assert 100_000 < size <= 160 * 1024 and size % 16 == 0            # fits one CU's LDS
    raw = (C.c_uint8 * size)()
    assert L.qf_debug_pack_lse_table(raw, size) == size
    raw = bytes(raw)
    tabp, cnt = C.POINTER(C.c_double)(), C.c_int()
    L.qf_get_lse_table(None, C.byref(tabp), C.byref(cnt))
    tab = np.ctypeslib.as_array(tabp, (cnt.value,))
    assert np.array_equal(tab, np.ctypeslib.as_array(O.lib().qo_lse_table(), (100001,)))
    P = 391                                                            # layout: csrc/qf_device.hpp kLsePack*
    coef = np.concatenate([np.frombuffer(raw[k * P * 16:(k + 1) * P * 16], dtype="<f8").reshape(P, 2) for k in range(3)], axis=1)
    meta = np.frombuffer(raw[3 * P * 16:3 * P * 16 + P * 8], dtype="<u4").reshape(P, 2)
    words = np.frombuffer(raw[(3 * P * 16 + P * 8 + 15) & ~15:], dtype="<u4")
    assert meta[:, 1].min() >= 1 and meta[:, 1].max() <= 16

    def entry(p, t):
        v = float(coef[p, 5])
        for i in range(4, -1, -1):
            v = float(Fraction(v) * (t - 128) + Fraction(float(coef[p, i])))          # one rounding: a fused multiply-add
        w, o = int(meta[p, 1]), int(meta[p, 0]) + t * int(meta[p, 1])
        f = ((int(words[o >> 5]) | (int(words[(o >> 5) + 1]) << 32)) >> (o & 31)) & ((1 << w) - 1)
        if f >= 1 << (w - 1):
            f -= 1 << w
        return struct.unpack("<d", struct.pack("<q", struct.unpack("<q", struct.pack("<d", v))[0] + f))[0]

    for n in list(range(0, 100000, 53)) + [255, 256, 511, 512, 99839, 99840, 99999]:
        assert entry(n >> 8, n & 255) == tab[n] and entry(n >> 8, (n & 255) + 1) == tab[n + 1], n


def test_exact_fixed_point_helpers(api):
    """qf_exact_from_double / qf_exact_add / qf_exact_to_double (the host side of qf_count_result.counts_exact): 64.64 two's
    complement words whose sums do not depend on the order; values that are not finite (a read without any path has log-likelihood
    -inf) become a marker that absorbs every sum and converts to -inf."""
    from fractions import Fraction
    rng = np.random.default_rng(7)
    v = np.concatenate([rng.random(400) * rng.choice([1e-12, 1e-3, 1.0, 1e6], 400), -rng.random(100) * 1e4, [0.0, 1.0, -1.0, 2.0 ** -64, 2.0 ** 40]])
    fx = api.exact_from_double(v)
    assert fx.shape == (len(v), 2) and fx.dtype == np.uint64
    # each value: truncated toward zero at 2^-64, exactly
    for x, (lo, hi) in zip(v, fx):
        w = (int(hi) << 64) | int(lo)
        if w >> 127:
            w -= 1 << 128
        mag = abs(Fraction(float(x)))
        assert abs(w) == int(mag * (1 << 64)) and (w < 0) == (x < 0 and abs(w) > 0)
    assert np.array_equal(api.exact_to_double(api.exact_from_double([1.5, -2.25, 3.0 ** -10])), np.array([1.5, -2.25, float(Fraction(int(Fraction(3.0 ** -10) * (1 << 64)), 1 << 64))]))
    # the sum is the same in any order and equals the exact rational sum
    def total(order):
        acc = np.zeros((1, 2), np.uint64)
        for k in order:
            acc = api.exact_add(acc, fx[k:k + 1])
        return acc
    a, b = total(range(len(v))), total(rng.permutation(len(v)))
    assert np.array_equal(a, b)
    exact = sum(((int(hi) << 64 | int(lo)) - ((1 << 128) if int(hi) >> 63 else 0)) for lo, hi in fx)
    got = (int(a[0, 1]) << 64 | int(a[0, 0])) - ((1 << 128) if int(a[0, 1]) >> 63 else 0)
    assert got == exact and abs(api.exact_to_double(a)[0] - float(Fraction(exact, 1 << 64))) <= 1e-9 * abs(float(Fraction(exact, 1 << 64)))
    # vectors add element-wise
    two = api.exact_add(fx, fx)
    np.testing.assert_allclose(api.exact_to_double(two), 2 * api.exact_to_double(fx), rtol=1e-15, atol=0)
    # not finite -> marker; it absorbs
    m = api.exact_from_double([float("-inf"), float("nan"), 1e19, 5.0])
    assert [int(x) for x in m[0]] == [0, 1 << 63] and np.array_equal(m[0], m[1]) and np.array_equal(m[0], m[2])
    out = api.exact_to_double(api.exact_add(m, api.exact_from_double([1.0, 2.0, 3.0, 4.0])))
    assert np.all(np.isneginf(out[:3])) and out[3] == 9.0
