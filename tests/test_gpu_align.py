"""GPU parity tests (run with -m gpu): the HIP path, called through the C ABI (quaff_amd.api is a
ctypes shim), against the CPU oracle on the same seeded inputs.  Bit-exact for everything Viterbi:
envelope diagonals, cell counts, scores (==, not approx), chosen reference, coordinates, CIGAR."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import rand_seq, make_reads, mutate, rand_qual

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
NULL_JSON = open(os.path.join(GOLDEN, "testquaffnullparams.json")).read()
DEFAULT_JSON = open(os.path.join(GOLDEN, "defaultparams.json")).read()


@pytest.fixture(scope="module")
def ctx():
    import quaff_amd as Q
    c = Q.Context(0)
    c.set_params_json(None)
    c.set_null_json(NULL_JSON)
    yield c
    c.close()


def oracle_model(params_json=DEFAULT_JSON, null_json=NULL_JSON):
    return O.Scores(O.Params.from_json(params_json)), O.NullParams.from_json(null_json)


def both_strands(ref):
    x = O.FastSeq("ref", ref)
    return [x, x.revcomp()]


def run_both(ctx, refs, reads, cfg_kw, sc, null, flags=0, quals=True):
    import quaff_amd as Q
    ctx.set_refs([x.seq for x in refs])
    ctx.upload_reads([r.seq for r in reads], [r.qual for r in reads] if quals else None)
    gcfg = Q.DPConfig(**cfg_kw)
    res = ctx.align_resident(gcfg, flags)
    ocfg = O.DPConfig(local=cfg_kw.get("local", True), sparse=cfg_kw.get("sparse", True), kmer_len=cfg_kw.get("kmer_len", 6),
                      kmer_threshold=cfg_kw.get("kmer_threshold", 20), band=cfg_kw.get("band_size", 64),
                      max_size=cfg_kw.get("max_size", 0))
    return res, ocfg


def check_against_oracle(ctx, refs, reads, cfg_kw, sc, null, quals=True, print_all=False):
    res, ocfg = run_both(ctx, refs, reads, cfg_kw, sc, null, flags=1 if print_all else 0, quals=quals)
    by_read = {}
    for a in res["alignments"]:
        by_read.setdefault(a["read"], []).append(a)
    total = 0
    for r, read in enumerate(reads):
        rd = read if quals else O.FastSeq(read.name, read.seq, "")
        rc = O.ReadCtx(rd, sc)
        nll = null.loglike(rd)
        assert res["null_loglike"][r] == nll, ("null loglike", r)
        for x, ref in enumerate(refs):
            xt = O.tokens(ref.seq)
            d = O.envelope(xt, rc.tok, ocfg, 24)
            assert res["n_diagonals"][r, x] == len(d), ("ndiag", r, x)
            cells = O.envelope_cells(d, len(xt), len(rc.tok))
            assert res["cells"][r, x] == cells, ("cells", r, x)
            total += cells
            v = O.viterbi(xt, rc, sc, d, ocfg.local, want_tb=False)
            assert res["viterbi"][r, x] == v["result"], ("viterbi", r, x, res["viterbi"][r, x], v["result"])
        kept = O.align_read(refs, rd, sc, null, ocfg, print_all=print_all)
        got = by_read.get(r, [])
        assert len(got) == len(kept), ("n alignments", r)
        for g, k in zip(got, kept):
            assert g["ref"] == k["ref"] and g["viterbi"] == k["raw"] and g["score"] == k["score"], (r, g, k)
            assert (g["xStart"], g["xEnd"]) == (k["xStart"], k["xEnd"]), (r, g["xStart"], g["xEnd"], k["xStart"], k["xEnd"])
            assert g["ops"] == k["ops"], ("cigar", r, g["cigar"], O.cigar(k["ops"]))
    assert res["total_cells"] == total
    return res


def test_score_tables_match_oracle(ctx):
    ml, gl, ins, mat, trans = ctx.get_scores()
    sc, _ = oracle_model()
    assert (ml, gl) == (1, 0)
    assert np.array_equal(ins, sc.ins) and np.array_equal(mat, sc.mat) and np.array_equal(trans, sc.trans)
    tab = np.ctypeslib.as_array(O.lib().qo_lse_table(), (100001,))
    assert np.array_equal(ctx.lse_table(), tab)


def test_envelopes_match_oracle(ctx):
    import quaff_amd as Q
    rng = np.random.default_rng(11)
    ref = rand_seq(rng, 1500)
    refs = both_strands(ref)
    reads = make_reads(rng, ref, 6, 260)
    reads.append(O.FastSeq("short", ref[40:80], rand_qual(rng, 40)))       # < 2(k+thr): full envelope
    ctx.set_refs([x.seq for x in refs])
    ctx.upload_reads([r.seq for r in reads], [r.qual for r in reads])
    diag = lambda xl, yl: min(xl, yl)
    cfgs = [dict(), dict(kmer_threshold=14), dict(kmer_len=5, band_size=16, kmer_threshold=3), dict(band_size=9, kmer_threshold=0),
            dict(sparse=False)]
    for nb in (0, 1, 30, 67, 68, 70, 140, 3000):
        cfgs.append(dict(kmer_threshold=-1, max_size=nb * 260 * 24))
    for kw in cfgs:
        ocfg = O.DPConfig(sparse=kw.get("sparse", True), kmer_len=kw.get("kmer_len", 6), kmer_threshold=kw.get("kmer_threshold", 20),
                          band=kw.get("band_size", 64), max_size=kw.get("max_size", 0))
        for r, read in enumerate(reads):
            for x, rf in enumerate(refs):
                want = O.envelope(O.tokens(rf.seq), O.tokens(read.seq), ocfg, 24)
                for dbg in (0, 1):      # 0: wavefront-per-pair seeding kernel, 1: workgroup-per-pair kernel
                    ctx.set_debug_flags(dbg)
                    got = ctx.envelope(r, x, Q.DPConfig(**kw))
                    ctx.set_debug_flags(0)
                    assert np.array_equal(got, want), (kw, dbg, r, x, len(got), len(want))


def test_long_kmers_sorted_index(ctx):
    """-kmatch 9 .. 32 (reference range 5..32, src/qmodel.cpp:773-779): sorted k-mer index + binary search."""
    import quaff_amd as Q
    rng = np.random.default_rng(12)
    ref = rand_seq(rng, 2500)
    sc, null = oracle_model()
    reads = make_reads(rng, ref, 6, 400, sub=0.02, ins=0.01, dele=0.01)
    refs = both_strands(ref)
    ctx.set_refs([x.seq for x in refs])
    ctx.upload_reads([r.seq for r in reads], [r.qual for r in reads])
    for k, thr in ((9, 10), (12, 6), (16, 4), (17, 3), (32, 1)):
        ocfg = O.DPConfig(kmer_len=k, kmer_threshold=thr)
        for r, read in enumerate(reads):
            for x, rf in enumerate(refs):
                got = ctx.envelope(r, x, Q.DPConfig(kmer_len=k, kmer_threshold=thr))
                want = O.envelope(O.tokens(rf.seq), O.tokens(read.seq), ocfg, 24)
                assert np.array_equal(got, want), (k, thr, r, x, len(got), len(want))
    check_against_oracle(ctx, refs, reads, dict(kmer_len=12, kmer_threshold=6), sc, null)
    check_against_oracle(ctx, refs, reads, dict(kmer_len=20, kmer_threshold=-1, max_size=80 * 400 * 24), sc, null)


def test_align_small_both_strands(ctx):
    rng = np.random.default_rng(21)
    ref = rand_seq(rng, 2000)
    sc, null = oracle_model()
    reads = make_reads(rng, ref, 24, 300)
    res = check_against_oracle(ctx, both_strands(ref), reads, dict(), sc, null)
    assert len(res["alignments"]) == 24
    # the true strand wins and its band is a multi-diagonal one
    assert all(a["ref"] == (a["read"] & 1) for a in res["alignments"])
    assert res["n_diagonals"].max() > 65


def test_align_ragged_lengths_and_bands(ctx):
    """ragged read lengths (1 .. 900), narrow and odd bands, low thresholds -> several (G,B) classes."""
    rng = np.random.default_rng(22)
    ref = rand_seq(rng, 3000)
    sc, null = oracle_model()
    reads = []
    for n, L in enumerate((1, 2, 5, 7, 51, 52, 60, 97, 128, 333, 600, 900)):
        s = int(rng.integers(0, len(ref) - L))
        seq = mutate(rng, ref[s:s + L]) or "A"
        reads.append(O.FastSeq("r%d" % n, seq, rand_qual(rng, len(seq))))
    refs = [O.FastSeq("ref", ref)]
    for kw in (dict(band_size=20, kmer_threshold=8), dict(band_size=33, kmer_threshold=5), dict(band_size=100, kmer_threshold=10),
               dict(band_size=4, kmer_threshold=2, kmer_len=5)):
        # reads shorter than 2(k+thr) fall back to the full envelope (3000+L-1 diagonals -> row-space kernel)
        check_against_oracle(ctx, refs, reads, kw, sc, null)


def test_align_global_noquals_printall(ctx):
    rng = np.random.default_rng(23)
    ref = rand_seq(rng, 700)
    sc, null = oracle_model()
    # global: reads spanning (almost) the whole reference
    reads = [O.FastSeq("g%d" % n, s, rand_qual(rng, len(s))) for n, s in enumerate(mutate(rng, ref) for _ in range(6))]
    check_against_oracle(ctx, [O.FastSeq("ref", ref)], reads, dict(local=False), sc, null)
    reads = make_reads(rng, ref, 8, 200)
    check_against_oracle(ctx, both_strands(ref), reads, dict(), sc, null, quals=False)
    check_against_oracle(ctx, both_strands(ref), reads, dict(kmer_threshold=4), sc, null, print_all=True)


def test_align_repeat_two_bands(ctx):
    rng = np.random.default_rng(24)
    unit = rand_seq(rng, 220)
    ref = rand_seq(rng, 300) + unit + rand_seq(rng, 500) + unit + rand_seq(rng, 200)
    sc, null = oracle_model()
    reads = [O.FastSeq("rep%d" % n, s, rand_qual(rng, len(s))) for n, s in
             enumerate(mutate(rng, unit, sub=0.02, ins=0.01, dele=0.01) for _ in range(5))]
    res = check_against_oracle(ctx, [O.FastSeq("ref", ref)], reads, dict(kmer_threshold=12, band_size=32), sc, null)
    assert res["n_units"] >= 3 * 5 - 2


def test_align_full_dp_and_wide_bands(ctx):
    """-kmatchoff (full DP), the short-read full-envelope fallback, and bands wider than 1024 diagonals: the row-space
    kernel.  Includes a reference longer than one 512-row stripe and reads longer than one stripe's column span."""
    rng = np.random.default_rng(26)
    sc, null = oracle_model()
    ref = rand_seq(rng, 1400)
    reads = make_reads(rng, ref, 6, 180)
    reads.append(O.FastSeq("tiny", ref[700:730], rand_qual(rng, 30)))
    reads.append(O.FastSeq("one", "G", "5"))
    check_against_oracle(ctx, both_strands(ref), reads, dict(sparse=False), sc, null)
    check_against_oracle(ctx, [O.FastSeq("ref", ref)], reads[:4], dict(sparse=False, local=False), sc, null)
    # sparse mode, reads shorter than 2(k+threshold) -> initFull against a 1400 bp reference (1400+L-1 diagonals)
    shorts = [O.FastSeq("s%d" % n, ref[s:s + L], rand_qual(rng, L)) for n, (s, L) in enumerate(((5, 40), (900, 51), (1349, 51)))]
    check_against_oracle(ctx, both_strands(ref), shorts, dict(), sc, null)
    # a very wide seeded band (band size 1500) on a longer read
    long_read = make_reads(rng, ref, 1, 1100)
    check_against_oracle(ctx, [O.FastSeq("ref", ref)], long_read, dict(band_size=1500, kmer_threshold=10), sc, null)
    check_against_oracle(ctx, [O.FastSeq("ref", ref)], long_read, dict(sparse=False), sc, null)


def synth_params_json(rng, match_len, gap_len):
    """random but valid parameter set of a given order (train -order k writes matchOrder k+1, gapOrder k)."""
    def sqd(p):
        return '{ "p": %.6g, "q": %.6g, "r": %.6g, "m": 1, "sd": 1 }' % (p, rng.uniform(.55, .95), rng.uniform(8, 90))
    Km, Kg = 4 ** match_len, 4 ** gap_len
    o = '{\n  "matchOrder": %d,\n  "gapOrder": %d,\n  "refBase": { "A": 0.25, "C": 0.25, "G": 0.25, "T": 0.25 },\n' % (match_len, gap_len)
    for name in ("beginInsert", "beginDelete"):
        o += '  "%s": {%s },\n' % (name, ",".join(' "%s": %.6g' % (O.kmer_string(g, gap_len), rng.uniform(.01, .08)) for g in range(Kg)))
    o += '  "extendInsert": 0.55,\n  "extendDelete": 0.6,\n  "insert": {\n'
    o += ",\n".join('    "%s": %s' % ("ACGT"[i], sqd(.25)) for i in range(4)) + " },\n"
    o += '  "match": {\n'
    blocks = []
    for jp in range(0, Km, 4):
        rows = []
        for i in range(4):
            ps = rng.dirichlet([1, 1, 1, 1]) * 0.1
            ps[i] += 0.9
            rows.append('    "%s": {\n%s }' % ("ACGT"[i], ",\n".join('      "%s": %s' % ("ACGT"[js], sqd(ps[js])) for js in range(4))))
        blocks.append('   "%s": {\n%s }' % (O.kmer_string(jp, match_len)[:match_len - 1], ",\n".join(rows)))
    return o + ",\n".join(blocks) + " } }\n"


def test_align_order1_contexts_lds_tables(ctx):
    """-order 1 parameters (match context 2, gap context 1): the largest emission tables that still live in LDS (51.7 KB per
    workgroup) together with per-step gap-context transitions; also full DP through the row-space kernel."""
    rng = np.random.default_rng(26)
    pj = synth_params_json(rng, 2, 1)
    sc = O.Scores(O.Params.from_json(pj))
    null = O.NullParams.from_json(NULL_JSON)
    ctx.set_params_json(pj)
    try:
        ref = rand_seq(rng, 1500)
        reads = make_reads(rng, ref, 8, 280)
        check_against_oracle(ctx, both_strands(ref), reads, dict(), sc, null)
        check_against_oracle(ctx, [O.FastSeq("ref", ref)], reads[:3], dict(sparse=False), sc, null)
    finally:
        ctx.set_params_json(None)


def test_align_order2_contexts(ctx):
    """-order 2 style parameters (match context 3, gap context 2): exercises context k-mers + GAPCTX kernels."""
    rng = np.random.default_rng(25)
    pj = synth_params_json(rng, 3, 2)
    sc = O.Scores(O.Params.from_json(pj))
    null = O.NullParams.from_json(NULL_JSON)
    ctx.set_params_json(pj)
    try:
        ml, gl, ins, mat, trans = ctx.get_scores()
        assert (ml, gl) == (3, 2) and np.array_equal(mat, sc.mat) and np.array_equal(trans, sc.trans)
        ref = rand_seq(rng, 1200)
        reads = make_reads(rng, ref, 10, 250)
        check_against_oracle(ctx, both_strands(ref), reads, dict(), sc, null)
    finally:
        ctx.set_params_json(None)


def test_c8f30_golden_through_gpu(ctx):
    """The reference's own integration golden (Makefile:149-150), with the HIP path doing the DP."""
    import quaff_amd as Q
    reads = O.read_fastx(os.path.join(GOLDEN, "c8f30.fastq.gz"))
    null = O.NullParams.fit(reads)
    ctx.set_null_json(null.to_json())
    try:
        ctx.set_refs([reads[0].seq])
        ctx.upload_reads([reads[0].seq], [reads[0].qual])
        res = ctx.align_resident(Q.DPConfig(kmer_threshold=-1, max_size=10 << 20))
        a = res["alignments"][0]
        assert res["n_diagonals"][0, 0] == 1 and a["cigar"] == "M6604"
        assert O.fmt(a["viterbi"]) == "-18710.3"
        # the null model round-trips through 6-s.f. JSON here, so compare the formatted score only
        al = dict(a, score=a["viterbi"] - null.loglike(reads[0]))
        assert O.stockholm(reads[0], reads[0], al) == open(os.path.join(GOLDEN, "c8f30-self-align.json")).read()
        res = ctx.align_resident(Q.DPConfig())           # default seeding: 65 diagonals, still M6604
        assert res["n_diagonals"][0, 0] == 65 and res["alignments"][0]["cigar"] == "M6604"
    finally:
        ctx.set_null_json(NULL_JSON)


def test_bad_symbol_and_errors(ctx):
    import quaff_amd as Q
    ctx.set_refs(["ACGTACGTACGTACGTACGTAAACCCGGGTTT" * 4])
    ctx.upload_reads(["ACGTNACGT" * 8])
    with pytest.raises(Q.QuaffHipError) as e:
        ctx.align_resident()
    assert e.value.code == -4
    with pytest.raises(Q.QuaffHipError):
        ctx.set_refs(["ACGTXX"])


def test_internal_chunking_matches_single_pass(ctx):
    """A batch whose traceback exceeds the device budget is processed in halves; results must not change."""
    import quaff_amd as Q
    rng = np.random.default_rng(29)
    ref = rand_seq(rng, 2500)
    reads = make_reads(rng, ref, 37, 300)
    refs = both_strands(ref)
    ctx.set_refs([x.seq for x in refs])
    ctx.upload_reads([r.seq for r in reads], [r.qual for r in reads])
    whole = ctx.align_resident(Q.DPConfig(), 1)
    try:
        ctx.set_memory_budget(whole["traceback_bytes"] // 7)
        parts = ctx.align_resident(Q.DPConfig(), 1)
    finally:
        ctx.set_memory_budget(0)
    for key in ("viterbi", "cells", "n_diagonals", "null_loglike"):
        assert np.array_equal(whole[key], parts[key]), key
    assert whole["total_cells"] == parts["total_cells"] and whole["traceback_bytes"] == parts["traceback_bytes"]
    assert len(whole["alignments"]) == len(parts["alignments"]) == 74
    for a, b in zip(whole["alignments"], parts["alignments"]):
        assert (a["read"], a["ref"], a["score"], a["xStart"], a["xEnd"], a["ops"]) == \
               (b["read"], b["ref"], b["score"], b["xStart"], b["xEnd"], b["ops"])
    # pipelined pieces (two host threads, two streams) with and without a tight budget
    try:
        ctx.set_pipeline_chunks(5)
        piped = ctx.align_resident(Q.DPConfig(), 1)
        ctx.set_memory_budget(whole["traceback_bytes"] // 3)
        piped2 = ctx.align_resident(Q.DPConfig(), 1)
    finally:
        ctx.set_memory_budget(0)
        ctx.set_pipeline_chunks(0)
    for other in (piped, piped2):
        for key in ("viterbi", "cells", "n_diagonals", "null_loglike"):
            assert np.array_equal(whole[key], other[key]), key
        assert whole["total_cells"] == other["total_cells"] and len(other["alignments"]) == 74
        for a, b in zip(whole["alignments"], other["alignments"]):
            assert (a["read"], a["ref"], a["score"], a["xStart"], a["xEnd"], a["ops"]) == \
                   (b["read"], b["ref"], b["score"], b["xStart"], b["xEnd"], b["ops"])
    # a budget below a single read's traceback is an error, not a silent truncation
    try:
        ctx.set_memory_budget(1024)
        with pytest.raises(Exception, match="over the memory budget"):
            ctx.align_resident(Q.DPConfig(), 0)
    finally:
        ctx.set_memory_budget(0)


def test_long_reference_global_seeding(ctx):
    """A reference too long for the LDS diagonal histogram (> ~190 k diagonals): seeding runs through global-memory
    workspaces; threshold mode with the bucket index (k = 6) and the sorted index (k = 11), and memory mode."""
    import quaff_amd as Q
    rng = np.random.default_rng(30)
    ref = rand_seq(rng, 260000)
    sc, null = oracle_model()
    reads = make_reads(rng, ref, 4, 450, sub=0.03, ins=0.02, dele=0.02)
    refs = both_strands(ref)
    res = check_against_oracle(ctx, refs, reads, dict(), sc, null)
    assert res["n_diagonals"].max() > 65
    check_against_oracle(ctx, refs, reads[:2], dict(kmer_len=11, kmer_threshold=6), sc, null)
    ctx.set_refs([x.seq for x in refs])
    ctx.upload_reads([r.seq for r in reads], [r.qual for r in reads])
    for nb in (1, 40, 200):
        kw = dict(kmer_threshold=-1, max_size=nb * 450 * 24)
        ocfg = O.DPConfig(kmer_threshold=-1, max_size=kw["max_size"])
        for r in (0, 3):
            for x, rf in enumerate(refs):
                got = ctx.envelope(r, x, Q.DPConfig(**kw))
                want = O.envelope(O.tokens(rf.seq), O.tokens(reads[r].seq), ocfg, 24)
                assert np.array_equal(got, want), (nb, r, x, len(got), len(want))


def test_ragged_batch_sorted_class_lists(ctx):
    """More than 64 reads of very different lengths: the class lists are sorted by read length before the fills (the bands
    a wavefront takes together run in lockstep); results must not care."""
    rng = np.random.default_rng(31)
    ref = rand_seq(rng, 2500)
    sc, null = oracle_model()
    reads = []
    for n in range(72):
        L = int(rng.integers(60, 700))
        s = int(rng.integers(0, len(ref) - L))
        src = ref[s:s + L]
        if n & 1:
            src = O.revcomp_str(src)
        seq = mutate(rng, src)
        reads.append(O.FastSeq("r%d" % n, seq, rand_qual(rng, len(seq))))
    res = check_against_oracle(ctx, both_strands(ref), reads, dict(), sc, null)
    assert len(res["alignments"]) == 72


def test_ragged_batch_many_classes_side_streams(ctx):
    """900 ragged reads x 2 strands whose bands span four fill classes with more than 64 units each (a 18..60-base deletion in
    the middle of two reads in three widens the seeded band from 65..80 to 85..125 diagonals; the wrong strand gives lone
    diagonals): every class list is
    sorted in place on the main stream while the other classes' fills run on side streams, which therefore have to wait
    for the sorts.  One piece and four pieces (two in flight), three rounds each, all against the oracle."""
    from concurrent.futures import ThreadPoolExecutor
    import quaff_amd as Q
    rng = np.random.default_rng(77)
    ref = rand_seq(rng, 4000)
    sc, null = oracle_model()
    reads = []
    for n in range(900):
        L = int(rng.integers(200, 900))
        s = int(rng.integers(0, len(ref) - L))
        src = ref[s:s + L]
        cut = (0, int(rng.integers(18, 30)), int(rng.integers(40, 60)))[n % 3]   # a deletion mid-read: two seed diagonals, one wider band
        if cut:
            src = src[:len(src) // 2] + src[len(src) // 2 + cut:]
        if n & 1:
            src = O.revcomp_str(src)
        seq = mutate(rng, src)
        reads.append(O.FastSeq("r%d" % n, seq, rand_qual(rng, len(seq))))
    refs = both_strands(ref)
    ocfg = O.DPConfig()
    O.lib()
    with ThreadPoolExecutor(8) as ex:
        want = list(ex.map(lambda rd: O.align_read(refs, rd, sc, null, ocfg), reads))
    ctx.set_refs([x.seq for x in refs])
    ctx.upload_reads([r.seq for r in reads], [r.qual for r in reads])
    try:
        for chunks in (1, 4):
            ctx.set_pipeline_chunks(chunks)
            for rnd in range(3):
                res = ctx.align_resident(Q.DPConfig(), 0)
                if chunks == 1 and rnd == 0:
                    big = [c for c in res["classes"] if c["units"] > 64]
                    assert len(big) >= 3, res["classes"]
                got = {a["read"]: a for a in res["alignments"]}
                assert len(got) == sum(1 for k in want if k)
                for r, kept in enumerate(want):
                    if not kept:
                        assert r not in got
                        continue
                    g, k = got[r], kept[0]
                    assert (g["ref"], g["viterbi"], g["score"], g["xStart"], g["xEnd"], g["ops"]) == \
                           (k["ref"], k["raw"], k["score"], k["xStart"], k["xEnd"], k["ops"]), (chunks, rnd, r)
    finally:
        ctx.set_pipeline_chunks(0)


def test_long_reads_wide_seed_counters(ctx):
    """Reads of 2 048+ bases switch the seeding to 32-bit coarse counters (a 32-diagonal bin could otherwise wrap 16 bits);
    a low-complexity pair puts ~95 000 matches into single bins."""
    rng = np.random.default_rng(33)
    ref = rand_seq(rng, 5000)
    sc, null = oracle_model()
    reads = make_reads(rng, ref, 3, 2600)
    check_against_oracle(ctx, both_strands(ref), reads, dict(), sc, null)
    lowc = "A" * 1500 + rand_seq(rng, 300) + "A" * 1500
    rd = O.FastSeq("polyA", "A" * 1400 + lowc[1500:1800] + "A" * 1400, rand_qual(rng, 3100))
    res = check_against_oracle(ctx, [O.FastSeq("lowc", lowc)], [rd], dict(), sc, null)
    assert res["n_diagonals"].max() > 3000


def test_diagonal_with_65536_matches_deep_counters(ctx):
    """An exact 65 550-base copy of the reference puts 65 541 10-mer matches on one diagonal: a 16-bit counter would wrap to
    5 (below the threshold, and the lowest level in memory mode); sequences this long get 32-bit counters."""
    rng = np.random.default_rng(35)
    ref = rand_seq(rng, 66000)
    sc, null = oracle_model()
    rd = O.FastSeq("copy", ref[200:200 + 65550], rand_qual(rng, 65550))
    res = check_against_oracle(ctx, both_strands(ref), [rd], dict(kmer_len=10, kmer_threshold=20, band_size=8), sc, null)
    assert res["n_diagonals"][0, 0] == 10 and res["alignments"][0]["cigar"] == "M65550"
    res = check_against_oracle(ctx, [O.FastSeq("ref", ref)], [rd], dict(kmer_len=10, kmer_threshold=-1, band_size=8, max_size=40 * 65550 * 24), sc, null)
    assert res["alignments"][0]["cigar"] == "M65550"


def test_many_bands_grow_unit_tables(ctx):
    """Low threshold + short k-mers + narrow bands: dozens of bands per pair, far more than the four per pair the unit table
    and the overflow list are provisioned for; the seeding is repeated with grown tables (found by a randomized soak)."""
    rng = np.random.default_rng(34)
    ref = rand_seq(rng, 2500)
    sc, null = oracle_model()
    reads = make_reads(rng, ref, 10, 350)
    res = check_against_oracle(ctx, both_strands(ref), reads, dict(kmer_len=4, kmer_threshold=3, band_size=6), sc, null)
    assert res["n_units"] > 4 * 20 + 1024


def test_score_threshold_filters_before_the_traceback(ctx):
    """qf_set_score_threshold = the printer's -threshold applied on the device: the surviving alignments are exactly the
    unfiltered ones with score >= threshold (best-per-read: the best is chosen first, then tested; -printall: each)."""
    import quaff_amd as Q
    rng = np.random.default_rng(36)
    ref = rand_seq(rng, 3000)
    reads = make_reads(rng, ref, 40, 300) + [O.FastSeq("junk%d" % k, rand_seq(rng, 250), rand_qual(rng, 250)) for k in range(8)]
    sc, null = oracle_model()
    key = lambda a: (a["read"], a["ref"], a["score"], a["xStart"], a["xEnd"], a["cigar"])
    try:
        for flags in (0, 1):
            ctx.set_score_threshold(float("-inf"))
            full, _ = run_both(ctx, both_strands(ref), reads, dict(), sc, null, flags=flags)
            scores = sorted(a["score"] for a in full["alignments"])
            thr = scores[len(scores) // 3]
            ctx.set_score_threshold(thr)
            cut, _ = run_both(ctx, both_strands(ref), reads, dict(), sc, null, flags=flags)
            want = [key(a) for a in full["alignments"] if a["score"] >= thr]
            assert 0 < len(want) < len(full["alignments"])
            assert [key(a) for a in cut["alignments"]] == want
            assert np.array_equal(cut["viterbi"], full["viterbi"]) and np.array_equal(cut["null_loglike"], full["null_loglike"])
        ctx.set_score_threshold(float("inf"))
        none, _ = run_both(ctx, both_strands(ref), reads, dict(), sc, null)
        assert none["alignments"] == []
    finally:
        ctx.set_score_threshold(float("-inf"))


def test_seeding_coarse_bin_widths_and_wide_counters(ctx):
    """The wavefront seeding picks coarse bins of 32 / 16 / 8 diagonals from the chance-match rate and 32-bit coarse counters
    when a bin could pass 65 535 matches.  One case per (width, wide) combination the other tests do not reach
    and a homopolymer pair that really puts more than 65 535 matches into one 32-diagonal bin."""
    sc, null = oracle_model()
    for seed, k, ref_len, read_len in ((61, 8, 6000, 2600),     # 32-diagonal bins, wide (reads >= 2040)
                                       (62, 7, 9000, 4300),     # 16-diagonal bins, wide (reads >= 4080)
                                       (63, 6, 10000, 8500)):   # 8-diagonal bins, wide (reads >= 8160)
        rng = np.random.default_rng(seed)
        ref = rand_seq(rng, ref_len)
        reads = make_reads(rng, ref, 2, read_len)
        res = check_against_oracle(ctx, both_strands(ref), reads, dict(kmer_len=k), sc, null)
        assert res["n_diagonals"].max() > 64
    # 32-diagonal bins with 32-bit counters really needed: 8-mers of a 2 400-base homopolymer, ~77 000 matches per bin
    rng = np.random.default_rng(64)
    lowc = rand_seq(rng, 150) + "A" * 2400 + rand_seq(rng, 150)
    rd = O.FastSeq("polyA", "A" * 2350 + rand_seq(rng, 40), rand_qual(rng, 2390))
    res = check_against_oracle(ctx, [O.FastSeq("lowc", lowc)], [rd], dict(kmer_len=8), sc, null)
    assert res["n_diagonals"].max() > 2400


def test_many_references_of_different_lengths(ctx):
    """Seven references (150 b ... 6 kb, one shorter than 2(k + n): full envelope against it) and their reverse complements:
    the best reference per read, ties to the earlier one, -printall order, per-reference k-mer indices."""
    rng = np.random.default_rng(37)
    sc, null = oracle_model()
    lens = [3000, 150, 6000, 800, 40, 2200, 1500]
    fwd = [O.FastSeq("ref%d" % k, rand_seq(rng, L)) for k, L in enumerate(lens)]
    fwd[5] = O.FastSeq("ref5", fwd[0].seq[500:2700])                      # shares 2.2 kb with ref0: ties and near-ties
    refs = fwd + [x.revcomp() for x in fwd]
    reads = []
    for k in range(36):
        x = fwd[int(rng.integers(0, len(fwd)))]
        L = int(rng.integers(30, min(400, len(x.seq)) + 1))
        s0 = int(rng.integers(0, len(x.seq) - L + 1))
        src = x.seq[s0:s0 + L]
        if k % 3 == 1:
            src = O.revcomp_str(src)
        seq = mutate(rng, src, sub=.04, ins=.02, dele=.02) or "A"
        reads.append(O.FastSeq("r%d" % k, seq, rand_qual(rng, len(seq))))
    res = check_against_oracle(ctx, refs, reads, dict(), sc, null)
    assert len({a["ref"] for a in res["alignments"]}) >= 6
    check_against_oracle(ctx, refs, reads[:12], dict(kmer_threshold=8, band_size=20), sc, null, print_all=True)


def test_secondary_buffers_that_do_not_fit_split_the_chunk(ctx):
    """Every per-chunk device buffer besides the traceback / Forward storage (unit tables, sort keys, alignment records, run lists
    ...) goes through the same release-split-retry path as the big one: a reserve that fails makes the chunk's caller cut it in two
    (qf_debug_fail_chunk_reserve injects the failure).  Results must not change -- for align, count and overlap, with the n-th
    reserve failing for a range of n, also under a memory budget that forces splits of its own and with three contexts at work."""
    import quaff_amd as Q
    from quaff_amd import api
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(606)
    ref = rand_seq(rng, 3000)
    refs = both_strands(ref)
    reads = make_reads(rng, ref, 400, 260)
    seqs, quals = [r.seq for r in reads], [r.qual for r in reads]
    others = [Q.Context(0), Q.Context(0)]
    ctxs = [ctx] + others
    try:
        for c in ctxs:
            c.set_params_json(None)
            c.set_null_json(NULL_JSON)
            c.set_refs([x.seq for x in refs])
            c.upload_reads(seqs, quals)
        cfg = Q.DPConfig()

        def align_key(res):
            return [(int(a["read"]), int(a["ref"]), float(a["viterbi"]), float(a["score"]), int(a["xStart"]), a["cigar"]) for a in res["alignments"]]

        base_al = ctx.align_resident(cfg)
        base_ct = ctx.count_resident(cfg)
        words = lambda r: np.concatenate([r["counts_exact"], r["loglike_exact"].reshape(1, 2)])
        fired = 0
        for nth in (0, 1, 3, 7, 12, 15, 17, 20):
            ctx.fail_chunk_reserve(nth)
            got = ctx.align_resident(cfg)
            fired += ctx.fail_chunk_reserve(-1) < 0
            assert align_key(got) == align_key(base_al), nth
            assert np.array_equal(got["viterbi"], base_al["viterbi"]), nth
        assert fired >= 6, fired                          # (the late ones too: alignment records, run lists)
        fired = 0
        for nth in (0, 2, 5, 9, 13, 16, 19):
            ctx.fail_chunk_reserve(nth)
            got = ctx.count_resident(cfg)
            fired += ctx.fail_chunk_reserve(-1) < 0
            assert np.array_equal(words(got), words(base_ct)), nth
        assert fired >= 5, fired
        # three contexts at once, a budget that forces splits of the big buffer too, failures injected while they run
        for c in ctxs:
            c.set_memory_budget(8 << 20)
        ctx.fail_chunk_reserve(5)
        with ThreadPoolExecutor(3) as ex:
            res = list(ex.map(lambda c: (c.align_resident(cfg), c.count_resident(cfg)), ctxs))
        assert ctx.fail_chunk_reserve(-1) < 0
        for al, ct in res:
            assert align_key(al) == align_key(base_al) and np.array_equal(words(ct), words(base_ct))
    finally:
        ctx.fail_chunk_reserve(-1)
        for c in ctxs:
            c.set_memory_budget(0)
        for c in others:
            c.close()
