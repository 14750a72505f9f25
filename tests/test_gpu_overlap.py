"""GPU parity for quaff overlap (read-vs-read Viterbi with log-sum-exp gap mixing): HIP path through the C ABI vs the
oracle, bit-exact (==) on result, adjusted score, end/start coordinates and the raw traceback state string, for both
strand flags."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import rand_seq, mutate, rand_qual
from tests.test_gpu_align import NULL_JSON, DEFAULT_JSON, synth_params_json

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import quaff_amd as Q
    c = Q.Context(0)
    c.set_params_json(None)
    c.set_null_json(NULL_JSON)
    yield c
    c.close()


def overlapping_reads(rng, genome_len, n, read_len):
    g = rand_seq(rng, genome_len)
    reads = []
    for k in range(n):
        s = int(rng.integers(0, genome_len - read_len))
        src = g[s:s + read_len]
        if k % 3 == 2:
            src = O.revcomp_str(src)
        seq = mutate(rng, src, sub=0.04, ins=0.02, dele=0.02)
        reads.append(O.FastSeq("r%d" % k, seq, rand_qual(rng, len(seq))))
    return reads


def check_overlap(ctx, reads, params_json, cfg_kw, with_revcomps=True):
    import quaff_amd as Q
    p = O.Params.from_json(params_json)
    sc = O.Scores(p)
    null = O.NullParams.from_json(NULL_JSON)
    osc = {False: O.OverlapScores(p, sc, False), True: O.OverlapScores(p, sc, True)}
    n = len(reads)
    seqs = reads + ([r.revcomp() for r in reads] if with_revcomps else [])
    pairs = O.overlap_task_pairs(n, len(seqs))
    ctx.upload_reads([s.seq for s in seqs], [s.qual for s in seqs])
    res = ctx.overlap_resident(pairs, Q.DPConfig(**cfg_kw))
    ocfg = O.DPConfig(kmer_len=cfg_kw.get("kmer_len", 6), kmer_threshold=cfg_kw.get("kmer_threshold", 14),
                      band=cfg_kw.get("band_size", 64), sparse=cfg_kw.get("sparse", True))
    nfinite = 0
    for k, (nx, ny, comp) in enumerate(pairs):
        want = O.overlap_pair(seqs[nx], seqs[ny], comp, osc[comp], sc, null, ocfg)
        got = res["alignments"].get(k)
        if want is None:
            assert got is None and np.isneginf(res["viterbi"][k]), k
            continue
        nfinite += 1
        assert got is not None, (k, nx, ny, comp)
        assert res["n_diagonals"][k] == want["ndiag"] and res["cells"][k] == want["cells"], k
        assert got["result"] == want["result"], (k, nx, ny, comp, got["result"], want["result"])
        assert got["score"] == want["score"], (k, got["score"], want["score"])
        assert (got["xStart"], got["xEnd"], got["yStart"], got["yEnd"]) == (want["xStart"], want["xEnd"], want["yStart"], want["yEnd"]), k
        assert got["ops"] == want["ops"], (k, O.cigar(got["ops"]), O.cigar(want["ops"]))
    return res, nfinite


def test_overlap_small_genome(ctx):
    rng = np.random.default_rng(41)
    reads = overlapping_reads(rng, 1200, 8, 400)
    res, nfinite = check_overlap(ctx, reads, DEFAULT_JSON, dict(kmer_threshold=14))
    assert nfinite == len(O.overlap_task_pairs(8, 16))        # diagonal 0 is always in the envelope: every pair has a path
    assert (res["n_diagonals"] > 60).sum() >= 6               # real overlaps seed multi-diagonal bands


def test_overlap_packed_lse_table_in_lds_and_global_table_agree(ctx):
    """The banded fills read the exact log-sum-exp table from its packed form in LDS (the device rebuilt all 100 001 entries
    from it bit for bit when the context uploaded it); the variant that gathers the table from global memory gives the same
    bits.  Both against the oracle, gap contexts on and off."""
    assert ctx.lse_pack_bytes() > 100_000
    rng = np.random.default_rng(141)
    reads = overlapping_reads(rng, 1500, 9, 500)
    pj = synth_params_json(rng, 1, 1)
    for params in (DEFAULT_JSON, pj):
        ctx.set_params_json(None if params is DEFAULT_JSON else params)
        try:
            a, _ = check_overlap(ctx, reads, params, dict(kmer_threshold=14))
            ctx.set_debug_flags(128)                                  # QF_DEBUG_GLOBAL_LSE
            b, _ = check_overlap(ctx, reads, params, dict(kmer_threshold=14))
        finally:
            ctx.set_debug_flags(0)
            ctx.set_params_json(None)
        assert np.array_equal(a["viterbi"], b["viterbi"]) and (a["n_diagonals"] > 60).sum() >= 6


def test_overlap_row_prefilter_settles_unrelated_pairs(ctx):
    """The scheduler's pair list is long runs (x, y0), (x, y0 + 1), ...: the seeding takes x against 8 - 64 consecutive y at a
    time (coarse k-mer-match counters per y in LDS) and gives every pair without a candidate bin its one forced diagonal
    there; only the rest go through the per-pair kernel.  Same results as the oracle and as the per-pair kernel alone, and
    most pairs of a set with few true overlaps are settled by the prefilter."""
    rng = np.random.default_rng(142)
    reads = overlapping_reads(rng, 9000, 36, 300)            # 72 sequences, rows of ~70 pairs; few of the pairs overlap
    try:
        ctx.set_debug_flags(512)                                 # QF_DEBUG_COUNT_SETTLED
        a, nfinite = check_overlap(ctx, reads, DEFAULT_JSON, dict(kmer_threshold=14))
        settled = ctx.rows_settled()
        ctx.set_debug_flags(256 | 512)                           # QF_DEBUG_NO_ROW_PREFILTER
        b, _ = check_overlap(ctx, reads, DEFAULT_JSON, dict(kmer_threshold=14))
        assert ctx.rows_settled() == 0
    finally:
        ctx.set_debug_flags(0)
    n_pairs = len(O.overlap_task_pairs(36, 72))
    assert nfinite == n_pairs and settled > 0.7 * n_pairs and (a["n_diagonals"] == 1).sum() >= settled
    for key in ("viterbi", "cells", "n_diagonals"):
        assert np.array_equal(a[key], b[key]), key
    assert (a["n_diagonals"] > 60).sum() >= 4                  # and the true overlaps kept their bands


def test_overlap_row_structures_survive_block_splits_and_other_orders(ctx):
    """The row prefilter's work list and the slotted single-diagonal list are built per internal block of the pair list: a
    memory budget that forces the call to run in pieces (blocks that start and end in the middle of an x row), a pair list in
    another order (no runs: everything through the per-pair kernel and the plain list) and a narrow band (several
    single-diagonal bands per pair: plain list) all give the whole call's results."""
    import quaff_amd as Q
    rng = np.random.default_rng(143)
    reads = overlapping_reads(rng, 4000, 40, 260)
    seqs = reads + [r.revcomp() for r in reads]
    pairs = O.overlap_task_pairs(40, 80)
    ctx.upload_reads([s.seq for s in seqs], [s.qual for s in seqs])
    cfg = Q.DPConfig(kmer_threshold=14)
    try:
        ctx.set_debug_flags(512)
        whole = ctx.overlap_resident(pairs, cfg)
        assert ctx.rows_settled() > len(pairs) // 2
        ctx.set_memory_budget(whole["traceback_bytes"] // 5)
        parts = ctx.overlap_resident(pairs, cfg)
        assert ctx.rows_settled() > len(pairs) // 2
    finally:
        ctx.set_memory_budget(0)
        ctx.set_debug_flags(0)
    order = rng.permutation(len(pairs))
    shuffled = ctx.overlap_resident([pairs[k] for k in order], cfg)

    def same(a, b, index=None):
        index = np.arange(len(pairs)) if index is None else index
        for key in ("viterbi", "cells", "n_diagonals"):
            assert np.array_equal(a[key][index], b[key]), key
        inv = {int(index[k]): k for k in range(len(index))}
        assert sorted(a["alignments"]) == sorted(int(index[k]) for k in b["alignments"])
        for k, al in a["alignments"].items():
            bl = b["alignments"][inv[k]]
            assert (al["result"], al["score"], al["xStart"], al["xEnd"], al["yStart"], al["yEnd"], al["ops"]) == \
                   (bl["result"], bl["score"], bl["xStart"], bl["xEnd"], bl["yStart"], bl["yEnd"], bl["ops"]), k

    same(whole, parts)
    same(whole, shuffled, order)
    assert len(whole["alignments"]) >= 20
    # narrow band: a seeded diagonal on its own is a second single-diagonal band of its pair
    check_overlap(ctx, reads[:36], DEFAULT_JSON, dict(kmer_threshold=14, band_size=0))


def test_overlap_bands_and_thresholds(ctx):
    rng = np.random.default_rng(42)
    reads = overlapping_reads(rng, 900, 6, 350)
    check_overlap(ctx, reads, DEFAULT_JSON, dict(kmer_threshold=8, band_size=20))
    check_overlap(ctx, reads, DEFAULT_JSON, dict(kmer_threshold=20, band_size=100), with_revcomps=False)


def test_overlap_long_kmers(ctx):
    rng = np.random.default_rng(44)
    check_overlap(ctx, overlapping_reads(rng, 700, 5, 300), DEFAULT_JSON, dict(kmer_len=11, kmer_threshold=5))


def test_overlap_order2(ctx):
    rng = np.random.default_rng(43)
    pj = synth_params_json(rng, 2, 1)
    ctx.set_params_json(pj)
    try:
        check_overlap(ctx, overlapping_reads(rng, 800, 5, 300), pj, dict(kmer_threshold=12))
    finally:
        ctx.set_params_json(None)


def test_c8f30_overlap_golden_through_gpu(ctx):
    """Makefile:152-156 golden with the HIP path: Stockholm text byte-for-byte."""
    import quaff_amd as Q
    golden = os.path.join(os.path.dirname(__file__), "golden")
    reads = O.read_fastx(os.path.join(golden, "c8f30.fastq.gz"))
    cp = O.FastSeq(reads[0].name.replace("channel", "copy", 1), reads[0].seq, reads[0].qual)
    seqs = [reads[0], cp]
    null = O.NullParams.fit(seqs)
    ctx.set_null_json(null.to_json())
    try:
        ctx.upload_reads([s.seq for s in seqs], [s.qual for s in seqs])
        res = ctx.overlap_resident([(0, 1, False)], Q.DPConfig(kmer_threshold=-1, max_size=10 << 20))
        a = res["alignments"][0]
        al = dict(a, score=a["result"] - null.loglike(seqs[0]) - null.loglike(seqs[1]))   # null went through 6-s.f. JSON
        assert O.overlap_stockholm(seqs[0], seqs[1], al) == open(os.path.join(golden, "c8f30-self-overlap.json")).read()
    finally:
        ctx.set_null_json(NULL_JSON)


def test_overlap_internal_chunking(ctx):
    """A pair list whose traceback exceeds the device budget runs in halves with identical results."""
    import quaff_amd as Q
    rng = np.random.default_rng(45)
    reads = overlapping_reads(rng, 1500, 7, 350)
    seqs = reads + [r.revcomp() for r in reads]
    pairs = O.overlap_task_pairs(7, 14)
    ctx.upload_reads([s.seq for s in seqs], [s.qual for s in seqs])
    whole = ctx.overlap_resident(pairs, Q.DPConfig(kmer_threshold=14))
    try:
        ctx.set_memory_budget(whole["traceback_bytes"] // 3)      # (no less than the largest pair's own bands)
        parts = ctx.overlap_resident(pairs, Q.DPConfig(kmer_threshold=14))
    finally:
        ctx.set_memory_budget(0)
    for key in ("viterbi", "cells", "n_diagonals"):
        assert np.array_equal(whole[key], parts[key]), key
    assert whole["total_cells"] == parts["total_cells"]
    assert sorted(whole["alignments"]) == sorted(parts["alignments"])
    for k, a in whole["alignments"].items():
        b = parts["alignments"][k]
        assert (a["result"], a["score"], a["xStart"], a["xEnd"], a["yStart"], a["yEnd"], a["ops"]) == \
               (b["result"], b["score"], b["xStart"], b["xEnd"], b["yStart"], b["yEnd"], b["ops"]), k


def test_overlap_wide_bands_row_space(ctx):
    """-kmatchoff (every diagonal: > 512 wide) and a sequence shorter than 2(k+threshold) (full-envelope fallback) run on
    the row-space overlap kernel; 700-base reads make two 512-row stripes."""
    rng = np.random.default_rng(46)
    reads = overlapping_reads(rng, 1600, 4, 700)
    res, nfinite = check_overlap(ctx, reads, DEFAULT_JSON, dict(sparse=False))
    assert nfinite == len(O.overlap_task_pairs(4, 8)) and res["n_diagonals"].min() > 1024
    g = reads[0].seq
    reads.append(O.FastSeq("short", g[100:135], rand_qual(rng, 35)))        # 35 < 2 * (6 + 14)
    check_overlap(ctx, reads, DEFAULT_JSON, dict(kmer_threshold=14))


def test_overlap_ragged_lengths_sorted_lists(ctx):
    """More than 64 sequences of very different lengths: class lists are sorted by length before the fills."""
    rng = np.random.default_rng(47)
    g = rand_seq(rng, 1500)
    reads = []
    for k in range(34):
        L = int(rng.integers(80, 500))
        s = int(rng.integers(0, len(g) - L))
        seq = mutate(rng, g[s:s + L], sub=0.04, ins=0.02, dele=0.02)
        reads.append(O.FastSeq("r%d" % k, seq, rand_qual(rng, len(seq))))
    res, nfinite = check_overlap(ctx, reads, DEFAULT_JSON, dict(kmer_threshold=14))
    assert nfinite == len(O.overlap_task_pairs(34, 68))


def test_overlap_score_threshold_filters_before_the_traceback(ctx):
    """Most pairs of an all-vs-all run do not overlap and score below the printer's default threshold of 0: with
    qf_set_score_threshold they are not traced back; the survivors and every per-pair score are unchanged."""
    import quaff_amd as Q
    rng = np.random.default_rng(47)
    reads = overlapping_reads(rng, 6000, 14, 500)
    seqs = reads + [r.revcomp() for r in reads]
    pairs = O.overlap_task_pairs(len(reads), len(seqs))
    ctx.upload_reads([s.seq for s in seqs], [s.qual for s in seqs])
    cfg = Q.DPConfig(kmer_threshold=14)
    try:
        full = ctx.overlap_resident(pairs, cfg)
        scores = sorted(a["score"] for a in full["alignments"].values())
        thr = scores[(2 * len(scores)) // 3]
        ctx.set_score_threshold(thr)
        cut = ctx.overlap_resident(pairs, cfg)
        want = {k: a for k, a in full["alignments"].items() if a["score"] >= thr}
        assert 0 < len(want) < len(full["alignments"]) and set(cut["alignments"]) == set(want)
        for k, a in want.items():
            g = cut["alignments"][k]
            assert (g["score"], g["result"], g["xStart"], g["xEnd"], g["yStart"], g["yEnd"], g["ops"]) == \
                   (a["score"], a["result"], a["xStart"], a["xEnd"], a["yStart"], a["yEnd"], a["ops"])
        assert np.array_equal(cut["viterbi"], full["viterbi"]) and np.array_equal(cut["score"], full["score"])
    finally:
        ctx.set_score_threshold(float("-inf"))


def test_overlap_read_preparation_is_cached_and_invalidated(ctx):
    """Blocks of one pair list reuse the reads' derived arrays (context words, insert sums, null log-likelihoods); anything
    that changes them — another entry point re-deriving them for its own k, new parameters, a new k — must not."""
    import quaff_amd as Q
    rng = np.random.default_rng(48)
    reads = overlapping_reads(rng, 1500, 8, 400)
    seqs = reads + [r.revcomp() for r in reads]
    pairs = O.overlap_task_pairs(len(reads), len(seqs))
    up = lambda: ctx.upload_reads([s.seq for s in seqs], [s.qual for s in seqs])
    cfg = Q.DPConfig(kmer_threshold=14)

    def same(a, b):
        for key in ("viterbi", "score", "cells", "n_diagonals"):
            assert np.array_equal(a[key], b[key]), key
        assert {k: (v["score"], v["ops"]) for k, v in a["alignments"].items()} == {k: (v["score"], v["ops"]) for k, v in b["alignments"].items()}

    up()
    first = ctx.overlap_resident(pairs, cfg)
    same(first, ctx.overlap_resident(pairs, cfg))                    # second block: cached
    ctx.set_refs([seqs[0].seq])
    ctx.align_resident(Q.DPConfig(kmer_len=8, kmer_threshold=5), 0)  # re-derives the read arrays with 8-mers
    same(first, ctx.overlap_resident(pairs, cfg))
    k7 = ctx.overlap_resident(pairs, Q.DPConfig(kmer_threshold=14, kmer_len=7))
    up()
    same(k7, ctx.overlap_resident(pairs, Q.DPConfig(kmer_threshold=14, kmer_len=7)))
    pj = synth_params_json(rng, 2, 1)
    try:
        ctx.set_params_json(pj)                                      # same reads, new model: no stale context words
        changed = ctx.overlap_resident(pairs, cfg)
        up()
        same(changed, ctx.overlap_resident(pairs, cfg))
        assert not np.array_equal(changed["viterbi"], first["viterbi"])
    finally:
        ctx.set_params_json(None)


def test_overlap_blocks_in_flight_on_two_contexts():
    """Two contexts with the same sequences, one host thread each, overlap calls running side by side on the device (how a
    caller keeps the GPU busy across the blocks of a pair list): every call returns what a lone call returns."""
    import quaff_amd as Q
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(144)
    reads = overlapping_reads(rng, 5000, 40, 300)
    seqs = reads + [r.revcomp() for r in reads]
    pairs = O.overlap_task_pairs(40, 80)
    ctxs = []
    for _ in range(2):
        c = Q.Context(0)
        c.set_params_json(None)
        c.set_null_json(NULL_JSON)
        c.upload_reads([s.seq for s in seqs], [s.qual for s in seqs])
        ctxs.append(c)
    try:
        cfg = Q.DPConfig(kmer_threshold=14)
        lone = ctxs[0].overlap_resident(pairs, cfg)
        with ThreadPoolExecutor(2) as pool:
            def run(c):
                return [c.overlap_resident(pairs, cfg) for _ in range(4)]
            results = [f.result() for f in [pool.submit(run, c) for c in ctxs]]
        for per_ctx in results:
            for r in per_ctx:
                for key in ("viterbi", "score", "cells", "n_diagonals"):
                    assert np.array_equal(r[key], lone[key]), key
                assert {k: (a["score"], a["ops"]) for k, a in r["alignments"].items()} == \
                       {k: (a["score"], a["ops"]) for k, a in lone["alignments"].items()}
        assert len(lone["alignments"]) >= 10
    finally:
        for c in ctxs:
            c.close()
