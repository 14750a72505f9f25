"""qf_overlap_rows: rows of QuaffOverlapScheduler's pair enumeration (src/qoverlap.cpp:475-480,528-547) generated, thresholded
and reduced on the device.  Checked against the explicit-pair-list entry point (same pairs, host-built list), against the oracle
(every hit with ==), across internal block sizes and row sub-ranges, with and without the reverse complements."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.test_gpu_align import NULL_JSON, DEFAULT_JSON
from tests.test_gpu_overlap import overlapping_reads

pytestmark = pytest.mark.gpu
MASK = (1 << 64) - 1


@pytest.fixture(scope="module")
def ctx():
    import quaff_amd as Q
    c = Q.Context(0)
    c.set_params_json(None)
    c.set_null_json(NULL_JSON)
    yield c
    c.set_overlap_block_pairs(0)
    c.set_score_threshold(float("-inf"))
    c.close()


def scheduler_pairs(n_orig, n_seqs, x0, x1):
    return [(nx, ny, ny >= n_orig) for nx in range(x0, x1) for ny in range(nx + 1, n_seqs)]


def checksum(viterbi):
    v = np.asarray(viterbi, np.float64)
    bits = v[np.isfinite(v)].view(np.uint64)
    return int(np.sum(bits, dtype=np.uint64)) & MASK       # numpy wraps mod 2^64 like the device's atomic adds


def compare_with_pair_list(ctx, n_orig, n_seqs, x0, x1, cfg, threshold):
    """The same rows through both entry points: totals and every returned alignment identical."""
    pairs = scheduler_pairs(n_orig, n_seqs, x0, x1)
    ctx.set_score_threshold(threshold)
    rows = ctx.overlap_rows(n_orig, x0, x1, cfg)
    lst = ctx.overlap_resident(pairs, cfg)
    assert rows["n_pairs"] == len(pairs) == ctx.L.qf_overlap_rows_pairs(n_seqs, x0, x1)
    assert rows["n_finite"] == int(np.isfinite(lst["viterbi"]).sum())
    assert rows["total_cells"] == lst["total_cells"] == int(lst["cells"].sum())
    assert rows["total_diagonals"] == int(lst["n_diagonals"].sum())
    assert rows["result_checksum"] == checksum(lst["viterbi"])
    want = [k for k in range(len(pairs)) if k in lst["alignments"]]
    assert len(rows["hits"]) == len(want)
    for h, k in zip(rows["hits"], want):
        a = lst["alignments"][k]
        assert (int(h["x"]), int(h["y"])) == pairs[k][:2]
        assert (h["viterbi"], h["score"], int(h["x_start"]), int(h["x_end"]), int(h["y_start"]), int(h["y_end"])) == \
               (a["result"], a["score"], a["xStart"], a["xEnd"], a["yStart"], a["yEnd"])
        assert ctx.hit_ops(h, rows["runs"]) == a["ops"]
    if len(rows["hits"]) > 1:       # the scheduler's order
        key = rows["hits"]["x"].astype(np.int64) * n_seqs + rows["hits"]["y"]
        assert np.all(np.diff(key) > 0)
    return rows, lst, pairs


def oracle_check(rows, seqs, n_orig, threshold, cfg_kw):
    p = O.Params.from_json(DEFAULT_JSON)
    sc = O.Scores(p)
    null = O.NullParams.from_json(NULL_JSON)
    osc = {False: O.OverlapScores(p, sc, False), True: O.OverlapScores(p, sc, True)}
    ocfg = O.DPConfig(kmer_len=cfg_kw.get("kmer_len", 6), kmer_threshold=cfg_kw.get("kmer_threshold", 14),
                      band=cfg_kw.get("band_size", 64), sparse=cfg_kw.get("sparse", True))
    for h in rows["hits"]:
        nx, ny = int(h["x"]), int(h["y"])
        comp = ny >= n_orig
        want = O.overlap_pair(seqs[nx], seqs[ny], comp, osc[comp], sc, null, ocfg)
        assert want is not None and want["score"] >= threshold
        assert (h["viterbi"], h["score"]) == (want["result"], want["score"]), (nx, ny)
        assert (int(h["x_start"]), int(h["x_end"]), int(h["y_start"]), int(h["y_end"])) == \
               (want["xStart"], want["xEnd"], want["yStart"], want["yEnd"])
        ops = rows["runs"][int(h["run_offset"]):int(h["run_offset"]) + int(h["n_runs"])]
        assert "".join("MID"[int(v) & 3] * (int(v) >> 2) for v in ops) == want["ops"]


def test_rows_match_pair_list_and_oracle_both_strands(ctx):
    import quaff_amd as Q
    rng = np.random.default_rng(301)
    reads = overlapping_reads(rng, 6000, 30, 320)
    seqs = reads + [r.revcomp() for r in reads]
    ctx.upload_reads([s.seq for s in seqs], [s.qual for s in seqs])
    cfg = Q.DPConfig(kmer_threshold=14)
    ctx.set_overlap_block_pairs(0)
    rows, lst, pairs = compare_with_pair_list(ctx, 30, 60, 0, 29, cfg, float("-inf"))
    assert rows["n_blocks"] == 1 and rows["n_finite"] == len(pairs) == len(rows["hits"])   # the forced diagonal: every pair has a path
    oracle_check(rows, seqs, 30, float("-inf"), {})
    # the printer's default threshold: only true overlaps come back, the totals still cover every pair
    thr, _, _ = compare_with_pair_list(ctx, 30, 60, 0, 29, cfg, 0.0)
    assert 0 < len(thr["hits"]) < len(pairs) / 4 and np.all(thr["hits"]["score"] >= 0)
    assert (thr["n_finite"], thr["total_cells"], thr["result_checksum"]) == (rows["n_finite"], rows["total_cells"], rows["result_checksum"])
    keep = rows["hits"][rows["hits"]["score"] >= 0]
    assert np.array_equal(keep[["x", "y", "viterbi", "score"]], thr["hits"][["x", "y", "viterbi", "score"]])


def test_rows_are_independent_of_internal_blocks_and_add_up_over_row_ranges(ctx):
    import quaff_amd as Q
    rng = np.random.default_rng(302)
    reads = overlapping_reads(rng, 5000, 26, 300)
    seqs = reads + [r.revcomp() for r in reads]
    ctx.upload_reads([s.seq for s in seqs], [s.qual for s in seqs])
    cfg = Q.DPConfig(kmer_threshold=14)
    ctx.set_score_threshold(0.0)
    ctx.set_overlap_block_pairs(0)
    whole = ctx.overlap_rows(26, 0, 25, cfg)
    try:
        for block in (1, 60, 333):        # one row per block ... several rows per block
            ctx.set_overlap_block_pairs(block)
            cut = ctx.overlap_rows(26, 0, 25, cfg)
            assert cut["n_blocks"] > 1
            for k in ("n_pairs", "n_finite", "total_cells", "total_diagonals", "result_checksum"):
                assert cut[k] == whole[k], (block, k)
            assert np.array_equal(cut["hits"][["x", "y", "viterbi", "score", "x_start", "x_end", "y_start", "y_end", "n_runs"]],
                                  whole["hits"][["x", "y", "viterbi", "score", "x_start", "x_end", "y_start", "y_end", "n_runs"]])
            assert np.array_equal(cut["runs"], whole["runs"])
    finally:
        ctx.set_overlap_block_pairs(0)
    # disjoint row ranges (what several GPUs take) add up to the whole; the last rows are the short ones
    parts = [ctx.overlap_rows(26, a, b, cfg) for a, b in ((0, 7), (7, 8), (8, 8), (8, 25))]
    for k in ("n_pairs", "n_finite", "total_cells", "total_diagonals"):
        assert sum(p[k] for p in parts) == whole[k], k
    assert sum(p["result_checksum"] for p in parts) & MASK == whole["result_checksum"]
    assert np.array_equal(np.concatenate([p["hits"] for p in parts])[["x", "y", "viterbi", "score"]], whole["hits"][["x", "y", "viterbi", "score"]])
    assert parts[2]["n_pairs"] == 0 and len(parts[2]["hits"]) == 0
    compare_with_pair_list(ctx, 26, 52, 20, 25, cfg, 0.0)       # the triangle's last rows through both entry points


def test_rows_forward_strand_only_ragged_lengths_and_full_dp(ctx):
    import quaff_amd as Q
    rng = np.random.default_rng(303)
    reads = overlapping_reads(rng, 2500, 14, 260)
    reads[3] = O.FastSeq("short", reads[3].seq[:40], reads[3].qual[:40])      # shorter than 2 (k + threshold): full envelope
    reads[9] = O.FastSeq("long", reads[9].seq + reads[2].seq, reads[9].qual + reads[2].qual)
    ctx.upload_reads([s.seq for s in reads], [s.qual for s in reads])          # -fwdstrand: no reverse complements resident
    for kw in (dict(kmer_threshold=14), dict(kmer_threshold=14, band_size=20), dict(sparse=False)):
        rows, lst, pairs = compare_with_pair_list(ctx, 14, 14, 0, 13, Q.DPConfig(**kw), float("-inf"))
        assert len(pairs) == 13 * 14 // 2
        oracle_check(rows, reads, 14, float("-inf"), kw)


def test_rows_argument_errors(ctx):
    import quaff_amd as Q
    from quaff_amd.api import QuaffHipError
    rng = np.random.default_rng(304)
    reads = overlapping_reads(rng, 1500, 6, 200)
    seqs = reads + [r.revcomp() for r in reads]
    ctx.upload_reads([s.seq for s in seqs], [s.qual for s in seqs])
    cfg = Q.DPConfig(kmer_threshold=14)
    for n_orig, x0, x1 in ((6, 0, 6), (6, 3, 2), (5, 0, 4), (0, 0, 0), (12, 0, 12)):
        with pytest.raises(QuaffHipError) as e:
            ctx.overlap_rows(n_orig, x0, x1, cfg)
        assert e.value.code == -2
    bad = Q.DPConfig(kmer_threshold=14)
    bad.reserved = 4
    with pytest.raises(QuaffHipError) as e:
        ctx.overlap_rows(6, 0, 5, bad)
    assert e.value.code == -2 and "reserved" in str(e.value)
    assert ctx.overlap_rows(6, 0, 5, cfg)["n_pairs"] == 11 + 10 + 9 + 8 + 7
    assert ctx.overlap_rows(12, 0, 11, cfg)["n_pairs"] == 66          # the same 12 sequences taken as 12 originals, no complements


def _oracle_rows(rows, seqs, n_orig, params_json, cfg_kw, limit=400):
    """every returned hit (at most `limit`, evenly spread) against the oracle with =="""
    p = O.Params.from_json(params_json)
    sc = O.Scores(p)
    null = O.NullParams.from_json(NULL_JSON)
    osc = {False: O.OverlapScores(p, sc, False), True: O.OverlapScores(p, sc, True)}
    ocfg = O.DPConfig(kmer_len=cfg_kw.get("kmer_len", 6), kmer_threshold=cfg_kw.get("kmer_threshold", 14), band=cfg_kw.get("band_size", 64))
    hits = rows["hits"]
    pick = range(len(hits)) if len(hits) <= limit else np.linspace(0, len(hits) - 1, limit).astype(int)
    for k in pick:
        h = hits[int(k)]
        nx, ny = int(h["x"]), int(h["y"])
        comp = ny >= n_orig
        want = O.overlap_pair(seqs[nx], seqs[ny], comp, osc[comp], sc, null, ocfg)
        assert want is not None and (h["viterbi"], h["score"]) == (want["result"], want["score"]), (nx, ny)
        assert (int(h["x_start"]), int(h["x_end"]), int(h["y_start"]), int(h["y_end"])) == (want["xStart"], want["xEnd"], want["yStart"], want["yEnd"])


@pytest.mark.parametrize("case", ["wide_qualities", "two_base_match_context", "no_qualities"])
def test_single_diagonal_rows_kernel_row_pitches(case):
    """k_overlap_single_rows keeps the pair-emission rows of the quality values in use, Km x nq entries, at a row pitch of 128,
    256 or 512 doubles: Phred 0..60 with one-base contexts takes 256 (244 entries), two-base match contexts with context-free
    gaps (-suborder 1 -gaporder 0) 512 (336 entries); reads without qualities have one quality row per context (4 entries).
    Rows vs the pair-list entry point (which runs the plain-list kernels) and vs the oracle."""
    import quaff_amd as Q
    from tests.helpers import mutate, rand_seq, rand_qual
    from tests.test_gpu_align import synth_params_json
    rng = np.random.default_rng({"wide_qualities": 311, "two_base_match_context": 312, "no_qualities": 313}[case])
    pj = synth_params_json(rng, 2, 0) if case == "two_base_match_context" else DEFAULT_JSON
    g = rand_seq(rng, 4000)
    reads = []
    for k in range(40):
        L = int(rng.integers(150, 330))
        s = int(rng.integers(0, len(g) - L))
        src = g[s:s + L] if k % 3 else O.revcomp_str(g[s:s + L])
        seq = mutate(rng, src, sub=0.04, ins=0.02, dele=0.02)
        qual = "" if case == "no_qualities" else (rand_qual(rng, len(seq), 0, 60) if case == "wide_qualities" else rand_qual(rng, len(seq)))
        reads.append(O.FastSeq("r%d" % k, seq, qual))
    seqs = reads + [r.revcomp() for r in reads]
    c = Q.Context(0)
    try:
        c.set_params_json(None if pj is DEFAULT_JSON else pj)
        c.set_null_json(NULL_JSON)
        c.upload_reads([s.seq for s in seqs], None if case == "no_qualities" else [s.qual for s in seqs])
        cfg = Q.DPConfig(kmer_threshold=14)
        rows, lst, pairs = compare_with_pair_list(c, 40, 80, 0, 39, cfg, float("-inf"))
        assert rows["n_finite"] == len(pairs)
        _oracle_rows(rows, seqs, 40, pj, {})
    finally:
        c.close()


def test_single_diagonal_rows_kernel_many_chunks_and_a_strand_boundary_inside_a_wavefront():
    """4 500 originals + their reverse complements = 9 000 resident sequences: 36 y chunks of 256 (both bands of a lane, several chunk
    sets per XCD), the first reverse complement (index 4 500) in the middle of a group of 64 -- a wavefront whose lanes read two
    strand tables, taken in two passes -- ragged lengths inside every group of 64 (blocks where some lanes' diagonals have ended),
    and the rows the triangle's chunks run out on.  All hits of rows 0..2, rows around the strand boundary and the last rows
    against the oracle; the totals of a row range against the pair-list entry point."""
    import quaff_amd as Q
    from tests.helpers import mutate, rand_seq, rand_qual
    rng = np.random.default_rng(314)
    n = 4500
    g = rand_seq(rng, 30000)
    reads = []
    for k in range(n):
        L = int(rng.integers(60, 200))
        s = int(rng.integers(0, len(g) - L))
        src = g[s:s + L] if k % 2 else O.revcomp_str(g[s:s + L])
        seq = mutate(rng, src, sub=0.03, ins=0.01, dele=0.01)
        reads.append(O.FastSeq("r%d" % k, seq, rand_qual(rng, len(seq))))
    seqs = reads + [r.revcomp() for r in reads]
    c = Q.Context(0)
    try:
        c.set_params_json(None)
        c.set_null_json(NULL_JSON)
        c.upload_reads([s.seq for s in seqs], [s.qual for s in seqs])
        cfg = Q.DPConfig(kmer_threshold=14)
        c.set_score_threshold(0.0)
        for x0, x1 in ((0, 3), (2240, 2243), (n - 4, n - 1)):
            rows = c.overlap_rows(n, x0, x1, cfg)
            assert rows["n_pairs"] == sum(2 * n - 1 - x for x in range(x0, x1)) == rows["n_finite"]
            assert len(rows["hits"]) > 0
            _oracle_rows(rows, seqs, n, DEFAULT_JSON, {}, limit=150)
        rows, lst, pairs = compare_with_pair_list(c, n, 2 * n, n - 3, n - 1, cfg, 0.0)
        # several rows in one block and one at a time: the same totals
        c.set_score_threshold(float("-inf"))
        whole = c.overlap_rows(n, 10, 14, cfg)
        one = [c.overlap_rows(n, x, x + 1, cfg) for x in range(10, 14)]
        for k in ("n_pairs", "n_finite", "total_cells", "total_diagonals"):
            assert sum(o[k] for o in one) == whole[k], k
        assert sum(o["result_checksum"] for o in one) & MASK == whole["result_checksum"]
    finally:
        c.close()

def test_row_prefilter_index_and_item_list_variants_agree(ctx):
    """The row prefilter's chunk index has 16-bit entries where the sequences are short enough and 32-bit ones otherwise
    (QF_DEBUG_ROW_INDEX_32 forces the latter), and for the scheduler's triangle its item list is formed by a kernel instead of the
    host (QF_DEBUG_HOST_ROW_ITEMS keeps the host's): every variant must settle the same pairs and return the same hits."""
    import quaff_amd as Q
    rng = np.random.default_rng(411)
    reads = overlapping_reads(rng, 6000, 90, 260)
    seqs = reads + [r.revcomp() for r in reads]
    ctx.upload_reads([s.seq for s in seqs], [s.qual for s in seqs])
    cfg = Q.DPConfig(kmer_threshold=14)
    ctx.set_score_threshold(0.0)
    cols = ["x", "y", "viterbi", "score", "x_start", "x_end", "y_start", "y_end", "n_runs"]
    try:
        ctx.set_debug_flags(512)                                   # QF_DEBUG_COUNT_SETTLED
        base = ctx.overlap_rows(90, 0, 89, cfg)
        settled = ctx.rows_settled()
        assert settled > 0
        for flags in (131072, 262144, 131072 | 262144):           # 32-bit entries; the host's item list; both
            ctx.set_debug_flags(512 | flags)
            got = ctx.overlap_rows(90, 0, 89, cfg)
            assert ctx.rows_settled() == settled, flags
            for k in ("n_pairs", "n_finite", "total_cells", "total_diagonals", "result_checksum"):
                assert got[k] == base[k], (flags, k)
            assert np.array_equal(got["hits"][cols], base["hits"][cols]) and np.array_equal(got["runs"], base["runs"])
    finally:
        ctx.set_debug_flags(0)


def test_rows_survive_a_secondary_buffer_that_does_not_fit(ctx):
    """The overlap path's per-chunk buffers (unit tables, slot list, item list, alignment records, run lists) split the chunk when
    they cannot be had, like the traceback budget does (tests/test_gpu_align.py has the align / count side): same hits."""
    import quaff_amd as Q
    rng = np.random.default_rng(517)
    reads = overlapping_reads(rng, 4000, 40, 250)
    seqs = reads + [r.revcomp() for r in reads]
    ctx.upload_reads([s.seq for s in seqs], [s.qual for s in seqs])
    cfg = Q.DPConfig(kmer_threshold=14)
    ctx.set_score_threshold(0.0)
    cols = ["x", "y", "viterbi", "score", "x_start", "x_end", "y_start", "y_end", "n_runs"]
    base = ctx.overlap_rows(40, 0, 39, cfg)
    fired = 0
    try:
        for nth in (0, 2, 6, 11, 14, 16, 18, 21, 24):
            ctx.fail_chunk_reserve(nth)
            got = ctx.overlap_rows(40, 0, 39, cfg)
            fired += ctx.fail_chunk_reserve(-1) < 0
            for k in ("n_pairs", "n_finite", "total_cells", "total_diagonals", "result_checksum"):
                assert got[k] == base[k], (nth, k)
            assert np.array_equal(got["hits"][cols], base["hits"][cols]) and np.array_equal(got["runs"], base["runs"]), nth
        assert fired >= 6, fired
    finally:
        ctx.fail_chunk_reserve(-1)
