"""GPU parity for the Forward-Backward E-step (quaff count / train): HIP path through the C ABI vs the oracle's
QuaffCountingTask restatement.  Tolerance 1e-4 relative (BASELINE.json north_star): the reference's own table
log-sum-exp carries that much error, and the GPU re-associates the count sums."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import rand_seq, make_reads, rand_qual, mutate
from tests.test_gpu_align import NULL_JSON, DEFAULT_JSON, both_strands, synth_params_json

pytestmark = pytest.mark.gpu
RTOL = 1e-4


@pytest.fixture(scope="module")
def ctx():
    import quaff_amd as Q
    c = Q.Context(0)
    c.set_params_json(None)
    c.set_null_json(NULL_JSON)
    yield c
    c.close()


def oracle_estep(refs, reads, sc, null, cfg, orders=None, use_null=True):
    """Sum of the oracle's per-read counts.  A read without any finite Forward likelihood under -force (no null model in the
    normalisation; e.g. -global with a band that misses the corners) has posterior weights exp(-inf - (-inf)) = NaN in the
    reference (src/qmodel.cpp:2258-2262) and poisons its E-step totals, and so does a pathless reference that comes first in a
    read's order (0 x the NaN counts of its Backward pass); the library gives neither any weight, so the expected
    totals leave them out."""
    tot = np.zeros(O.counts_size(sc.Km, sc.Kg))
    ylogs, new_orders, fwd = [], [], []
    for r, read in enumerate(reads):
        d = {}
        c, yl, no = O.count_read(refs, read, sc, null, cfg, None if orders is None else orders[r], use_null, skip_pathless=not use_null, details=d)
        fwd.append(d)
        if np.isfinite(yl):
            tot += c
        else:
            assert not use_null
        ylogs.append(yl)
        new_orders.append(no)
    oracle_estep.details = fwd      # per read: the oracle's per-reference Forward values and which references it gave a Backward pass
    return tot, np.array(ylogs), new_orders


def assert_counts_close(got, want, what=""):
    """Every entry above 1e-6 at RTOL relative -- no floor: a rarely used (context, quality) cell is held to the same 1e-4 as the
    big ones; what the oracle puts below 1e-6 must be below 2e-6 here too."""
    big = np.abs(want) > 1e-6
    rel = np.abs(got - want)[big] / np.abs(want)[big]
    if rel.size:      # (a case in which the null model takes every read has no counts at all)
        assert rel.max() <= RTOL, (what, float(rel.max()), np.flatnonzero(big)[np.argsort(-rel)[:5]])
    assert np.all(np.abs(got[~big]) <= 2e-6), what
    return float(rel.max()) if rel.size else 0.0


def run_case(ctx, refs, reads, sc, null, cfg_kw=None, orders=None, force=False):
    import quaff_amd as Q
    cfg_kw = cfg_kw or {}
    ctx.set_refs([x.seq for x in refs])
    ctx.upload_reads([r.seq for r in reads], [r.qual for r in reads])
    res = ctx.count_resident(Q.DPConfig(**cfg_kw), force=force, sort_order=orders)
    ocfg = O.DPConfig(local=cfg_kw.get("local", True), kmer_threshold=cfg_kw.get("kmer_threshold", 20),
                      band=cfg_kw.get("band_size", 64), kmer_len=cfg_kw.get("kmer_len", 6), sparse=cfg_kw.get("sparse", True))
    want, ylogs, new_orders = oracle_estep(refs, reads, sc, null, ocfg, orders, use_null=not force)
    fin = np.isfinite(ylogs)
    assert np.array_equal(np.isfinite(res["read_loglike"]), fin)
    # WHICH pairs carry no weight is pinned, not only the totals: a pair is pathless here exactly where the oracle finds no path,
    # and has posterior weight exactly where the oracle ran a Backward pass over it (and the read has a likelihood at all)
    for r, d in enumerate(oracle_estep.details):
        order = list(range(len(refs))) if orders is None else list(orders[r])
        pathless = np.array([d["forward"][x] == -np.inf for x in range(len(refs))])
        seen = np.zeros(len(refs), bool)
        seen[order] = True
        assert np.array_equal(np.isneginf(res["forward"][r])[seen], pathless[seen]), (r, res["forward"][r], d["forward"])
        weighted = set(int(x) for x in np.flatnonzero(res["weight"][r] > 0))
        counted = set(d["counted"]) if fin[r] else set()
        must = set(x for x in counted if np.exp(d["forward"][x] - ylogs[r]) > 1e-290)   # (a weight may underflow to 0 on either side)
        assert must <= weighted <= counted, (r, weighted, d["counted"])
    np.testing.assert_allclose(res["read_loglike"][fin], ylogs[fin], rtol=RTOL)
    assert res["loglike"] == ylogs.sum() or abs(res["loglike"] - ylogs.sum()) <= RTOL * abs(ylogs.sum())
    assert_counts_close(res["counts"], want, "counts")
    assert res["sort_order"] == new_orders
    return res, want


def test_count_small_both_strands(ctx):
    rng = np.random.default_rng(31)
    ref = rand_seq(rng, 1500)
    sc, null = O.Scores(O.Params.from_json(DEFAULT_JSON)), O.NullParams.from_json(NULL_JSON)
    reads = make_reads(rng, ref, 10, 280)
    refs = both_strands(ref)
    res, want = run_case(ctx, refs, reads, sc, null)
    # Forward per pair against the oracle, and emitted-base conservation
    for r, read in enumerate(reads):
        rc = O.ReadCtx(read, sc)
        for x, rf in enumerate(refs):
            xt = O.tokens(rf.seq)
            f, _, _ = O.forward_backward(xt, rc, sc, O.envelope(xt, rc.tok, O.DPConfig(), 48), want_back=False)
            assert abs(res["forward"][r, x] - f) <= RTOL * abs(f), (r, x)
    ne = (4 + 4 * sc.Km) * O.NQUAL
    assert abs(res["counts"][:ne].sum() - want[:ne].sum()) < 1e-3 * want[:ne].sum()
    # second EM iteration: feed the pruned order back (wrong-strand references drop out)
    orders = res["sort_order"]
    assert all(len(o) == 1 for o in orders)
    res2, _ = run_case(ctx, refs, reads, sc, null, orders=orders)
    assert np.all(np.isneginf(res2["forward"][np.arange(10), 1 - np.array([o[0] for o in orders])]))


def test_count_force_global_and_bands(ctx):
    rng = np.random.default_rng(32)
    ref = rand_seq(rng, 900)
    sc, null = O.Scores(O.Params.from_json(DEFAULT_JSON)), O.NullParams.from_json(NULL_JSON)
    reads = make_reads(rng, ref, 6, 300)[::2]
    run_case(ctx, [O.FastSeq("ref", ref)], reads, sc, null, force=True)
    run_case(ctx, [O.FastSeq("ref", ref)], reads, sc, null, cfg_kw=dict(band_size=24, kmer_threshold=10))
    from tests.helpers import mutate, rand_qual
    greads = [O.FastSeq("g%d" % n, s, rand_qual(rng, len(s))) for n, s in enumerate(mutate(rng, ref) for _ in range(3))]
    run_case(ctx, [O.FastSeq("ref", ref)], greads, sc, null, cfg_kw=dict(local=False))


def test_count_order2(ctx):
    rng = np.random.default_rng(33)
    pj = synth_params_json(rng, 3, 2)
    sc, null = O.Scores(O.Params.from_json(pj)), O.NullParams.from_json(NULL_JSON)
    ctx.set_params_json(pj)
    try:
        ref = rand_seq(rng, 1000)
        run_case(ctx, both_strands(ref), make_reads(rng, ref, 6, 250), sc, null)
    finally:
        ctx.set_params_json(None)


def test_c8f30_counts_golden_through_gpu(ctx):
    """The reference's count golden (Makefile:146-147) with the HIP path doing Forward-Backward: 6 s.f. text."""
    import quaff_amd as Q
    golden = os.path.join(os.path.dirname(__file__), "golden")
    reads = O.read_fastx(os.path.join(golden, "c8f30.fastq.gz"))
    null = O.NullParams.fit(reads)
    ctx.set_null_json(null.to_json())
    try:
        ctx.set_refs([reads[0].seq])
        ctx.upload_reads([reads[0].seq], [reads[0].qual])
        res = ctx.count_resident(Q.DPConfig(kmer_threshold=-1, max_size=10 << 20))
        want = open(os.path.join(golden, "c8f30-self-counts.json")).read()
        got = O.param_counts_json(res["counts"], 1, 0)
        if got != want:   # allow last-digit differences of the 6-s.f. rendering (tolerance 1e-4)
            import re
            g, w = (list(map(float, re.findall(r"-?\d+\.?\d*(?:e[-+]?\d+)?", t))) for t in (got, want))
            assert len(g) == len(w)
            np.testing.assert_allclose(g, w, rtol=RTOL, atol=1e-6)
    finally:
        ctx.set_null_json(NULL_JSON)


def test_count_internal_chunking(ctx):
    """Forward matrices over the device budget: the E-step runs in halves, same per-read results, counts to 1e-9."""
    import quaff_amd as Q
    rng = np.random.default_rng(35)
    ref = rand_seq(rng, 1200)
    reads = make_reads(rng, ref, 13, 260)
    refs = both_strands(ref)
    ctx.set_refs([x.seq for x in refs])
    ctx.upload_reads([r.seq for r in reads], [r.qual for r in reads])
    whole = ctx.count_resident(Q.DPConfig())
    try:
        ctx.set_memory_budget(whole["forward_bytes"] // 5)
        parts = ctx.count_resident(Q.DPConfig())
        parts2 = ctx.count_resident(Q.DPConfig(), sort_order=whole["sort_order"])
    finally:
        ctx.set_memory_budget(0)
    whole2 = ctx.count_resident(Q.DPConfig(), sort_order=whole["sort_order"])
    for w, p in ((whole, parts), (whole2, parts2)):
        assert np.array_equal(w["forward"], p["forward"]) and np.array_equal(w["read_loglike"], p["read_loglike"])
        assert w["sort_order"] == p["sort_order"] and w["loglike"] == p["loglike"]
        assert w["total_cells"] == p["total_cells"] and w["forward_bytes"] == p["forward_bytes"]
        np.testing.assert_allclose(p["counts"], w["counts"], rtol=1e-9, atol=1e-12)
    # the packed form of the per-read reference order (arrays instead of per-read lists) means the same thing
    packed = ctx.count_resident(Q.DPConfig(), packed_order=True)["sort_order"]
    assert [list(map(int, packed[0][r, :packed[1][r]])) for r in range(len(reads))] == whole["sort_order"]
    whole3 = ctx.count_resident(Q.DPConfig(), sort_order=packed, packed_order=True)
    assert np.array_equal(whole3["forward"], whole2["forward"]) and whole3["loglike"] == whole2["loglike"]
    assert [list(map(int, whole3["sort_order"][0][r, :whole3["sort_order"][1][r]])) for r in range(len(reads))] == whole2["sort_order"]


def test_count_wide_bands_row_space(ctx):
    """Bands wider than 1024 diagonals (-kmatchoff, and the full-envelope fallback of reads shorter than 2(k+threshold))
    run on the row-space Forward/Backward kernels: several 512-row stripes, both strands, local and global."""
    rng = np.random.default_rng(36)
    ref = rand_seq(rng, 1400)
    sc, null = O.Scores(O.Params.from_json(DEFAULT_JSON)), O.NullParams.from_json(NULL_JSON)
    reads = make_reads(rng, ref, 4, 240)
    reads.append(O.FastSeq("short", ref[700:745], rand_qual(rng, 45)))     # 45 < 2 * (6 + 20): full envelope
    res, _ = run_case(ctx, both_strands(ref), reads, sc, null, cfg_kw=dict(sparse=False))
    assert res["total_cells"] == 2 * 1400 * sum(len(r.seq) for r in reads)
    run_case(ctx, both_strands(ref), reads, sc, null)                       # mixed: narrow bands + one full envelope
    run_case(ctx, [O.FastSeq("ref", ref)], reads[:2], sc, null, cfg_kw=dict(sparse=False, local=False), force=True)


def test_count_global_mode_no_paths(ctx):
    """-global -force with reads that are fragments of the reference: no global path exists, every log-likelihood is -inf;
    the next-iteration order keeps both references with the later one first (ties of the reference's ascending sort,
    reversed: src/util.h:115-124, src/qmodel.cpp:2264-2265).  Found by a randomized soak."""
    rng = np.random.default_rng(37)
    ref = rand_seq(rng, 1200)
    sc, null = O.Scores(O.Params.from_json(DEFAULT_JSON)), O.NullParams.from_json(NULL_JSON)
    reads = make_reads(rng, ref, 5, 200)
    res, _ = run_case(ctx, both_strands(ref), reads, sc, null, cfg_kw=dict(local=False), force=True)
    assert np.isneginf(res["loglike"]) and all(o == [1, 0] for o in res["sort_order"])


def test_count_many_bands_one_running_end_sum(ctx):
    """Dozens of bands per pair (4-mers, threshold 4): the reference's Forward `end` is one running table-lse sum down the whole
    last column (src/qmodel.cpp:1379-1381), and the table drops terms more than 10 below the running total, so a band-by-band
    sum combined afterwards differed by 1.1e-4 on two counts of this case (soak seed 250)."""
    from tests.helpers import mutate
    rng = np.random.default_rng(5250)
    order = int(rng.integers(0, 3))
    pj = synth_params_json(rng, order + 1, order)
    ctx.set_params_json(pj)
    try:
        sc, null = O.Scores(O.Params.from_json(pj)), O.NullParams.from_json(NULL_JSON)
        ref = rand_seq(rng, int(rng.integers(600, 3000)))
        n = int(rng.integers(2, 70))
        reads = []
        for k in range(n):
            L = int(rng.integers(25, min(700, len(ref) - 10)))
            s0 = int(rng.integers(0, len(ref) - L)); src = ref[s0:s0 + L]
            if rng.random() < 0.5:
                src = O.revcomp_str(src)
            seq = mutate(rng, src, sub=rng.uniform(0, .08), ins=rng.uniform(0, .05), dele=rng.uniform(0, .05)) or "A"
            reads.append(O.FastSeq("r%d" % k, seq, rand_qual(rng, len(seq))))
        res, want = run_case(ctx, both_strands(ref), reads, sc, null, cfg_kw=dict(kmer_len=4, kmer_threshold=4, band_size=42))
        assert assert_counts_close(res["counts"], want, "ragged") < 5e-5
    finally:
        ctx.set_params_json(None)


def test_count_many_references_and_pruned_orders(ctx):
    """Ten reference sequences (five + reverse complements, two of them near-duplicates so that several references stay within
    20 of a read's running log-likelihood): posterior weights over more than one reference, the next iteration's order
    (log-likelihood descending, ties to the later reference, cut at -20), and a second E-step on the pruned orders."""
    from tests.helpers import mutate
    rng = np.random.default_rng(38)
    sc, null = O.Scores(O.Params.from_json(DEFAULT_JSON)), O.NullParams.from_json(NULL_JSON)
    base = rand_seq(rng, 1800)
    fwd = [O.FastSeq("a", base), O.FastSeq("b", mutate(rng, base, sub=.01, ins=.002, dele=.002)), O.FastSeq("c", rand_seq(rng, 900)),
           O.FastSeq("d", base[600:1500]), O.FastSeq("e", rand_seq(rng, 2500))]
    refs = fwd + [x.revcomp() for x in fwd]
    reads = make_reads(rng, base, 14, 260) + make_reads(rng, fwd[4].seq, 6, 300)
    res, _ = run_case(ctx, refs, reads, sc, null)
    assert max(len(o) for o in res["sort_order"]) >= 3 and min(len(o) for o in res["sort_order"]) >= 1
    assert ((res["weight"] > 0.01).sum(axis=1) >= 2).any()          # some reads split their weight over several references
    res2, _ = run_case(ctx, refs, reads, sc, null, orders=res["sort_order"])
    run_case(ctx, refs, reads, sc, null, orders=res2["sort_order"], force=True)


# ---------------------------------------------------------------------------------------------------------------------
# Adversarial inputs for the E-step's reduced-precision storage: Forward rows are kept as fp32 offsets from a per-row anchor and
# the count terms are formed in fp32.  The error of a stored state grows with its distance from its row's best
# state, so these cases put probability mass far from the row maximum or split it evenly.  Same bar as everything else: every
# count entry above 1e-6 at 1e-4 relative, no floor; per-read log-likelihoods at 1e-4.
# ---------------------------------------------------------------------------------------------------------------------
def test_forced_estep_over_wrong_strand_pairs(ctx):
    """-force (no null model in the normalisation, src/qmodel.cpp:2244-2246) with only the forward reference resident: reverse-strand reads
    have no good path at all, their posterior is spread over junk alignments whose states sit far below their rows' maxima, and each
    still gets weight 1."""
    rng = np.random.default_rng(71)
    ref = rand_seq(rng, 2500)
    sc, null = O.Scores(O.Params.from_json(DEFAULT_JSON)), O.NullParams.from_json(NULL_JSON)
    reads = []
    for k in range(24):
        s0 = int(rng.integers(0, len(ref) - 400))
        src = ref[s0:s0 + 400]
        if k % 3 != 0:
            src = O.revcomp_str(src)                      # wrong strand for the only reference
        seq = mutate(rng, src)
        reads.append(O.FastSeq("r%d" % k, seq, rand_qual(rng, len(seq))))
    res, want = run_case(ctx, [O.FastSeq("ref", ref)], reads, sc, null, force=True)
    assert np.all(np.isfinite(res["read_loglike"])) and np.all(res["weight"] > 0.999)
    res, want = run_case(ctx, [O.FastSeq("ref", ref)], reads, sc, null, cfg_kw=dict(sparse=False), force=True)   # and through the row-space kernels


def test_read_with_a_long_phred2_stretch(ctx):
    """200 consecutive bases of quality 2 inside otherwise ordinary reads: over the stretch the emissions barely distinguish the bases, many
    paths through the band carry comparable mass, and the per-row anchor of the Forward storage moves between states."""
    rng = np.random.default_rng(72)
    ref = rand_seq(rng, 3000)
    sc, null = O.Scores(O.Params.from_json(DEFAULT_JSON)), O.NullParams.from_json(NULL_JSON)
    reads = []
    for k in range(16):
        s0 = int(rng.integers(0, len(ref) - 600))
        seq = mutate(rng, ref[s0:s0 + 600], sub=0.06, ins=0.04, dele=0.04)
        a = int(rng.integers(50, len(seq) - 260))
        noisy = "".join("ACGT"[i] for i in rng.integers(0, 4, 200))            # and what the low qualities cover is noise
        seq = seq[:a] + noisy + seq[a + 200:]
        qual = rand_qual(rng, len(seq))
        qual = qual[:a] + chr(33 + 2) * 200 + qual[a + 200:]
        if k & 1:
            seq, qual = O.revcomp_str(seq), qual[::-1]
        reads.append(O.FastSeq("r%d" % k, seq, qual))
    run_case(ctx, both_strands(ref), reads, sc, null)
    pj = synth_params_json(rng, 2, 1)                                            # context-dependent tables (global-memory emission rows)
    ctx.set_params_json(pj)
    try:
        res, want = run_case(ctx, both_strands(ref), reads[:8], O.Scores(O.Params.from_json(pj)), null, force=True)
        assert (np.abs(want) > 1e-6).sum() > 500          # (-force: random parameters would lose every read to the null model)
    finally:
        ctx.set_params_json(None)


def test_two_equal_weight_repeat_bands(ctx):
    """A reference with an exact tandem duplicate of the read's source region: two seeded bands per pair with (nearly) equal Forward mass,
    each taking half the posterior -- neither band is negligible, and the pair's Forward result is a log-sum of two equal terms."""
    rng = np.random.default_rng(73)
    unit = rand_seq(rng, 700)
    ref = rand_seq(rng, 400) + unit + rand_seq(rng, 350) + unit + rand_seq(rng, 300)
    sc, null = O.Scores(O.Params.from_json(DEFAULT_JSON)), O.NullParams.from_json(NULL_JSON)
    reads = []
    for k in range(12):
        s0 = int(rng.integers(0, 250))
        seq = mutate(rng, unit[s0:s0 + 420], sub=0.03, ins=0.02, dele=0.02)
        qual = rand_qual(rng, len(seq))
        if k & 1:
            seq, qual = O.revcomp_str(seq), qual[::-1]
        reads.append(O.FastSeq("r%d" % k, seq, qual))
    import quaff_amd as Q
    res, want = run_case(ctx, both_strands(ref), reads, sc, null)
    nd = [len(ctx.envelope(r, r & 1, Q.DPConfig())) for r in range(len(reads))]
    assert min(nd) > 100                                                        # two bands of ~65+ diagonals on the true strand


def test_estep_totals_do_not_depend_on_order_pieces_or_partition(ctx):
    """The count terms are added as 128-bit fixed-point integers (two 64-bit integer atomics with carry), so an E-step's totals are a
    function of the reads alone: the same words run to run, for every way the call is cut into pieces (pipeline pieces, a memory
    budget that forces splits), and -- added with qf_exact_add -- for the whole batch, two halves, and three contexts taking a
    third each.  (src/qmodel.cpp:2416-2422 adds per-read counts in read order: one fixed answer; this is one too.)  Order-2 contexts
    (LDS accumulators for the context-dependent transitions) and the row-space kernels included."""
    import quaff_amd as Q
    from quaff_amd import api
    rng = np.random.default_rng(81)
    ref = rand_seq(rng, 4000)
    refs = both_strands(ref)
    reads = make_reads(rng, ref, 300, 350)
    for k in range(0, 300, 3):            # ragged lengths (40 .. 700): which bands share a wavefront changes with every way of cutting the batch
        L = int(rng.integers(40, 700))
        s0 = int(rng.integers(0, len(ref) - L))
        src = ref[s0:s0 + L] if k & 1 == 0 else O.revcomp_str(ref[s0:s0 + L])
        seq = mutate(rng, src)
        reads[k] = O.FastSeq("x%d" % k, seq, rand_qual(rng, len(seq)))
    null = O.NullParams.from_json(NULL_JSON)
    pj = synth_params_json(rng, 2, 1)

    def words(res):
        return np.concatenate([res["counts_exact"], res["loglike_exact"].reshape(1, 2)])

    def run(c, lo, hi, cfg, force=False):
        c.upload_reads([r.seq for r in reads[lo:hi]], [r.qual for r in reads[lo:hi]])
        return c.count_resident(cfg, force=force)

    others = [Q.Context(0), Q.Context(0)]
    try:
        for params, cfg, force in ((None, Q.DPConfig(), False), (pj, Q.DPConfig(), True), (None, Q.DPConfig(sparse=False), False)):
            n = 300 if cfg.sparse else 24
            for c in [ctx] + others:
                c.set_params_json(params)
                c.set_null_json(NULL_JSON)
                c.set_refs([x.seq for x in refs])
            whole = run(ctx, 0, n, cfg, force)
            assert (whole["counts"] > 1e-6).sum() > 300 and np.array_equal(api.exact_to_double(whole["counts_exact"]), whole["counts"])
            again = run(ctx, 0, n, cfg, force)
            assert np.array_equal(words(again), words(whole))                       # run to run
            for pieces, budget in ((2, 0), (5, 0), (1, (48 << 20) if cfg.sparse else (400 << 20))):   # internal pieces; a budget that forces splits
                ctx.set_pipeline_chunks(pieces)
                ctx.set_memory_budget(budget)
                cut = run(ctx, 0, n, cfg, force)
                ctx.set_pipeline_chunks(0)
                ctx.set_memory_budget(0)
                assert np.array_equal(words(cut), words(whole)), (pieces, budget)
                assert np.array_equal(cut["counts"], whole["counts"]) and np.array_equal(cut["read_loglike"], whole["read_loglike"])
            half = api.exact_add(words(run(ctx, 0, n // 2, cfg, force)), words(run(ctx, n // 2, n, cfg, force)))
            assert np.array_equal(half, words(whole))                               # two halves
            cuts = [0, n // 3, 2 * n // 3, n]
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(3) as ex:                                       # three contexts at once on the device
                parts = list(ex.map(lambda k: words(run(([ctx] + others)[k], cuts[k], cuts[k + 1], cfg, force)), range(3)))
            third = api.exact_add(api.exact_add(parts[0], parts[1]), parts[2])
            assert np.array_equal(third, words(whole))
            vals = api.exact_to_double(third)
            assert np.array_equal(vals[:-1], whole["counts"]) and abs(vals[-1] - whole["loglike"]) <= 1e-12 * abs(whole["loglike"])
    finally:
        for c in others:
            c.close()
        ctx.set_params_json(None)
        ctx.set_pipeline_chunks(0)
        ctx.set_memory_budget(0)

def test_count_flush_paths_and_split_backward_give_the_same_words(ctx):
    """Backward leaves its per-column count sums in the Forward rows and k_count_flush adds them up: through a table in
    LDS that holds all the match-emission rows in use, through slices of them (a 12 KB table: QF_DEBUG_FLUSH_SLICES), or straight to
    the global accumulators (tables too large to slice: QF_DEBUG_FLUSH_GLOBAL); the class with the most cells runs its Backward as
    two launches once it has 8 192 bands (QF_DEBUG_NO_BACKWARD_SPLIT: one).  The terms are the same fixed-point words on every path,
    so the totals must be the same 128-bit integers -- and close to the oracle's."""
    import quaff_amd as Q
    rng = np.random.default_rng(97)
    ref = rand_seq(rng, 3000)
    refs = both_strands(ref)
    reads = make_reads(rng, ref, 9000, 110)              # 9 000 short reads: one banded unit each on the strand they come from
    null = O.NullParams.from_json(NULL_JSON)
    pj = synth_params_json(rng, 2, 1)                    # order-2 match contexts (many emission rows), gap contexts

    def words(res):
        return np.concatenate([res["counts_exact"], res["loglike_exact"].reshape(1, 2)])

    try:
        ctx.set_params_json(pj)
        ctx.set_null_json(NULL_JSON)
        ctx.set_refs([x.seq for x in refs])
        ctx.upload_reads([r.seq for r in reads], [r.qual for r in reads])
        cfg = Q.DPConfig()
        base = ctx.count_resident(cfg)
        assert max(c["units"] for c in base["classes"]) >= 8192           # the split applies
        assert (base["counts"] > 1e-6).sum() > 300
        for flags, what in ((1048576, "global"), (2097152, "slices"), (524288, "one launch"), (524288 | 2097152, "one launch, slices")):
            ctx.set_debug_flags(flags)
            got = ctx.count_resident(cfg)
            ctx.set_debug_flags(0)
            assert np.array_equal(words(got), words(base)), what
        # against the oracle on a sample of the reads (the whole batch would take minutes of CPU)
        sample = reads[:60]
        ctx.upload_reads([r.seq for r in sample], [r.qual for r in sample])
        got = ctx.count_resident(cfg)
        want, _, _ = oracle_estep(refs, sample, O.Scores(O.Params.from_json(pj)), null, O.DPConfig())
        assert_counts_close(got["counts"], want, "flush sample")
    finally:
        ctx.set_debug_flags(0)
        ctx.set_params_json(None)
