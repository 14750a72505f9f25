"""BASELINE config 2 at full size (10 kb reference + reverse complement x 100 000 x 1 kb reads, band 64) on the GPU,
checked through size-independent properties on EVERY read and against the oracle on samples:
  * every read gets exactly one alignment, on its true strand (odd reads are reverse-complemented by the generator);
  * its CIGAR consumes the whole read, and M+D equals the reference span;
  * re-scoring the returned path with the oracle's O(path) recurrence reproduces the Viterbi score bit-for-bit (2 000
    reads), i.e. traceback, coordinates and score are mutually consistent;
  * a full oracle run agrees on 150 reads (score, reference, coordinates, CIGAR);
  * the cell total equals the sum of the per-pair counts; running the same batch twice is bit-identical."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_config2_full_size_properties():
    import quaff_amd as Q
    from quaff_amd import api
    n_reads, read_len = 100000, 1000
    ref = api.synth_ref(1, 10000)
    seq, qual, off = api.synth_reads(2, ref, n_reads, read_len)
    ctx = Q.Context(0)
    null_json = open(os.path.join(GOLDEN, "testquaffnullparams.json")).read()
    ctx.set_params_json(None)
    ctx.set_null_json(null_json)
    refs = [ref, api.revcomp(ref)]
    ctx.set_refs(refs)
    ctx.upload_reads_packed(seq, qual, off)
    raw = ctx.align_resident(Q.DPConfig(), 0, raw=True)
    assert raw.n_alignments == n_reads
    cells = np.ctypeslib.as_array(raw.cells, (2 * n_reads,)).reshape(n_reads, 2).copy()
    vit = np.ctypeslib.as_array(raw.viterbi, (2 * n_reads,)).reshape(n_reads, 2).copy()
    assert int(raw.total_cells) == int(cells.sum()) and 7.0e9 < raw.total_cells < 8.2e9
    lens = np.diff(off).astype(np.int64)
    al = raw.alignments
    runs_all = np.ctypeslib.as_array(raw.cigar_runs, (int(max(al[a].run_offset + al[a].n_runs for a in range(0, n_reads, 997))) + 4096,))
    first = {}
    for a in range(n_reads):
        x = al[a]
        assert x.read == a and x.ref == (a & 1), a                       # one alignment per read, on the true strand
        assert x.viterbi == vit[a, x.ref] and x.viterbi > vit[a, 1 - x.ref]
        assert 1 <= x.x_start <= x.x_end <= 10000
        if a % 50 == 0:                                                   # CIGAR accounting on a 2 000-read sample
            runs = np.ctypeslib.as_array(raw.cigar_runs, (int(x.run_offset + x.n_runs),))[int(x.run_offset):]
            ops, ln = runs & 3, runs >> 2
            assert ln[ops != 2].sum() == lens[a] and ln[ops != 1].sum() == x.x_end - x.x_start + 1
            assert ln.sum() == x.n_columns and ops[0] == 0 and ops[-1] == 0
            first[a] = (x.ref, x.viterbi, x.score, x.x_start, x.x_end, "".join("MID"[int(o)] * int(l) for o, l in zip(ops, ln)))
    # oracle checks on samples
    sc = O.Scores(O.Params.from_json(open(os.path.join(GOLDEN, "defaultparams.json")).read()))
    null = O.NullParams.from_json(null_json)
    xf = O.FastSeq("ref", ref.decode())
    orefs = [xf, xf.revcomp()]
    xtoks = [O.tokens(r.seq) for r in orefs]
    for n, a in enumerate(sorted(first)):
        rd = O.FastSeq("read%d" % a, seq[int(off[a]):int(off[a + 1])].decode(), qual[int(off[a]):int(off[a + 1])].decode())
        rf, v, s, xs, xe, ops = first[a]
        rc = O.ReadCtx(rd, sc)
        assert O.rescore_path(xtoks[rf], rc, sc, xs, ops) == v, a     # path, coordinates and score are consistent
        if n % 14 == 0:                                                # ~150 full oracle runs
            k = O.align_read(orefs, rd, sc, null, O.DPConfig())[0]
            assert (k["ref"], k["raw"], k["score"], k["xStart"], k["xEnd"], k["ops"]) == (rf, v, s, xs, xe, ops), a
    # determinism: same batch again, bit-identical scores
    raw2 = ctx.align_resident(Q.DPConfig(), 0, raw=True)
    vit2 = np.ctypeslib.as_array(raw2.viterbi, (2 * n_reads,)).reshape(n_reads, 2)
    assert np.array_equal(vit, vit2)
    ctx.close()


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs 3, 4 and 5 at their stated shapes.  Each goes through the bench job that produces the published line
# (bench.py), so the parity leg of the bench is what is being tested, plus size-independent properties on everything.
# ---------------------------------------------------------------------------------------------------------------------
def _job(workload, *extra):
    import bench
    import quaff_amd as Q
    from quaff_amd import api, dist
    a = bench.parse(["--workload", workload] + list(extra))
    job = bench.JOBS[workload](a, 0, 1, 0)
    job.setup(Q, api, dist)
    return job


def test_config5_full_dp_100kb_by_5kb():
    """-kmatchoff, 100 kb reference (+ reverse complement) x 5 kb reads: 49 stripes of the row-space Viterbi kernel per pair.
    Two reads x both strands against the oracle with == (score, strand choice, coordinates, CIGAR: 4 x 5e8 cells on the CPU);
    every alignment of a 64-read batch re-scored along its path with the oracle's O(path) recurrence."""
    job = _job("fulldp", "--reads", "64")
    cells = job.step()
    assert cells == 2 * 100000 * int(np.diff(job.off).sum())
    cpu = job.cpu_baseline(2, 4, job.ctx.align_resident(job.cfg, 0, reads_below=2), cfg_kw=dict(sparse=False))
    assert cpu["gpu_parity_mismatches"] == 0, cpu
    res = job.ctx.align_resident(job.cfg, 0)
    assert len(res["alignments"]) == 64
    sc = O.Scores(O.Params.from_json(open(os.path.join(GOLDEN, "defaultparams.json")).read()))
    xf = O.FastSeq("ref", job.ref.decode())
    xtoks = [O.tokens(xf.seq), O.tokens(xf.revcomp().seq)]
    for al in res["alignments"]:
        r = al["read"]
        assert al["ref"] == (r & 1)                                      # the generator reverse-complements odd reads
        rd = O.FastSeq("r", job.seq[int(job.off[r]):int(job.off[r + 1])].decode(), job.qual[int(job.off[r]):int(job.off[r + 1])].decode())
        assert O.rescore_path(xtoks[al["ref"]], O.ReadCtx(rd, sc), sc, al["xStart"], al["ops"]) == al["viterbi"], r
        assert al["viterbi"] == res["viterbi"][r, al["ref"]] and al["viterbi"] > res["viterbi"][r, 1 - al["ref"]]
    job.ctx.close()


def test_config4_train_estep_20k_reads_order2():
    """quaff train E-step, -order 2 (matchOrder 3, gapOrder 2), 10 kb reference x 20 000 x 1 kb reads, two EM iterations (the second on
    the pruned reference order).  Oracle on a 200-read sample: per-read log-likelihood, pruned order and the sample's summed
    counts at 1e-4 relative for every entry above 1e-6 (no floor), max error reported.  On all 20 000 reads: the emission counts
    add up to every read base once per posterior unit; two half batches add up to the whole."""
    job = _job("train")
    n = job.n
    r1 = job.ctx.count_resident(job.cfg, packed_order=True)
    order1 = r1["sort_order"]
    assert int(order1[1].min()) >= 1 and int(order1[1].max()) <= 2
    r2 = job.ctx.count_resident(job.cfg, sort_order=order1, packed_order=True)
    lens = np.diff(job.off).astype(np.float64)
    for res in (r1, r2):
        Km, Kg = 64, 16
        ne = (4 + 4 * Km) * 94
        assert len(res["counts"]) == ne + 4 * Kg + 4
        expect = float((res["weight"].sum(axis=1) * lens).sum())          # each aligned read base is emitted once, by match or insert
        # (the reference's table log-sum-exp drops terms more than 10 below the running sum, up to 4.5e-5 each, so its Forward
        # and Backward passes - and these - agree with each other only to ~1e-4: 1.15e-4 measured here)
        assert abs(res["counts"][:ne].sum() - expect) <= 1e-3 * expect
        assert np.all(np.isfinite(res["read_loglike"])) and abs(res["loglike"] - res["read_loglike"].sum()) <= 1e-12 * abs(res["loglike"])
    # the pruned order drops the wrong strand, whose posterior weight was ~0: the second iteration agrees with the first
    np.testing.assert_allclose(r2["read_loglike"], r1["read_loglike"], rtol=1e-9)
    np.testing.assert_allclose(r2["counts"], r1["counts"], rtol=1e-6, atol=1e-9)
    # two halves add up to the whole (count terms are added with floating-point atomics: reproducible to rounding)
    half = n // 2
    tot = np.zeros_like(r1["counts"])
    ll = 0.0
    for lo, hi in ((0, half), (half, n)):
        b0, b1 = int(job.off[lo]), int(job.off[hi])
        job.ctx.upload_reads_packed(job.seq[b0:b1], job.qual[b0:b1], (job.off[lo:hi + 1] - job.off[lo]).astype(np.uint64))
        rh = job.ctx.count_resident(job.cfg)
        tot += rh["counts"]
        ll += rh["loglike"]
    np.testing.assert_allclose(tot, r1["counts"], rtol=1e-9, atol=1e-12)
    assert abs(ll - r1["loglike"]) <= 1e-12 * abs(ll)
    job.ctx.upload_reads_packed(job.seq, job.qual, job.off)
    # oracle, both iterations, on the first 200 reads
    cpu = job.cpu_baseline(200, 8)
    assert cpu["gpu_parity_mismatches"] == 0, cpu
    assert cpu["max_rel_err_counts"] < 1e-4 and cpu["max_rel_err_read_loglike"] < 1e-4
    print("config 4: max relative error of the count entries above 1e-6: %.3e; of the per-read log-likelihoods: %.3e"
          % (cpu["max_rel_err_counts"], cpu["max_rel_err_read_loglike"]))
    from concurrent.futures import ThreadPoolExecutor
    sc = O.Scores(O.Params.from_json(job.params_json))
    null = O.NullParams.from_json(open(os.path.join(GOLDEN, "testquaffnullparams.json")).read())
    x = O.FastSeq("ref", job.ref.decode())
    refs = [x, x.revcomp()]
    reads = [O.FastSeq("r%d" % k, job.seq[int(job.off[k]):int(job.off[k + 1])].decode(), job.qual[int(job.off[k]):int(job.off[k + 1])].decode())
             for k in range(200)]
    orders = [list(map(int, order1[0][k, :int(order1[1][k])])) for k in range(200)]
    with ThreadPoolExecutor(8) as ex:
        out = list(ex.map(lambda k: O.count_read(refs, reads[k], sc, null, O.DPConfig(), orders[k]), range(200)))
    ylogs = np.array([o[1] for o in out])
    assert np.max(np.abs(r2["read_loglike"][:200] - ylogs) / np.abs(ylogs)) < 1e-4
    for k in range(200):
        assert list(map(int, r2["sort_order"][0][k, :int(r2["sort_order"][1][k])])) == out[k][2], k
    job.ctx.close()


def test_config3_overlap_block_of_50k_reads():
    """quaff overlap, 50 000 x 2 kb reads of a 1 Mb genome, both strands: rows 0..33 of the all-vs-all pair triangle (3.4 M pairs).
    Every pair that yields an alignment with score >= 0 and a random 2 000 of the others against the oracle with ==."""
    job = _job("overlap")
    cells = job.step()
    xs, ys, cs = job.pairs
    assert len(xs) == sum(2 * 50000 - 1 - r for r in range(34)) and cells > 6e9
    cpu = job.cpu_baseline(2000, 8)
    assert cpu["gpu_parity_mismatches"] == 0, cpu
    res = job.ctx.overlap_resident(job.pairs, job.cfg)
    assert len(res["alignments"]) > 1000                               # ~100x coverage: ~40 true overlaps per row and strand
    assert int(res["cells"].sum()) == res["total_cells"] == cells
    for p, al in res["alignments"].items():
        assert al["score"] >= 0 and al["score"] == res["score"][p]
        ops = al["ops"]
        assert ops.count("M") + ops.count("D") == al["xEnd"] - al["xStart"] + 1
        assert ops.count("M") + ops.count("I") == al["yEnd"] - al["yStart"] + 1
    job.ctx.close()
