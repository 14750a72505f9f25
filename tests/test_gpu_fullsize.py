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


def test_config5_full_dp_all_10000_reads():
    """BASELINE config 5 at its stated extent: -kmatchoff, 100 kb reference (+ reverse complement) x 10 000 x 5 kb reads = 1e13 cells,
    49 stripes of the row-space Viterbi kernel per pair, the batch cut into pieces by the device's memory (its 5 TB of packed
    traceback does not fit at once: the budget-halving path under a full batch).  On every read: one alignment, on its true
    strand, better than the other strand, CIGAR accounting; the cell total equals the closed form.  The first 64 alignments equal
    those of the same reads run alone (other pieces, same bits) and are re-scored along their paths with the oracle's O(path)
    recurrence; two reads x both strands against the full oracle with == (4 x 5e8 cells on the CPU)."""
    import time
    import quaff_amd as Q
    from quaff_amd import api
    job = _job("fulldp")
    assert job.n == 10000
    t0 = time.time()
    raw = job.ctx.align_resident(job.cfg, 0, raw=True)
    print("config 5: 10 000 reads, %.3e cells in %.1f s (%.3e cells/s)" % (raw.total_cells, time.time() - t0, raw.total_cells / (time.time() - t0)))
    n = job.n
    lens = np.diff(job.off).astype(np.int64)
    assert int(raw.total_cells) == 2 * 100000 * int(lens.sum()) and raw.total_cells > 9.5e12
    assert raw.n_alignments == n
    vit = np.ctypeslib.as_array(raw.viterbi, (2 * n,)).reshape(n, 2).copy()
    cells = np.ctypeslib.as_array(raw.cells, (2 * n,)).reshape(n, 2)
    assert np.array_equal(cells, np.repeat((100000 * lens)[:, None], 2, axis=1).astype(np.uint64))
    from quaff_amd.api import _Alignment
    AL = np.dtype([("read", "<u4"), ("ref", "<u4"), ("viterbi", "<f8"), ("score", "<f8"), ("x_start", "<u4"), ("x_end", "<u4"),
                   ("n_columns", "<u4"), ("n_runs", "<u4"), ("run_offset", "<u8")])
    import ctypes as C
    assert AL.itemsize == C.sizeof(_Alignment)
    al = np.frombuffer(C.string_at(raw.alignments, n * AL.itemsize), dtype=AL).copy()
    assert np.array_equal(al["read"], np.arange(n)) and np.array_equal(al["ref"], np.arange(n) & 1)   # the generator reverse-complements odd reads
    assert np.array_equal(al["viterbi"], vit[np.arange(n), al["ref"]]) and np.all(al["viterbi"] > vit[np.arange(n), 1 - al["ref"]])
    assert np.all((1 <= al["x_start"]) & (al["x_start"] <= al["x_end"]) & (al["x_end"] <= 100000))
    n_runs = int(al["run_offset"].max() + al["n_runs"][np.argmax(al["run_offset"])])
    runs = np.ctypeslib.as_array(raw.cigar_runs, (n_runs,)).copy()
    order = np.argsort(al["run_offset"], kind="stable")
    starts = al["run_offset"][order].astype(np.int64)
    assert np.array_equal(starts[1:], starts[:-1] + al["n_runs"][order][:-1]) and starts[0] == 0      # the runs tile the array
    ops, ln = runs & 3, (runs >> 2).astype(np.int64)
    seg = lambda v: np.add.reduceat(v, starts)
    y_used, x_used, cols = np.empty(n, np.int64), np.empty(n, np.int64), np.empty(n, np.int64)
    y_used[order], x_used[order], cols[order] = seg(ln * (ops != 2)), seg(ln * (ops != 1)), seg(ln)
    assert np.array_equal(y_used, lens) and np.array_equal(x_used, al["x_end"].astype(np.int64) - al["x_start"] + 1)
    assert np.array_equal(cols, al["n_columns"])
    first = {r: (int(al["ref"][r]), al["viterbi"][r], al["score"][r], int(al["x_start"][r]), int(al["x_end"][r]),
                 "".join("MID"[int(o)] * int(l) for o, l in zip(ops[int(al["run_offset"][r]):int(al["run_offset"][r] + al["n_runs"][r])],
                                                                ln[int(al["run_offset"][r]):int(al["run_offset"][r] + al["n_runs"][r])])))
             for r in range(64)}
    # the same 64 reads alone (one piece instead of a slice of a 300-read piece): bit-identical records
    small = Q.Context(0)
    small.set_params_json(None)
    small.set_null_json(open(os.path.join(GOLDEN, "testquaffnullparams.json")).read())
    small.set_refs([job.ref, api.revcomp(job.ref)])
    b1 = int(job.off[64])
    small.upload_reads_packed(job.seq[:b1], job.qual[:b1], job.off[:65].copy())
    res = small.align_resident(job.cfg, 0)
    small.close()
    sc = O.Scores(O.Params.from_json(open(os.path.join(GOLDEN, "defaultparams.json")).read()))
    xf = O.FastSeq("ref", job.ref.decode())
    xtoks = [O.tokens(xf.seq), O.tokens(xf.revcomp().seq)]
    assert len(res["alignments"]) == 64
    for a in res["alignments"]:
        r = a["read"]
        assert (a["ref"], a["viterbi"], a["score"], a["xStart"], a["xEnd"], a["ops"]) == first[r], r
        rd = O.FastSeq("r", job.seq[int(job.off[r]):int(job.off[r + 1])].decode(), job.qual[int(job.off[r]):int(job.off[r + 1])].decode())
        assert O.rescore_path(xtoks[a["ref"]], O.ReadCtx(rd, sc), sc, a["xStart"], a["ops"]) == a["viterbi"], r
    cpu = job.cpu_baseline(2, 4, res, cfg_kw=dict(sparse=False))
    assert cpu["gpu_parity_mismatches"] == 0, cpu
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


def test_config3_whole_pair_triangle_of_50k_reads():
    """BASELINE config 3 at its stated extent: quaff overlap, 50 000 x 2 kb reads of a 1 Mb genome, both strands, ALL 49 999 rows of
    the scheduler's pair triangle (3 749 925 000 pairs) through qf_overlap_rows, the pairs enumerated and thresholded on the
    device.  On everything: pair count = closed form; every pair has a finite result (the forced diagonal); every returned
    alignment scores >= 0, lies inside its reads, accounts for its CIGAR, and the list is in the scheduler's order.  The
    sampled rows (first 34, last 34 -- the short rows of the triangle -- and 24 between) give the same records alone as inside
    their block, and every one of their alignments, plus 10 000 random pairs through the pair-list entry point, == the oracle
    (bench.py's parity leg)."""
    import time
    job = _job("overlap", "--inflight", "2")
    n, total = job.n, 2 * job.n
    assert (n, job.rows) == (50000, 49999) and job.n_pairs == 3749925000
    ctx = job.ctxs[0]
    t0 = time.time()
    tot = {k: 0 for k in ("n_pairs", "n_finite", "total_cells", "total_diagonals", "n_blocks")}
    checksum = 0
    hits, runs, run_base = [], [], 0
    for b0, b1 in job.blocks:
        res = ctx.overlap_rows(n, b0, b1, job.cfg)
        for k in tot:
            tot[k] += res[k]
        checksum = (checksum + res["result_checksum"]) & ((1 << 64) - 1)
        h = res["hits"]
        h["run_offset"] += run_base
        run_base += len(res["runs"])
        hits.append(h)
        runs.append(res["runs"])
    hits, runs = np.concatenate(hits), np.concatenate(runs)
    dt = time.time() - t0
    print("config 3: whole triangle, %d pairs, %.3e cells, %d alignments in %.1f s on one context (%.3e cells/s, %.3e pairs/s)"
          % (tot["n_pairs"], tot["total_cells"], len(hits), dt, tot["total_cells"] / dt, tot["n_pairs"] / dt))
    assert tot["n_pairs"] == 3749925000 == ctx.L.qf_overlap_rows_pairs(total, 0, n - 1)
    assert tot["n_finite"] == tot["n_pairs"]                       # diagonal 0 is always in the envelope
    assert tot["total_diagonals"] >= tot["n_pairs"] and tot["total_cells"] > 1900 * tot["n_pairs"]
    assert len(hits) > 1_000_000                                   # ~100x coverage: ~40 true overlaps per row and strand
    lens = np.array([len(s) for s in job.seqs], np.int64)
    x, y = hits["x"].astype(np.int64), hits["y"].astype(np.int64)
    assert np.all((x < y) & (x < n - 1) & (y < total)) and np.all(np.diff(x * total + y) > 0)   # the scheduler's order, no pair twice
    assert np.all(hits["score"] >= 0) and np.all(np.isfinite(hits["viterbi"]))
    assert np.all((1 <= hits["x_start"]) & (hits["x_start"] <= hits["x_end"] + 1) & (hits["x_end"] <= lens[x]))
    assert np.all((1 <= hits["y_start"]) & (hits["y_start"] <= hits["y_end"] + 1) & (hits["y_end"] <= lens[y]))
    starts = hits["run_offset"].astype(np.int64)
    assert starts[0] == 0 and np.array_equal(starts[1:], starts[:-1] + hits["n_runs"][:-1]) and starts[-1] + hits["n_runs"][-1] == len(runs)
    ops, ln = runs & 3, (runs >> 2).astype(np.int64)
    seg = lambda v: np.add.reduceat(v, starts)
    assert np.array_equal(seg(ln * (ops != 1)), hits["x_end"].astype(np.int64) - hits["x_start"] + 1)    # M + D = x span
    assert np.array_equal(seg(ln * (ops != 2)), hits["y_end"].astype(np.int64) - hits["y_start"] + 1)    # M + I = y span
    assert np.array_equal(seg(ln), hits["n_columns"])
    # an overlap ends at an end of one of the reads on each side (both ends are free, src/qoverlap.cpp:141,153)
    assert np.all((hits["x_end"] == lens[x]) | (hits["y_end"] == lens[y]))
    # sampled rows alone == the same rows inside their blocks (other blocking, other y chunks in flight)
    rng = np.random.default_rng(5)
    fields = ["x", "y", "viterbi", "score", "x_start", "x_end", "y_start", "y_end", "n_columns", "n_runs"]
    for r in job.sample_rows(rng):
        alone = ctx.overlap_rows(n, r, r + 1, job.cfg)
        sel = hits[hits["x"] == r]
        assert np.array_equal(alone["hits"][fields], sel[fields]), r
        if len(sel):
            assert np.array_equal(alone["runs"], runs[int(sel["run_offset"][0]):int(sel["run_offset"][-1] + sel["n_runs"][-1])]), r
    # the timed step of the bench (raw results, two contexts pulling blocks off one list) sees the same totals
    cells = job.step()
    assert cells == tot["total_cells"] and job.tot["n_hits"] == len(hits) and job.tot["checksum"] == checksum
    cpu = job.cpu_baseline(10000, 8)
    assert cpu["gpu_parity_mismatches"] == 0, cpu
    for c in job.ctxs:
        c.close()
