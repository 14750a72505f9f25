"""Randomized GPU-vs-oracle soak for the overlap structures that only long pair lists switch on (not collected by pytest):
the row prefilter of the seeding, the slotted single-diagonal list, sorted band lists, blocks cut by a memory budget.
`python tests/soak_overlap_rows.py FIRST LAST` — 36..72 reads of ragged lengths (some below 2 (k + threshold): full envelope),
random k / threshold / band, parameter orders 0..2; every pair against the oracle with ==, then the same call in pieces, then the
same pairs through qf_overlap_rows (pairs enumerated on the device) with a random internal block size and score threshold: totals,
checksum and every returned record equal to the pair-list entry point's."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import quaff_amd as Q
from oracle import oracle as O
from tests.helpers import rand_seq, mutate, rand_qual
from tests.test_gpu_align import NULL_JSON, DEFAULT_JSON, synth_params_json
from tests.test_gpu_overlap import check_overlap
c = Q.Context(0); c.set_params_json(None); c.set_null_json(NULL_JSON)
t0 = time.time(); ok = 0; bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(9000 + seed)
    order = int(rng.integers(0, 3))
    pj = DEFAULT_JSON if order == 0 else synth_params_json(rng, order + 1, order)
    c.set_params_json(None if order == 0 else pj)
    genome = rand_seq(rng, int(rng.integers(1500, 12000)))
    n = int(rng.integers(36, 73))
    reads = []
    for k in range(n):
        L = int(rng.integers(20, min(600, len(genome) - 10)))
        s = int(rng.integers(0, len(genome) - L)); src = genome[s:s + L]
        if rng.random() < 0.5: src = O.revcomp_str(src)
        seq = mutate(rng, src, sub=rng.uniform(0, .08), ins=rng.uniform(0, .05), dele=rng.uniform(0, .05)) or "A"
        reads.append(O.FastSeq("r%d" % k, seq, rand_qual(rng, len(seq))))
    kw = dict(kmer_len=int(rng.integers(4, 8)), kmer_threshold=int(rng.integers(3, 20)), band_size=int(rng.integers(0, 100)))
    try:
        c.set_debug_flags(512)
        full, _ = check_overlap(c, reads, pj, kw)
        settled = c.rows_settled()
        if rng.random() < 0.6 and full["traceback_bytes"] > 4096:
            seqs = reads + [r.revcomp() for r in reads]
            pairs = O.overlap_task_pairs(n, len(seqs))
            parts = None
            try:
                c.set_memory_budget(max(1024, full["traceback_bytes"] // int(rng.integers(2, 9))))
                parts = c.overlap_resident(pairs, Q.DPConfig(**kw))
            except Q.QuaffHipError as e:
                assert "over the memory budget" in str(e)
            finally:
                c.set_memory_budget(0)
            if parts is not None:
                for key in ("viterbi", "score", "cells", "n_diagonals"):
                    assert np.array_equal(parts[key], full[key]), key
                assert {k: (a["score"], a["ops"]) for k, a in parts["alignments"].items()} == \
                       {k: (a["score"], a["ops"]) for k, a in full["alignments"].items()}
        # the scheduler's enumeration on the device: same pairs, same totals, same records
        thr = float(rng.choice([float("-inf"), 0.0, -50.0]))
        blk = int(rng.choice([0, 1, 37, 500]))
        c.set_score_threshold(thr)
        c.set_overlap_block_pairs(blk)
        try:
            rows = c.overlap_rows(n, 0, n - 1, Q.DPConfig(**kw))
        finally:
            c.set_score_threshold(float("-inf"))
            c.set_overlap_block_pairs(0)
        pairs = O.overlap_task_pairs(n, 2 * n)
        fin = np.isfinite(full["viterbi"])
        assert rows["n_pairs"] == len(pairs) and rows["n_finite"] == int(fin.sum()) and rows["total_cells"] == full["total_cells"]
        assert rows["total_diagonals"] == int(full["n_diagonals"].sum())
        assert rows["result_checksum"] == int(np.sum(full["viterbi"][fin].view(np.uint64), dtype=np.uint64))
        want = [k for k in range(len(pairs)) if k in full["alignments"] and full["alignments"][k]["score"] >= thr]
        assert len(rows["hits"]) == len(want), (len(rows["hits"]), len(want), thr, blk)
        for h, k in zip(rows["hits"], want):
            a = full["alignments"][k]
            assert (int(h["x"]), int(h["y"]), h["viterbi"], h["score"], int(h["x_start"]), int(h["x_end"]), int(h["y_start"]), int(h["y_end"])) == \
                   (pairs[k][0], pairs[k][1], a["result"], a["score"], a["xStart"], a["xEnd"], a["yStart"], a["yEnd"]), k
            assert c.hit_ops(h, rows["runs"]) == a["ops"], k
        ok += 1
        print("seed", seed, "order", order, kw, "n", n, "settled", settled, "of", len(O.overlap_task_pairs(n, 2 * n)), "ok", ok, "bad", bad,
              "%.0fs" % (time.time() - t0)); sys.stdout.flush()
    except Exception as e:
        bad += 1
        print("FAIL seed", seed, "order", order, kw, "n", n, type(e).__name__, str(e)[:300]); sys.stdout.flush()
    finally:
        c.set_debug_flags(0)
print("done ok", ok, "bad", bad)
