"""Randomized GPU-vs-oracle soak for count (even seeds) and overlap (odd seeds), parameter orders 0..2 (not collected by
pytest): `python tests/soak_count_overlap.py FIRST LAST`.  Found the tie order of the next-iteration reference list."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import quaff_amd as Q
from oracle import oracle as O
from tests.helpers import rand_seq, mutate, rand_qual
from tests.test_gpu_align import both_strands, NULL_JSON, DEFAULT_JSON, synth_params_json
from tests.test_gpu_count import run_case
from tests.test_gpu_overlap import check_overlap
c = Q.Context(0); c.set_params_json(None); c.set_null_json(NULL_JSON)
null = O.NullParams.from_json(NULL_JSON)
t0 = time.time(); ok = 0; bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(5000 + seed)
    order = int(rng.integers(0, 3))
    pj = DEFAULT_JSON if order == 0 else synth_params_json(rng, order + 1, order)
    c.set_params_json(None if order == 0 else pj)
    sc = O.Scores(O.Params.from_json(pj))
    ref = rand_seq(rng, int(rng.integers(600, 3000)))
    n = int(rng.integers(2, 70))
    reads = []
    for k in range(n):
        L = int(rng.integers(25, min(700, len(ref) - 10)))
        s = int(rng.integers(0, len(ref) - L)); src = ref[s:s + L]
        if rng.random() < 0.5: src = O.revcomp_str(src)
        seq = mutate(rng, src, sub=rng.uniform(0, .08), ins=rng.uniform(0, .05), dele=rng.uniform(0, .05)) or "A"
        reads.append(O.FastSeq("r%d" % k, seq, rand_qual(rng, len(seq))))
    try:
        if seed % 2 == 0:
            # (seed 250 - k = 4, threshold 4, dozens of bands per pair - found that Forward's end sum must be one running sum
            # across the bands, as in the reference: DESIGN.md section 4, Forward-Backward)
            kw = dict(kmer_len=int(rng.integers(4, 8)), kmer_threshold=int(rng.integers(3, 25)), band_size=int(rng.integers(6, 100)),
                      local=bool(rng.random() < 0.8))
            if rng.random() < 0.1: kw = dict(sparse=False)
            force = bool(rng.random() < 0.3)
            res, _ = run_case(c, both_strands(ref), reads, sc, null, cfg_kw=kw, force=force)
            if rng.random() < 0.35 and res["forward_bytes"] > 4096:   # the same E-step cut into pieces by a small memory budget
                parts = None
                try:
                    c.set_memory_budget(max(1024, res["forward_bytes"] // int(rng.integers(2, 9))))
                    parts = c.count_resident(Q.DPConfig(**kw), force=force)
                except Q.QuaffHipError as e:   # a budget below one read's own need is refused, as documented
                    assert "over the memory budget" in str(e)
                finally:
                    c.set_memory_budget(0)
                if parts is not None:
                    assert np.array_equal(parts["forward"], res["forward"]) and parts["sort_order"] == res["sort_order"]
                    assert np.array_equal(parts["counts"], res["counts"]) and np.array_equal(parts["counts_exact"], res["counts_exact"])   # fixed-point totals: bit for bit
        else:
            kw = dict(kmer_len=int(rng.integers(4, 8)), kmer_threshold=int(rng.integers(3, 20)), band_size=int(rng.integers(6, 100)))
            if rng.random() < 0.1: kw = dict(sparse=False)
            full, _ = check_overlap(c, reads[:min(n, 14)], pj, kw)
            if rng.random() < 0.5 and full["alignments"]:   # the printer's threshold on the device: exactly the survivors
                sub = reads[:min(n, 14)]
                seqs = sub + [r.revcomp() for r in sub]
                pairs = O.overlap_task_pairs(len(sub), len(seqs))
                scores = sorted(a["score"] for a in full["alignments"].values())
                thr = scores[int(rng.integers(0, len(scores)))]
                try:
                    c.set_score_threshold(thr)
                    cut = c.overlap_resident(pairs, Q.DPConfig(**kw))
                finally:
                    c.set_score_threshold(float("-inf"))
                want = {k: (a["score"], a["ops"]) for k, a in full["alignments"].items() if a["score"] >= thr}
                assert {k: (a["score"], a["ops"]) for k, a in cut["alignments"].items()} == want
                assert np.array_equal(cut["score"], full["score"])
        ok += 1
    except Exception as e:
        bad += 1
        print("FAIL seed", seed, "count" if seed % 2 == 0 else "overlap", "order", order, kw, "n", n, type(e).__name__, str(e)[:300]); sys.stdout.flush()
    if seed % 5 == 0: print("seed", seed, "ok", ok, "bad", bad, "%.0fs" % (time.time() - t0)); sys.stdout.flush()
print("done ok", ok, "bad", bad)
