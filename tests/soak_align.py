"""Randomized GPU-vs-oracle soak for the align path (not collected by pytest): `python tests/soak_align.py FIRST LAST` runs
seeds FIRST..LAST-1, each a random reference (sometimes with a tandem repeat), 3-90 ragged reads on both strands, random
-kmatch / -kmatchn / -kmatchband (or memory mode, or -kmatchoff), local or global, with or without qualities, best or
-printall, and compares everything check_against_oracle compares.  This is how the unit-table growth was found."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import quaff_amd as Q
from oracle import oracle as O
from tests.helpers import rand_seq, mutate, rand_qual
from tests.test_gpu_align import check_against_oracle, run_both, both_strands, oracle_model, NULL_JSON, DEFAULT_JSON, synth_params_json
c = Q.Context(0); c.set_params_json(None); c.set_null_json(NULL_JSON)
sc, null = oracle_model()
t0 = time.time()
n_ok = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(1000 + seed)
    order = int(rng.integers(0, 3)) if rng.random() < 0.4 else 0
    pj = DEFAULT_JSON if order == 0 else synth_params_json(rng, order + 1, order)
    c.set_params_json(None if order == 0 else pj)
    sc = O.Scores(O.Params.from_json(pj))
    ref = rand_seq(rng, int(rng.integers(1500, 12000)) if rng.random() < 0.93 else int(rng.integers(200000, 280000)))
    if rng.random() < 0.3:   # repeats
        p = int(rng.integers(0, len(ref) - 400)); ref = ref[:p] + ref[p:p + 300] * 2 + ref[p + 300:]
    n = int(rng.integers(3, 90))
    reads = []
    for k in range(n):
        L = int(rng.integers(30, min(2600, len(ref) - 10)))
        s = int(rng.integers(0, len(ref) - L)); src = ref[s:s + L]
        if rng.random() < 0.5: src = O.revcomp_str(src)
        seq = mutate(rng, src, sub=rng.uniform(0, .1), ins=rng.uniform(0, .06), dele=rng.uniform(0, .06))
        if len(seq) == 0: seq = "A"
        reads.append(O.FastSeq("r%d" % k, seq, rand_qual(rng, len(seq))))
    kw = dict(kmer_len=int(rng.integers(4, 9)) if rng.random() < 0.85 else int(rng.integers(9, 20)), kmer_threshold=int(rng.integers(2, 30)), band_size=int(rng.integers(4, 140)),
              local=bool(rng.random() < 0.8))
    if rng.random() < 0.15: kw = dict(kmer_threshold=-1, max_size=int(rng.integers(1, 300)) * 600 * 24)
    if rng.random() < 0.08 and len(ref) < 20000: kw = dict(sparse=False)   # (full DP of a 250 kb reference: minutes of oracle)
    gpu_only = len(ref) >= 20000 and kw.get('kmer_len', 6) < 8 and kw.get('kmer_threshold', 0) >= 0
    quals = rng.random() < 0.85
    pall = rng.random() < 0.3
    try:
        if gpu_only:   # short k-mers on a 250 kb reference: ~1e9 oracle cells (minutes); the GPU leg alone must still finish
            t1 = time.time()
            res, _ = run_both(c, both_strands(ref), reads, kw, sc, null, flags=1 if pall else 0, quals=quals)
            print("seed", seed, "gpu only: %d records, %d cells in %.1fs" % (len(res["alignments"]), res["total_cells"], time.time() - t1)); sys.stdout.flush()
            continue
        res = check_against_oracle(c, both_strands(ref), reads, kw, sc, null, quals=quals, print_all=pall)
        if rng.random() < 0.4 and res["traceback_bytes"] > 4096:   # the same batch in pieces: small memory budget and / or two slots
            key = lambda a: (a["read"], a["ref"], a["score"], a["xStart"], a["xEnd"], a["cigar"])
            again = None
            try:
                c.set_memory_budget(max(1024, res["traceback_bytes"] // int(rng.integers(2, 9))))
                c.set_pipeline_chunks(int(rng.integers(0, 4)))
                again, _ = run_both(c, both_strands(ref), reads, kw, sc, null, flags=1 if pall else 0, quals=quals)
            except Q.QuaffHipError as e:   # a budget below one read's own need is refused, as documented
                assert "over the memory budget" in str(e)
            finally:
                c.set_memory_budget(0); c.set_pipeline_chunks(0)
            if again is not None:
                assert [key(a) for a in again["alignments"]] == [key(a) for a in res["alignments"]]
                assert np.array_equal(again["viterbi"], res["viterbi"]) and again["total_cells"] == res["total_cells"]
        n_ok += 1
    except Exception as e:
        print("FAIL seed", seed, kw, "n", n, "quals", quals, "printall", pall, type(e).__name__, str(e)[:300]); sys.stdout.flush()
    print("seed", seed, "ok so far", n_ok, "elapsed %.0fs" % (time.time() - t0)); sys.stdout.flush()
print("done", n_ok)
