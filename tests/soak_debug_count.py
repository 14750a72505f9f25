"""Diagnostic companion of tests/soak_count_overlap.py (not collected by pytest): re-runs given count seeds of that soak and prints the
worst count entries against the oracle, with the emission tables in LDS and in global memory, and the same call in pieces.
`python tests/soak_debug_count.py SEED ...` from the repo root."""
import sys, numpy as np
sys.path.insert(0, '.')
import quaff_amd as Q
from oracle import oracle as O
from tests.helpers import rand_seq, mutate, rand_qual
from tests.test_gpu_align import both_strands, NULL_JSON, DEFAULT_JSON, synth_params_json
from tests.test_gpu_count import oracle_estep
c = Q.Context(0); c.set_params_json(None); c.set_null_json(NULL_JSON)
null = O.NullParams.from_json(NULL_JSON)
for seed in map(int, sys.argv[1:]):
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
    kw = dict(kmer_len=int(rng.integers(4, 8)), kmer_threshold=int(rng.integers(3, 25)), band_size=int(rng.integers(6, 100)), local=bool(rng.random() < 0.8))
    if rng.random() < 0.1: kw = dict(sparse=False)
    force = bool(rng.random() < 0.3)
    refs = both_strands(ref)
    c.set_refs([x.seq for x in refs]); c.upload_reads([r.seq for r in reads], [r.qual for r in reads])
    ocfg = O.DPConfig(local=kw.get("local", True), kmer_threshold=kw.get("kmer_threshold", 20), band=kw.get("band_size", 64), kmer_len=kw.get("kmer_len", 6), sparse=kw.get("sparse", True))
    want, ylogs, orders = oracle_estep(refs, reads, sc, null, ocfg, None, use_null=not force)
    for flags in (0, 8):
        c.set_debug_flags(flags)
        res = c.count_resident(Q.DPConfig(**kw), force=force)
        got = res["counts"]
        big = np.abs(want) > 1e-6
        rel = np.abs(got - want)[big] / np.abs(want)[big]
        worst = np.flatnonzero(big)[np.argsort(-rel)[:4]]
        small_bad = np.flatnonzero(~big & (np.abs(got) > 2e-6))
        print("seed", seed, "order", order, kw, "force", force, "flags", flags, "max rel", rel.max() if rel.size else None,
              [(int(i), got[i], want[i]) for i in worst], "small_bad", [(int(i), got[i], want[i]) for i in small_bad[:4]],
              "ll err", float(np.max(np.abs(res["read_loglike"] - ylogs) / np.abs(ylogs))))
    c.set_debug_flags(0)
    # pieces under a memory budget vs the whole call
    if len(sys.argv) > 1 and res["forward_bytes"] > 4096:
        for div in (2, 3, 5, 8):
            try:
                c.set_memory_budget(max(1024, res["forward_bytes"] // div))
                parts = c.count_resident(Q.DPConfig(**kw), force=force)
            except Q.QuaffHipError as e:
                print("  div", div, "refused:", str(e)[:80]); continue
            finally:
                c.set_memory_budget(0)
            dif = np.flatnonzero(parts["counts"] != res["counts"])
            print("  div", div, "forward equal", np.array_equal(parts["forward"], res["forward"]), "orders equal", parts["sort_order"] == res["sort_order"],
                  "counts differing", len(dif), [(int(i), parts["counts"][i], res["counts"][i]) for i in dif[:5]],
                  "exact differing", int((parts["counts_exact"] != res["counts_exact"]).any(axis=1).sum()), "rll equal", np.array_equal(parts["read_loglike"], res["read_loglike"]))
