"""Developer aid (not collected by pytest): re-run one even seed of soak_count_overlap.py with and without debug bit 6
(E-step band shortcuts) and print the worst relative count error against the oracle.  python tests/soak_seed_ab.py SEED"""
import sys, numpy as np
sys.path.insert(0, '.')
import quaff_amd as Q
from oracle import oracle as O
from tests.helpers import rand_seq, mutate, rand_qual
from tests.test_gpu_align import both_strands, NULL_JSON, DEFAULT_JSON, synth_params_json
from tests.test_gpu_count import oracle_estep
seed = int(sys.argv[1])
c = Q.Context(0); c.set_params_json(None); c.set_null_json(NULL_JSON)
null = O.NullParams.from_json(NULL_JSON)
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
ocfg = O.DPConfig(local=kw.get("local", True), kmer_threshold=kw.get("kmer_threshold", 20), band=kw.get("band_size", 64),
                  kmer_len=kw.get("kmer_len", 6), sparse=kw.get("sparse", True))
want, ylogs, _ = oracle_estep(refs, reads, sc, null, ocfg, None, use_null=not force)
scale = max(1.0, np.abs(want).max())
for flags in (0, 64):
    c.set_debug_flags(flags)
    res = c.count_resident(Q.DPConfig(**kw), force=force)
    err = np.abs(res["counts"] - want) / np.maximum(np.abs(want), 1e-3 * scale)
    k = int(err.argmax())
    print("flags", flags, kw, "order", order, "worst rel err %.3e at %d (got %.8g want %.8g)" % (err.max(), k, res["counts"][k], want[k]),
          "loglike err %.2e" % abs(res["loglike"] - ylogs.sum()))
