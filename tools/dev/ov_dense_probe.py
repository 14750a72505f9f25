#!/usr/bin/env python3
"""Developer aid: the 5 000-read slice of config 3 tools/cli_bench.py runs `quaff overlap` on (2 kb reads, 100x coverage of a
100 kb genome: one pair in five overlaps), through qf_overlap_rows, a few times and with different block sizes: per call the
phases, blocks and traceback bytes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quaff_amd as Q
from quaff_amd import api
n = 5000
g = api.synth_ref(3, 20 * n)
seq, qual, off = api.synth_reads(4, g, n, 2000)
seqs = [seq[int(off[k]):int(off[k + 1])] for k in range(n)]
quals = [qual[int(off[k]):int(off[k + 1])] for k in range(n)]
seqs += [api.revcomp(s) for s in seqs]
quals += [q[::-1] for q in quals[:n]]
golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "testquaffnullparams.json")
for blk in (0, 0, 1 << 23, 1 << 24, 18750000):
    ctx = Q.Context(0)
    ctx.set_params_json(None)
    ctx.set_null_json(open(golden).read())
    ctx.upload_reads(seqs, quals)
    ctx.set_score_threshold(0.0)
    if blk: ctx.set_overlap_block_pairs(blk)
    cfg = Q.DPConfig(kmer_threshold=14, band_size=64)
    for rep in range(2):
        t = time.perf_counter()
        r = ctx.overlap_rows(n, 0, n - 1, cfg)
        dt = time.perf_counter() - t
        print("block_pairs %9d rep %d wall %.3f s  blocks %d  tb %.1f GB  ms %s hits %d" % (
            blk, rep, dt, r["n_blocks"], r["traceback_bytes"] / 1e9, {k: round(v) for k, v in r["ms"].items()}, len(r["hits"]) if "hits" in r else -1), flush=True)
    del ctx
