#!/usr/bin/env python3
"""The seeding kernels the headline workloads never reach, once through the library for a profile (src/diagenv.cpp:56-96):
memory mode (`-kmatchmb`: kmer_threshold < 0, max_size set -> k_seed<true>) and a genome-length reference (5 Mb: the diagonal
histogram does not fit LDS -> k_seed_global).  Run under rocprofv3 --kernel-trace --stats (tools/profile.sh has no workload for them):
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_seedmodes -- python3 tools/dev/seed_modes.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import quaff_amd as Q
from quaff_amd import api
null_json = open(os.path.join(ROOT, "tests", "golden", "testquaffnullparams.json")).read()
ctx = Q.Context(0)
ctx.set_params_json(None)
ctx.set_null_json(null_json)
for label, ref_len, n_reads, read_len, cfg in (
        ("memory mode (-kmatchmb 10), 10 kb reference", 10000, 20000, 1000, Q.DPConfig(kmer_threshold=-1, max_size=10 << 20)),
        ("5 Mb reference, threshold mode (global-workspace seeding)", 5000000, 2000, 1000, Q.DPConfig())):
    ref = api.synth_ref(11, ref_len)
    seq, qual, off = api.synth_reads(12, ref, n_reads, read_len)
    ctx.set_refs([ref.decode(), api.revcomp(ref).decode()])
    ctx.upload_reads([seq[int(off[k]):int(off[k + 1])] for k in range(n_reads)], [qual[int(off[k]):int(off[k + 1])] for k in range(n_reads)])
    for rep in range(3):
        t = time.time()
        res = ctx.align_resident(cfg)
        dt = time.time() - t
    print("%s: %d reads, %d cells, %d alignments, %.1f ms per call (seed %.2f ms, fill %.2f ms)" % (
        label, n_reads, res["total_cells"], len(res["alignments"]), dt * 1e3, res["ms"]["seed"], res["ms"]["fill"]))
ctx.close()
