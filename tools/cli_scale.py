"""Developer aid: `quaff overlap` end to end on 3 000 synthetic 2 kb reads (13.5 M pairs, seven device calls): wall time
including process start, FASTQ parsing and the null-model fit.  python3 tools/cli_scale.py (on a GPU box)"""
import sys, time, subprocess, os
sys.path.insert(0, '.')
from quaff_amd import api
n = 3000
g = api.synth_ref(3, 60000)
seq, qual, off = api.synth_reads(4, g, n, 2000)
with open('/tmp/reads.fq', 'w') as f:
    for k in range(n):
        f.write("@r%d\n%s\n+\n%s\n" % (k, seq[int(off[k]):int(off[k+1])].decode(), qual[int(off[k]):int(off[k+1])].decode()))
for extra in ([], ["-gpus", "1"]):
    t = time.time()
    out = subprocess.run(["quaff_amd/bin/quaff", "overlap", "/tmp/reads.fq"] + extra, capture_output=True, text=True)
    dt = time.time() - t
    print("rc", out.returncode, "pairs", n * (2 * n - 1) - n * (n + 1) // 2, "alignments", out.stdout.count("# STOCKHOLM"), "wall %.2fs" % dt, out.stderr[-300:])
    sys.stdout.flush()
