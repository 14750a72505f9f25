#!/usr/bin/env python3
"""End-to-end timing of the drop-in itself: `quaff_amd/bin/quaff align|overlap` on files, wall time split into the phases the
binary reports under QUAFF_HIP_TIMING=1 (parse / null-model fit / pack / device calls / collect / write).  Run on a GPU box:
    python3 tools/cli_bench.py [--reads 100000] [--overlap-reads 5000] [--out profiles/r04_cli_bench.json]
Config 2's shape for align (one 10 kb reference, 100 k x 1 kb reads, SAM output) and a 5 k-read slice of config 3's for
overlap (2 kb reads from a 100 kb genome: 100x coverage as in config 3; Stockholm output).  Reference: t/quaff.cpp:610-636
(the same commands), src/qmodel.cpp:2570-2600 (the printer)."""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quaff_amd import api

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=100000)
ap.add_argument("--overlap-reads", type=int, default=5000)
ap.add_argument("--repeat", type=int, default=3, help="runs per case: the one with the median wall time is reported, all wall times listed")
ap.add_argument("--out", default="")
ap.add_argument("--tmp", default="/tmp/quaff_cli_bench")
a = ap.parse_args()
os.makedirs(a.tmp, exist_ok=True)
QUAFF = os.path.join(ROOT, "quaff_amd", "bin", "quaff")


def write_fastq(path, seq, qual, off, n):
    with open(path, "w") as f:
        for k in range(n):
            f.write("@r%d\n%s\n+\n%s\n" % (k, seq[int(off[k]):int(off[k + 1])].decode(), qual[int(off[k]):int(off[k + 1])].decode()))


def run(args, env_extra=None):
    recs = sorted((run_once(args, env_extra) for _ in range(max(1, a.repeat))), key=lambda r: r["wall_s"])
    rec = recs[len(recs) // 2]
    rec["wall_s_all"] = [round(r["wall_s"], 4) for r in recs]
    rec["device_call_s_all"] = [round(r.get("device_call_s", 0), 4) for r in recs]
    return rec


def run_once(args, env_extra=None):
    env = dict(os.environ, QUAFF_HIP_TIMING="1")
    env.update(env_extra or {})
    out_path = os.path.join(a.tmp, "out.txt")
    t = time.time()
    with open(out_path, "w") as out:
        p = subprocess.run([QUAFF] + args, stdout=out, stderr=subprocess.PIPE, text=True, env=env)
    wall = time.time() - t
    if p.returncode != 0:
        raise SystemExit("quaff %s failed: %s" % (args[0], p.stderr[-2000:]))
    line = [l for l in p.stderr.splitlines() if l.startswith('{"quaff_hip_timing"')][-1]
    rec = json.loads(line)
    rec["process_wall_s"] = wall
    rec["output_bytes"] = os.path.getsize(out_path)
    return rec


res = {"tool": "tools/cli_bench.py", "binary": "quaff_amd/bin/quaff", "runs": []}
ref = api.synth_ref(1, 10000)
seq, qual, off = api.synth_reads(2, ref, a.reads, 1000)
fa, fq = os.path.join(a.tmp, "ref.fa"), os.path.join(a.tmp, "reads.fq")
open(fa, "w").write(">ref\n" + ref.decode() + "\n")
write_fastq(fq, seq, qual, off, a.reads)
null = os.path.join(ROOT, "tests", "golden", "testquaffnullparams.json")
for label, extra, env in (("align -format sam, one context", [], None),
                          ("align -format sam, QUAFF_HIP_DEVICES=0,0,0", [], {"QUAFF_HIP_DEVICES": "0,0,0"})):
    r = run(["align", fa, fq, "-null", null, "-format", "sam"] + extra, env)
    r.update(label=label, reads=a.reads, input_bytes=os.path.getsize(fq))
    res["runs"].append(r)
    print(json.dumps(r)); sys.stdout.flush()
g = api.synth_ref(3, 20 * a.overlap_reads)
seq, qual, off = api.synth_reads(4, g, a.overlap_reads, 2000)
fq2 = os.path.join(a.tmp, "ovreads.fq")
write_fastq(fq2, seq, qual, off, a.overlap_reads)
for label, env in (("overlap, one context", None), ("overlap, QUAFF_HIP_DEVICES=0,0,0", {"QUAFF_HIP_DEVICES": "0,0,0"})):
    r = run(["overlap", fq2, "-null", null], env)
    n = a.overlap_reads
    r.update(label=label, reads=n, pairs=n * (2 * n - 1) - n * (n + 1) // 2 - 0, input_bytes=os.path.getsize(fq2))
    res["runs"].append(r)
    print(json.dumps(r)); sys.stdout.flush()
if a.out:
    json.dump(res, open(os.path.join(ROOT, a.out) if not os.path.isabs(a.out) else a.out, "w"), indent=1)
