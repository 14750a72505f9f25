#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric (DP cells/s of the k-mer-seeded banded pair-HMM DP) on BASELINE.json's configs.

Default (the headline line, `--workload align`) = BASELINE config 2: banded Viterbi align, 1 synthetic 10 kb reference (+ its
reverse complement) x 100 k synthetic 1 kb reads per GPU, -kmatchband 64 (k = 6, threshold 20).  `--workload train | overlap |
fulldp` = configs 4 / 3 / 5 at their stated extents (supplementary lines, same JSON shape; the driver runs only the default):
train = one E-step over 20 k reads at -order 2; overlap = the WHOLE pair triangle of 50 k reads (3.75e9 pairs) through
qf_overlap_rows (a step takes ~20 s: default --steps 1 --warmup 1); fulldp = all 10 000 reads (1e13 cells, ~15 s a step).

A "step" is one pass of the whole hot path (read prep, k-mer seeding, DP fill, selection, traceback / count reduction, results
on the host) over one batch that is already resident in HBM.  N > 1: one process per GPU.  `python bench.py --gpus N` starts
the N ranks itself (torch.distributed.run as a child process, before anything touches HIP); under the driver's own
`python -m torch.distributed.run ... bench.py --gpus N` the ranks come from the environment.  Read x reference pairs are
independent, so there is no data-path collective: torch.distributed (backend nccl = RCCL) carries the barrier and the
max-over-ranks time; `train` adds the E-step's one exchange, an RCCL all-reduce of the counts' fixed-point words through the
library's own qf_allreduce_counts_exact (the same totals for any number of ranks).

Prints ONE JSON line (rank 0) with
  roofline      the dominant kernel against the roof that binds it: fp64 vector issue for the Viterbi kernels (DESIGN.md 4:
                20 f64 operations per cell, 39.3 T op/s), HBM at the algorithmic 24 B/cell for Forward / Backward (which do
                materialise the matrix); the HBM view (algorithmic GB/s) and the fp64 issue rate MEASURED in the same run are
                always given beside it; `traffic` / `replayed_counters` are replayed, labelled, from the committed rocprofv3
                capture of the same build (profiles/), never measured here;
  cpu_baseline  the oracle (oracle/, a bit-exact CPU port of the reference's algorithm) timed on a bounded sample on this
                box's host cores, which also re-checks the GPU's results for that sample (`gpu_parity_mismatches`).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
F64_PEAK_TOPS = 39.3         # 256 CUs x 4 SIMDs x 16 fp64 lanes x 2.4 GHz (= the 78.6 TFLOP/s FMA peak / 2)
MEASURED_F64_TOPS = None     # set by main(): the fp64 add rate this device sustains (T lane-operations/s), measured live
BYTES_PER_CELL = 24.0        # 3 fp64 states per DP cell
# fp64-rate operations the recurrence needs per cell, whatever the kernel ("arithmetic floors")
F64_OPS_PER_CELL = {"viterbi": 20.0, "forward": 65.0, "backward": 101.0, "overlap": 52.0, "overlap_single": 3.0}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 10 (align, train), 1 (overlap: a step is the whole pair triangle), 2 (fulldp)")
    ap.add_argument("--warmup", type=int, default=None, help="default 3 (align, train), 1 (overlap, fulldp)")
    ap.add_argument("--workload", default="align", choices=["align", "train", "overlap", "fulldp"],
                    help="align = BASELINE config 2 (the headline metric, default); train / overlap / fulldp = configs 4 / 3 / 5")
    ap.add_argument("--reads", type=int, default=0,
                    help="reads: per GPU for align (default 100000, weak scaling); in all for train (20000), overlap (50000) and "
                         "fulldp (256), which are sharded over the ranks (strong scaling, as BASELINE.json states them)")
    ap.add_argument("--read-len", type=int, default=0, help="default 1000 (align, train), 2000 (overlap), 5000 (fulldp)")
    ap.add_argument("--ref-len", type=int, default=0, help="default 10000 (align, train), 100000 (fulldp); overlap: genome = 20 x reads")
    ap.add_argument("--band", type=int, default=64)
    ap.add_argument("--train-order", type=int, default=2, choices=[1, 2], help="train: -order of the parameters (BASELINE config 4: 2)")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="units in the CPU-baseline / parity sample (reads; pairs for overlap); 0 = skip; default per workload")
    ap.add_argument("--overlap-rows", type=int, default=-1,
                    help="overlap: the job is rows [0, R) of the all-vs-all pair triangle (row nx = pairs (nx, ny > nx), both "
                         "strands), cut into per-rank blocks of equal pair count; default: the whole triangle (all n - 1 rows)")
    ap.add_argument("--overlap-block-pairs", type=int, default=1 << 26,
                    help="overlap: pairs per qf_overlap_rows call (the unit the contexts of a rank pull off their shared list)")
    ap.add_argument("--overlap-threshold", type=float, default=0.0,
                    help="overlap: alignments scoring below it are not traced back (`quaff overlap` prints only score >= 0 by "
                         "default, -threshold; pass -inf for -nothreshold)")
    ap.add_argument("--serial-classes", action="store_true", help="A/B: fill classes one after another on one stream")
    ap.add_argument("--debug-flags", type=int, default=0, help="developer: csrc/qf_internal.h switches (A/B)")
    ap.add_argument("--chunks", type=int, default=0, help="pieces per batch kept two in flight (0 = library default)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="align: batches in flight per GPU, one context (its own streams and result buffers) and one host thread each: "
                         "a batch's result copy and its latency-bound kernels run beside the next batch's fill.  1 = one step at a time")
    ap.add_argument("--align-flags", type=int, default=0, help="developer: QF_ALIGN_* flags (2 = scores only, not a valid bench)")
    ap.add_argument("--single-device", action="store_true",
                    help="testing only: every rank uses GPU 0 and the process group is gloo (RCCL wants one GPU per rank)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = this process's CPU share (at most 16)")
    ap.add_argument("--master-port", type=int, default=0, help="--gpus N launcher: rendezvous port (0 = pick a free one)")
    a = ap.parse_args(argv)
    d_steps, d_warm = {"align": (10, 3), "train": (10, 3), "overlap": (1, 1), "fulldp": (2, 1)}[a.workload]
    a.steps = d_steps if a.steps is None else a.steps
    a.warmup = d_warm if a.warmup is None else a.warmup
    return a


def spawn_ranks(a):
    """`python bench.py --gpus N` with no rank environment: start N ranks (one per GPU) as a child torch.distributed.run and
    relay its output.  This process has not imported anything that touches HIP and never will."""
    import socket
    port = a.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------ helpers
def golden(name):
    return open(os.path.join(ROOT, "tests", "golden", name)).read()


def order2_params_json(order=2):
    """-order 2 shaped parameters (matchOrder 3, gapOrder 2; `order` = 1: matchOrder 2, gapOrder 1) made by repeating the built-in
    order-0/1 values for every context, as `quaff train -order 2` would start from a context-free prior."""
    import re
    import itertools
    base = golden("defaultparams.json")
    bi = re.search(r'"beginInsert": \{ "": ([0-9.e-]+)', base).group(1)
    bd = re.search(r'"beginDelete": \{ "": ([0-9.e-]+)', base).group(1)
    block = base[base.index('"match": {') + len('"match": {'):]
    block = block[block.index('{', 1) + 1:block.rindex('} } }')]           # the four reference-base rows of the "" context
    ctxs = ["".join(t) for t in itertools.product("ACGT", repeat=order)]
    gaps = ",".join(' "%s": %s' % (c, bi) for c in ctxs), ",".join(' "%s": %s' % (c, bd) for c in ctxs)
    head = base[:base.index('"beginInsert"')]
    mid = base[base.index('"extendInsert"'):base.index('"match": {')]
    match = ",\n".join('   "%s": {%s }' % (c, block.rstrip().rstrip("}").rstrip() + " }") for c in ctxs)
    return ('{\n  "matchOrder": %d,\n  "gapOrder": %d,\n' % (order + 1, order) + head[head.index('"refBase"') - 2:] + '"beginInsert": {%s },\n  "beginDelete": {%s },\n  '
            % gaps + mid + '"match": {\n' + match + " } }\n")


def api_revcomp(seq):
    from quaff_amd import api
    return api.revcomp(seq)


def cpu_threads(a):
    # the GPU box's CPU share for one GPU is 16 cores; never oversubscribe beyond the affinity mask
    return a.cpu_threads or min(16, len(os.sched_getaffinity(0)))


def pmc_for(workload, kernel_prefix):
    """HBM bytes per launch and issue counters of one kernel from the newest committed rocprofv3 --pmc summary of this workload
    (profiles/r<NN>_pmc_<workload>.json, written by tools/pmc_summary.py): (row, provenance) or (None, None).  These are NOT
    measured in the run that prints them -- counter passes serialise kernels and need the profiler -- so they are replayed,
    labelled as such, and only while the device code is still the one they were captured on (`source_hash`)."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_%s.json" % workload)))
    if not paths:
        return None, None
    doc = json.load(open(paths[-1]))
    from quaff_amd import api
    prov = {"file": os.path.relpath(paths[-1], ROOT), "captured_on_source_hash": doc.get("source_hash"), "captured_at_git": doc.get("git_head"),
            "current_source_hash": api.kernel_source_hash()}
    if doc.get("source_hash") != prov["current_source_hash"]:
        prov["stale"] = True           # the kernels changed since the capture: nothing is replayed
        return None, prov
    want = kernel_prefix.replace(" ", "")
    rows = doc.get("kernels", [])
    for k in rows:
        if want in k["kernel"].replace(" ", ""):
            return k, prov
    if want.endswith(">"):            # a template whose trailing arguments the caller did not spell out
        for k in rows:
            if want[:-1] + "," in k["kernel"].replace(" ", ""):
                return k, prov
    return None, prov


# A single-diagonal overlap band materialises neither the matrix nor a traceback, and its x emission rows and its own column
# offsets come out of L2 (2.8 GB of HBM traffic per 2^24-pair block, profiles/r03_pmc_overlap.json): no HBM roof binds it.  What
# every cell does need is one 8-byte gather of its pair emission from the staged rows in LDS and the recurrence's
# fp64 operations; its roofline is quoted against the smaller of the two hardware fractions, the LDS one: 256 B per clock per CU
# (MI355X_MICROARCH.md, LDS) x 256 CUs x 2.4 GHz.
LDS_BYTES_PER_CELL = {"overlap_single": 8.0}
LDS_PEAK_GBS = 256 * 256 * 2.4


def roofline_entry(workload, kernel, kind, cells, ms, bound, extra=None):
    """The roofline object of one kernel launch: `bound` says which roof `achieved / peak / frac` are quoted against; both
    views are spelled out beside it."""
    ops = F64_OPS_PER_CELL[kind]
    t = ms * 1e-3
    alg_gbs = BYTES_PER_CELL * cells / t / 1e9
    lds_bpc = LDS_BYTES_PER_CELL.get(kind)
    valu = ops * cells / t / 1e12
    pmc, prov = pmc_for(workload, kernel.replace(" ", ""))
    if pmc and abs(pmc.get("cells_per_launch", 0) - cells) > 0.01 * cells:
        pmc = None                    # the committed capture is of another problem size (other --reads, another rank count)
        prov = dict(prov, other_problem_size=True)
    traffic = pmc.get("hbm_bytes_per_pass", pmc.get("hbm_bytes_per_launch")) if pmc else None   # (per_pass: a class run as several launches per step)
    out = {"bound": bound, "kernel": kernel, "cells_per_launch": int(cells), "ms_per_launch": round(ms, 4)}
    if bound == "fp64_valu":
        out.update(achieved=round(valu, 3), peak=F64_PEAK_TOPS, unit="TFLOP/s", frac=round(valu / F64_PEAK_TOPS, 4),
                   f64_ops_per_cell=ops)
    elif bound == "lds" and lds_bpc:
        gbs = lds_bpc * cells / t / 1e9
        out.update(achieved=round(gbs, 1), peak=LDS_PEAK_GBS, unit="GB/s", frac=round(gbs / LDS_PEAK_GBS, 4), lds_bytes_per_cell=lds_bpc,
                   note="one ds_read_b64 of the pair emission per cell against the CUs' LDS rate (256 B/clk/CU); the smaller of this and "
                        "the fp64 fraction below; nothing this kernel reads has to come from HBM (`hbm` keeps SURVEY 8(d)'s 24 B/cell view)")
    else:
        out.update(achieved=round(alg_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(alg_gbs / HBM_PEAK_GBS, 4),
                   bytes_per_cell=BYTES_PER_CELL)
    out["traffic"] = traffic
    out["traffic_source"] = ("replayed from %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this build, not measured in this run)" % prov["file"]
                             if traffic else None)
    out["hbm"] = {"algorithmic_GBs": round(alg_gbs, 1), "algorithmic_frac": round(alg_gbs / HBM_PEAK_GBS, 4),
                  "bytes_per_cell": BYTES_PER_CELL, "traffic_bytes_per_launch": traffic,
                  "traffic_GBs": round(traffic / t / 1e9, 1) if traffic else None,
                  "traffic_frac": round(traffic / t / 1e9 / HBM_PEAK_GBS, 4) if traffic else None}
    out["fp64_valu"] = {"f64_ops_per_cell": ops, "achieved_Tops": round(valu, 3), "peak_Tops": F64_PEAK_TOPS,
                        "frac": round(valu / F64_PEAK_TOPS, 4)}
    if MEASURED_F64_TOPS:
        # the spec-sheet roof (one fp64 wave-instruction per 4 clocks at 2.4 GHz) beside what this chip sustains on a stream of
        # independent v_add_f64 (measured in this run: qf_debug_measure_f64_rate)
        out["fp64_valu"].update(measured_add_f64_peak_Tops=round(MEASURED_F64_TOPS, 2), frac_of_measured_peak=round(valu / MEASURED_F64_TOPS, 4))
    if prov:
        out["replayed_counters"] = dict(prov, note="counters of a separate rocprofv3 --pmc run on the same build (tools/profile.sh), replayed here; "
                                                   "everything outside this object and `traffic` is measured live")
        if pmc:
            out["replayed_counters"]["counters"] = {k: pmc[k] for k in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
                                                                         "GRBM_GUI_ACTIVE", "valu_busy_frac", "clock_GHz", "valu_insts_per_cell",
                                                                         "wave_cycles_parked_frac", "wave_cycles_issue_stalled_frac",
                                                                         "wave_cycles_lds_stalled_frac") if k in pmc}
    if extra:
        out.update(extra)
    return out


def run_in_flight(job, k):
    """k steps of `job` over its contexts, one host thread per context: the thread of context i runs its share of the steps one
    after another (a context is not re-entrant), the contexts' calls overlap on the device."""
    n = len(job.ctxs)
    share = [k // n + (1 if (i - job.turn) % n < k % n else 0) for i in range(n)]
    job.turn += k
    futs = [job.pool.submit(lambda c=job.ctxs[i], m=share[i]: sum(job.step(c) for _ in range(m))) for i in range(n) if share[i]]
    return sum(f.result() for f in futs)


class Job:
    """One workload on one rank.  setup() leaves everything resident; step() runs the hot path once and returns the cells
    it processed; finish() (rank 0, after the timed region) builds the roofline and the CPU-baseline / parity leg."""
    metric = "DP cells/sec"
    scaling = "weak"

    def __init__(self, a, rank, world, local_rank):
        self.a, self.rank, self.world, self.local_rank = a, rank, world, local_rank
        self.phase = {}
        self.extra = {}

    def add_phases(self, res, names):
        for k in names:
            self.phase[k] = self.phase.get(k, 0.0) + float(getattr(res, "ms_" + k))

    def run_steps(self, k):
        """k steps; returns the cells they processed."""
        return sum(self.step() for _ in range(k))


# ------------------------------------------------------------------------------------------------ config 2: align
class AlignJob(Job):
    metric = "DP cells/sec (banded Viterbi)"

    def setup(self, Q, api, dist):
        a = self.a
        self.n = a.reads or 100000
        self.read_len = a.read_len or 1000
        self.ref_len = a.ref_len or 10000
        self.ref = api.synth_ref(1, self.ref_len)
        self.seq, self.qual, self.off = api.synth_reads(2 + self.rank, self.ref, self.n, self.read_len)
        self.ctxs = []
        for _ in range(max(1, 1 if a.serial_classes else a.inflight)):
            ctx = Q.Context(self.local_rank)
            ctx.set_params_json(None)
            ctx.set_null_json(golden("testquaffnullparams.json"))
            ctx.set_refs([self.ref, api.revcomp(self.ref)])
            ctx.upload_reads_packed(self.seq, self.qual, self.off)          # resident in HBM before the timed region
            ctx.set_pipeline_chunks(a.chunks)
            ctx.set_debug_flags((4 if a.serial_classes else 0) | a.debug_flags)
            self.ctxs.append(ctx)
        self.ctx = self.ctxs[0]
        self.cfg = Q.DPConfig(band_size=a.band)
        self.cls_ms, self.cls_cells = {}, {}
        self.Q = Q
        self.turn = 0
        import threading
        self.lock = threading.Lock()
        if len(self.ctxs) > 1:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(len(self.ctxs))
            for c in self.ctxs:                       # every context allocates its buffers before the warm-up steps
                c.align_resident(self.cfg, a.align_flags, raw=True)

    def reset(self):
        self.cls_ms, self.cls_cells, self.phase = {}, {}, {}

    def run_steps(self, k):
        """k steps, len(self.ctxs) of them in flight: step s runs on context s mod n (each call is synchronous and returns with
        its results on the host; the calls of different contexts overlap on the device)."""
        if len(self.ctxs) == 1:
            return sum(self.step() for _ in range(k))
        return run_in_flight(self, k)

    def step(self, ctx=None):
        res = (ctx or self.ctx).align_resident(self.cfg, self.a.align_flags, raw=True)   # synchronous: returns with the results on the host
        with self.lock:
            for k in range(res.n_fill_classes):
                if res.units_class[k]:
                    self.cls_ms[k] = self.cls_ms.get(k, 0.0) + res.ms_fill_class[k]
                    self.cls_cells[k] = int(res.cells_class[k])
            self.add_phases(res, ("prep", "seed", "fill", "traceback", "total"))
            self.last = (int(res.traceback_bytes), int(res.n_units), int(res.n_alignments))
        return int(res.total_cells)

    def describe(self):
        return ("BASELINE config 2: quaff align, 1 x %d bp ref (+revcomp) x %d x %d bp reads per GPU, -kmatchband %d, k=6, threshold 20, "
                "local, best alignment per read + CIGAR" % (self.ref_len, self.n, self.read_len, self.a.band))

    def kernel_symbol(self, cls):
        name = self.ctx.L.qf_fill_class_name(cls).decode()
        if name.startswith("k_viterbi_fill<"):     # the instantiation that runs for order-0/1 parameters: <G, B, GAPCTX=false, EMLDS=true>
            return "qf::" + name[:-1] + ",false,true>"
        return "qf::" + name + ("<true>" if name in ("k_viterbi_single", "k_viterbi_rows") else "")

    def finish(self, steps):
        a, ctx = self.a, self.ctx
        dom = max(self.cls_cells, key=lambda k: self.cls_cells[k])   # the class that does most of the work (the others run beside it)
        dom_ms = self.cls_ms[dom] / steps
        sym = self.kernel_symbol(dom)
        roof = roofline_entry(a.workload, sym, "viterbi", self.cls_cells[dom], dom_ms, "fp64_valu")
        if not a.serial_classes and len(self.cls_ms) > 1:
            # In the timed region the fill classes run on concurrent streams, so the dominant kernel's event-bracketed duration
            # includes the share of the GPU the other classes took.  One extra, untimed pass with the classes one after
            # another gives the same kernel's duration alone (what rocprofv3 --stats shows for a serialised run).
            ctx.set_debug_flags(4 | a.debug_flags)
            sres = ctx.align_resident(self.cfg, a.align_flags, raw=True)
            ctx.set_debug_flags(a.debug_flags)
            iso_ms = float(sres.ms_fill_class[dom])
            iso = roofline_entry(a.workload, sym, "viterbi", self.cls_cells[dom], iso_ms, "fp64_valu")
            roof["concurrent_kernels"] = sorted(self.kernel_symbol(k) for k in self.cls_ms if k != dom)
            roof["isolated"] = {"ms_per_launch": iso["ms_per_launch"], "achieved": iso["achieved"], "frac": iso["frac"]}
        tb_bytes, n_units, n_align = self.last
        # one step at a time, and one step that also uploads its reads (200 MB of characters and qualities over PCIe)
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.align_resident(self.cfg, a.align_flags, raw=True)
        seq_ms = (time.perf_counter() - t0) / 3 * 1e3
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.upload_reads_packed(self.seq, self.qual, self.off)
            ctx.align_resident(self.cfg, a.align_flags, raw=True)
        up_ms = (time.perf_counter() - t0) / 3 * 1e3
        # ... and the same with the batches in flight: every step uploads its own reads (200 MB over PCIe) into its context and
        # aligns them, len(self.ctxs) contexts at a time -- the end-to-end rate of a caller that streams batches from the host
        up_inflight_ms = None
        if len(self.ctxs) > 1:
            def up_step(c):
                c.upload_reads_packed(self.seq, self.qual, self.off)
                c.align_resident(self.cfg, a.align_flags, raw=True)
            for c in self.ctxs:
                up_step(c)
            t0 = time.perf_counter()
            for f in [self.pool.submit(lambda c=c: [up_step(c) for _ in range(4)]) for c in self.ctxs]:
                f.result()
            up_inflight_ms = (time.perf_counter() - t0) / (4 * len(self.ctxs)) * 1e3
        self.extra = {"reads_per_gpu": self.n, "pairs_per_gpu": 2 * self.n, "bands": n_units, "alignments": n_align,
                      "traceback_bytes": tb_bytes, "batches_in_flight": len(self.ctxs),
                      "ms_per_step_one_at_a_time": round(seq_ms, 3), "ms_per_step_with_upload_one_at_a_time": round(up_ms, 3),
                      "ms_per_step_with_upload_in_flight": round(up_inflight_ms, 3) if up_inflight_ms else None,
                      "fill_kernels": {self.kernel_symbol(k): {"ms": round(self.cls_ms[k] / steps, 4), "cells": self.cls_cells[k]}
                                       for k in sorted(self.cls_ms)}}
        cpu = None
        n_s = min(30000 if a.cpu_sample < 0 else a.cpu_sample, self.n)   # ~10 s of oracle time on the box's 16 host cores
        if n_s > 0:
            full = ctx.align_resident(self.cfg, 0, reads_below=n_s)   # unpacked view of the same batch for the parity check
            cpu = self.cpu_baseline(n_s, cpu_threads(a), full)
        return roof, cpu

    def cpu_baseline(self, n_sample, threads, gpu_res, cfg_kw=None):
        """Oracle (oracle/, kind "port") on the first n_sample reads, one read per task on a thread pool — the reference's
        execution model (runQuaffAlignmentTasks, src/qmodel.cpp:2870-2882).  Also checks the GPU's alignments for those
        reads bit-for-bit."""
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as O
        sc = O.Scores(O.Params.from_json(golden("defaultparams.json")))
        null = O.NullParams.from_json(golden("testquaffnullparams.json"))
        x = O.FastSeq("ref", self.ref.decode())
        refs = [x, x.revcomp()]
        cfg = O.DPConfig(band=self.a.band, **(cfg_kw or {}))
        seq, qual, off = self.seq, self.qual, self.off
        reads = [O.FastSeq("read%d" % n, seq[int(off[n]):int(off[n + 1])].decode(), qual[int(off[n]):int(off[n + 1])].decode())
                 for n in range(n_sample)]
        O.lib()
        t0 = time.time()
        with ThreadPoolExecutor(threads) as ex:
            out = list(ex.map(lambda r: O.align_read(refs, r, sc, null, cfg), reads))
        dt = time.time() - t0
        by_read = {al["read"]: al for al in gpu_res["alignments"]}
        cells = 0
        mismatches = 0
        for n, kept in enumerate(out):
            cells += int(gpu_res["cells"][n].sum())
            g = by_read.get(n)
            if not kept:
                mismatches += g is not None
                continue
            k = kept[0]
            if g is None or (g["ref"], g["viterbi"], g["score"], g["xStart"], g["xEnd"], g["cigar"]) != \
                    (k["ref"], k["raw"], k["score"], k["xStart"], k["xEnd"], O.cigar(k["ops"])):
                mismatches += 1
        return {"value": cells / dt, "unit": "DP cells/s", "cores": threads, "kind": "port",
                "sample": "first %d of the rank-0 reads x 2 strands (%d cells), oracle/quaff_oracle.c, %d threads, one read per task; "
                          "GPU reference choice, score, coordinates and CIGAR compared with == for every one of them"
                          % (n_sample, cells, threads),
                "seconds": round(dt, 3), "gpu_parity_mismatches": mismatches}


# ------------------------------------------------------------------------------------------------ config 5: full DP
class FullDPJob(AlignJob):
    metric = "DP cells/sec (unbanded Viterbi)"
    scaling = "strong"

    def setup(self, Q, api, dist):
        a = self.a
        total = a.reads or 10000                 # BASELINE config 5: all 10 000 reads (1e13 cells), cut into pieces by the memory budget
        self.read_len = a.read_len or 5000
        self.ref_len = a.ref_len or 100000
        self.ctx = ctx = Q.Context(self.local_rank)
        ctx.set_params_json(None)
        ctx.set_null_json(golden("testquaffnullparams.json"))
        self.ref = api.synth_ref(1, self.ref_len)
        ctx.set_refs([self.ref, api.revcomp(self.ref)])
        seq, qual, off = api.synth_reads(2, self.ref, total, self.read_len)      # the whole job; every rank cuts out its own block
        import numpy as np
        cuts = dist.balanced_blocks(np.diff(off).astype(np.float64), self.world)   # cells of a read = its length x the references' lengths
        lo, hi = int(cuts[self.rank]), int(cuts[self.rank + 1])
        b0, b1 = int(off[lo]), int(off[hi])
        self.seq, self.qual, self.off = seq[b0:b1], qual[b0:b1], (off[lo:hi + 1] - off[lo]).astype(np.uint64)
        self.n, self.total_reads = hi - lo, total
        ctx.upload_reads_packed(self.seq, self.qual, self.off)
        ctx.set_debug_flags((4 if a.serial_classes else 0) | a.debug_flags)
        self.cfg = Q.DPConfig(sparse=False)
        self.cls_ms, self.cls_cells = {}, {}
        self.Q = Q
        self.ctxs, self.turn = [ctx], 0
        import threading
        self.lock = threading.Lock()

    def describe(self):
        return ("BASELINE config 5%s: -kmatchoff full DP, %d bp ref (+revcomp) x %d x %d bp reads sharded over %d GPU(s) by cell count "
                "(%d reads on rank 0; a rank's batch is processed in pieces that fit the device's memory)"
                % ("" if (self.total_reads, self.ref_len, self.read_len) == (10000, 100000, 5000) else " shape (10 000 reads in the stated config)",
                   self.ref_len, self.total_reads, self.read_len, self.world, self.n))

    def finish(self, steps):
        a, ctx = self.a, self.ctx
        dom = max(self.cls_cells, key=lambda k: self.cls_cells[k])
        sym = self.kernel_symbol(dom)
        roof = roofline_entry(a.workload, sym, "viterbi", self.cls_cells[dom], self.cls_ms[dom] / steps, "fp64_valu")
        tb_bytes, n_units, n_align = self.last
        self.extra = {"reads_this_rank": self.n, "alignments": n_align, "traceback_bytes": tb_bytes}
        cpu = None
        n_s = min(1 if a.cpu_sample < 0 else a.cpu_sample, self.n)
        if n_s > 0:
            # the sample alone on a second context (the job's context holds the whole batch; running it again for one read's
            # alignment would cost another full step)
            small = self.Q.Context(self.local_rank)
            small.set_params_json(None)
            small.set_null_json(golden("testquaffnullparams.json"))
            small.set_refs([self.ref, api_revcomp(self.ref)])
            b1 = int(self.off[n_s])
            small.upload_reads_packed(self.seq[:b1], self.qual[:b1], self.off[:n_s + 1].copy())
            full = small.align_resident(self.cfg, 0)
            small.close()
            cpu = self.cpu_baseline(n_s, min(cpu_threads(a), 2 * n_s), full, cfg_kw=dict(sparse=False))
            cpu["sample"] = cpu["sample"].replace("one read per task", "one read per task (a read's two strands run one after the other)")
        return roof, cpu


# ------------------------------------------------------------------------------------------------ config 4: train E-step
class TrainJob(Job):
    metric = "DP cells/sec (Forward + Backward E-step)"
    scaling = "strong"

    def setup(self, Q, api, dist):
        a = self.a
        total = a.reads or 20000
        self.read_len = a.read_len or 1000
        self.ref_len = a.ref_len or 10000
        self.ctx = ctx = Q.Context(self.local_rank)
        self.params_json = order2_params_json(a.train_order)
        ctx.set_params_json(self.params_json)
        ctx.set_null_json(golden("testquaffnullparams.json"))
        self.ref = api.synth_ref(1, self.ref_len)
        ctx.set_refs([self.ref, api.revcomp(self.ref)])
        seq, qual, off = api.synth_reads(2, self.ref, total, self.read_len)
        import numpy as np
        lo, hi = dist.shard_range(total, self.rank, self.world)            # read ownership is fixed across EM iterations
        b0, b1 = int(off[lo]), int(off[hi])
        self.seq, self.qual, self.off = seq[b0:b1], qual[b0:b1], (off[lo:hi + 1] - off[lo]).astype(np.uint64)
        self.n, self.total_reads = hi - lo, total
        ctx.upload_reads_packed(self.seq, self.qual, self.off)
        ctx.set_pipeline_chunks(a.chunks)
        ctx.set_debug_flags((4 if a.serial_classes else 0) | a.debug_flags)
        self.cfg = Q.DPConfig(band_size=a.band)
        self.dist = dist
        self.rccl = dist.attach_rccl(ctx) if self.world > 1 or os.environ.get("WORLD_SIZE") else False
        self.order = None
        self.cls = {}
        self.Q = Q

    def reset(self):
        self.phase, self.cls = {}, {}

    def step(self):
        res = self.ctx.count_resident(self.cfg, sort_order=self.order, packed_order=True)
        if self.order is None:
            self.order = res["sort_order"]     # later EM iterations run on the pruned reference order (the warm-up provides it)
        if self.world > 1 or self.rccl:
            # the E-step's only exchange, on the order-free fixed-point words: the same totals for any number of ranks
            self.global_counts, self.global_ll, self.global_fx = self.dist.estep_allreduce_exact(res["counts_exact"], res["loglike_exact"], self.ctx)
        for k, v in res["ms"].items():
            self.phase[k] = self.phase.get(k, 0.0) + v
        for c in res["classes"]:
            e = self.cls.setdefault(c["geometry"], {"ms_forward": 0.0, "ms_backward": 0.0, "cells": c["cells"], "units": c["units"]})
            e["ms_forward"] += c["ms_forward"]
            e["ms_backward"] += c["ms_backward"]
        self.last = res
        return res["total_cells"] + res["backward_cells"]

    def describe(self):
        return ("BASELINE config 4%s: quaff train E-step (Forward-Backward), -order %d, %d bp ref (+revcomp) x %d x %d bp reads sharded over "
                "%d GPU(s), band %d, counts all-reduced by %s" % ("" if (self.total_reads, self.a.train_order) == (20000, 2) else " shape", self.a.train_order, self.ref_len, self.total_reads,
                                                                  self.read_len, self.world, self.a.band,
                                                                  "RCCL (qf_allreduce_counts)" if self.rccl else "the host (one rank)" if self.world == 1 else "gloo (one-GPU rehearsal)"))

    def finish(self, steps):
        a = self.a
        geo = max(self.cls, key=lambda g: self.cls[g]["cells"])
        e = self.cls[geo]
        G, B = geo[geo.index("<") + 1:-1].split(",")
        back_cells = self.last["backward_cells"]
        # the dominant kernel: Backward of the dominant class (it re-reads the Forward matrix: 24 B/cell algorithmic)
        bsym = "qf::k_backward_fill<%s,%s>" % (G, B)
        fsym = "qf::k_forward_fill<%s,%s>" % (G, B)
        bcells = e["cells"] * back_cells / max(1, self.last["total_cells"])     # pruned pairs get no Backward pass
        roof = roofline_entry(a.workload, bsym, "backward", bcells, e["ms_backward"] / steps, "hbm")
        fwd = roofline_entry(a.workload, fsym, "forward", e["cells"], e["ms_forward"] / steps, "hbm")
        roof["forward_kernel"] = {k: fwd[k] for k in ("kernel", "cells_per_launch", "ms_per_launch", "achieved", "frac", "traffic", "fp64_valu")}
        es_ms = (self.phase["forward"] + self.phase["backward"]) / steps
        es_bytes = BYTES_PER_CELL * (self.last["total_cells"] + back_cells)
        roof["estep"] = {"ms": round(es_ms, 3), "algorithmic_bytes": es_bytes, "achieved": round(es_bytes / (es_ms * 1e-3) / 1e9, 1),
                         "frac": round(es_bytes / (es_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "forward_bytes_allocated": self.last["forward_bytes"]}
        self.extra = {"reads_this_rank": self.n, "forward_bytes": self.last["forward_bytes"],
                      "kernels": {g: {"ms_forward": round(v["ms_forward"] / steps, 4), "ms_backward": round(v["ms_backward"] / steps, 4),
                                      "cells": v["cells"], "units": v["units"]} for g, v in sorted(self.cls.items())}}
        cpu = None
        n_s = min(10000 if a.cpu_sample < 0 else a.cpu_sample, self.n)
        if n_s > 0:
            cpu = self.cpu_baseline(n_s, cpu_threads(a))
        return roof, cpu

    def cpu_baseline(self, n_sample, threads):
        """The oracle's QuaffCountingTask (one read per task) on the first n_sample reads of rank 0, timed; the GPU runs the same
        sample alone and its per-read log-likelihoods, next-iteration orders and summed counts are compared at 1e-4 relative."""
        from concurrent.futures import ThreadPoolExecutor
        import numpy as np
        from oracle import oracle as O
        sc = O.Scores(O.Params.from_json(self.params_json))
        null = O.NullParams.from_json(golden("testquaffnullparams.json"))
        x = O.FastSeq("ref", self.ref.decode())
        refs = [x, x.revcomp()]
        cfg = O.DPConfig(band=self.a.band)
        seq, qual, off = self.seq, self.qual, self.off
        reads = [O.FastSeq("read%d" % n, seq[int(off[n]):int(off[n + 1])].decode(), qual[int(off[n]):int(off[n + 1])].decode())
                 for n in range(n_sample)]
        O.lib()
        t0 = time.time()
        with ThreadPoolExecutor(threads) as ex:
            out = list(ex.map(lambda r: O.count_read(refs, r, sc, null, cfg), reads))
        dt = time.time() - t0
        want = np.sum([o[0] for o in out], axis=0)
        ylogs = np.array([o[1] for o in out])
        b1 = int(off[n_sample])
        self.ctx.upload_reads_packed(seq[:b1], qual[:b1], off[:n_sample + 1].copy())
        res = self.ctx.count_resident(self.cfg)
        self.ctx.upload_reads_packed(seq, qual, off)
        mism = int(np.sum(np.abs(res["read_loglike"] - ylogs) > 1e-4 * np.abs(ylogs)))
        mism += sum(1 for r in range(n_sample) if res["sort_order"][r] != out[r][2])
        big = np.abs(want) > 1e-6
        rel = np.abs(res["counts"] - want)[big] / np.abs(want)[big]
        mism += int(np.sum(rel > 1e-4))
        cells = res["total_cells"] + res["backward_cells"]
        return {"value": cells / dt, "unit": "DP cells/s", "cores": threads, "kind": "port",
                "sample": "first %d of the rank-0 reads x 2 strands (%d Forward + Backward cells), oracle/quaff_oracle.c, %d threads, one read "
                          "per task; GPU per-read log-likelihood, pruned reference order and every count entry above 1e-6 compared at 1e-4 relative"
                          % (n_sample, cells, threads),
                "seconds": round(dt, 3), "gpu_parity_mismatches": mism,
                "max_rel_err_counts": float(rel.max()) if rel.size else 0.0,
                "max_rel_err_read_loglike": float(np.max(np.abs(res["read_loglike"] - ylogs) / np.abs(ylogs)))}


# ------------------------------------------------------------------------------------------------ config 3: overlap
class OverlapJob(Job):
    """quaff overlap over rows of QuaffOverlapScheduler's pair enumeration (src/qoverlap.cpp:475-480,528-547), through
    qf_overlap_rows: the pairs are enumerated, thresholded and reduced on the device; what comes back is the alignments scoring
    at least the threshold plus totals.  The default job is the WHOLE triangle of BASELINE config 3 (3.75e9 pairs)."""
    metric = "DP cells/sec (overlap Viterbi)"
    scaling = "strong"

    def setup(self, Q, api, dist):
        import numpy as np
        a = self.a
        self.n = n = a.reads or 50000            # BASELINE config 3: 50 k reads x 2 kb, 100x coverage of a 1 Mb genome
        self.read_len = a.read_len or 2000
        rows = a.overlap_rows if a.overlap_rows > 0 else n - 1     # the scheduler's rows: nx = 0 ... n - 2
        self.rows = rows = min(rows, n - 1)
        self.genome = api.synth_ref(3, max(a.ref_len, 20 * n))
        seq, qual, off = api.synth_reads(4, self.genome, n, self.read_len)
        # SeqList::loadSequences: originals followed by their reverse complements
        self.seqs = [seq[int(off[k]):int(off[k + 1])] for k in range(n)]
        self.quals = [qual[int(off[k]):int(off[k + 1])] for k in range(n)]
        self.seqs += [api.revcomp(s) for s in self.seqs]
        self.quals += [q[::-1] for q in self.quals[:n]]
        # Row nx holds 2n - 1 - nx pairs; ranks take contiguous row blocks of equal pair count, and inside a rank a few contexts
        # (one host thread each; calls are synchronous) pull sub-blocks of rows off a shared list, as the reference's worker
        # threads pull tasks off the scheduler: a block's tails (wide bands, selection, traceback, copies) run beside the next
        # block's seeding.
        self.row_block, self.blocks, self.n_pairs = dist.overlap_row_plan(n, 2 * n, rows, self.rank, self.world, a.overlap_block_pairs)
        self.ctxs = []
        for _ in range(max(1, 1 if a.serial_classes else min(a.inflight, len(self.blocks)))):
            ctx = Q.Context(self.local_rank)
            ctx.set_params_json(None)
            ctx.set_null_json(golden("testquaffnullparams.json"))
            ctx.upload_reads(self.seqs, self.quals)
            ctx.set_score_threshold(a.overlap_threshold)
            ctx.set_debug_flags((4 if a.serial_classes else 0) | a.debug_flags)
            self.ctxs.append(ctx)
        self.ctx = self.ctxs[0]
        self.cfg = Q.DPConfig(kmer_threshold=14, band_size=a.band)
        self.cls = {}
        self.tot = {}
        self.Q = Q
        import threading
        from concurrent.futures import ThreadPoolExecutor
        self.lock = threading.Lock()
        self.pool = ThreadPoolExecutor(len(self.ctxs))
        if self.blocks:
            for c in self.ctxs:                       # every context prepares its reads and buffers before the warm-up steps
                c.overlap_rows(n, self.blocks[0][0], min(self.blocks[0][0] + 2, self.blocks[0][1]), self.cfg, raw=True)

    def reset(self):
        self.phase, self.cls, self.tot = {}, {}, {}

    def run_steps(self, k):
        return sum(self.step() for _ in range(k))

    def step(self):
        """The rank's whole row range once: its sub-blocks dealt to the contexts from a shared list."""
        todo = list(reversed(self.blocks))
        cells = [0]

        def work(ctx):
            while True:
                with self.lock:
                    if not todo:
                        return
                    b0, b1 = todo.pop()
                res = ctx.overlap_rows(self.n, b0, b1, self.cfg, raw=True)
                with self.lock:
                    self.add_phases(res, ("prep", "seed", "fill", "traceback", "total"))
                    for key in ("n_pairs", "n_finite", "n_hits", "total_cells", "total_diagonals", "n_blocks", "traceback_bytes"):
                        self.tot[key] = self.tot.get(key, 0) + int(getattr(res, key))
                    self.tot["checksum"] = (self.tot.get("checksum", 0) + int(res.result_checksum)) & ((1 << 64) - 1)
                    for k in range(res.n_fill_classes):
                        if res.units_class[k]:
                            e = self.cls.setdefault(k, {"ms": 0.0, "cells": 0, "units": 0})
                            e["ms"] += res.ms_fill_class[k]
                            e["cells"] += int(res.cells_class[k])
                            e["units"] += int(res.units_class[k])
                    cells[0] += int(res.total_cells)
        for f in [self.pool.submit(work, c) for c in self.ctxs]:
            f.result()
        return cells[0]

    def describe(self):
        whole = self.n == 50000 and self.rows == self.n - 1
        return ("BASELINE config 3%s: quaff overlap, %d x %d bp reads from a %d bp genome, both strands, %s all-vs-all pair triangle "
                "(%d pairs in all) through qf_overlap_rows, rows sharded over %d GPU(s) by pair count (rows [%d, %d) = %d pairs on rank 0)"
                % ("" if whole else " shape", self.n, self.read_len, len(self.genome),
                   "the whole" if self.rows == self.n - 1 else "rows [0, %d) of the" % self.rows,
                   sum(2 * self.n - 1 - r for r in range(self.rows)), self.world, self.row_block[0], self.row_block[1], self.n_pairs))

    def kernel_symbol(self, cls):
        name = self.ctx.L.qf_fill_class_name(cls).decode()
        if cls == 0:
            return "qf::k_overlap_single_rows"
        if name == "k_viterbi_rows":
            return "qf::k_overlap_rows"
        return "qf::" + name.replace("k_viterbi_fill", "k_overlap_fill")

    def finish(self, steps):
        a = self.a
        # the kernel that fills most of the cells (the classes run side by side: a small class's event-bracketed duration is
        # the length of the whole phase, not its own work); per launch = per internal row block
        dom = max(self.cls, key=lambda k: self.cls[k]["cells"])
        e = self.cls[dom]
        launches = max(1, self.tot["n_blocks"])
        sym = self.kernel_symbol(dom)
        kind = "overlap_single" if dom == 0 else "overlap"
        roof = roofline_entry(a.workload, sym, kind, e["cells"] / launches, e["ms"] / launches, "fp64_valu" if dom else "lds")
        roof["launches_per_step"] = launches // steps
        if not a.serial_classes and self.blocks:
            # the dominant kernel alone, on one internal block: in the timed region it shares the GPU with the other fill classes
            # and with the other contexts' blocks
            b0 = self.blocks[0][0]
            b1 = b0
            while b1 < self.row_block[1] and sum(2 * self.n - 1 - r for r in range(b0, b1 + 1)) <= (1 << 24):
                b1 += 1
            b1 = max(b1, b0 + 1)
            self.ctx.set_debug_flags(4 | a.debug_flags)
            sres = self.ctx.overlap_rows(self.n, b0, b1, self.cfg, raw=True)
            self.ctx.set_debug_flags(a.debug_flags)
            if sres.units_class[dom]:
                iso = roofline_entry(a.workload, sym, kind, int(sres.cells_class[dom]), float(sres.ms_fill_class[dom]), "fp64_valu" if dom else "lds")
                roof["isolated"] = {"rows": [b0, b1], "cells_per_launch": iso["cells_per_launch"], "ms_per_launch": iso["ms_per_launch"],
                                    "achieved": iso["achieved"], "frac": iso["frac"]}
        t = {k: v // steps if k != "checksum" else v for k, v in self.tot.items()}
        self.extra = {"pairs_this_rank": self.n_pairs, "rows_this_rank": list(self.row_block), "row_blocks": len(self.blocks),
                      "score_threshold": a.overlap_threshold, "contexts_in_flight": len(self.ctxs),
                      "per_step": {"pairs": t["n_pairs"], "pairs_with_a_finite_result": t["n_finite"], "alignments_returned": t["n_hits"],
                                   "envelope_diagonals": t["total_diagonals"], "internal_blocks": t["n_blocks"], "traceback_bytes": t["traceback_bytes"]},
                      "fill_kernels": {self.kernel_symbol(k): {"ms": round(v["ms"] / steps, 3), "cells": v["cells"] // steps, "bands": v["units"] // steps}
                                       for k, v in sorted(self.cls.items())}}
        assert t["n_pairs"] == self.n_pairs, (t["n_pairs"], self.n_pairs)
        cpu = None
        n_s = 10000 if a.cpu_sample < 0 else a.cpu_sample
        if n_s > 0 and self.n_pairs:
            cpu = self.cpu_baseline(n_s, cpu_threads(a))
        return roof, cpu

    def sample_rows(self, rng, n_rows=24, edge=34):
        """Rows whose hits the parity leg checks in full: the first and the last `edge` rows of this rank's range (long rows; the
        short rows at the end of the triangle) and a random n_rows in between."""
        r0, r1 = self.row_block
        rows = set(range(r0, min(r1, r0 + edge))) | set(range(max(r0, r1 - edge), r1))
        mid = [r for r in range(r0, r1) if r not in rows]
        if mid:
            rows |= set(int(r) for r in rng.choice(mid, size=min(n_rows, len(mid)), replace=False))
        return sorted(rows)

    def cpu_baseline(self, n_sample, threads, max_hits=3000):
        """Oracle overlap (QuaffOverlapTask::run, one pair per task), timed, on (a) the alignments qf_overlap_rows returned for a
        sample of rows -- the first 34 and the last 34 rows of the rank's range and 24 random rows between (at most max_hits of
        them, chosen at random) -- compared record for record with ==, and (b) n_sample random pairs of the rank's range pushed through
        the explicit-pair-list entry point (qf_overlap_resident), whose per-pair result, score and cell count are compared with ==
        (most of these score below the threshold: they must have no alignment).  Also checks the two entry points against each
        other on the sampled rows: every pair of a row the oracle scores >= threshold is among the returned hits."""
        from concurrent.futures import ThreadPoolExecutor
        import numpy as np
        from oracle import oracle as O
        params = O.Params.from_json(golden("defaultparams.json"))
        sc = O.Scores(params)
        null = O.NullParams.from_json(golden("testquaffnullparams.json"))
        osc = [O.OverlapScores(params, sc, False), O.OverlapScores(params, sc, True)]
        cfg = O.DPConfig(kmer_threshold=14, band=self.a.band)
        rng = np.random.default_rng(5)
        n, total = self.n, 2 * self.n
        thr = self.a.overlap_threshold
        ctx = self.ctx
        # (a) hits of the sampled rows
        hit_recs = []
        for r in self.sample_rows(rng):
            res = ctx.overlap_rows(n, r, r + 1, self.cfg)
            for h in res["hits"]:
                hit_recs.append((h, ctx.hit_ops(h, res["runs"])))
        n_hits_all = len(hit_recs)
        if len(hit_recs) > max_hits:
            hit_recs = [hit_recs[k] for k in sorted(rng.choice(len(hit_recs), size=max_hits, replace=False))]
        # (b) random pairs of the rank's rows, row chosen by its length
        r0, r1 = self.row_block
        row_len = (total - 1 - np.arange(r0, r1)).astype(np.float64)
        rr = r0 + rng.choice(r1 - r0, size=n_sample, p=row_len / row_len.sum())
        yy = rr + 1 + (rng.random(n_sample) * (total - 1 - rr)).astype(np.int64)
        rnd = sorted(set(zip(rr.tolist(), yy.tolist())))
        xs = np.array([p[0] for p in rnd], np.uint32)
        ys = np.array([p[1] for p in rnd], np.uint32)
        lst = ctx.overlap_resident((xs, ys, (ys >= n).astype(np.uint8)), self.cfg)
        fs = {}

        def seq_of(k):
            if k not in fs:
                fs[k] = O.FastSeq("s%d" % k, self.seqs[k].decode(), self.quals[k].decode())
            return fs[k]
        for h, _ in hit_recs:
            seq_of(int(h["x"])), seq_of(int(h["y"]))
        for x, y in rnd:
            seq_of(x), seq_of(y)
        tasks = [(int(h["x"]), int(h["y"])) for h, _ in hit_recs] + rnd
        O.lib()
        t0 = time.time()
        with ThreadPoolExecutor(threads) as ex:
            out = list(ex.map(lambda p: O.overlap_pair(fs[p[0]], fs[p[1]], p[1] >= n, osc[int(p[1] >= n)], sc, null, cfg), tasks))
        dt = time.time() - t0
        mism, cells = 0, 0
        for (h, ops), want in zip(hit_recs, out[:len(hit_recs)]):
            if want is None or want["score"] < thr:
                mism += 1
                continue
            cells += want["cells"]
            if (h["viterbi"], h["score"], int(h["x_start"]), int(h["x_end"]), int(h["y_start"]), int(h["y_end"]), ops) != \
                    (want["result"], want["score"], want["xStart"], want["xEnd"], want["yStart"], want["yEnd"], want["ops"]):
                mism += 1
        for p, want in enumerate(out[len(hit_recs):]):
            cells += int(lst["cells"][p])
            g = lst["alignments"].get(p)
            if want is None:
                mism += g is not None or np.isfinite(lst["viterbi"][p])
                continue
            if lst["viterbi"][p] != want["result"] or lst["score"][p] != want["score"] or int(lst["cells"][p]) != want["cells"]:
                mism += 1
            elif want["score"] >= thr:
                if g is None or (g["xStart"], g["xEnd"], g["yStart"], g["yEnd"], g["ops"]) != \
                        (want["xStart"], want["xEnd"], want["yStart"], want["yEnd"], want["ops"]):
                    mism += 1
            elif g is not None:
                mism += 1
        return {"value": cells / dt, "unit": "DP cells/s", "cores": threads, "kind": "port",
                "sample": "%d of the %d alignments qf_overlap_rows returned for %d sampled rows of the rank-0 range (first / last 34 rows + 24 random) and "
                          "%d random pairs of the range through qf_overlap_resident (%d cells in all), oracle/quaff_oracle.c, %d threads, one pair per "
                          "task; result, score, coordinates and state path compared with =="
                          % (len(hit_recs), n_hits_all, len(self.sample_rows(np.random.default_rng(5))), len(rnd), cells, threads),
                "seconds": round(dt, 3), "gpu_parity_mismatches": int(mism)}


JOBS = {"align": AlignJob, "fulldp": FullDPJob, "train": TrainJob, "overlap": OverlapJob}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from quaff_amd import dist
    if a.single_device:
        local_rank = 0
        os.environ["LOCAL_RANK"] = "0"
    if "WORLD_SIZE" in os.environ:
        # one process per GPU under torch.distributed.run; nccl == RCCL on ROCm.  torch is imported (and brings its HIP runtime)
        # before libquaffhip is loaded.  RCCL wants a GPU per rank, so the one-GPU rehearsal uses gloo.
        dist.init("gloo" if a.single_device else "nccl")
    import numpy as np
    import quaff_amd as Q
    from quaff_amd import api

    job = JOBS[a.workload](a, rank, world, local_rank)
    job.setup(Q, api, dist)

    def sync_all():
        if "WORLD_SIZE" in os.environ:
            dist.barrier()                    # barrier + torch.cuda.synchronize() on both sides

    rccl = dist.rccl_report(job.ctx, a.single_device) if "WORLD_SIZE" in os.environ else {"ranks": 1, "distinct_devices": 1, "backend": None}
    job.run_steps(a.warmup)
    if hasattr(job, "reset"):
        job.reset()
    sync_all()
    t0 = time.perf_counter()
    total_cells = job.run_steps(a.steps)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = dist.allreduce_max(dt)
        total_cells = int(dist.allreduce_sum(np.array([float(total_cells)]))[0])
    if rank != 0:
        dist.finalize()
        return
    global MEASURED_F64_TOPS
    try:
        MEASURED_F64_TOPS = job.ctx.measure_f64_rate() / 1e12
    except Exception:
        MEASURED_F64_TOPS = None
    roof, cpu = job.finish(a.steps)
    cfg = {"workload": job.describe(), "cells_per_step": total_cells // a.steps,
           "phase_ms": {k: round(v / a.steps, 3) for k, v in job.phase.items()}}
    cfg.update(job.extra)
    out = {"metric": job.metric, "value": total_cells / dt, "unit": "DP cells/s", "n_gpus": world, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": job.scaling,
           "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": cfg, "roofline": roof, "cpu_baseline": cpu, "rccl": rccl}
    if world > 1 and not a.single_device:
        assert rccl["ranks"] == world and rccl["distinct_devices"] == world, rccl
    print(json.dumps(out))
    sys.stdout.flush()
    for c in getattr(job, "ctxs", [job.ctx]):
        c.close()
    dist.finalize()


if __name__ == "__main__":
    main()
