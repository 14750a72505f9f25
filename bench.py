#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's config 2:

    banded Viterbi align, 1 synthetic 10 kb reference (+ its reverse complement) x 100 k synthetic
    1 kb reads, -kmatchband 64 (k=6, threshold 20), DP cells/s.

A "step" is one pass of the whole hot path (read prep, k-mer seeding, banded Viterbi fill, best
reference per read, traceback to CIGAR runs, results copied to the host) over one batch of reads that
is already resident in HBM.  Weak scaling: every rank owns its own --reads reads (different seeds) and
one GPU; there is no data-path collective (read x reference pairs are independent), torch.distributed is
used only for the barrier and the max-over-ranks time.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel against the HBM roof with the
reference's algorithmic 24 B/cell; `cpu_baseline` times the oracle (a bit-exact CPU port
of the reference algorithm, oracle/) on a bounded sample on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_CELL = 24.0        # 3 fp64 states per DP cell


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=100000, help="reads per GPU (config 2: 100000)")
    ap.add_argument("--read-len", type=int, default=1000)
    ap.add_argument("--ref-len", type=int, default=10000)
    ap.add_argument("--band", type=int, default=64)
    ap.add_argument("--cpu-sample", type=int, default=3000, help="reads in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--workload", default="align", choices=["align", "train", "overlap", "fulldp"],
                    help="align = BASELINE config 2 (the headline metric, default); train / overlap / fulldp = scaled "
                         "versions of configs 4 / 3 / 5 (supplementary lines, same JSON shape)")
    ap.add_argument("--overlap-rows", type=int, default=-1,
                    help="overlap workload: rows [rank*R, rank*R+R) of the pair triangle per step (a rank's block of rows when "
                         "config 3 is sharded by rows); 0 = all pairs of the read set; default 34 (3.4 M pairs) for the "
                         "50 k-read config, all pairs for small --reads")
    ap.add_argument("--overlap-threshold", type=float, default=0.0,
                    help="overlap workload: alignments scoring below it are not traced back (`quaff overlap` prints only "
                         "score >= 0 by default, -threshold; pass -inf for -nothreshold)")
    ap.add_argument("--reference-kernel", action="store_true", help="A/B: use the first-generation fill kernel")
    ap.add_argument("--serial-classes", action="store_true", help="A/B: fill classes one after another on one stream")
    ap.add_argument("--debug-flags", type=int, default=0, help="developer: extra qf_dp_config.reserved bits (A/B switches)")
    ap.add_argument("--chunks", type=int, default=0, help="pieces per batch kept two in flight (0 = library default)")
    ap.add_argument("--align-flags", type=int, default=0, help="developer: QF_ALIGN_* flags (2 = scores only, not a valid bench)")
    ap.add_argument("--single-device", action="store_true",
                    help="testing only: every rank uses GPU 0 (rehearse the N>1 path on a one-GPU box)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = this process's CPU share (at most 16)")
    return ap.parse_args()


def cpu_baseline(ref, seq, qual, off, n_sample, threads, band, gpu_res):
    """Oracle (oracle/, kind "port") on the first n_sample reads, one read per task on a thread pool —
    the reference's execution model (runQuaffAlignmentTasks, src/qmodel.cpp:2870-2882).  Also checks the
    GPU's alignments for those reads bit-for-bit."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    golden = os.path.join(ROOT, "tests", "golden")
    sc = O.Scores(O.Params.from_json(open(os.path.join(golden, "defaultparams.json")).read()))
    null = O.NullParams.from_json(open(os.path.join(golden, "testquaffnullparams.json")).read())
    x = O.FastSeq("ref", ref.decode())
    refs = [x, x.revcomp()]
    cfg = O.DPConfig(band=band)
    reads = [O.FastSeq("read%d" % n, seq[int(off[n]):int(off[n + 1])].decode(), qual[int(off[n]):int(off[n + 1])].decode())
             for n in range(n_sample)]
    O.lib()
    t0 = time.time()
    with ThreadPoolExecutor(threads) as ex:
        out = list(ex.map(lambda r: O.align_read(refs, r, sc, null, cfg), reads))
    dt = time.time() - t0
    cells = 0
    mismatches = 0
    for n, kept in enumerate(out):
        for xi in range(2):
            cells += int(gpu_res["cells"][n, xi])
        g = gpu_res["by_read"].get(n)
        if not kept:
            mismatches += g is not None
            continue
        k = kept[0]
        if g is None or (g["ref"], g["viterbi"], g["xStart"], g["xEnd"], g["cigar"]) != \
                (k["ref"], k["raw"], k["xStart"], k["xEnd"], O.cigar(k["ops"])):
            mismatches += 1
    return {"value": cells / dt, "unit": "DP cells/s", "cores": threads, "kind": "port",
            "sample": "first %d of the rank-0 reads x 2 strands (%d cells), oracle/quaff_oracle.c, %d threads, one read per task"
                      % (n_sample, cells, threads),
            "seconds": round(dt, 3), "gpu_parity_mismatches": mismatches}


def order2_params_json():
    """-order 2 shaped parameters (matchOrder 3, gapOrder 2) made by repeating the built-in order-0/1 values for every
    context, as `quaff train -order 2` would start from a context-free prior."""
    import re
    base = open(os.path.join(ROOT, "tests", "golden", "defaultparams.json")).read()
    bi = re.search(r'"beginInsert": \{ "": ([0-9.e-]+)', base).group(1)
    bd = re.search(r'"beginDelete": \{ "": ([0-9.e-]+)', base).group(1)
    block = base[base.index('"match": {') + len('"match": {'):]
    block = block[block.index('{', 1) + 1:block.rindex('} } }')]           # the four reference-base rows of the "" context
    ctxs = [a + b2 for a in "ACGT" for b2 in "ACGT"]
    gaps = ",".join(' "%s": %s' % (c, bi) for c in ctxs), ",".join(' "%s": %s' % (c, bd) for c in ctxs)
    head = base[:base.index('"beginInsert"')]
    mid = base[base.index('"extendInsert"'):base.index('"match": {')]
    match = ",\n".join('   "%s": {%s }' % (c, block.rstrip().rstrip("}").rstrip() + " }") for c in ctxs)
    return ('{\n  "matchOrder": 3,\n  "gapOrder": 2,\n' + head[head.index('"refBase"') - 2:] + '"beginInsert": {%s },\n  "beginDelete": {%s },\n  '
            % gaps + mid + '"match": {\n' + match + " } }\n")


def extra_workload(a, rank, world, local_rank):
    """Supplementary workloads (scaled configs 3, 4, 5).  One JSON line, same keys as the headline line."""
    import numpy as np
    import quaff_amd as Q
    from quaff_amd import api, dist
    ctx = Q.Context(local_rank)
    null_json = open(os.path.join(ROOT, "tests", "golden", "testquaffnullparams.json")).read()
    ctx.set_null_json(null_json)

    def sync_all():
        if world > 1:
            dist.barrier()

    if a.workload == "train":
        ctx.set_params_json(order2_params_json())
        ref = api.synth_ref(1, a.ref_len)
        ctx.set_refs([ref, api.revcomp(ref)])
        n = a.reads if a.reads != 100000 else 20000
        seq, qual, off = api.synth_reads(2 + rank, ref, n, a.read_len)
        ctx.upload_reads_packed(seq, qual, off)
        cfg = Q.DPConfig(band_size=a.band)
        order = None
        for _ in range(a.warmup):
            order = ctx.count_resident(cfg, packed_order=True)["sort_order"]   # later EM iterations run on the pruned reference order
        sync_all()
        t0 = time.perf_counter()
        cells = 0
        ph = {}
        for _ in range(a.steps):
            res = ctx.count_resident(cfg, sort_order=order, packed_order=True)
            if world > 1:
                dist.estep_allreduce(res["counts"], res["loglike"])   # the E-step's only exchange (RCCL all-reduce)
            cells += res["total_cells"] + res["backward_cells"]
            for k, v in res["ms"].items():
                ph[k] = ph.get(k, 0.0) + v
        sync_all()
        dt = time.perf_counter() - t0
        desc = "config 4 shape: quaff train E-step, -order 2, %d bp ref (+revcomp) x %d x %d bp reads per GPU, band %d" % (a.ref_len, n, a.read_len, a.band)
        metric = "DP cells/sec (Forward + Backward E-step)"
        extra = {"forward_bytes": res["forward_bytes"], "phase_ms": {k: round(v / a.steps, 3) for k, v in ph.items()}}
    elif a.workload == "overlap":
        ctx.set_params_json(None)
        n = a.reads if a.reads != 100000 else 50000            # BASELINE config 3: 50 k reads x 2 kb, 100x coverage of a 1 Mb genome
        rows_per_step = a.overlap_rows if a.overlap_rows >= 0 else (34 if n >= 10000 else 0)
        genome = api.synth_ref(3, max(a.ref_len, 20 * n))
        seq, qual, off = api.synth_reads(4 + rank, genome, n, 2000)
        # SeqList::loadSequences: originals followed by their reverse complements
        seqs = [seq[int(off[k]):int(off[k + 1])] for k in range(n)]
        quals = [qual[int(off[k]):int(off[k + 1])] for k in range(n)]
        seqs += [api.revcomp(s) for s in seqs]
        quals += [q[::-1] for q in quals]
        ctx.upload_reads(seqs, quals)
        if rows_per_step:                        # rows [rank*R, rank*R + R) of the triangle: one block of what a rank of the sharded config 3 owns
            rows = np.arange(rank * rows_per_step, min(n - 1, (rank + 1) * rows_per_step))
            xs = np.concatenate([np.full(2 * n - 1 - r, r) for r in rows])
            ys = np.concatenate([np.arange(r + 1, 2 * n) for r in rows])
            keep = np.ones(len(xs), bool)
        else:
            xs, ys = np.triu_indices(2 * n, 1)   # QuaffOverlapScheduler order: nx < ny, nx an original
            keep = xs < n - 1
        pairs = (xs[keep].astype(np.uint32), ys[keep].astype(np.uint32), (ys[keep] >= n).astype(np.uint8))
        cfg = Q.DPConfig(kmer_threshold=14, band_size=a.band)
        ctx.set_score_threshold(a.overlap_threshold)
        for _ in range(a.warmup):
            ctx.overlap_resident(pairs, cfg, raw=True)
        sync_all()
        t0 = time.perf_counter()
        cells = 0
        ph = {}
        for _ in range(a.steps):
            res = ctx.overlap_resident(pairs, cfg, raw=True)
            cells += int(res.total_cells)
            for k in ("prep", "seed", "fill", "traceback", "total"):
                ph[k] = ph.get(k, 0.0) + getattr(res, "ms_" + k)
        sync_all()
        dt = time.perf_counter() - t0
        desc = "config 3%s: quaff overlap, %d x 2 kb reads from a %d bp genome, both strands, %s (%d pairs per step per GPU)" % (
            "" if n == 50000 else " shape", n, len(genome),
            "%d rows of the all-vs-all pair triangle per step" % rows_per_step if rows_per_step else "all-vs-all", len(pairs[0]))
        metric = "DP cells/sec (overlap Viterbi)"
        extra = {"pairs": len(pairs[0]), "score_threshold": a.overlap_threshold, "alignments": int(res.n_alignments), "phase_ms": {k: round(v / a.steps, 3) for k, v in ph.items()}}
    else:
        ctx.set_params_json(None)
        ref_len = a.ref_len if a.ref_len != 10000 else 100000
        ref = api.synth_ref(1, ref_len)
        ctx.set_refs([ref, api.revcomp(ref)])
        n = a.reads if a.reads != 100000 else 256
        seq, qual, off = api.synth_reads(2 + rank, ref, n, 5000)
        ctx.upload_reads_packed(seq, qual, off)
        cfg = Q.DPConfig(sparse=False)
        for _ in range(a.warmup):
            ctx.align_resident(cfg, 0, raw=True)
        sync_all()
        t0 = time.perf_counter()
        cells = 0
        ph = {}
        for _ in range(a.steps):
            res = ctx.align_resident(cfg, 0, raw=True)
            cells += int(res.total_cells)
            for k in ("prep", "seed", "fill", "traceback", "total"):
                ph[k] = ph.get(k, 0.0) + getattr(res, "ms_" + k)
        sync_all()
        dt = time.perf_counter() - t0
        desc = "config 5 shape: -kmatchoff full DP, %d bp ref (+revcomp) x %d x 5 kb reads per GPU" % (ref_len, n)
        metric = "DP cells/sec (unbanded Viterbi)"
        extra = {"traceback_bytes": int(res.traceback_bytes), "phase_ms": {k: round(v / a.steps, 3) for k, v in ph.items()}}
    if world > 1:
        dt = dist.allreduce_max(dt)
        cells = int(dist.allreduce_sum(np.array([float(cells)]))[0])
    if rank == 0:
        print(json.dumps({"metric": metric, "value": cells / dt, "unit": "DP cells/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                          "config": dict({"workload": desc}, **extra)}))
    ctx.close()
    dist.finalize()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import numpy as np
    import quaff_amd as Q
    from quaff_amd import api, dist
    if a.single_device:
        local_rank = 0
        os.environ["LOCAL_RANK"] = "0"
    if world > 1:
        # nccl == RCCL on ROCm, one process per GPU (RCCL refuses two ranks on one GPU, so the rehearsal uses gloo)
        dist.init("gloo" if a.single_device else "nccl")

    if a.workload != "align":
        return extra_workload(a, rank, world, local_rank)
    ctx = Q.Context(local_rank)
    ctx.set_params_json(None)
    ctx.set_null_json(open(os.path.join(ROOT, "tests", "golden", "testquaffnullparams.json")).read())
    ref = api.synth_ref(1, a.ref_len)
    ctx.set_refs([ref, api.revcomp(ref)])
    seq, qual, off = api.synth_reads(2 + rank, ref, a.reads, a.read_len)
    ctx.upload_reads_packed(seq, qual, off)          # resident in HBM before the timed region
    ctx.set_pipeline_chunks(a.chunks)
    cfg = Q.DPConfig(band_size=a.band, debug_flags=(2 if a.reference_kernel else 0) | (4 if a.serial_classes else 0) | a.debug_flags)

    def sync_all():
        if world > 1:
            dist.barrier()                    # barrier + torch.cuda.synchronize() on both sides

    for _ in range(a.warmup):
        ctx.align_resident(cfg, a.align_flags, raw=True)
    sync_all()
    t0 = time.perf_counter()
    cls_ms, cls_cells, cls_names = {}, {}, {}
    phases = {"prep": 0.0, "seed": 0.0, "fill": 0.0, "traceback": 0.0, "total": 0.0}
    total_cells = 0
    for _ in range(a.steps):
        res = ctx.align_resident(cfg, a.align_flags, raw=True)   # synchronous: returns after results are on the host
        total_cells += int(res.total_cells)
        for k in range(res.n_fill_classes):
            if res.units_class[k]:
                cls_ms[k] = cls_ms.get(k, 0.0) + res.ms_fill_class[k]
                cls_cells[k] = int(res.cells_class[k])
        for p in phases:
            phases[p] += getattr(res, "ms_" + p)
        tb_bytes, n_units, n_align = int(res.traceback_bytes), int(res.n_units), int(res.n_alignments)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = dist.allreduce_max(dt)
        total_cells = int(dist.allreduce_sum(np.array([float(total_cells)]))[0])
    if rank != 0:
        dist.finalize()
        return

    dom = max(cls_cells, key=lambda k: cls_cells[k])   # the class that does most of the work (the others run beside it)
    dom_ms = cls_ms[dom] / a.steps
    dom_name = ctx.L.qf_fill_class_name(dom).decode()
    achieved = BYTES_PER_CELL * cls_cells[dom] / (dom_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("kernel") == dom_name and tj.get("reads") == a.reads and tj.get("read_len") == a.read_len:
            traffic = tj.get("hbm_bytes_per_launch")
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "cells_per_launch": cls_cells[dom], "bytes_per_cell": BYTES_PER_CELL, "ms_per_launch": round(dom_ms, 4)}
    if not a.serial_classes and len(cls_ms) > 1:
        # In the timed region the fill classes run on concurrent streams, so the dominant kernel's event-bracketed duration
        # includes the share of the GPU the other classes took.  One extra, untimed pass with the classes one after
        # another gives the same kernel's duration alone (what rocprofv3 --stats shows for a serialised run).
        scfg = Q.DPConfig(band_size=a.band, debug_flags=cfg.reserved | 4)
        sres = ctx.align_resident(scfg, a.align_flags, raw=True)
        iso_ms = float(sres.ms_fill_class[dom])
        iso = BYTES_PER_CELL * cls_cells[dom] / (iso_ms * 1e-3) / 1e9
        roofline["concurrent_classes"] = sorted(ctx.L.qf_fill_class_name(k).decode() for k in cls_ms if k != dom)
        roofline["isolated"] = {"ms_per_launch": round(iso_ms, 4), "achieved": round(iso, 1), "frac": round(iso / HBM_PEAK_GBS, 4)}
    # What actually bounds the kernel (informational; the contract's roofline above stays the HBM one): fp64 vector issue.
    # 161 VALU instructions per wavefront step of 5 cells per lane in the fast-path loop (tools/kernel_asm.sh) = 32.2 lane
    # operations per cell; peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (the chip runs ~2.1 GHz under this load).
    if dom_name == "k_viterbi_fill<16,5>":
        ops = 32.2 * cls_cells[dom] / ((roofline.get("isolated", roofline)["ms_per_launch"]) * 1e-3) / 1e12
        roofline["valu_issue"] = {"lane_ops_per_cell": 32.2, "achieved": round(ops, 2), "peak": 39.3, "unit": "T lane-ops/s",
                                  "frac": round(ops / 39.3, 3), "of": "isolated" if "isolated" in roofline else "in-region"}

    cpu = None
    if a.cpu_sample > 0:
        n_s = min(a.cpu_sample, a.reads)
        full = ctx.align_resident(cfg, 0, reads_below=n_s)   # unpacked view of the same batch for the parity check
        full["by_read"] = {al["read"]: al for al in full["alignments"]}
        # the GPU box's CPU share for one GPU is 16 cores; never oversubscribe beyond the affinity mask
        threads = a.cpu_threads or min(16, len(os.sched_getaffinity(0)))
        cpu = cpu_baseline(ref, seq, qual, off, n_s, threads, a.band, full)

    out = {
        "metric": "DP cells/sec (banded Viterbi)", "value": total_cells / dt, "unit": "DP cells/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE config 2: quaff align, 1 x %d bp ref (+revcomp) x %d x %d bp reads per GPU, "
                               "-kmatchband %d, k=6, threshold 20, local, best alignment per read + CIGAR"
                               % (a.ref_len, a.reads, a.read_len, a.band),
                   "reads_per_gpu": a.reads, "pairs_per_gpu": 2 * a.reads, "cells_per_step_per_gpu": total_cells // (a.steps * world),
                   "bands": n_units, "alignments": n_align, "traceback_bytes": tb_bytes,
                   "phase_ms": {k: round(v / a.steps, 3) for k, v in phases.items()},
                   "fill_kernels": {ctx.L.qf_fill_class_name(k).decode(): {"ms": round(cls_ms[k] / a.steps, 4), "cells": cls_cells[k]}
                                    for k in sorted(cls_ms)}},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(out))
    ctx.close()
    dist.finalize()


if __name__ == "__main__":
    main()
