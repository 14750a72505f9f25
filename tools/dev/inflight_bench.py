import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quaff_amd as Q
from quaff_amd import api
from concurrent.futures import ThreadPoolExecutor
null = open(os.path.join(sys.path[0], "tests/golden/testquaffnullparams.json")).read()
ref = api.synth_ref(1, 10000)
seq, qual, off = api.synth_reads(2, ref, 100000, 1000)
def mk():
    c = Q.Context(0); c.set_params_json(None); c.set_null_json(null); c.set_refs([ref, api.revcomp(ref)]); c.upload_reads_packed(seq, qual, off); return c
for nctx in (1, 2, 3):
    ctxs = [mk() for _ in range(nctx)]
    cfg = Q.DPConfig()
    for c in ctxs: c.align_resident(cfg, 0, raw=True)
    K = 12
    ex = ThreadPoolExecutor(nctx)
    t0 = time.perf_counter()
    futs = [ex.submit(ctxs[k % nctx].align_resident, cfg, 0, True) for k in range(K)]
    cells = sum(int(f.result().total_cells) for f in futs)
    dt = time.perf_counter() - t0
    print("contexts in flight", nctx, "ms/step %.2f" % (dt / K * 1e3), "cells/s %.3e" % (cells / dt))
    for c in ctxs: c.close()
