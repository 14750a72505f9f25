"""world_size-2 gloo test of the N>1 plumbing bench.py and the train E-step use (quaff_amd/dist.py): read sharding is a
partition, the counts all-reduce is a sum, the timing reduction is a max."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import sys, json
    import numpy as np
    sys.path.insert(0, %r)
    from quaff_amd import dist
    rank, world, local = dist.init("gloo")
    lo, hi = dist.shard_range(1001, rank, world)
    counts = np.arange(1888, dtype=np.float64) * (rank + 1)
    tot, ll = dist.estep_allreduce(counts, -100.0 * (rank + 1))
    dist.barrier()
    mx = dist.allreduce_max(1.5 + rank)
    n = dist.allreduce_sum(np.array([hi - lo], dtype=np.float64))[0]
    # one file per rank: two processes printing to one pipe can interleave inside a line
    open(sys.argv[1] + "/rank%%d.json" %% rank, "w").write(json.dumps(
        {"rank": rank, "world": world, "lo": lo, "hi": hi, "sum_ok": bool(np.array_equal(tot, np.arange(1888) * 3.0)),
         "ll": ll, "max": mx, "n": n}))
    dist.finalize()
""") % ROOT


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + (os.getpid() % 2000)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), str(tmp_path)],
                         capture_output=True, text=True, timeout=240, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    rows = [json.loads((tmp_path / ("rank%d.json" % r)).read_text()) for r in range(2)]
    assert sorted(r["rank"] for r in rows) == [0, 1]
    rows.sort(key=lambda r: r["rank"])
    assert rows[0]["lo"] == 0 and rows[0]["hi"] == rows[1]["lo"] and rows[1]["hi"] == 1001
    for r in rows:
        assert r["world"] == 2 and r["sum_ok"] and r["ll"] == -300.0 and r["max"] == 2.5 and r["n"] == 1001


def test_shard_range_partition():
    from quaff_amd.dist import shard_range
    for n in (0, 1, 7, 100, 100001):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_balanced_blocks_partition():
    """Strong-scaled workloads cut their items into contiguous blocks of nearly equal weight (rows of the overlap pair
    triangle by pair count, full-DP reads by cell count): a partition, monotone, and balanced to within one item."""
    from quaff_amd.dist import balanced_blocks
    rng = np.random.default_rng(3)
    n = 50000
    rows = (2 * n - 1 - np.arange(n - 1)).astype(np.float64)          # config 3: row nx holds 2n - 1 - nx pairs
    for world in (1, 2, 3, 8):
        cuts = balanced_blocks(rows, world)
        assert cuts[0] == 0 and cuts[-1] == len(rows) and len(cuts) == world + 1 and np.all(np.diff(cuts) >= 0)
        sums = np.array([rows[cuts[k]:cuts[k + 1]].sum() for k in range(world)])
        assert sums.max() - sums.min() <= 2 * rows.max()
    w = rng.integers(4000, 6000, 257).astype(np.float64)                # config 5: reads of ~5 kb
    cuts = balanced_blocks(w, 8)
    sums = np.array([w[cuts[k]:cuts[k + 1]].sum() for k in range(8)])
    assert cuts[-1] == 257 and sums.max() - sums.min() <= 2 * w.max()
    assert list(balanced_blocks([1, 1, 1], 1)) == [0, 3]
