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
    # the exact (order-free) reduction: terms whose floating-point sum depends on the order, as 128-bit fixed-point words
    from quaff_amd import api
    terms = np.array([1e15 + 0.25, -1e15, 3.0 ** -30, 7.0]) if rank == 0 else np.array([-(1e15 + 0.25), 1e15, 5.0, 2.0 ** -40])
    ex_c, ex_ll, ex_fx = dist.estep_allreduce_exact(api.exact_from_double(terms), api.exact_from_double([-250.5 * (rank + 1)])[0])
    # one file per rank: two processes printing to one pipe can interleave inside a line
    open(sys.argv[1] + "/rank%%d.json" %% rank, "w").write(json.dumps(
        {"rank": rank, "world": world, "lo": lo, "hi": hi, "sum_ok": bool(np.array_equal(tot, np.arange(1888) * 3.0)),
         "ll": ll, "max": mx, "n": n, "ex_c": ex_c.tolist(), "ex_ll": ex_ll, "ex_fx": [[int(a), int(b)] for a, b in ex_fx]}))
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
        # (1e15 + .25) - (1e15 + .25) = 0 exactly, 3^-30 + 5 and 7 + 2^-40 to the last bit (3^-30 itself truncated at 2^-64): no
        # floating-point order of these six additions gives all four
        assert r["ex_c"] == [0.0, 0.0, 5.0 + float(int(3.0 ** -30 * 2 ** 64)) / 2 ** 64, 7.0 + 2.0 ** -40] and r["ex_ll"] == -751.5
        assert r["ex_fx"] == rows[0]["ex_fx"]


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


def test_overlap_row_plan_covers_the_triangle_once():
    """config 3 over 1 / 2 / 3 / 8 ranks: the ranks' row ranges and their sub-blocks tile rows [0, n - 1) exactly once, the pair
    counts add up to the closed form (n - 1)(2n - 1) - (n - 2)(n - 1) / 2, and the ranks' shares differ by less than two rows."""
    from quaff_amd.dist import overlap_row_plan
    for n, rows in ((50000, 49999), (50000, 34), (7, 6), (2, 1), (300, 299)):
        want = sum(2 * n - 1 - r for r in range(rows))
        if rows == n - 1:
            assert want == (n - 1) * (2 * n - 1) - (n - 2) * (n - 1) // 2
        for world in (1, 2, 3, 8):
            covered, pairs, shares = [], 0, []
            for rank in range(world):
                (r0, r1), blocks, np_rank = overlap_row_plan(n, 2 * n, rows, rank, world, 1 << 22)
                assert np_rank == sum(2 * n - 1 - r for r in range(r0, r1))
                if blocks:
                    assert blocks[0][0] == r0 and blocks[-1][1] == r1 and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
                    assert all(b1 > b0 for b0, b1 in blocks)
                else:
                    assert r0 == r1
                covered += blocks
                pairs += np_rank
                shares.append(np_rank)
            assert pairs == want
            flat = [r for b0, b1 in covered for r in range(b0, b1)] if rows < 1000 else None
            if flat is not None:
                assert flat == list(range(rows))
            else:
                assert covered[0][0] == 0 and covered[-1][1] == rows and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
            if rows >= world:
                assert max(shares) - min(shares) <= 2 * (2 * n - 1)
