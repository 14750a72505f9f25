"""N > 1 on the device: the HIP E-step sharded over two ranks (one process each, both on this box's one GPU, process group
gloo — RCCL wants a GPU per rank) must reproduce the one-rank result; the library's own RCCL path (qf_comm_*,
qf_allreduce_counts) is exercised with a one-rank communicator; `bench.py --gpus 2` launches its ranks itself."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

WORKER = textwrap.dedent("""
    import sys, json, os
    import numpy as np
    sys.path.insert(0, %r)
    from quaff_amd import dist
    rank, world, local = dist.init("gloo")
    import quaff_amd as Q
    from quaff_amd import api
    ctx = Q.Context(0)
    ctx.set_params_json(None)
    ctx.set_null_json(open(%r).read())
    ref = api.synth_ref(1, 6000)
    ctx.set_refs([ref, api.revcomp(ref)])
    seq, qual, off = api.synth_reads(2, ref, 3001, 600)
    lo, hi = dist.shard_range(3001, rank, world)
    b0, b1 = int(off[lo]), int(off[hi])
    ctx.upload_reads_packed(seq[b0:b1], qual[b0:b1], (off[lo:hi + 1] - off[lo]).astype(np.uint64))
    order = None
    out = []
    for it in range(2):                                   # second iteration on the pruned reference order
        res = ctx.count_resident(Q.DPConfig(), sort_order=order, packed_order=True)
        order = res["sort_order"]
        counts, ll, fx = dist.estep_allreduce_exact(res["counts_exact"], res["loglike_exact"], ctx)
        out.append({"counts": counts.tolist(), "ll": ll, "local_ll": res["loglike"], "fx": [[int(a), int(b)] for a, b in fx]})
    dist.barrier()
    open(sys.argv[1] + "/rank%%d.json" %% rank, "w").write(json.dumps({"rank": rank, "lo": lo, "hi": hi, "its": out}))
    ctx.close()
    dist.finalize()
""") % (ROOT, os.path.join(GOLDEN, "testquaffnullparams.json"))


def test_hip_estep_on_two_ranks_matches_one_rank(tmp_path):
    import quaff_amd as Q
    from quaff_amd import api
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + (os.getpid() % 2000)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), str(tmp_path)],
                         capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr[-3000:]
    rows = [json.loads((tmp_path / ("rank%d.json" % r)).read_text()) for r in range(2)]
    assert rows[0]["lo"] == 0 and rows[0]["hi"] == rows[1]["lo"] and rows[1]["hi"] == 3001
    # one rank, whole batch
    ctx = Q.Context(0)
    ctx.set_params_json(None)
    ctx.set_null_json(open(os.path.join(GOLDEN, "testquaffnullparams.json")).read())
    ref = api.synth_ref(1, 6000)
    ctx.set_refs([ref, api.revcomp(ref)])
    seq, qual, off = api.synth_reads(2, ref, 3001, 600)
    ctx.upload_reads_packed(seq, qual, off)
    order = None
    for it in range(2):
        res = ctx.count_resident(Q.DPConfig(), sort_order=order, packed_order=True)
        order = res["sort_order"]
        for r in rows:
            got = np.array(r["its"][it]["counts"])
            assert np.array_equal(got, np.array(rows[0]["its"][it]["counts"]))          # every rank holds the same sum
            big = np.abs(res["counts"]) > 1e-6
            rel = np.abs(got - res["counts"])[big] / np.abs(res["counts"])[big]
            assert rel.max() < 1e-4, (it, rel.max())
            # in fact bit for bit: the count terms are added as 128-bit integers on the device and across the ranks, so the
            # two-rank totals ARE the one-rank totals (counts and log-likelihood words alike)
            assert np.array_equal(got, res["counts"]), (it, rel.max())
            one = np.concatenate([res["counts_exact"], res["loglike_exact"].reshape(1, 2)])
            assert [[int(a), int(b)] for a, b in one] == r["its"][it]["fx"]
            assert abs(r["its"][it]["ll"] - res["loglike"]) <= 1e-12 * abs(res["loglike"])
        assert abs(rows[0]["its"][it]["local_ll"] + rows[1]["its"][it]["local_ll"] - res["loglike"]) <= 1e-12 * abs(res["loglike"])
    ctx.close()


def test_rccl_allreduce_counts_one_rank_communicator():
    """qf_comm_unique_id -> qf_comm_init_rank -> qf_allreduce_counts through the C ABI with a communicator of one rank (the most
    this box's single GPU allows): loads RCCL, creates the communicator, runs ncclAllReduce(sum, fp64) on the context's stream."""
    import quaff_amd as Q
    ctx = Q.Context(0)
    ctx.set_params_json(None)
    assert ctx.comm_size() == 0
    with pytest.raises(Exception, match="no communicator"):
        ctx.allreduce_counts(np.zeros(4), 0.0)
    uid = Q.Context.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx.comm_init_rank(uid, 0, 1)
    assert ctx.comm_size() == 1
    v = np.random.default_rng(1).random(24508)
    got, ll = ctx.allreduce_counts(v, -1234.5)
    assert np.array_equal(got, v) and ll == -1234.5
    got, ll = ctx.allreduce_counts(np.zeros(0), 7.0)
    assert len(got) == 0 and ll == 7.0
    ctx.close()


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 --single-device`: the parent starts two ranks (torch.distributed.run as a child process) and
    relays rank 0's JSON line; n_gpus = 2, whole-job cells = both ranks' work."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-device", "--reads", "4000",
                          "--steps", "2", "--warmup", "1", "--cpu-sample", "200"], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["cpu_baseline"]["gpu_parity_mismatches"] == 0
    assert j["config"]["cells_per_step"] > 2 * 4000 * 60000 and 0 < j["roofline"]["frac"] <= 1
    assert j["roofline"]["bound"] == "fp64_valu" and "k_viterbi_fill<16,5,false,true>" in j["roofline"]["kernel"]


def test_comm_init_all_wants_one_gpu_per_context():
    """qf_comm_init_all is the one-process form (one context per device, `quaff train -gpus n`): RCCL takes one rank per GPU,
    so two contexts on one device are refused with a message (the CLI then sums on the host)."""
    import ctypes as C
    import quaff_amd as Q
    a, b = Q.Context(0), Q.Context(0)
    arr = (C.c_void_p * 2)(a.h, b.h)
    L = a.L
    L.qf_comm_init_all.argtypes = [C.c_void_p, C.c_int]
    rc = L.qf_comm_init_all(arr, 2)
    assert rc == -5 and b"one rank per GPU" in L.qf_last_error(a.h)
    one = (C.c_void_p * 1)(a.h)
    assert L.qf_comm_init_all(one, 1) == 0 and a.comm_size() == 1
    v, ll = a.allreduce_counts(np.arange(10.0), 2.5)
    assert np.array_equal(v, np.arange(10.0)) and ll == 2.5
    a.close(); b.close()


LONELY = textwrap.dedent("""
    import sys, time
    sys.path.insert(0, %r)
    import quaff_amd as Q
    from quaff_amd.api import QuaffHipError
    ctx = Q.Context(0)
    ctx.set_params_json(None)
    uid = Q.Context.comm_unique_id()           # (an id of its own: the other process holds a different one)
    t0 = time.time()
    try:
        ctx.comm_init_rank(uid, int(sys.argv[1]), 2)
    except QuaffHipError as e:
        print("STATUS", e.code, "%%.1f" %% (time.time() - t0), str(e))
        assert ctx.comm_size() == 0
        sys.exit(3)
    print("JOINED")
""") % ROOT


def test_missing_peer_returns_a_status_instead_of_hanging(tmp_path):
    """qf_comm_init_rank when a peer never joins (it died before the E-step, or holds another id): the library gives up after
    QUAFF_HIP_COMM_TIMEOUT seconds with QF_ERR_DEVICE and a message, it does not hang (ncclCommInitRank itself blocks until all
    ranks have called it; src/qmodel.cpp:2416-2422 is the reduce this guards).  Two processes, each the only member of its own
    two-rank communicator (mismatched ids): both must exit non-zero by themselves; the parent's own timeout is only the backstop
    and, if it strikes, kills the child (no exec from a process that has touched the GPU)."""
    script = tmp_path / "lonely.py"
    script.write_text(LONELY)
    env = dict(os.environ, QUAFF_HIP_COMM_TIMEOUT="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(rank)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for rank in (0, 1)]
    for rank, p in enumerate(procs):
        try:
            out, err = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            p.kill()
            p.communicate()
            pytest.fail("rank %d was still waiting for its peer after 150 s" % rank)
        assert p.returncode == 3, (rank, p.returncode, out[-500:], err[-1500:])
        status = [l for l in out.splitlines() if l.startswith("STATUS")][0].split(None, 3)
        assert int(status[1]) == -1 and 3.0 <= float(status[2]) < 60.0 and "waiting for the other ranks" in status[3]


SKIPPER = textwrap.dedent("""
    import sys, time
    sys.path.insert(0, %r)
    import numpy as np
    import quaff_amd as Q
    from quaff_amd.api import QuaffHipError
    rank, uid_hex = int(sys.argv[1]), sys.argv[2]
    ctx = Q.Context(rank)                        # one rank per GPU
    ctx.set_params_json(None)
    if uid_hex == "-":
        print("UID", Q.Context.comm_unique_id().hex()); sys.stdout.flush()
        uid_hex = sys.stdin.readline().strip()
    ctx.comm_init_rank(bytes.fromhex(uid_hex), rank, 2)
    print("JOINED"); sys.stdout.flush()
    if rank == 1:
        time.sleep(20)                           # never enters the collective
        sys.exit(0)
    t0 = time.time()
    try:
        ctx.allreduce_counts_exact(np.zeros((8, 2), np.uint64))
    except QuaffHipError as e:
        print("STATUS", e.code, "%%.1f" %% (time.time() - t0), str(e))
        sys.exit(3)
    print("REDUCED")
""") % ROOT


def test_all_reduce_with_a_peer_that_skips_it_returns_a_status(tmp_path):
    """The all-reduce deadline (qf_allreduce_counts_exact, src/qmodel.cpp:2416-2422): two ranks form a communicator, rank 1 never
    enters the collective; rank 0 must come back with QF_ERR_DEVICE after QUAFF_HIP_COMM_TIMEOUT, with nothing of its own still in
    flight (the staging buffer belongs to the context).  Needs two devices: on the one-GPU boxes of this pool it is skipped."""
    from quaff_amd import api
    if api.load_library().qf_device_count() < 2:
        pytest.skip("needs two GPUs (RCCL takes one rank per device)")
    script = tmp_path / "skipper.py"
    script.write_text(SKIPPER)
    env = dict(os.environ, QUAFF_HIP_COMM_TIMEOUT="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p0 = subprocess.Popen([sys.executable, str(script), "0", "-"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    uid = p0.stdout.readline().split()[1]
    p1 = subprocess.Popen([sys.executable, str(script), "1", uid], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    p0.stdin.write(uid + "\n"); p0.stdin.flush()
    try:
        out, err = p0.communicate(timeout=120)
    except subprocess.TimeoutExpired:
        p0.kill(); p1.kill()
        pytest.fail("rank 0 was still inside the all-reduce after 120 s")
    p1.communicate(timeout=60)
    assert p0.returncode == 3, (p0.returncode, out[-500:], err[-1500:])
    status = [l for l in out.splitlines() if l.startswith("STATUS")][0].split(None, 3)
    assert int(status[1]) == -1 and 3.0 <= float(status[2]) < 60.0 and "did not complete" in status[3]
