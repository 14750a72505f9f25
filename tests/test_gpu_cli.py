"""The `quaff` command-line shell (quaff_amd/bin/quaff, C++ over the C ABI) against the reference's own integration
goldens (Makefile:146-156) — the same commands `make test` runs — and against the oracle for SAM output."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import rand_seq, make_reads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUAFF = os.path.join(ROOT, "quaff_amd", "bin", "quaff")
GOLDEN = os.path.join(ROOT, "tests", "golden")
C8 = os.path.join(GOLDEN, "c8f30.fastq.gz")


def run(*args, devices=None, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    if devices:
        env["QUAFF_HIP_DEVICES"] = devices
    out = subprocess.run([QUAFF] + list(args), capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def test_align_self_golden():
    assert run("align", C8, C8, "-kmatchmb", "10", "-fwdstrand") == open(os.path.join(GOLDEN, "c8f30-self-align.json")).read()


def test_kmatchmax_divides_system_memory_by_threads(tmp_path):
    """-kmatchmax = system RAM / -threads per matrix (src/qmodel.cpp:801-807,1058-1060): with an absurd thread count the
    budget admits no band (the -kmatchmb 10 golden: diagonal 0 only); with one thread it equals `-kmatchmb 0` (all RAM)."""
    assert run("align", C8, C8, "-kmatchmax", "-threads", "100000000", "-fwdstrand") == open(os.path.join(GOLDEN, "c8f30-self-align.json")).read()
    rng = np.random.default_rng(52)
    ref = rand_seq(rng, 1500)
    fa, fq = tmp_path / "ref.fasta", tmp_path / "reads.fastq"
    fa.write_text(">ref\n%s\n" % ref)
    fq.write_text("".join("@%s\n%s\n+\n%s\n" % (r.name, r.seq, r.qual) for r in make_reads(rng, ref, 6, 300)))
    whole = run("align", str(fa), str(fq), "-kmatchmb", "0")
    assert run("align", str(fa), str(fq), "-kmatchmax") == whole
    assert run("align", str(fa), str(fq), "-kmatchmax", "-threads", "1") == whole
    assert whole.count("#=GF Score") == 6


def test_count_self_golden():
    got = run("count", C8, C8, "-kmatchmb", "10", "-fwdstrand")
    want = open(os.path.join(GOLDEN, "c8f30-self-counts.json")).read()
    if got != want:       # 1e-4 tolerance on Forward-Backward counts: allow last-digit differences of the 6 s.f. text
        import re
        g, w = (list(map(float, re.findall(r"-?\d+\.?\d*(?:e[-+]?\d+)?", t))) for t in (got, want))
        assert len(g) == len(w)
        np.testing.assert_allclose(g, w, rtol=1e-4, atol=1e-6)


def test_overlap_self_golden(tmp_path):
    copy = tmp_path / "copy-of-c8f30.fastq"
    copy.write_text(gzip.open(C8, "rt").read().replace("channel", "copy", 1))
    assert run("overlap", C8, str(copy), "-kmatchmb", "10", "-fwdstrand") == open(os.path.join(GOLDEN, "c8f30-self-overlap.json")).read()


def test_align_sam_both_strands(tmp_path):
    """SAM output incl. the reverse-strand POS/CIGAR quirk, explicit -params/-null files."""
    rng = np.random.default_rng(51)
    ref = rand_seq(rng, 2500)
    reads = make_reads(rng, ref, 10, 300)
    (tmp_path / "ref.fa").write_text(">chr1 test\n" + "\n".join(ref[i:i + 70] for i in range(0, len(ref), 70)) + "\n")
    (tmp_path / "reads.fq").write_text("".join("@%s\n%s\n+\n%s\n" % (r.name, r.seq, r.qual) for r in reads))
    nullf = os.path.join(GOLDEN, "testquaffnullparams.json")
    got = run("align", str(tmp_path / "ref.fa"), str(tmp_path / "reads.fq"), "-null", nullf, "-format", "sam",
              "-params", os.path.join(GOLDEN, "defaultparams.json"))
    sc = O.Scores(O.Params.from_json(open(os.path.join(GOLDEN, "defaultparams.json")).read()))
    null = O.NullParams.from_json(open(nullf).read())
    x = O.FastSeq("chr1", ref, "", "test")
    refs = [x, x.revcomp()]
    want = "@HD\tVN:1.0\tSO:unknown\n@SQ\tSN:chr1\tLN:2500\n"
    for r in reads:
        al = O.align_read(refs, r, sc, null, O.DPConfig())[0]
        if al["score"] >= 0:
            want += O.sam_line(refs[al["ref"]], r, al)
    assert got == want
    assert "\t16\t" in got and "\t0\tchr1" in got


def test_train_runs_and_improves(tmp_path):
    rng = np.random.default_rng(52)
    ref = rand_seq(rng, 3000)
    reads = make_reads(rng, ref, 40, 250)
    (tmp_path / "ref.fa").write_text(">ref\n" + ref + "\n")
    (tmp_path / "reads.fq").write_text("".join("@%s\n%s\n+\n%s\n" % (r.name, r.seq, r.qual) for r in reads))
    out = subprocess.run([QUAFF, "train", str(tmp_path / "ref.fa"), str(tmp_path / "reads.fq"), "-maxiter", "3"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lls = [float(l.split("log-likelihood (")[1].split(")")[0]) for l in out.stderr.splitlines() if "EM iteration" in l]
    assert len(lls) >= 2 and lls[1] > lls[0]
    p = O.Params.from_json(out.stdout)          # parses as a quaff params file
    assert 0 < p.extendInsert < 1 and abs(p.match[0, 0, 0] + p.match[0, 1, 0] + p.match[0, 2, 0] + p.match[0, 3, 0] - 1) < 1e-5


def _em_lines(stderr):
    """(log-likelihood, log-prior, sum) of every "EM iteration" line (src/qmodel.cpp:2203)."""
    import re
    out = []
    for l in stderr.splitlines():
        m = re.match(r"EM iteration (\d+): log-likelihood \((\S+)\) \+ log-prior \((\S+)\) = (\S+)", l)
        if m:
            assert int(m.group(1)) == len(out) + 1
            out.append(tuple(float(m.group(k)) for k in (2, 3, 4)))
    return out


def _numbers(text):
    import re
    return np.array(list(map(float, re.findall(r"-?\d+\.?\d*(?:e[-+]?\d+)?", text))))


def _layout(text):
    import re
    return re.sub(r"-?\d+\.?\d*(?:e[-+]?\d+)?", "#", text).split()


@pytest.mark.parametrize("flags", [("-order", 0), ("-order", 1), ("-order", 2), ("-suborder", 2, "-gaporder", 0), ("-suborder", 0, "-gaporder", 2)],
                         ids=["order0", "order1", "order2", "sub2gap0", "sub0gap2"])
def test_train_files_and_stopping_iteration_match_the_oracles_em_loop(tmp_path, flags):
    """SURVEY 8(f) #2 end to end: `quaff train -maxiter 3 -saveprior -savecounts -savecountswithprior -saveparams` on 40 reads
    against the oracle's own EM loop (QuaffTrainer::fitUnlimited, src/qmodel.cpp:2186-2231) driving the oracle's own E-step:
    the auto-prior (initCounts(9, 9, 5, 1, null), incl. the `i == j` quirk for -order 1) byte for byte; the last E-step's
    counts and counts + prior at 1e-4 relative; the fitted parameters (6 s.f. text; q, r come out of a Newton iteration that
    stops at a relative 1e-4 step); log-likelihood and log-prior of every iteration; and `-mininc` stops both at the same
    iteration.  -order k = match contexts of k + 1 bases, gap contexts of k (t/quaff.cpp:441-450: config 4 trains at -order 2, 64
    match prefixes and 24 508 counts); -suborder / -gaporder set them separately (:452-468), so that the emission and the
    transition tables have different context lengths."""
    opt = dict(zip(flags[::2], flags[1::2]))
    order = opt.get("-order", 0)
    ml, gl = 1 + opt.get("-suborder", order), opt.get("-gaporder", order)
    rng = np.random.default_rng(57 + 10 * ml + gl)
    ref = rand_seq(rng, 3000)
    reads = make_reads(rng, ref, 40, 250)
    fa, fq = tmp_path / "ref.fa", tmp_path / "reads.fq"
    fa.write_text(">ref\n" + ref + "\n")
    fq.write_text("".join("@%s\n%s\n+\n%s\n" % (r.name, r.seq, r.qual) for r in reads))
    null_path = os.path.join(GOLDEN, "testquaffnullparams.json")
    null = O.NullParams.from_json(open(null_path).read())
    x = O.FastSeq("ref", ref)
    refs = [x, x.revcomp()]
    prior = O.init_counts(ml, gl, 9, 9, 5, 1, null)
    seed = O.m_step(prior, ml, gl)                          # requireParamsOrUsePrior, t/quaff.cpp:370-376
    cfg = O.DPConfig()
    files = {k: tmp_path / (k + ".json") for k in ("prior", "counts", "withprior", "params")}
    extra = [str(f) for f in flags] if (ml, gl) != (1, 0) else []
    out = subprocess.run([QUAFF, "train", str(fa), str(fq), "-null", null_path, "-maxiter", "3", "-mininc", "0", "-saveprior", str(files["prior"]),
                          "-savecounts", str(files["counts"]), "-savecountswithprior", str(files["withprior"]),
                          "-saveparams", str(files["params"])] + extra, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout == ""                                  # -saveparams: nothing on stdout (src/qmodel.cpp:2224-2227; t/quaff.cpp)
    want_p, log = O.train(refs, reads, null, prior, seed, cfg, max_iter=3, min_inc=0.0)
    assert len(log) == 3 and "params" in log[-1]
    assert files["prior"].read_text() == O.param_counts_json(prior, ml, gl)
    em = _em_lines(out.stderr)
    assert len(em) == len(log)
    for (ll, lp, tot), rec in zip(em, log):
        assert abs(ll - rec["loglike"]) <= 1e-4 * abs(rec["loglike"]) and abs(lp - rec["logprior"]) <= 1e-4 * abs(rec["logprior"])
        assert abs(tot - (rec["loglike"] + rec["logprior"])) <= 1e-4 * abs(tot)
    for name, want in (("counts", log[-1]["counts"]), ("withprior", log[-1]["counts_with_prior"])):
        got_text, want_text = files[name].read_text(), O.param_counts_json(want, ml, gl)
        assert _layout(got_text) == _layout(want_text), name
        g, w = _numbers(got_text), _numbers(want_text)
        big = np.abs(w) > 1e-6
        np.testing.assert_allclose(g[big], w[big], rtol=1.2e-4, err_msg=name)       # 1e-4 + the 6 s.f. of the text
        assert np.all(np.abs(g[~big]) < 2e-6)
    got_text, want_text = files["params"].read_text(), O.params_json(want_p)
    assert _layout(got_text) == _layout(want_text)
    gp, wp = O.Params.from_json(got_text), O.Params.from_json(want_text)
    np.testing.assert_allclose(gp.match[:, :, 0], wp.match[:, :, 0], rtol=2e-4, atol=1e-9)           # symbol probabilities
    np.testing.assert_allclose(gp.insert[:, 0], wp.insert[:, 0], rtol=2e-4)
    np.testing.assert_allclose(gp.match[:, :, 1:], wp.match[:, :, 1:], rtol=5e-3)                     # (q, r): Newton stops at 1e-4
    np.testing.assert_allclose(gp.insert[:, 1:], wp.insert[:, 1:], rtol=5e-3)
    for a, b in ((gp.beginInsert, wp.beginInsert), (gp.beginDelete, wp.beginDelete), ([gp.extendInsert, gp.extendDelete], [wp.extendInsert, wp.extendDelete])):
        np.testing.assert_allclose(a, b, rtol=2e-4)
    # the stopping rule: a -mininc between two consecutive relative rises of the oracle's series stops both after the same E-step
    _, free = O.train(refs, reads, null, prior, seed, cfg, max_iter=5, min_inc=-1.0)
    v = [r["loglike"] + r["logprior"] for r in free]
    rise = [(v[k] - v[k - 1]) / abs(v[k - 1]) for k in range(1, len(v))]
    k = max(range(len(rise) - 1), key=lambda q: rise[q] / max(rise[q + 1], 1e-300) if rise[q] > 0 else 0)
    assert rise[k] > 1.5 * rise[k + 1], rise
    min_inc = (rise[k] * max(rise[k + 1], rise[k] * 1e-3)) ** 0.5 if rise[k + 1] > 0 else rise[k] / 2
    _, stopped = O.train(refs, reads, null, prior, seed, cfg, max_iter=10, min_inc=min_inc)
    assert len(stopped) == k + 3 and "params" not in stopped[-1]       # E-steps 1 .. k + 3; the last one found the rise too small
    out2 = subprocess.run([QUAFF, "train", str(fa), str(fq), "-null", null_path, "-maxiter", "10", "-mininc", repr(float(min_inc))] + extra,
                          capture_output=True, text=True, timeout=600)
    assert out2.returncode == 0, out2.stderr[-2000:]
    assert len(_em_lines(out2.stderr)) == len(stopped)
    final = O.Params.from_json(out2.stdout)                           # the parameters of the last M-step that ran
    np.testing.assert_allclose(final.match[:, :, 0], stopped[-2]["params"].match[:, :, 0], rtol=2e-4, atol=1e-9)


def test_batches_spread_over_devices_give_the_same_output(tmp_path):
    """`-gpus n` spreads read blocks (align, count/train) or pair-list blocks (overlap) over one context per device and
    prints in input order.  QUAFF_HIP_DEVICES names the device of each context explicitly; "0,0,0" puts three contexts
    (three host threads) on this box's one GPU, which exercises the same sharding and ordering code."""
    import re
    rng = np.random.default_rng(53)
    ref = rand_seq(rng, 3000)
    reads = make_reads(rng, ref, 41, 400)
    (tmp_path / "ref.fa").write_text(">ref\n" + ref + "\n")
    (tmp_path / "reads.fq").write_text("".join("@%s\n%s\n+\n%s\n" % (r.name, r.seq, r.qual) for r in reads))
    fa, fq = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fq")
    one = run("align", fa, fq, "-format", "sam")
    assert one.count("\n") > 41
    assert run("align", fa, fq, "-format", "sam", devices="0,0,0") == one
    assert run("align", fa, fq, "-format", "sam", "-gpus", "1") == one
    (tmp_path / "few.fq").write_text("".join("@%s\n%s\n+\n%s\n" % (r.name, r.seq, r.qual) for r in make_reads(rng, ref[:1200], 12, 800, sub=.01, ins=.005, dele=.005)))
    few = str(tmp_path / "few.fq")
    ov = run("overlap", few, "-nothreshold")   # (random qualities: the null-adjusted overlap scores are negative)
    assert ov.count("#=GF Score") > 10 and run("overlap", few, "-nothreshold", devices="0,0") == ov
    assert run("overlap", few, "-nothreshold", devices="0,0,0", env_extra={"QUAFF_HIP_OVERLAP_CHUNK": "17"}) == ov   # 13 blocks, 5 rounds
    nums = lambda t: np.array(list(map(float, re.findall(r"-?\d+\.?\d*(?:e[-+]?\d+)?", t))))
    c1, c3 = run("count", fa, fq), run("count", fa, fq, devices="0,0,0")
    assert len(nums(c1)) == len(nums(c3)) > 1000
    np.testing.assert_allclose(nums(c3), nums(c1), rtol=1e-4, atol=1e-6)
    t1 = subprocess.run([QUAFF, "train", fa, fq, "-maxiter", "2"], capture_output=True, text=True, timeout=300)
    t3 = subprocess.run([QUAFF, "train", fa, fq, "-maxiter", "2"], capture_output=True, text=True, timeout=300,
                        env=dict(os.environ, QUAFF_HIP_DEVICES="0,0,0"))
    assert t1.returncode == 0 and t3.returncode == 0, t3.stderr[-1000:]
    np.testing.assert_allclose(nums(t3.stdout), nums(t1.stdout), rtol=1e-3, atol=1e-6)


def test_gpus_flag_rejects_more_devices_than_visible():
    out = subprocess.run([QUAFF, "align", C8, C8, "-gpus", "99"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "HIP device" in out.stderr


def test_fasta_and_refseq_formats_agree_with_stockholm(tmp_path):
    """-format fasta = the two gapped rows as FASTA records plus a blank line (Alignment::writeGappedFasta,
    src/qmodel.cpp:548-551, :2572-2575); -format refseq = the ungapped reference row with "matches(Read)" prepended to its
    comment (:2585-2590).  Both are rebuilt here from the Stockholm output of the same run."""
    rng = np.random.default_rng(54)
    ref = rand_seq(rng, 2000)
    reads = make_reads(rng, ref, 8, 350)
    (tmp_path / "ref.fa").write_text(">chrT a reference\n" + ref + "\n")
    (tmp_path / "reads.fq").write_text("".join("@%s\n%s\n+\n%s\n" % (r.name, r.seq, r.qual) for r in reads))
    fa, fq = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fq")
    sto = run("align", fa, fq)
    recs = []
    for block in sto.split("//\n")[:-1]:
        rows, cc = {}, {}
        for line in block.splitlines():
            if line.startswith("#=GS CC "):
                name, comment = line[8:].split(" ", 1)
                cc[name] = comment
            elif line and not line.startswith("#"):
                name, data = line.split(None, 1)
                rows[name] = rows.get(name, "") + data
        recs.append((rows, cc))
    assert len(recs) == 8 and all(set(r) == {"Ref", "Read"} for r, _ in recs)
    want_fasta = "".join(">Ref %s\n%s\n>Read %s\n%s\n\n" % (cc["Ref"], rows["Ref"], cc["Read"], rows["Read"]) for rows, cc in recs)
    assert run("align", fa, fq, "-format", "fasta") == want_fasta
    want_refseq = "".join(">Ref matches(Read) %s\n%s\n" % (cc["Ref"], rows["Ref"].replace("-", "")) for rows, cc in recs)
    assert run("align", fa, fq, "-format", "refseq") == want_refseq
    out_file = tmp_path / "saved.sto"
    assert run("align", fa, fq, "-savealign", str(out_file)) == "" and out_file.read_text() == sto


def test_auto_fitted_null_model_matches_oracle(tmp_path):
    """Without -null the null model is fitted to the reads (QuaffNullParams(seqs), src/qmodel.cpp:1811-1843, with the
    negative-binomial fit of src/negbinom.cpp); -savenull writes it.  Two independent restatements (C++ here, Python in the
    oracle) must print the same file at the reference's 6 significant figures."""
    import re
    rng = np.random.default_rng(55)
    ref = rand_seq(rng, 2500)
    reads = make_reads(rng, ref, 60, 300)
    (tmp_path / "ref.fa").write_text(">ref\n" + ref + "\n")
    (tmp_path / "reads.fq").write_text("".join("@%s\n%s\n+\n%s\n" % (r.name, r.seq, r.qual) for r in reads))
    saved = tmp_path / "null.json"
    run("align", str(tmp_path / "ref.fa"), str(tmp_path / "reads.fq"), "-savenull", str(saved))
    got, want = saved.read_text(), O.NullParams.fit(reads).to_json()
    nums = lambda t: np.array(list(map(float, re.findall(r"-?\d+\.?\d*(?:e[-+]?\d+)?", t))))
    assert re.sub(r"[-\d.e+]+", "#", got).split() == re.sub(r"[-\d.e+]+", "#", want).split()     # same layout
    np.testing.assert_allclose(nums(got), nums(want), rtol=2e-5)                                    # same numbers (6 s.f. text)


def test_cli_errors_exit_nonzero_with_the_reference_messages(tmp_path):
    """Require/Fail print a message and exit(1) in the reference (src/util.cpp:80-98); an unknown base terminates it
    (src/fastseq.cpp:76-79).  Same here: non-zero exit status, message on stderr, nothing on stdout."""
    rng = np.random.default_rng(56)
    ref = rand_seq(rng, 800)
    (tmp_path / "ref.fa").write_text(">ref\n" + ref + "\n")
    (tmp_path / "bad.fq").write_text("@r0\n%sN%s\n+\n%s\n" % (ref[100:200], ref[201:300], "5" * 200))
    (tmp_path / "noqual.fa").write_text(">r0\n" + ref[100:300] + "\n")
    fa = str(tmp_path / "ref.fa")

    def fails(*args):
        out = subprocess.run([QUAFF] + list(args), capture_output=True, text=True, timeout=120)
        assert out.returncode != 0 and out.stdout == "", (args, out.stdout[:200])
        return out.stderr

    assert "Unknown symbol" in fails("align", fa, str(tmp_path / "bad.fq"))
    assert "Couldn't open" in fails("align", fa, str(tmp_path / "missing.fq"))
    assert "does not have quality scores" in fails("train", fa, str(tmp_path / "noqual.fa"))
    assert "out of range" in fails("align", fa, str(tmp_path / "noqual.fa"), "-kmatch", "3")
    assert "Unknown format" in fails("align", fa, str(tmp_path / "noqual.fa"), "-format", "bam")
    assert "does not have quality scores" in fails("align", fa, str(tmp_path / "noqual.fa"))     # reads.wantQualScores, t/quaff.cpp:117
    # with -noquals reads without qualities are fine, and so is a read shorter than k
    (tmp_path / "short.fa").write_text(">tiny\nACG\n>r1\n" + ref[300:500] + "\n")
    ok = subprocess.run([QUAFF, "align", fa, str(tmp_path / "short.fa"), "-noquals", "-nothreshold"], capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0 and ok.stdout.count("# STOCKHOLM") == 2, ok.stderr[-500:]


def _overlap_expected(reads, cfg_kw=None):
    """`quaff overlap` output as the reference would print it single-threaded: QuaffOverlapScheduler's pair order
    (src/qoverlap.cpp:475-480,528-547) over originals + reverse complements, one Stockholm block per finite alignment,
    rows re-paired by the indel squashing of src/qoverlap.cpp:231-267 (oracle: overlap_rows / overlap_stockholm)."""
    params = O.Params.from_json(open(os.path.join(GOLDEN, "defaultparams.json")).read())
    sc = O.Scores(params)
    seqs = list(reads) + [r.revcomp() for r in reads]
    null = O.NullParams.fit(seqs)      # without -null it is fitted to the loaded sequences, complements included (t/quaff.cpp:230-232)
    osc = [O.OverlapScores(params, sc, False), O.OverlapScores(params, sc, True)]
    cfg = O.DPConfig(kmer_threshold=14, **(cfg_kw or {}))
    out, pairs = "", []
    for nx, ny, comp in O.overlap_task_pairs(len(reads), len(seqs)):
        al = O.overlap_pair(seqs[nx], seqs[ny], comp, osc[int(comp)], sc, null, cfg)
        if al is not None:
            out += O.overlap_stockholm(seqs[nx], seqs[ny], al)
            pairs.append((seqs[nx].name, seqs[ny].name))
    return out, pairs


def test_overlap_rows_with_real_indels_match_the_oracle(tmp_path):
    """12 reads (both strands) of one 1.6 kb stretch with 4 % insertions and 4 % deletions each: the overlaps' gap states come
    in mixed insert / delete stretches, which the writer re-pairs (squashes).  Every Stockholm block byte for byte."""
    rng = np.random.default_rng(57)
    ref = rand_seq(rng, 1600)
    reads = make_reads(rng, ref, 12, 900, sub=0.03, ins=0.04, dele=0.04)
    fq = tmp_path / "reads.fq"
    fq.write_text("".join("@%s\n%s\n+\n%s\n" % (r.name, r.seq, r.qual) for r in reads))
    want, _ = _overlap_expected(reads)
    got = run("overlap", str(fq), "-nothreshold")
    assert want.count("# STOCKHOLM") > 60
    # squashing happened: some block pairs bases that the raw state path had as separate insert and delete columns
    assert got == want
    assert run("overlap", str(fq), "-nothreshold", env_extra={"QUAFF_HIP_OVERLAP_CHUNK": "7"}) == want


def test_overlap_pair_order_five_reads(tmp_path):
    """Pair enumeration for N = 5 originals (+ 5 reverse complements): (nx, ny) with nx < N - 1... the order of the blocks
    in the output is the order of O.overlap_task_pairs, which restates QuaffOverlapScheduler."""
    import re
    rng = np.random.default_rng(58)
    ref = rand_seq(rng, 700)
    reads = make_reads(rng, ref, 5, 500, sub=0.02, ins=0.01, dele=0.01)
    fq = tmp_path / "reads.fq"
    fq.write_text("".join("@%s\n%s\n+\n%s\n" % (r.name, r.seq, r.qual) for r in reads))
    want, pairs = _overlap_expected(reads)
    got = run("overlap", str(fq), "-nothreshold")
    assert got == want
    names = re.findall(r"#=GS CC read_x substr\((.+),\d+\.\.\d+\)\n#=GS CC read_y substr\((.+),\d+\.\.\d+\)", got)
    assert names == pairs and len(pairs) == len(O.overlap_task_pairs(5, 10)) == 4 * 9 - 6   # every pair of overlapping 500-base reads aligns


def test_config1_tiny_sequences_shorter_than_k():
    """BASELINE config 1: `quaff align data/tiny.fasta data/tiny.fastq` — 4-base sequences, shorter than k = 6.  The reference
    underflows an unsigned loop bound there (src/fastseq.cpp:247) and crashes; here such sequences simply have no k-mers, the
    short-sequence rule (src/diagenv.cpp:23-29) gives them the full envelope, and the run is defined: exit 0, the four
    alignments the oracle finds, byte for byte."""
    fa, fq = os.path.join(GOLDEN, "tiny.fasta"), os.path.join(GOLDEN, "tiny.fastq")
    got = run("align", fa, fq, "-nothreshold")
    refs, reads = O.read_fastx(fa), O.read_fastx(fq)
    refs = refs + [r.revcomp() for r in refs]
    sc = O.Scores(O.Params.from_json(open(os.path.join(GOLDEN, "defaultparams.json")).read()))
    null = O.NullParams.fit(reads)
    want = ""
    for rd in reads:
        for al in O.align_read(refs, rd, sc, null, O.DPConfig()):
            want += O.stockholm(refs[al["ref"]], rd, al)
    assert got == want and got.count("# STOCKHOLM") == len(reads)
    assert run("align", fa, fq) == "".join(b + "//\n" for b in want.split("//\n")[:-1]
                                            if float(b.split("#=GF Score ")[1].split()[0]) >= 0)
