"""The reference's host-side unit tests (`make unit-tests`, Makefile:103-133) over this build's FASTA/FASTQ reader and
writers, params / null / counts JSON round trips and negative-binomial fitter, through `quaff selftest` (no device needed):
  testfasta / testfastq with data/tiny.* (perl/testexpect.pl: output must equal the expected file byte for byte),
  testquaffjsonio, testquaffnulljsonio, testquaffcountsjsonio (read, write, compare with the input file),
  testnegbinom .1 5 10000 .1 (fit must come back within 10 %; exact expected frequencies stand in for GSL's sampler)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUAFF = os.path.join(ROOT, "quaff_amd", "bin", "quaff")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def selftest(*args):
    if not os.path.exists(QUAFF):
        import __graft_entry__ as g
        g.build()
    out = subprocess.run([QUAFF, "selftest"] + list(args), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def golden(name):
    return open(os.path.join(GOLDEN, name)).read()


@pytest.mark.parametrize("prog,inp,expect", [
    ("fasta", "tiny.fasta", "tiny.fasta"), ("fasta", "tiny.fastq", "tiny.fasta"),                    # Makefile:110-111
    ("fastq", "tiny.fastq", "tiny.fastq"), ("fastq", "tiny.fasta", "tiny.noqual.fastq"),            # Makefile:112-113
    ("fastq", "tiny.noqual.fastq", "tiny.noqual.fastq"), ("fastq", "tiny.truncated.fastq", "tiny.noqual.fastq"),  # :114-115
])
def test_fast_readers_and_writers(prog, inp, expect):
    assert selftest(prog, os.path.join(GOLDEN, inp)) == golden(expect)


def test_gzipped_input_reads_like_plain(tmp_path):
    import gzip
    plain = tmp_path / "c8f30.fastq"
    plain.write_bytes(gzip.open(os.path.join(GOLDEN, "c8f30.fastq.gz")).read())
    assert selftest("fastq", os.path.join(GOLDEN, "c8f30.fastq.gz")) == selftest("fastq", str(plain)) == plain.read_text()


def _kseq_records(text):
    """kseq's conventions (kseq/kseq.h; src/fastseq.cpp:133-171), restated independently of the C++ reader: a header starts with
    '>' or '@' (other lines before one are skipped); name = up to the first blank, comment = the rest; sequence lines run to the
    next line starting with '>', '@' or '+', white space dropped; after '+', quality lines are taken until they hold as many
    characters as the sequence, and kept only if exactly as many."""
    lines = [ln[:-1] if ln.endswith("\r") else ln for ln in text.split("\n")]
    if lines and lines[-1] == "":
        lines.pop()
    recs, k = [], 0
    while k < len(lines):
        if lines[k][:1] not in (">", "@"):
            k += 1
            continue
        head = lines[k][1:]
        cut = min([head.index(c) for c in " \t" if c in head] or [len(head)])
        name, comment = head[:cut], head[cut + 1:]
        k += 1
        seq = ""
        while k < len(lines) and lines[k][:1] not in (">", "@", "+"):
            seq += "".join(lines[k].split())
            k += 1
        qual = ""
        if k < len(lines) and lines[k][:1] == "+":
            k += 1
            while k < len(lines) and len(qual) < len(seq):
                qual += lines[k]
                k += 1
            if len(qual) != len(seq):
                qual = ""
        recs.append((name, comment, seq, qual))
    return recs


def test_reader_edge_cases_across_threads(tmp_path):
    """The reader builds its strings in blocks of records on several threads: 6 000 records of every awkward shape (CR LF, wrapped
    sequence and quality, blanks inside sequence lines, quality lines that start with '@' or '+', a missing quality, a short
    quality, FASTA records in between, junk before the first header, no newline at the end) must come out as kseq's rules say."""
    import random
    rng = random.Random(11)
    parts = ["junk before any header\n", "\n"]
    for r in range(6000):
        L = rng.choice([0, 1, 7, 60, 61, 200])
        seq = "".join(rng.choice("ACGTN") for _ in range(L))
        qual = "".join(chr(rng.randrange(33, 127)) for _ in range(L))
        if L and r % 5 == 0:
            qual = "@" + qual[1:]                                  # a quality line that looks like a header
        if L and r % 7 == 0:
            qual = "+" + qual[1:]
        eol = "\r\n" if r % 3 == 0 else "\n"
        wrap = rng.choice([0, 0, 13, 60])
        def wrapped(t):
            return eol.join(t[i:i + wrap] for i in range(0, len(t), wrap)) if wrap and t else t
        head = "read%d" % r + (rng.choice(["", " a comment", "\ttabbed comment here"]))
        kind = r % 11
        if kind == 0:                                              # FASTA
            parts.append(">" + head + eol + wrapped(seq) + eol)
        elif kind == 1 and L > 2:                                  # blanks inside a sequence line
            parts.append("@" + head + eol + seq[:2] + " " + seq[2:] + "\t" + eol + "+" + eol + qual + eol)
        elif kind == 2 and L > 1:                                  # quality one character short: dropped
            parts.append("@" + head + eol + seq + eol + "+" + eol + qual[:-1] + eol)
        elif kind == 3:                                            # '+' line repeats the name
            parts.append("@" + head + eol + wrapped(seq) + eol + "+read%d" % r + eol + wrapped(qual) + eol)
        else:
            parts.append("@" + head + eol + wrapped(seq) + eol + "+" + eol + wrapped(qual) + eol)
    text = "".join(parts) + "@last\nACGT\n+\nIIII"           # no newline at the end
    f = tmp_path / "edge.fastq"
    f.write_bytes(text.encode())
    got = selftest("fastq", str(f))
    # an empty sequence "has quality" of equal (zero) length: the writer prints "+" and an empty line for it
    want = "".join("@" + n + (" " + c if c else "") + "\n" + s + "\n" + ("+\n" + q + "\n" if len(q) == len(s) else "")
                   for n, c, s, q in _kseq_records(text))
    assert got == want


@pytest.mark.parametrize("what,name", [("params", "testquaffparams.json"), ("params", "defaultparams.json"),   # Makefile:118-119
                                       ("null", "testquaffnullparams.json"), ("counts", "testquaffcounts.json")])  # :122, :125
def test_json_round_trips(what, name):
    assert selftest(what, os.path.join(GOLDEN, name)) == golden(name)


def test_negbinom_fit_recovers_parameters():
    assert selftest("negbinom", ".1", "5", "10000", ".1").startswith("ok:")           # Makefile:132
    assert selftest("negbinom", ".6", "30", "5000", ".1").startswith("ok:")


def test_m_step_fit_from_counts():
    """QuaffParamCounts::fit (src/qmodel.cpp:1731-1768) over the reference's counts fixture: transition probabilities are
    yes / (yes + no), symbol probabilities the normalised count sums within a context, and every (q, r) the
    negative-binomial fit of that context's quality counts — checked against the oracle's independent fitter."""
    import numpy as np
    from oracle import oracle as O
    import json
    cj = O.gason_loads(golden("testquaffcounts.json"))
    # the fixture (a self-alignment) has no insertions and no mismatches: 0/0 there, in the reference too.  Fill those in.
    for k, x in enumerate("ACGT"):
        diag = cj["match"][""][x][x]
        for m, y in enumerate("ACGT"):
            if y != x:
                cj["match"][""][x][y] = [0.0] * (2 + m) + [0.02 * (1 + m) * v for v in diag[:len(diag) - 2 - m]]
        cj["insert"][x] = [0.1 * (1 + k) * v for v in diag[3:]] + [0.0] * 3
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(cj, f)
    try:
        pj = O.gason_loads(selftest("fit", f.name))
    finally:
        os.unlink(f.name)
    isum = {x: float(np.sum(cj["insert"][x])) for x in "ACGT"}
    for x in "ACGT":
        q, r = O.fit_negbinom(np.array(cj["insert"][x], float))
        got = pj["insert"][x]
        assert abs(got["p"] - isum[x] / sum(isum.values())) <= 2e-5 * got["p"]
        assert abs(got["q"] - q) <= 2e-3 * q and abs(got["r"] - r) <= 2e-3 * r, (x, got, q, r)
    close = lambda a, b: abs(a - b) <= 2e-5 * max(abs(a), abs(b), 1e-300)     # 6 s.f. text
    for name in ("Insert", "Delete"):
        yes, no = cj["begin%sYes" % name][""], cj["begin%sNo" % name][""]
        assert close(pj["begin" + name][""], 1 / (1 + no / yes))
        assert close(pj["extend" + name], 1 / (1 + cj["extend%sNo" % name] / cj["extend%sYes" % name]))
    for x in "ACGT":
        sums = {y: float(np.sum(cj["match"][""][x][y])) for y in "ACGT"}
        for y in "ACGT":
            got = pj["match"][""][x][y]
            assert close(got["p"], sums[y] / sum(sums.values()))
            if sums[y] > 0:
                q, r = O.fit_negbinom(np.array(cj["match"][""][x][y], float))
                assert abs(got["q"] - q) <= 2e-3 * q and abs(got["r"] - r) <= 2e-3 * r, (x, y, got, q, r)   # both stop at a relative 1e-4 Newton step


def test_negbinom_fit_degenerate_count_vectors():
    """Count vectors the E-step can hand the M-step for a rarely used (context, base) cell: a single occupied bin (zero
    variance), variance below the mean, flat, nearly empty.  The reference keeps GSL's last accepted root when a Newton step
    fails (src/negbinom.cpp:262-322) and ignores the status; the fit must stay finite and agree with the oracle's."""
    import numpy as np
    from oracle import oracle as O
    for vec in ([0, 0, 5, 0, 0, 0], [3, 0, 0, 0], [1, 2, 3, 2, 1, 0, 0, 0], [5] * 8, [0, 0, 0, 0, 7], [1e-300, 0, 0], [10, 1],
                [0.0] * 20 + [2.5] + [0.0] * 73, [1e-12] * 94):
        status, p, r = selftest("fitvec", *[repr(float(v)) for v in vec]).split()
        p, r = float(p), float(r)
        assert np.isfinite(p) and np.isfinite(r) and 0 < p <= 1 and r > 0, (vec, status, p, r)
        qo, ro = O.fit_negbinom(np.array(vec, float))
        assert abs(p - qo) <= 2e-3 * qo and abs(r - ro) <= 2e-3 * ro, (vec, status, p, r, qo, ro)


# ---------------------------------------------------------------------------------------------------------------------
# the prior, its log-density and the EM stopping rule (src/qmodel.cpp:431-456, 1681-1710, 2204-2206) against
# the oracle's restatement; no device.
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ml,gl", [(1, 0), (2, 1), (3, 2)])
@pytest.mark.parametrize("with_null", [True, False])
def test_auto_prior_init_counts(ml, gl, with_null, tmp_path):
    """QuaffParamCounts::initCounts(9, 9, 5, 1, null) -- the prior `quaff train` sets itself (t/quaff.cpp:490-512) -- for order 0,
    1 and 2, with and without a null model, byte for byte as counts JSON.  For match contexts longer than one base the
    reference's `i == j` test (src/qmodel.cpp:446,448) gives the identity pseudocount only to contexts with an all-A prefix."""
    import numpy as np
    from oracle import oracle as O
    null_path = os.path.join(GOLDEN, "testquaffnullparams.json")
    null = O.NullParams.from_json(golden("testquaffnullparams.json")) if with_null else None
    want = O.init_counts(ml, gl, 9, 9, 5, 1, null)
    got = selftest("initcounts", str(ml), str(gl), "9", "9", "5", "1", *([null_path] if with_null else []))
    assert got == O.param_counts_json(want, ml, gl)
    cv = O.CountsView(want, ml, gl)
    sums = cv.mat.sum(axis=2)                      # negbinom pdf sums to ~1 over the 94 qualities: the row sum is the pseudocount
    for i in range(4):
        for j in range(4 ** ml):
            ident = abs(sums[i, j] - 5) < 0.05
            assert ident == (i == j), (i, j, sums[i, j])          # j < 4 <=> the context's prefix is all A
    back, ml2, gl2 = O.counts_from_json(got)
    assert (ml2, gl2) == (ml, gl) and np.allclose(back, want, rtol=2e-5, atol=0)


def test_log_prior_and_expected_loglike(tmp_path):
    """QuaffParamCounts::logPrior / expectedLogLike of the auto-prior under the default parameters, under the M-step's own fit of
    the prior (the seed `quaff train` starts from without -params, t/quaff.cpp:370-376) and under -order 2 parameters: product
    == oracle to rounding; the beta / Dirichlet terms also against scipy's densities."""
    import math
    import sys
    import numpy as np
    from scipy.stats import beta, dirichlet
    from oracle import oracle as O
    sys.path.insert(0, ROOT)
    import bench
    null_path = os.path.join(GOLDEN, "testquaffnullparams.json")
    for ml, gl, params_text in ((1, 0, golden("defaultparams.json")), (1, 0, None), (3, 2, bench.order2_params_json())):
        prior_text = selftest("initcounts", str(ml), str(gl), "9", "9", "5", "1", null_path)
        pc, _, _ = O.counts_from_json(prior_text)
        cpath, ppath = tmp_path / "prior.json", tmp_path / "params.json"
        cpath.write_text(prior_text)
        if params_text is None:
            params_text = selftest("fit", str(cpath))                 # the seed: prior.fit()
            seed = O.m_step(pc, ml, gl)
            assert params_text == O.params_json(seed)
        ppath.write_text(params_text)
        p = O.Params.from_json(params_text)
        got_lp, got_ell = (float(v) for v in selftest("logprior", str(cpath), str(ppath)).split())
        want = O.log_prior(pc, p)
        assert math.isfinite(want) and abs(got_lp - want) <= 1e-11 * abs(want), (ml, gl, got_lp, want)
        # third opinion on the closed-form pieces (scipy): transitions and the symbol Dirichlets
        cv = O.CountsView(pc, ml, gl)
        tr = sum(beta.logpdf(p.beginInsert[g], cv.beginInsertYes[g] + 1, cv.beginInsertNo[g] + 1) +
                 beta.logpdf(p.beginDelete[g], cv.beginDeleteYes[g] + 1, cv.beginDeleteNo[g] + 1) for g in range(p.Kg))
        tr += beta.logpdf(p.extendInsert, cv.ext[1] + 1, cv.ext[0] + 1) + beta.logpdf(p.extendDelete, cv.ext[3] + 1, cv.ext[2] + 1)
        di = dirichlet.logpdf(np.array(p.insert[:, 0]) / p.insert[:, 0].sum(), 1 + cv.ins.sum(axis=1))
        ours = sum(O.log_beta_pdf(p.beginInsert[g], cv.beginInsertYes[g], cv.beginInsertNo[g]) +
                   O.log_beta_pdf(p.beginDelete[g], cv.beginDeleteYes[g], cv.beginDeleteNo[g]) for g in range(p.Kg))
        ours += O.log_beta_pdf(p.extendInsert, cv.ext[1], cv.ext[0]) + O.log_beta_pdf(p.extendDelete, cv.ext[3], cv.ext[2])
        assert abs(ours - tr) <= 1e-9 * max(1.0, abs(tr))
        if abs(p.insert[:, 0].sum() - 1) < 1e-9:
            assert abs(O.log_dirichlet_pdf(list(1 + cv.ins.sum(axis=1)), list(p.insert[:, 0])) - di) <= 1e-9 * max(1.0, abs(di))
        assert math.isfinite(got_ell)


def test_em_stopping_rule():
    """fitUnlimited's test (src/qmodel.cpp:2204-2206): never before the second E-step; stop when logLike + logPrior < previous +
    |previous| * minInc, i.e. also on a rise smaller than the fraction."""
    from oracle import oracle as O

    def oracle_steps(min_inc, vals):
        prev, it = float("-inf"), 0
        for it, v in enumerate(vals):
            if O.em_converged(it, v, prev, min_inc):
                return it
            prev = v
        return len(vals)
    for min_inc, vals in ((0.01, [-1000, -900, -895, -894]), (0.01, [-1000, -1100]), (0.0, [-5, -4, -4, -3]), (0.0, [-5, -4, -4.5]),
                          (0.5, [-100, -40, -30]), (0.01, [-1000]), (0.01, [10, 10.05, 10.2, 10.21]), (1e-4, [-1e6, -999950, -999900, -999899])):
        got = int(selftest("converge", repr(min_inc), *[repr(float(v)) for v in vals]))
        assert got == oracle_steps(min_inc, vals), (min_inc, vals, got)
