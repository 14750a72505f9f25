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
