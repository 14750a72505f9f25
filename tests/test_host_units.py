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
