"""Pins the oracle's DP (score tables, Viterbi + traceback, Forward/Backward counts, null model)
against the reference's own integration goldens (Makefile:146-150):
  quaff align c8f30 c8f30 -kmatchmb 10 -fwdstrand  == data/c8f30-self-align.json   (byte-for-byte)
  quaff count c8f30 c8f30 -kmatchmb 10 -fwdstrand  == data/c8f30-self-counts.json  (byte-for-byte)
These files carry 6 significant figures and (because -kmatchmb 10 admits no band for a 6.6 kb
read) a 1-diagonal envelope; that is the precision and coverage at which the reference itself
pins this path."""
import os

import numpy as np
import pytest

from oracle import oracle as O


@pytest.fixture(scope="module")
def c8f30(golden):
    reads = O.read_fastx(os.path.join(golden, "c8f30.fastq.gz"))
    sc = O.Scores(O.Params.from_json(open(os.path.join(golden, "defaultparams.json")).read()))
    null = O.NullParams.fit(reads)          # requireNullModelOrFit, t/quaff.cpp:419-429
    cfg = O.DPConfig(kmer_threshold=-1, max_size=10 << 20)   # -kmatchmb 10, qmodel.cpp:788-799
    return reads, sc, null, cfg


def test_self_align_golden(c8f30, golden):
    reads, sc, null, cfg = c8f30
    al = O.align_read(reads, reads[0], sc, null, cfg)[0]
    assert al["ndiag"] == 1 and al["cells"] == 6604          # SURVEY 8c anchor values
    assert O.cigar(al["ops"]) == "M6604"
    assert O.fmt(al["raw"]) == "-18710.3" and O.fmt(al["score"]) == "7981.84"
    assert O.stockholm(reads[0], reads[0], al) == open(os.path.join(golden, "c8f30-self-align.json")).read()


def test_self_counts_golden(c8f30, golden):
    reads, sc, null, cfg = c8f30
    tot, ylog, order = O.count_read(reads, reads[0], sc, null, cfg)
    assert order == [0]
    assert O.param_counts_json(tot, 1, 0) == open(os.path.join(golden, "c8f30-self-counts.json")).read()


def test_self_overlap_golden(golden):
    """quaff overlap c8f30 copy-of-c8f30 -kmatchmb 10 -fwdstrand == data/c8f30-self-overlap.json (Makefile:152-156;
    the copy is the same read with `channel` -> `copy` in its name)."""
    reads = O.read_fastx(os.path.join(golden, "c8f30.fastq.gz"))
    cp = O.FastSeq(reads[0].name.replace("channel", "copy", 1), reads[0].seq, reads[0].qual)
    seqs = [reads[0], cp]
    p = O.Params.from_json(open(os.path.join(golden, "defaultparams.json")).read())
    sc = O.Scores(p)
    osc = O.OverlapScores(p, sc, False)
    null = O.NullParams.fit(seqs)
    assert O.overlap_task_pairs(2, 2) == [(0, 1, False)]
    al = O.overlap_pair(seqs[0], seqs[1], False, osc, sc, null, O.DPConfig(kmer_threshold=-1, max_size=10 << 20))
    assert O.fmt(al["score"]) == "6876.76" and O.cigar(al["ops"]) == "M6604"
    assert O.overlap_stockholm(seqs[0], seqs[1], al) == open(os.path.join(golden, "c8f30-self-overlap.json")).read()


def test_default_threshold_seeding_anchor(c8f30):
    """with the align default (threshold 20) c8f30 vs itself seeds one diagonal -> 65."""
    reads, sc, null, cfg = c8f30
    t = O.tokens(reads[0].seq)
    d = O.envelope(t, t, O.DPConfig())
    assert len(d) == 65 and d[0] == -32 and d[-1] == 32
    v = O.viterbi(t, O.ReadCtx(reads[0], sc), sc, d)
    assert O.cigar(v["ops"]) == "M6604"


def test_forward_backward_consistency():
    """Backward's result re-derives Forward's within the table log-sum-exp error the reference
    itself tolerates (MAX_FRACTIONAL_FWDBACK_ERROR, qmodel.cpp:19), on a multi-diagonal band
    with indels; and per-column posterior mass sums to ~1."""
    from tests.helpers import rand_seq, make_reads
    rng = np.random.default_rng(5)
    ref = rand_seq(rng, 1200)
    sc = O.Scores(O.Params.from_json(open(os.path.join(os.path.dirname(__file__), "golden", "defaultparams.json")).read()))
    for read in make_reads(rng, ref, 4, 300)[::2]:
        rc = O.ReadCtx(read, sc)
        xt = O.tokens(ref)
        d = O.envelope(xt, rc.tok, O.DPConfig())
        assert len(d) > 60
        f, b, cnt = O.forward_backward(xt, rc, sc, d)
        assert abs(f - b) <= 1e-4 * abs(f) * 2
        ne = (4 + 4 * sc.Km) * O.NQUAL
        emitted = cnt[:ne].sum()                 # every read base is emitted by a match or an insert state
        assert abs(emitted - len(read.seq)) < 0.02 * len(read.seq)
        v = O.viterbi(xt, rc, sc, d)
        assert v["result"] <= f + 1e-9
