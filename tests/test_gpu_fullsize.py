"""BASELINE config 2 at full size (10 kb reference + reverse complement x 100 000 x 1 kb reads, band 64) on the GPU,
checked through size-independent properties on EVERY read and against the oracle on samples:
  * every read gets exactly one alignment, on its true strand (odd reads are reverse-complemented by the generator);
  * its CIGAR consumes the whole read, and M+D equals the reference span;
  * re-scoring the returned path with the oracle's O(path) recurrence reproduces the Viterbi score bit-for-bit (2 000
    reads), i.e. traceback, coordinates and score are mutually consistent;
  * a full oracle run agrees on 150 reads (score, reference, coordinates, CIGAR);
  * the cell total equals the sum of the per-pair counts; running the same batch twice is bit-identical."""
import os

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_config2_full_size_properties():
    import quaff_amd as Q
    from quaff_amd import api
    n_reads, read_len = 100000, 1000
    ref = api.synth_ref(1, 10000)
    seq, qual, off = api.synth_reads(2, ref, n_reads, read_len)
    ctx = Q.Context(0)
    null_json = open(os.path.join(GOLDEN, "testquaffnullparams.json")).read()
    ctx.set_params_json(None)
    ctx.set_null_json(null_json)
    refs = [ref, api.revcomp(ref)]
    ctx.set_refs(refs)
    ctx.upload_reads_packed(seq, qual, off)
    raw = ctx.align_resident(Q.DPConfig(), 0, raw=True)
    assert raw.n_alignments == n_reads
    cells = np.ctypeslib.as_array(raw.cells, (2 * n_reads,)).reshape(n_reads, 2).copy()
    vit = np.ctypeslib.as_array(raw.viterbi, (2 * n_reads,)).reshape(n_reads, 2).copy()
    assert int(raw.total_cells) == int(cells.sum()) and 7.0e9 < raw.total_cells < 8.2e9
    lens = np.diff(off).astype(np.int64)
    al = raw.alignments
    runs_all = np.ctypeslib.as_array(raw.cigar_runs, (int(max(al[a].run_offset + al[a].n_runs for a in range(0, n_reads, 997))) + 4096,))
    first = {}
    for a in range(n_reads):
        x = al[a]
        assert x.read == a and x.ref == (a & 1), a                       # one alignment per read, on the true strand
        assert x.viterbi == vit[a, x.ref] and x.viterbi > vit[a, 1 - x.ref]
        assert 1 <= x.x_start <= x.x_end <= 10000
        if a % 50 == 0:                                                   # CIGAR accounting on a 2 000-read sample
            runs = np.ctypeslib.as_array(raw.cigar_runs, (int(x.run_offset + x.n_runs),))[int(x.run_offset):]
            ops, ln = runs & 3, runs >> 2
            assert ln[ops != 2].sum() == lens[a] and ln[ops != 1].sum() == x.x_end - x.x_start + 1
            assert ln.sum() == x.n_columns and ops[0] == 0 and ops[-1] == 0
            first[a] = (x.ref, x.viterbi, x.score, x.x_start, x.x_end, "".join("MID"[int(o)] * int(l) for o, l in zip(ops, ln)))
    # oracle checks on samples
    sc = O.Scores(O.Params.from_json(open(os.path.join(GOLDEN, "defaultparams.json")).read()))
    null = O.NullParams.from_json(null_json)
    xf = O.FastSeq("ref", ref.decode())
    orefs = [xf, xf.revcomp()]
    xtoks = [O.tokens(r.seq) for r in orefs]
    for n, a in enumerate(sorted(first)):
        rd = O.FastSeq("read%d" % a, seq[int(off[a]):int(off[a + 1])].decode(), qual[int(off[a]):int(off[a + 1])].decode())
        rf, v, s, xs, xe, ops = first[a]
        rc = O.ReadCtx(rd, sc)
        assert O.rescore_path(xtoks[rf], rc, sc, xs, ops) == v, a     # path, coordinates and score are consistent
        if n % 14 == 0:                                                # ~150 full oracle runs
            k = O.align_read(orefs, rd, sc, null, O.DPConfig())[0]
            assert (k["ref"], k["raw"], k["score"], k["xStart"], k["xEnd"], k["ops"]) == (rf, v, s, xs, xe, ops), a
    # determinism: same batch again, bit-identical scores
    raw2 = ctx.align_resident(Q.DPConfig(), 0, raw=True)
    vit2 = np.ctypeslib.as_array(raw2.viterbi, (2 * n_reads,)).reshape(n_reads, 2)
    assert np.array_equal(vit, vit2)
    ctx.close()
