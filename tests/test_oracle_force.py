"""The reference's own behaviour in the -force corner, pinned on the oracle WITHOUT the switch that follows the library
(oracle.count_read(skip_pathless=False), the default): under -force the running log-likelihood starts at -inf
(src/qmodel.cpp:2244-2246), a reference without any path passes the `>= yLogLike - 20` test (:2252), gets a Backward pass
whose counts are NaN, and its weight exp(-inf - -inf) is NaN too (:2258-2262) -- the read's totals, and with them the E-step's,
are NaN.  The library deliberately diverges: such pairs carry no weight and the totals stay finite; the GPU
tests (tests/test_gpu_count.py) compare against the oracle WITH the switch and pin which pairs were left out."""
import numpy as np

from oracle import oracle as O
from tests.helpers import make_reads, rand_seq
from tests.test_gpu_align import DEFAULT_JSON, NULL_JSON


def both_strands(ref):
    x = O.FastSeq("ref", ref)
    return [x, x.revcomp()]


def test_reference_semantics_pathless_read_under_force_gives_nan_totals():
    """-global -force, reads that are fragments of the reference: no global path on either strand."""
    rng = np.random.default_rng(37)
    ref = rand_seq(rng, 1200)
    sc, null = O.Scores(O.Params.from_json(DEFAULT_JSON)), O.NullParams.from_json(NULL_JSON)
    reads = make_reads(rng, ref, 2, 200)
    cfg = O.DPConfig(local=False)
    for read in reads:
        d = {}
        tot, ylog, _ = O.count_read(both_strands(ref), read, sc, null, cfg, use_null=False, details=d)
        assert ylog == -np.inf and all(f == -np.inf for f in d["forward"])
        assert d["counted"] == [0, 1]                 # both pass `>= -inf - 20` and get a Backward pass
        assert np.isnan(tot).any()                    # the reference's result
        d2 = {}
        tot2, ylog2, order2 = O.count_read(both_strands(ref), read, sc, null, cfg, use_null=False, skip_pathless=True, details=d2)
        assert ylog2 == -np.inf and d2["counted"] == [] and np.all(tot2 == 0)   # the library's: no weight, finite totals


def test_reference_semantics_pathless_reference_first_in_the_order_poisons_a_good_read():
    """One reference that has a global path (the read itself) behind one that has none: the pathless one comes first, passes the
    test against -inf, and 0 x NaN makes the read's totals NaN although its log-likelihood is finite."""
    rng = np.random.default_rng(41)
    body = rand_seq(rng, 150)
    good = O.FastSeq("good", body)
    bad = O.FastSeq("bad", rand_seq(rng, 900))        # a global alignment of a 150-base read to it lies outside every band
    read = make_reads(rng, body, 1, 150, sub=0.02, ins=0.0, dele=0.0)[0]
    read = O.FastSeq(read.name, body, read.qual[:len(body)] if len(read.qual) >= len(body) else "5" * len(body))
    sc, null = O.Scores(O.Params.from_json(DEFAULT_JSON)), O.NullParams.from_json(NULL_JSON)
    cfg = O.DPConfig(local=False)
    d = {}
    tot, ylog, _ = O.count_read([bad, good], read, sc, null, cfg, use_null=False, details=d)
    if d["forward"][0] != -np.inf:                    # (the construction relies on `bad` having no global path)
        raise AssertionError("test input: the first reference has a path")
    assert np.isfinite(ylog) and d["counted"] == [0, 1] and np.isnan(tot).any()
    d2 = {}
    tot2, ylog2, _ = O.count_read([bad, good], read, sc, null, cfg, use_null=False, skip_pathless=True, details=d2)
    assert ylog2 == ylog and d2["counted"] == [1] and np.all(np.isfinite(tot2)) and tot2.sum() > 0
