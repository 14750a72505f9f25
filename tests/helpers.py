"""Shared test helpers: small synthetic read sets (numpy RNG; independent of the product's
own C++ generator, which has its own tests)."""
import numpy as np

from oracle import oracle as O


def rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, n))


def mutate(rng, src, sub=0.05, ins=0.03, dele=0.03):
    out = []
    for c in src:
        if rng.random() < dele:
            continue
        if rng.random() < ins:
            out.append("ACGT"[rng.integers(0, 4)])
        if rng.random() < sub:
            c = "ACGT"[rng.integers(0, 4)]
        out.append(c)
    return "".join(out)


def rand_qual(rng, n, lo=5, hi=25):
    return "".join(chr(33 + int(q)) for q in rng.integers(lo, hi + 1, n))


def make_reads(rng, ref, n, read_len, **kw):
    """n reads sampled from ref (odd ones reverse-complemented), mutated, with qualities."""
    reads = []
    for r in range(n):
        s = int(rng.integers(0, max(1, len(ref) - read_len + 1)))
        src = ref[s: s + read_len]
        if r & 1:
            src = O.revcomp_str(src)
        seq = mutate(rng, src, **kw)
        reads.append(O.FastSeq("read%d" % r, seq, rand_qual(rng, len(seq))))
    return reads
