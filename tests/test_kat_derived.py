"""Known-answer cases for the cross-check's corner, DERIVED BY HAND from the reference's text
(src/feature_matcher/brute_force_matcher.cc:64-84), not recorded from a run of the reference: see the
provenance note in tests/golden/kat_derived.json.  The oracle (CPU) and the HIP path (GPU, through the C
ABI, both engines) must both give the derived lists."""
import numpy as np
import pytest

from conftest import kat_derived_cases, kat_descriptors


@pytest.mark.parametrize("case", kat_derived_cases(), ids=lambda c: c["name"])
def test_oracle_gives_the_derived_answers(case):
    from oracle import oracle
    A, B = kat_descriptors(case["A"]), kat_descriptors(case["B"])
    l, r, _ = oracle.match_pair(A, B, 0.8, case["cross"])
    assert [[int(a), int(b)] for a, b in zip(l, r)] == case["matches"]


def test_independent_model_gives_the_derived_answers():
    """The same lists from a few lines of numpy that share nothing with the oracle's C: distances in
    float64 (the cases have wide margins or exact ties), std::map::operator[] as dict.setdefault."""
    for case in kat_derived_cases():
        A, B = kat_descriptors(case["A"]).astype(np.float64), kat_descriptors(case["B"]).astype(np.float64)

        def direction(X, Y):
            out = []
            for i, x in enumerate(X):
                d = np.sqrt(((Y - x) ** 2).sum(1))
                order = np.argsort(d, kind="stable")
                second = d[order[1]] if len(d) > 1 else np.finfo(np.float32).max
                if d[order[0]] < 0.8 * second:
                    out.append((i, int(order[0])))
            return out
        fwd = direction(A, B)
        if case["cross"]:
            rev = {}
            for j, i in direction(B, A):
                assert j not in rev          # a reverse list never repeats a key: one match per query at most
                rev[j] = i
            fwd = [(i, j) for i, j in fwd if rev.setdefault(j, 0) == i]
        assert [list(m) for m in fwd] == case["matches"], case["name"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", kat_derived_cases(), ids=lambda c: c["name"])
def test_gpu_gives_the_derived_answers(matcher, case):
    A, B = kat_descriptors(case["A"]), kat_descriptors(case["B"])
    matcher.upload(0, A)
    matcher.upload(1, B)
    (l, r, _), = matcher.match_pairs([(0, 1)], 0.8, case["cross"])
    assert [[int(a), int(b)] for a, b in zip(l, r)] == case["matches"], case["name"]
