"""CPU-only: pins the oracle (test infrastructure) to the reference and to the committed fixtures."""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import ROOT, kat_cases, kat_descriptors
from oracle import oracle
from sfm_amd import synth


@pytest.mark.parametrize("case", kat_cases(), ids=lambda c: c["name"])
def test_reference_known_answers(case):
    # outputs of the reference's BruteForceMatcher::Match recorded in tests/golden/kat.json
    A, B = kat_descriptors(case["A"]), kat_descriptors(case["B"])
    for cross in ([case["cross"]] if case["cross"] is not None else [False, True]):
        l, r, _ = oracle.match_pair(A, B, 0.8, cross)
        assert [[int(a), int(b)] for a, b in zip(l, r)] == case["matches"]


def test_l2_metric_equals_reference_build_bit_for_bit():
    ref = oracle.ref_metric_lib()
    if ref is None:
        pytest.skip("oracle/_ref/libref_metric.so not built (needs the reference checkout)")
    fp = ctypes.POINTER(ctypes.c_float)
    rng = np.random.default_rng(2024)
    for scale in (1.0, 1e-3, 255.0):
        X = (rng.standard_normal((3000, 128)) * scale).astype(np.float32)
        Y = (rng.standard_normal((3000, 128)) * scale).astype(np.float32)
        Y[::7] = X[::7]                 # exact zeros
        Y[1::7] = X[1::7] + np.float32(1e-4 * scale)
        for a, b in zip(X, Y):
            got = oracle.l2_metric(a, b)
            exp = np.float32(ref.ref_L2_metric(a.ctypes.data_as(fp), b.ctypes.data_as(fp)))
            assert got.view(np.uint32) == exp.view(np.uint32)
    R = synth.rootsift_rows(2000, 99)
    for a, b in zip(R[:-1], R[1:]):
        got = oracle.l2_metric(a, b)
        exp = np.float32(ref.ref_L2_metric(a.ctypes.data_as(fp), b.ctypes.data_as(fp)))
        assert got.view(np.uint32) == exp.view(np.uint32)


def _golden():
    with open(os.path.join(ROOT, "tests", "golden", "oracle_cases.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _golden(), ids=lambda c: c["name"])
def test_oracle_reproduces_committed_fixtures(case):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    got = make_golden.run_case((case["name"], case["n_a"], case["n_b"], case["planted"], case["seed"],
                                case["ratio"], case["cross"]))
    assert got == case


def _golden_sets():
    with open(os.path.join(ROOT, "tests", "golden", "oracle_cases.json")) as f:
        return json.load(f)["set_cases"]


@pytest.mark.parametrize("case", _golden_sets(), ids=lambda c: c["name"])
def test_oracle_reproduces_committed_set_fixtures(case):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden
    assert make_golden.run_set_case((case["name"], case["n_images"], case["n_features"], case["seed"])) == case


def test_threaded_oracle_equals_serial():
    A, B, _ = synth.make_pair(300, 260, 120, seed=8)
    for cross in (True, False):
        s = oracle.match_pair(A, B, 0.8, cross, threads=1)
        for t in (2, 3, 8):
            p = oracle.match_pair(A, B, 0.8, cross, threads=t)
            assert all(np.array_equal(x, y) for x, y in zip(s, p))


def test_reference_test_recipe_exact_copies_are_found():
    # src/feature_matcher/match_test.cc:9-63: rhs rows 0..511 are exact copies of shuffled lhs rows,
    # so shuffle[i] -> i must be matched (distance exactly 0) for every i < 512.
    rng = np.random.default_rng(7)
    feats = rng.standard_normal((1024, 128)).astype(np.float32)
    outl = rng.standard_normal((512, 128)).astype(np.float32)
    shuffle = rng.permutation(1024)
    rhs = np.concatenate([feats[shuffle[:512]], outl])
    l, r, d = oracle.match_pair(feats, rhs, 0.8, True, threads=8)
    found = dict(zip(l.tolist(), r.tolist()))
    assert all(found.get(int(shuffle[i])) == i for i in range(512))
    assert all(d[k] == 0.0 for k in range(len(l)) if r[k] < 512)


def test_edge_cases():
    E = np.zeros((0, 128), np.float32)
    A, B, _ = synth.make_pair(10, 12, 4, seed=3)
    for X, Y in ((E, B), (A, E), (E, E)):
        for cross in (True, False):
            l, r, d = oracle.match_pair(X, Y, 0.8, cross)
            assert len(l) == 0
    # a single candidate always matches (second distance stays FLT_MAX)
    l, r, d = oracle.match_pair(A, B[:1], 0.8, False)
    assert l.tolist() == list(range(10)) and set(r.tolist()) == {0}


def test_exhaustive_pairs_order():
    # src/feature_matcher/exhaustive_pair_builder.cc:3-19 through a std::set: lexicographic (min, max)
    p = oracle.exhaustive_pairs([7, 3, 5, 11])
    assert p.tolist() == [[3, 5], [3, 7], [3, 11], [5, 7], [5, 11], [7, 11]]
    assert len(oracle.exhaustive_pairs(range(100))) == 4950


def _numpy_model(A, B, ratio, cross):
    """A second, independent restatement of brute_force_matcher.cc:13-41,62-86 and metric.hpp:15-22
    (numpy float32 arithmetic, Python control flow), to catch a misreading shared by nobody."""
    def direction(X, Y):
        out = []
        if len(Y) == 0:
            return out
        # metric.hpp:15-22: one float accumulator, dimensions in order, every operation rounded to float
        acc = np.zeros((len(X), len(Y)), np.float32)
        for k in range(128):
            d = (X[:, k][:, None] - Y[:, k][None, :]).astype(np.float32)
            acc = (acc + (d * d).astype(np.float32)).astype(np.float32)
        dist = np.sqrt(acc).astype(np.float32)
        fmax = np.float32(np.finfo(np.float32).max)
        for i in range(len(X)):
            best, second, bi = fmax, fmax, -1
            for j in range(len(Y)):                       # :22-33, strict comparisons
                v = dist[i, j]
                if v < best:
                    second, best, bi = best, v, j
                elif v < second:
                    second = v
            if float(best) < ratio * float(second):      # :35, in double
                out.append((i, bi, best))
        return out
    fwd = direction(A, B)
    if not cross:
        return fwd
    rev = {i: j for i, j, _ in direction(B, A)}           # :69-72
    kept = []
    for i, j, d in fwd:
        if j not in rev:
            rev[j] = 0                                    # :75-78: map_[j] default-inserts 0
        if rev[j] == i:
            kept.append((i, j, d))
    return kept


@pytest.mark.parametrize("na,nb,planted,seed", [(1, 1, 1, 1), (7, 1, 0, 2), (23, 40, 12, 3), (40, 23, 12, 4), (64, 64, 40, 5)])
@pytest.mark.parametrize("cross", [True, False])
def test_oracle_equals_an_independent_numpy_model(na, nb, planted, seed, cross):
    A, B, _ = synth.make_pair(na, nb, planted, seed=seed)
    if seed == 5:                      # exact duplicates and ties as well
        B[3] = B[9]
        A[0] = B[3]
        A[11] = A[12]
    l, r, d = oracle.match_pair(A, B, 0.8, cross)
    exp = _numpy_model(A, B, 0.8, cross)
    assert l.tolist() == [e[0] for e in exp] and r.tolist() == [e[1] for e in exp]
    assert d.view(np.uint32).tolist() == [int(np.float32(e[2]).view(np.uint32)) for e in exp]
