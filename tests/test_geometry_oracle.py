"""CPU-only checks of the geometric-verification oracle (oracle/geometry_oracle.cc): the fit against
numpy's SVD, the error function against a direct evaluation, the bookkeeping constants, and the
two samplers against each other (statistically -- which is all the reference itself offers: it seeds
a fresh generator from std::random_device in every round)."""
import math

import numpy as np

from oracle import geometry
from sfm_amd import synth


def test_fit_equals_the_svd_route():
    l, r, _ = synth.make_two_view_matches(300, 0.7, seed=5)
    for k in range(6):
        idx = geometry.sample(300, 1, 0, k)
        assert len(set(idx.tolist())) == 8 and idx.min() >= 0 and idx.max() < 300
        F = geometry.fit(l, r, idx)
        # fundamental_solver.cc:16-52 with numpy: null vector of A^T A, then the two largest singular values
        A = np.array([[r[i, 0] * l[i, 0], r[i, 0] * l[i, 1], r[i, 0], r[i, 1] * l[i, 0], r[i, 1] * l[i, 1], r[i, 1],
                       l[i, 0], l[i, 1], 1.0] for i in idx])
        G = np.linalg.svd(A.T @ A)[2][-1].reshape(3, 3)
        U, S, Vt = np.linalg.svd(G)
        want = U @ np.diag([S[0], S[1], 0.0]) @ Vt
        want *= np.sign((want * F).sum())
        assert np.abs(F / np.linalg.norm(F) - want / np.linalg.norm(want)).max() < 1e-9
        assert np.linalg.svd(F)[1][2] < 1e-15 * np.linalg.svd(F)[1][0]
        # (The sample's own errors under F are NOT small: the fit works on raw pixel positions and the
        # rank-two step moves the model away from them -- fundamental_solver.cc:10-11 notes the same.)


def test_error_is_the_epipolar_line_error():
    rng = np.random.default_rng(3)
    F = rng.normal(size=(3, 3))
    for _ in range(20):
        x, xp = rng.uniform(0, 1000, 2), rng.uniform(0, 1000, 2)
        lh, rh = np.array([x[0], x[1], 1.0]), np.array([xp[0], xp[1], 1.0])
        lf, rf = F @ lh, F.T @ rh
        want = (rh @ F @ lh) ** 2 * (1 / (lf[0] ** 2 + lf[1] ** 2) + 1 / (rf[0] ** 2 + rf[1] ** 2))
        assert math.isclose(geometry.error(x, xp, F), want, rel_tol=1e-9)


def test_rounds_needed_follows_the_formula():
    # ransac.hpp:128-131: ceil(log(0.01) / log(1 - w^8)), never above the 4096 rounds there are
    assert geometry.rounds_needed(500, 1000) == math.ceil(math.log(0.01) / math.log(1 - 0.5 ** 8)) == 1177
    assert geometry.rounds_needed(990, 1000) == 2
    assert geometry.rounds_needed(100, 1000) == 4096
    assert geometry.rounds_needed(8, 1000) == 4096      # w^8 vanishes against 1: N stays


def test_both_samplers_find_the_same_structure():
    l, r, is_in = synth.make_two_view_matches(800, 0.8, seed=21, noise=0.3)
    found = {}
    for sampler in ("counter", "std"):
        ok, F, flags, rounds = geometry.compute_fundamental(l, r, seed=5, pair=0, sampler=sampler)
        assert ok and rounds < 4096
        found[sampler] = flags.astype(bool)
        assert (found[sampler] & is_in).sum() > 0.6 * is_in.sum()
        assert (found[sampler] & ~is_in).sum() <= 6
    both = (found["counter"] & found["std"]).sum()
    assert both > 0.6 * min(found["counter"].sum(), found["std"].sum())


def test_fewer_than_eight_matches():
    l, r, _ = synth.make_two_view_matches(7, 1.0, seed=1)
    ok, F, flags, rounds = geometry.compute_fundamental(l, r)
    assert not ok and rounds == 0 and not F.any() and not flags.any()


def test_golden_cases_reproduce():
    """tests/golden/geometry_cases.json: the pair (generator, oracle) gives the same bits here as where
    the file was written (tests/golden/make_golden_next.py)."""
    import json
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_next as g
    with open(os.path.join(ROOT, "tests", "golden", "geometry_cases.json")) as f:
        want = json.load(f)
    got = [g.geometry_case(c) for c in g.GEOMETRY_CASES]
    assert got == want["cases"]
    by = {w["name"]: w for w in want["cases"]}
    assert by["clean"]["rounds"] == 1 and by["clean"]["n_inliers"] == 50      # everything fits: stop at once
    assert by["too_few"]["rounds"] == 0 and not by["too_few"]["ok"]
    assert by["hopeless"]["rounds"] == 4096 and not by["hopeless"]["ok"]
    assert by["mostly_inliers"]["ok"] and by["mostly_inliers"]["rounds"] < 200
