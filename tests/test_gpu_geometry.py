"""GPU parity of the geometric verification (sfm_hip_geometry_verify_pairs) against the CPU oracle
(oracle/geometry_oracle.cc) through the C ABI: the fundamental matrix bit for bit, the inlier flags,
the verdict and the number of RANSAC rounds, with the same counter-based sampler on both sides.

Parity with the reference itself is unpinned and could only be statistical (it seeds its sampler
from std::random_device every round and takes its singular vectors from Eigen's JacobiSVD); see the
oracle's header.
"""
import numpy as np
import pytest

import sfm_amd
from sfm_amd import synth
from oracle import geometry

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher():
    m = sfm_amd.Matcher(0)
    yield m
    m.close()


def _same(got, want):
    ok, F, flags, rounds = got
    wok, wF, wflags, wrounds = want
    assert rounds == wrounds
    assert F.view(np.uint64).tolist() == wF.view(np.uint64).tolist()
    assert flags.tolist() == wflags.tolist()
    assert ok == wok


@pytest.mark.parametrize("n,frac", [(200, 0.7), (1000, 0.9), (1000, 0.5), (60, 0.9), (8, 1.0), (33, 0.95)])
def test_one_pair_identical_to_oracle(matcher, n, frac):
    l, r, _ = synth.make_two_view_matches(n, frac, seed=1000 + n)
    got, = matcher.geometry_verify_pairs([(l, r)], seed=99, first_pair_index=5)
    _same(got, geometry.compute_fundamental(l, r, seed=99, pair=5))


def test_many_pairs_and_degenerate_ones(matcher):
    sets = []
    for k, (n, frac) in enumerate([(300, 0.8), (0, 1.0), (7, 1.0), (500, 0.3), (120, 0.6), (2000, 0.7), (40, 0.2)]):
        l, r, _ = synth.make_two_view_matches(n, frac, seed=70 + k)
        sets.append((l, r))
    # a pair where every match fits one model exactly: RANSAC stops in its first round
    l, r, _ = synth.make_two_view_matches(50, 1.0, seed=3, noise=0.0)
    sets.append((l, r))
    got = matcher.geometry_verify_pairs(sets, seed=4)
    for k, ((l, r), g) in enumerate(zip(sets, got)):
        _same(g, geometry.compute_fundamental(l, r, seed=4, pair=k))
    assert got[1][0] is False and got[2][0] is False and got[1][3] == 0      # fewer than eight matches
    assert got[0][0] and got[5][0] and not got[6][0]
    # the same pairs handed over in two calls: a pair's stream does not depend on the cut
    again = matcher.geometry_verify_pairs(sets[:3], seed=4) + matcher.geometry_verify_pairs(sets[3:], seed=4, first_pair_index=3)
    for a, b in zip(got, again):
        _same(a, b)


def test_inliers_are_the_planted_ones(matcher):
    l, r, is_in = synth.make_two_view_matches(1500, 0.8, seed=11, noise=0.3)
    (ok, F, flags, rounds), = matcher.geometry_verify_pairs([(l, r)], seed=1)
    assert ok and rounds < 4096
    found = flags.astype(bool)
    assert (found & is_in).sum() > 0.6 * is_in.sum()
    assert (found & ~is_in).sum() < 0.05 * (~is_in).sum() + 3
    # rank two, and the planted matches lie on their epipolar lines
    assert abs(np.linalg.det(F / np.linalg.norm(F))) < 1e-12


def test_bad_arguments_are_refused(matcher):
    from sfm_amd import binding
    l, r, _ = synth.make_two_view_matches(20, 1.0, seed=1)
    with pytest.raises(binding.SfmHipError, match="start at 0"):
        matcher.geometry_verify_flat(np.array([1, 20], np.int64), l, r)
    with pytest.raises(binding.SfmHipError, match="ascend"):
        matcher.geometry_verify_flat(np.array([0, 20, 10], np.int64), l, r)
    F, flags, counts, rounds = matcher.geometry_verify_flat(np.array([0], np.int64), l[:0], r[:0])
    assert len(counts) == 0 and len(flags) == 0


def test_statistics_count_the_work_of_timed_calls(matcher):
    """With profiling on a call reports, through sfm_hip_geometry_get_stats, the rounds its launches
    worked on (at least the rounds the pairs report: a launch takes 256 rounds of a pair at once),
    the (match, round) pairs measured and kernel durations; the results are the same either way."""
    sets = [synth.make_two_view_matches(n, f, seed=70 + k)[:2] for k, (n, f) in enumerate([(300, 0.8), (500, 0.4), (5, 1.0)])]
    plain = matcher.geometry_verify_pairs(sets, seed=2)
    matcher.set_profiling(True)
    matcher.geometry_reset_stats()
    timed = matcher.geometry_verify_pairs(sets, seed=2)
    st = matcher.geometry_stats()
    matcher.set_profiling(False)
    for a, b in zip(plain, timed):
        _same(a, b)
    rounds = [g[3] for g in timed]
    assert st["calls"] == 1 and st["pairs"] == 2          # the pair with five matches cannot be sampled
    assert st["timed_launches"] >= 1 and st["fit_ms"] > 0 and st["count_ms"] > 0
    assert st["rounds"] >= sum(rounds) and st["rounds"] % 1 == 0
    assert st["rounds"] <= sum(-(-r // 256) * 256 for r in rounds)
    assert 300 * rounds[0] + 500 * rounds[1] <= st["match_rounds"] <= 500 * st["rounds"]
    # without profiling nothing but the call and pair counts moves
    matcher.geometry_verify_pairs(sets, seed=2)
    again = matcher.geometry_stats()
    assert again["calls"] == 2 and again["rounds"] == st["rounds"] and again["timed_launches"] == st["timed_launches"]


@pytest.mark.parametrize("n_pairs", [6000, 40000])
def test_calls_of_many_pairs_get_fewer_rounds_per_launch(matcher, n_pairs):
    """A launch holds as many runs of 256 rounds per pair as a fixed amount of device memory allows: eight for
    a few thousand pairs, six for 6,000, one for 40,000.  The result does not depend on it: a sample of
    the pairs against the oracle, and the first pairs against a call of their own (eight runs)."""
    rng = np.random.default_rng(n_pairs)
    base = [synth.make_two_view_matches(int(rng.integers(8, 40)), float(rng.uniform(0.3, 1.0)), seed=900 + k, noise=0.5)[:2]
            for k in range(50)]
    sets = [base[k % 50] for k in range(n_pairs)]
    got = matcher.geometry_verify_pairs(sets, seed=17)
    for k in list(range(0, n_pairs, n_pairs // 23)) + [n_pairs - 1]:
        l, r = sets[k]
        _same(got[k], geometry.compute_fundamental(l, r, seed=17, pair=k))
    alone = matcher.geometry_verify_pairs(sets[:60], seed=17)
    for a, b in zip(alone, got[:60]):
        _same(a, b)
    assert max(g[3] for g in got) > 1024      # some pairs go through several launches
