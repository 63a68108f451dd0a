"""Parity of the HIP path (through the C ABI) with the CPU oracle.

Bar: match indices identical, distances bit-identical (the exact re-rank
reproduces the reference's fp32 arithmetic, so no tolerance is needed; the
stated tolerance of 1e-4 relative is therefore met with room to spare).
"""
import numpy as np
import pytest

from conftest import kat_cases, kat_descriptors
from oracle import oracle
from sfm_amd import synth

pytestmark = pytest.mark.gpu


def run_pair(m, A, B, ratio=0.8, cross=True, ids=(0, 1)):
    m.upload(ids[0], A)
    m.upload(ids[1], B)
    (l, r, d), = m.match_pairs([ids], ratio, cross)
    return l, r, d


def check_pair(m, A, B, cross, ratio=0.8):
    got = run_pair(m, A, B, ratio, cross)
    exp = oracle.match_pair(A, B, ratio, cross, threads=8)
    assert_same(got, exp)
    return got


def assert_same(got, exp, what=""):
    gl, gr, gd = got
    el, er, ed = exp
    assert gl.tolist() == el.tolist(), what + " lhs indices differ"
    assert gr.tolist() == er.tolist(), what + " rhs indices differ"
    assert gd.view(np.uint32).tolist() == ed.view(np.uint32).tolist(), what + " distances differ in bits"


@pytest.mark.parametrize("case", kat_cases(), ids=lambda c: c["name"])
def test_known_answers(matcher, case):
    A, B = kat_descriptors(case["A"]), kat_descriptors(case["B"])
    for cross in ([case["cross"]] if case["cross"] is not None else [False, True]):
        l, r, d = run_pair(matcher, A, B, 0.8, cross)
        assert [[int(a), int(b)] for a, b in zip(l, r)] == case["matches"], (case["name"], cross)


@pytest.mark.parametrize("na,nb,planted", [
    (1, 1, 1), (1, 7, 0), (5, 3, 2), (31, 33, 10), (64, 64, 32), (127, 129, 60), (128, 128, 64),
    (200, 1000, 100), (1000, 200, 100), (513, 777, 300), (1024, 1024, 512), (2048, 1536, 700),
])
@pytest.mark.parametrize("cross", [True, False])
def test_random_pairs_match_oracle(matcher, na, nb, planted, cross):
    A, B, _ = synth.make_pair(na, nb, planted, seed=1000 + na * 7 + nb)
    got = run_pair(matcher, A, B, 0.8, cross)
    exp = oracle.match_pair(A, B, 0.8, cross, threads=8)
    assert_same(got, exp, "%dx%d cross=%s" % (na, nb, cross))


def test_empty_images(matcher):
    A, B, _ = synth.make_pair(50, 60, 20, seed=5)
    E = np.zeros((0, 128), np.float32)
    for X, Y in ((E, B), (A, E), (E, E)):
        for cross in (True, False):
            l, r, d = run_pair(matcher, X, Y, 0.8, cross)
            assert len(l) == 0 and len(r) == 0


def test_gaussian_descriptors_like_reference_test(matcher):
    # The recipe of the reference's own matcher test (src/feature_matcher/match_test.cc:9-63):
    # 1024 N(0,1) rows; the second image is 512 exact copies of shuffled rows plus 512 outliers.
    rng = np.random.default_rng(12345)
    feats = rng.standard_normal((1024, 128)).astype(np.float32)
    outl = rng.standard_normal((512, 128)).astype(np.float32)
    shuffle = rng.permutation(1024)
    rhs = np.concatenate([feats[shuffle[:512]], outl])
    got = run_pair(matcher, feats, rhs, 0.8, True)
    exp = oracle.match_pair(feats, rhs, 0.8, True, threads=8)
    assert_same(got, exp)
    # the 512 exact copies have distance 0 and must all be present: shuffle[i] -> i
    found = dict(zip(got[0].tolist(), got[1].tolist()))
    for i in range(512):
        assert found.get(int(shuffle[i])) == i


def test_duplicates_and_ties(matcher):
    # exact duplicates on both sides: ties for nearest are rejected (d < 0.8*d is false),
    # and the lowest index must win wherever an index is reported.
    A, B, _ = synth.make_pair(300, 300, 150, seed=77)
    B[10] = B[20]
    B[200:210] = A[5]
    A[100] = A[101]
    for cross in (True, False):
        got = run_pair(matcher, A, B, 0.8, cross)
        exp = oracle.match_pair(A, B, 0.8, cross)
        assert_same(got, exp)
    # ratio above 1 accepts nearly everything, which exposes the chosen index of every query
    got = run_pair(matcher, A, B, 5.0, False)
    exp = oracle.match_pair(A, B, 5.0, False)
    assert_same(got, exp)


def test_all_identical_forces_full_rescan(matcher):
    # every distance is equal: no candidate list can be proven complete, so every query
    # goes through the exact full scan; result: nothing passes the ratio test.
    A = np.tile(synth.rootsift_rows(1, 3), (257, 1))
    B = np.tile(synth.rootsift_rows(1, 3), (300, 1))
    matcher.reset_stats()
    got = run_pair(matcher, A, B, 0.8, True)
    assert len(got[0]) == 0
    got = run_pair(matcher, A, B, 1.5, False)
    exp = oracle.match_pair(A, B, 1.5, False)
    assert_same(got, exp)
    assert matcher.stats()["rescanned_queries"] > 0


def test_unnormalised_large_values(matcher):
    # descriptors in the range an un-normalised SIFT or the padded ORB of
    # src/feature_extractor/opencv_orb_extractor.cc:22-27 would produce (0..255)
    rng = np.random.default_rng(9)
    A = rng.integers(0, 256, (400, 128)).astype(np.float32)
    B = rng.integers(0, 256, (500, 128)).astype(np.float32)
    B[:200] = A[rng.permutation(400)[:200]] + rng.integers(-2, 3, (200, 128)).astype(np.float32)
    for cross in (True, False):
        got = run_pair(matcher, A, B, 0.8, cross)
        exp = oracle.match_pair(A, B, 0.8, cross)
        assert_same(got, exp)


def test_many_pairs_one_call(matcher):
    imgs = synth.make_image_set(6, 500, 0.5, seed=42)
    sizes = [500, 300, 129, 500, 64, 411]
    imgs = [im[:n] for im, n in zip(imgs, sizes)]
    for i, im in enumerate(imgs):
        matcher.upload(100 + i, im)
    pairs = [(100 + i, 100 + j) for i in range(6) for j in range(i + 1, 6)]
    res = matcher.match_pairs(pairs, 0.8, True)
    for (i, j), got in zip(pairs, res):
        exp = oracle.match_pair(imgs[i - 100], imgs[j - 100], 0.8, True)
        assert_same(got, exp, "pair %d-%d" % (i, j))


def test_error_reporting(matcher):
    import sfm_amd
    with pytest.raises(sfm_amd.SfmHipError, match="not uploaded|unknown image id"):
        matcher.match_pairs([(987654, 987655)])
    with pytest.raises(ValueError):
        matcher.upload(1, np.zeros((4, 64), np.float32))


def test_unknown_engine_is_rejected(matcher):
    import sfm_amd
    lib = sfm_amd.library()
    assert lib.sfm_hip_set_engine(matcher._h, 7) != 0
    assert b"unknown engine" in lib.sfm_hip_last_error(matcher._h)


@pytest.mark.parametrize("scale", [1e-6, 1.0 / 512.0, 300.0, 6e4, 1e7, 1e15, 3e17])
def test_descriptor_scale_does_not_change_matches(matcher, scale):
    """The split-fp16 engine rescales every image by a power of two before splitting; values far
    outside the fp16 range (or far below it) must give the reference's matches all the same.  At
    1e15 and 3e17 the squared norms (1e30, 9e34) are beyond what the candidate keys can express: the
    reference's arithmetic (src/metric.hpp:15-22) is still finite there, and so must the answer be."""
    A, B, _ = synth.make_pair(700, 900, 300, seed=77)
    A = (A * np.float32(scale)).astype(np.float32)
    B = (B * np.float32(scale)).astype(np.float32)
    matcher.reset_stats()
    l, _, _ = check_pair(matcher, A, B, True)
    assert len(l) >= 290                      # the planted correspondences survive at every scale
    if scale >= 1e15:                         # ... by the exact scan, not by luck
        assert matcher.stats()["rescanned_queries"] >= 700 + 900


@pytest.mark.parametrize("side", ["first", "second", "both"])
def test_a_few_rows_with_huge_norms(matcher, side):
    """Rows whose squared norm lies between 1e29 and the fp32 maximum, among ordinary rows, on either
    side of the pair: to the reference they are ordinary descriptors with finite distances (and near
    each other they match); the distance kernel cannot tell them from padding rows, so the pair has
    to be answered exactly."""
    A, B, _ = synth.make_pair(600, 800, 250, seed=81)
    A, B = A.copy(), B.copy()
    rng = np.random.default_rng(3)
    big = (rng.random((6, 128), dtype=np.float32) * np.float32(2e15)).astype(np.float32)
    near = (big * (np.float32(1) + rng.normal(0, 1e-4, big.shape).astype(np.float32))).astype(np.float32)
    if side in ("first", "both"):
        A[[0, 17, 100, 256, 400, 599]] = big
    if side in ("second", "both"):
        B[[3, 64, 255, 256, 511, 799]] = near if side == "both" else big
    matcher.reset_stats()
    for cross in (False, True):
        l, r, _ = check_pair(matcher, A, B, cross)
    if side == "both":
        assert {(0, 3), (17, 64), (599, 799)} <= set(zip(l.tolist(), r.tolist()))
    assert matcher.stats()["rescanned_queries"] >= 600


def test_images_of_very_different_scale(matcher):
    """Each image has its own scale factor; a pair of images whose magnitudes differ by 2^20 still
    goes through both engines exactly (nothing matches, as in the reference)."""
    A, B, _ = synth.make_pair(300, 400, 100, seed=78)
    B = (B * np.float32(2.0 ** 20)).astype(np.float32)
    check_pair(matcher, A, B, True)


def test_one_huge_element_among_tiny_rows(matcher):
    """An image whose scale is set by a single outlier element: the other rows lose precision in
    the split planes, the completeness bound notices, and the exact path answers."""
    A, B, _ = synth.make_pair(600, 600, 300, seed=79)
    A = (A * np.float32(1e-3)).astype(np.float32)
    B = (B * np.float32(1e-3)).astype(np.float32)
    A[17, 5] = np.float32(900.0)
    check_pair(matcher, A, B, True)


@pytest.mark.parametrize("na,nb,planted", [(8192, 8192, 4096), (16384, 16384, 6000)])
def test_full_size_pairs_match_oracle(matcher, na, nb, planted):
    """The headline sizes (BASELINE.json configs 1 and 4), index lists and distance bits."""
    A, B, truth = synth.make_pair(na, nb, planted, seed=2024)
    l, r, d = check_pair(matcher, A, B, True)
    found = set(zip(l.tolist(), r.tolist()))
    assert set(map(tuple, truth.tolist())) <= found          # every planted correspondence survives
    assert l.tolist() == sorted(l.tolist())                  # brute_force_matcher.cc:16: ascending lhs


@pytest.mark.parametrize("na,nb,planted", [(300, 40000, 200), (5000, 20000, 2500), (20000, 700, 500)])
def test_more_columns_than_one_chunk_holds(matcher, na, nb, planted):
    """Second images beyond 16384 rows are cut into several column chunks per row block; first
    images beyond 256 rows into several row blocks: the merged lists must still give the
    reference's answer."""
    A, B, _ = synth.make_pair(na, nb, planted, seed=31 + na)
    check_pair(matcher, A, B, True)
    check_pair(matcher, A, B, False)


def test_swapped_pair_is_the_mirror_image(matcher):
    """Cross-checked matches of (A, B) and (B, A) mirror each other, except for the reference's
    map_[j] quirk, which can only add matches whose lhs index is 0 (brute_force_matcher.cc:75-78)."""
    A, B, _ = synth.make_pair(3000, 2500, 1200, seed=99)
    l, r, _ = run_pair(matcher, A, B, 0.8, True)
    l2, r2, _ = run_pair(matcher, B, A, 0.8, True)
    ab = {(i, j) for i, j in zip(l.tolist(), r.tolist()) if i != 0}
    ba = {(j, i) for i, j in zip(l2.tolist(), r2.tolist()) if i != 0}
    assert {p for p in ab if p[1] != 0} == {p for p in ba if p[0] != 0}


def test_ragged_batch_in_one_call(matcher):
    """Pairs of very different sizes (and an empty image) in a single sfm_hip_match_pairs call."""
    sizes = [0, 1, 255, 256, 257, 1000, 3000]
    base = synth.make_image_set(len(sizes), max(sizes), 0.6, seed=500)
    imgs = [np.ascontiguousarray(base[k][:n]) for k, n in enumerate(sizes)]
    for k, im in enumerate(imgs):
        matcher.upload(100 + k, im)
    pairs = [(100 + a, 100 + b) for a in range(len(sizes)) for b in range(a + 1, len(sizes))]
    res = matcher.match_pairs(pairs, 0.8, True)
    for (a, b), got in zip(pairs, res):
        exp = oracle.match_pair(imgs[a - 100], imgs[b - 100], 0.8, True, threads=4)
        assert_same(got, exp, "pair %s" % ((a, b),))


def test_ragged_batch_with_first_images_of_several_sweeps(matcher):
    """One call whose first images range from empty to several compaction sweeps (2,048 rows each):
    the compaction's workgroups past a pair's last row must leave its count and lists alone."""
    sizes = [5000, 100, 2049, 4096, 0, 2048]
    base = synth.make_image_set(len(sizes), max(sizes), 0.6, seed=501)
    imgs = [np.ascontiguousarray(base[k][:n]) for k, n in enumerate(sizes)]
    for k, im in enumerate(imgs):
        matcher.upload(200 + k, im)
    pairs = [(200 + a, 200 + b) for a in range(len(sizes)) for b in range(len(sizes)) if a != b]
    for cross in (True, False):
        res = matcher.match_pairs(pairs, 0.8, cross)
        for (a, b), got in zip(pairs, res):
            exp = oracle.match_pair(imgs[a - 200], imgs[b - 200], 0.8, cross, threads=4)
            assert_same(got, exp, "pair %s cross %s" % ((a, b), cross))


@pytest.mark.parametrize("scale", [1.0, 37.0])
def test_ratio_test_at_its_threshold(matcher, scale):
    """Queries whose nearest / second-nearest distance ratio straddles 0.8 within a few parts per
    million: the decision must come from the exact arithmetic, never from the approximate keys."""
    rng = np.random.default_rng(5)
    n = 600
    A = synth.rootsift_rows(n, 123)
    B = synth.rootsift_rows(2 * n + 300, 124)
    for i in range(n):
        e1 = rng.standard_normal(128)
        e1 /= np.linalg.norm(e1)
        e2 = rng.standard_normal(128)
        e2 -= e2.dot(e1) * e1
        e2 /= np.linalg.norm(e2)
        delta = 0.05 + 0.1 * rng.random()
        eps = (rng.random() - 0.5) * 2e-5                      # ratio = 0.8 * (1 + eps)
        B[2 * i] = (A[i] + delta * 0.8 * (1 + eps) * e1).astype(np.float32)
        B[2 * i + 1] = (A[i] + delta * e2).astype(np.float32)
    A = (A * np.float32(scale)).astype(np.float32)
    B = (B * np.float32(scale)).astype(np.float32)
    l, r, d = check_pair(matcher, A, B, False)
    assert 0.2 * n < len(l) < 0.8 * n                          # the cases really fall on both sides
    check_pair(matcher, A, B, True)


def _queue_call(m, lib, pair, stride, bufs):
    import ctypes
    p = np.array([pair], np.int64)
    cnt, lhs, rhs, dist = bufs
    rc = lib.sfm_hip_match_pairs(m._h, p.ctypes.data, 1, ctypes.c_double(0.8), 1, cnt.ptr, lhs.ptr, rhs.ptr, dist.ptr, stride)
    assert rc == 0, lib.sfm_hip_last_error(m._h)


def test_many_calls_in_flight_each_with_buffers_of_its_own(matcher):
    """Twelve single-pair calls queued without a sync in between (more than the context has lanes),
    every call with output buffers of its own: every call's result must be the oracle's."""
    import sfm_amd
    from sfm_amd.binding import _Pinned
    lib = sfm_amd.library()
    sizes = [700, 333, 1024, 512, 90]
    imgs = [im[:n] for im, n in zip(synth.make_image_set(5, 1024, 0.5, seed=77), sizes)]
    for i, im in enumerate(imgs):
        matcher.upload(300 + i, im)
    pairs = [(300 + a, 300 + b) for a in range(5) for b in range(5) if a != b][:12]
    stride = 1024
    sets = [[_Pinned(lib, s, t) for s, t in (((1,), np.int32), ((1, stride), np.int32), ((1, stride), np.int32),
                                             ((1, stride), np.float32))] for _ in pairs]
    try:
        for pair, bufs in zip(pairs, sets):
            _queue_call(matcher, lib, pair, stride, bufs)
        matcher.sync()
        for (a, b), (cnt, lhs, rhs, dist) in zip(pairs, sets):
            c = int(cnt.array[0])
            got = (lhs.array[0, :c].astype(np.int64), rhs.array[0, :c].astype(np.int64), dist.array[0, :c].copy())
            assert_same(got, oracle.match_pair(imgs[a - 300], imgs[b - 300], 0.8, True), "pair %s" % ((a, b),))
    finally:
        matcher.sync()
        for bufs in sets:
            for b in bufs:
                b.free()


def test_calls_in_flight_that_reuse_output_buffers_land_in_call_order(matcher):
    """Two sets of output buffers used in turn by ten calls in flight.  The last but one call on
    either set has a slow tail (identical descriptors: every query is answered by the exact full
    scan) and the small pair after it is done long before that tail is: each set must still end up
    holding its LAST call's result."""
    import sfm_amd
    from sfm_amd.binding import _Pinned
    lib = sfm_amd.library()
    big = [np.tile(synth.rootsift_rows(1, 3), (n, 1)) for n in (3000, 2900)]
    small = [im[:n] for im, n in zip(synth.make_image_set(4, 256, 0.5, seed=92), [200, 256, 64, 130])]
    for i, im in enumerate(big):
        matcher.upload(400 + i, im)
    for i, im in enumerate(small):
        matcher.upload(410 + i, im)
    where = {400: big[0], 401: big[1], 410: small[0], 411: small[1], 412: small[2], 413: small[3]}
    calls = [(410, 411), (412, 413), (411, 410), (413, 412), (410, 413), (413, 410), (400, 401), (401, 400),
             (411, 412), (412, 410)]
    stride = 4096
    sets = [[_Pinned(lib, s, t) for s, t in (((1,), np.int32), ((1, stride), np.int32), ((1, stride), np.int32),
                                             ((1, stride), np.float32))] for _ in range(2)]
    try:
        for k, pair in enumerate(calls):
            _queue_call(matcher, lib, pair, stride, sets[k & 1])
        matcher.sync()
        for which in (0, 1):
            a, b = [p for k, p in enumerate(calls) if (k & 1) == which][-1]
            cnt, lhs, rhs, dist = sets[which]
            c = int(cnt.array[0])
            got = (lhs.array[0, :c].astype(np.int64), rhs.array[0, :c].astype(np.int64), dist.array[0, :c].copy())
            assert_same(got, oracle.match_pair(where[a], where[b], 0.8, True), "set %d" % which)
    finally:
        matcher.sync()
        for bufs in sets:
            for b in bufs:
                b.free()


def test_profiling_times_one_launch_in_eight(matcher):
    A, B, _ = synth.make_pair(512, 512, 100, seed=3)
    matcher.upload(0, A)
    matcher.upload(1, B)
    matcher.set_profiling(True)
    matcher.reset_stats()
    for _ in range(17):
        matcher.match_pairs_async([(0, 1)], 0.8, True)
    st = matcher.stats()
    matcher.set_profiling(False)
    assert st["pairs"] == 17
    assert st["distance_launches"] == 3            # launches 0, 8 and 16
    assert st["distance_ms"] > 0.0
    assert st["distance_flop"] == 3 * 2.0 * 512 * 512 * 128


def test_profiling_of_batch_launches_matches_the_wall_clock(matcher):
    """Launches of a batch are timed from the completion of the launch before them (no marker
    packet): the timed launches' average must agree with the wall clock of a run of back-to-back
    calls, none of them standing out."""
    import time
    A, B, _ = synth.make_pair(4096, 4096, 1000, seed=5)
    matcher.upload(0, A)
    matcher.upload(1, B)
    pairs = [(0, 1)] * 32
    for _ in range(4):
        matcher.match_pairs_async(pairs, 0.8, True)
    matcher.sync()
    matcher.set_profiling(True)
    matcher.reset_stats()
    t0 = time.perf_counter()
    for _ in range(25):
        matcher.match_pairs_async(pairs, 0.8, True)
    matcher.sync()
    wall_ms = (time.perf_counter() - t0) * 1e3 / 25
    st = matcher.stats()
    matcher.set_profiling(False)
    assert st["distance_launches"] == 4            # launches 0, 8, 16 and 24
    per_launch = st["distance_ms"] / st["distance_launches"]
    # (a launch stretched to 1.6 times its length, as a timing marker once did, would read 1.12 here;
    # the lower bound only says that the figure is a launch's duration at all: a host that falls
    # behind lengthens the wall clock, not the launches)
    assert 0.5 * wall_ms < per_launch < 1.08 * wall_ms, (per_launch, wall_ms)


def test_profiling_switched_on_and_off_between_calls(matcher):
    """Timing state (the event a timed launch is timed from) survives profiling being switched off
    and on again and a drain in between; results are unaffected."""
    A, B, _ = synth.make_pair(1500, 1200, 400, seed=11)
    matcher.upload(0, A)
    matcher.upload(1, B)
    exp = oracle.match_pair(A, B, 0.8, True, threads=4)
    matcher.reset_stats()
    total = 0
    for rounds, on in ((7, True), (5, False), (9, True), (3, False), (12, True)):
        matcher.set_profiling(on)
        outs = [matcher.match_pairs_async([(0, 1)], 0.8, True) for _ in range(rounds)]
        if rounds % 2:
            matcher.sync()
        total += rounds
    st = matcher.stats()
    matcher.set_profiling(False)
    assert st["pairs"] == total
    assert 1 <= st["distance_launches"] <= 7 + 9 + 12
    assert st["distance_ms"] > 0.0
    assert_same(matcher.match_pairs([(0, 1)], 0.8, True)[0], exp, "after the switches")


def test_nan_and_inf_in_descriptors_fall_back_to_the_exact_scan(matcher):
    """A NaN or an infinity in an image makes its norms useless for the candidate bounds: every query
    of the pair is then answered by the exact full scan, and the result is still the oracle's (a
    candidate at a NaN distance never wins a '<', brute_force_matcher.cc:26-33)."""
    A, B, _ = synth.make_pair(700, 900, 300, seed=11)
    B = B.copy()
    B[5, 17] = np.nan
    B[77, :] = np.nan
    B[300, 3] = np.inf
    matcher.reset_stats()
    for cross in (False, True):
        check_pair(matcher, A, B, cross)
    assert matcher.stats()["rescanned_queries"] >= 700


def test_contexts_come_and_go():
    """Forty contexts created, used and destroyed in turn (streams, events and work space go with
    them), and three alive at once: the last one still answers like the first."""
    import sfm_amd
    A, B, _ = synth.make_pair(300, 400, 120, seed=21)
    exp = oracle.match_pair(A, B, 0.8, True)
    for _ in range(40):
        m = sfm_amd.Matcher(0)
        m.upload(0, A)
        m.upload(1, B)
        (got,) = m.match_pairs([(0, 1)], 0.8, True)
        m.close()
    assert_same(got, exp)
    ms = [sfm_amd.Matcher(0) for _ in range(3)]
    try:
        for m in ms:
            m.upload(0, A)
            m.upload(1, B)
        outs = [m.match_pairs_async([(0, 1)], 0.8, True) for m in ms]
        for m, (cnt, lhs, rhs, dist) in zip(ms, outs):
            m.sync()
            c = int(cnt[0])
            assert_same((lhs[0, :c].astype(np.int64), rhs[0, :c].astype(np.int64), dist[0, :c].copy()), exp)
    finally:
        for m in ms:
            m.close()


def test_two_nearest_in_one_lane_list_and_ratio_at_the_threshold(matcher):
    """The distance kernel keeps two candidates per lane and row; here the two nearest columns of a
    query sit 32 apart (the same lane's list) and their ratio straddles 0.8 within parts per million,
    so neither the keys nor the lists can settle the query: it must come out right by the exact scan."""
    rng = np.random.default_rng(8)
    n = 256
    A = synth.rootsift_rows(n, 223)
    B = synth.rootsift_rows(64 * n, 224)
    for i in range(n):
        e1 = rng.standard_normal(128)
        e1 /= np.linalg.norm(e1)
        e2 = rng.standard_normal(128)
        e2 -= e2.dot(e1) * e1
        e2 /= np.linalg.norm(e2)
        delta = 0.05 + 0.1 * rng.random()
        eps = (rng.random() - 0.5) * 2e-5
        B[64 * i] = (A[i] + delta * 0.8 * (1 + eps) * e1).astype(np.float32)
        B[64 * i + 32] = (A[i] + delta * e2).astype(np.float32)
    matcher.reset_stats()
    l, r, d = check_pair(matcher, A, B, False)
    assert 0.2 * n < len(l) < 0.8 * n
    check_pair(matcher, A, B, True)
    assert matcher.stats()["rescanned_queries"] > 0


def test_two_nearest_in_one_column_list_and_ratio_at_the_threshold(matcher):
    """The mirror image for the reverse direction: the two nearest ROWS of a second-image query are
    neighbours in the first image (the same half-wave's column list) with a ratio at the threshold."""
    rng = np.random.default_rng(9)
    n = 256
    Y = synth.rootsift_rows(n, 323)
    X = synth.rootsift_rows(64 * n, 324)
    for i in range(n):
        e1 = rng.standard_normal(128)
        e1 /= np.linalg.norm(e1)
        e2 = rng.standard_normal(128)
        e2 -= e2.dot(e1) * e1
        e2 /= np.linalg.norm(e2)
        delta = 0.05 + 0.1 * rng.random()
        eps = (rng.random() - 0.5) * 2e-5
        X[64 * i] = (Y[i] + delta * 0.8 * (1 + eps) * e1).astype(np.float32)
        X[64 * i + 1] = (Y[i] + delta * e2).astype(np.float32)
    matcher.reset_stats()
    l, r, d = check_pair(matcher, X, Y, True)
    assert 0.1 * n < len(l) < 0.9 * n
    assert matcher.stats()["rescanned_queries"] > 0


@pytest.mark.parametrize("ratio", [0.99999, 1.0, 0.999])
def test_near_equal_neighbours_with_ratio_next_to_one(matcher, ratio):
    """Ratios within 1e-5 of 1: the two nearest of a query are almost equally far, in both directions
    (neighbouring rows of the first image share a column list, columns 32 apart share a lane's row
    list).  A candidate that reaches the exact re-rank under two keys must count once, or the second
    distance would equal the first and the test below the ratio would fail where the reference's
    passes."""
    rng = np.random.default_rng(10)
    n = 192
    Y = synth.rootsift_rows(n, 523)
    X = synth.rootsift_rows(64 * n, 524)
    for i in range(n):
        e1 = rng.standard_normal(128)
        e1 /= np.linalg.norm(e1)
        e2 = rng.standard_normal(128)
        e2 -= e2.dot(e1) * e1
        e2 /= np.linalg.norm(e2)
        delta = 0.05 + 0.1 * rng.random()
        eps = (rng.random() - 0.5) * 4e-5
        X[64 * i] = (Y[i] + delta * (1 + eps) * e1).astype(np.float32)
        X[64 * i + (1 if i % 2 else 32)] = (Y[i] + delta * e2).astype(np.float32)
    l, r, d = check_pair(matcher, X, Y, True, ratio)
    check_pair(matcher, X, Y, False, ratio)
    l2, _, _ = check_pair(matcher, Y, X, True, ratio)
    check_pair(matcher, Y, X, False, ratio)
    if ratio == 0.99999:                        # the queries sit on both sides of this ratio
        assert 0.2 * n < len(l) < n and 0.2 * n < len(l2) < n


def test_wait_outputs_makes_one_calls_results_valid(matcher):
    """sfm_hip_wait_outputs: a host loop that keeps the next call queued behind the running one and
    consumes results call by call, without ever draining the context."""
    sizes = [700, 333, 1024, 512, 90, 2048]
    imgs = [im[:n] for im, n in zip(synth.make_image_set(6, 2048, 0.5, seed=79), sizes)]
    for i, im in enumerate(imgs):
        matcher.upload(800 + i, im)
    calls = [[(800 + a, 800 + b) for b in range(6) if b != a] for a in range(6)]     # five pairs per call
    exp = {(a, b): oracle.match_pair(imgs[a - 800], imgs[b - 800], 0.8, True, threads=4) for c in calls for a, b in c}
    try:
        prev = None
        for call in calls + [None]:
            out = matcher.match_pairs_async(call, 0.8, True, stride=2048) if call else None
            if prev is not None:
                pcall, (cnt, lhs, rhs, dist) = prev
                matcher.wait_outputs(cnt)
                for k, pair in enumerate(pcall):
                    c = int(cnt[k])
                    assert_same((lhs[k, :c].astype(np.int64), rhs[k, :c].astype(np.int64), dist[k, :c].copy()), exp[pair],
                                "pair %s" % (pair,))
            prev = (call, out)
        matcher.wait_outputs(np.zeros(4, np.int32))      # memory no call writes: returns at once
    finally:
        matcher.sync()
        for i in range(6):
            matcher.drop(800 + i)


def test_calls_without_distances_return_the_same_matches(matcher):
    """out_dist = NULL (what the C++ matcher class passes: the reference's Match record holds no
    distance) lets the re-rank skip the exact distance of every match its keys already decide; the
    index lists must not change.  A 64-pair call (the batch variant of the re-rank) and single pairs."""
    imgs = [im[:n] for im, n in zip(synth.make_image_set(12, 2048, 0.5, seed=83), [2048, 1500, 2048, 700, 2048, 2048, 1, 1024, 2048, 333, 2048, 2000])]
    for i, im in enumerate(imgs):
        matcher.upload(900 + i, im)
    pairs = [(900 + a, 900 + b) for a in range(12) for b in range(12) if a != b][:64]
    try:
        for cross in (True, False):
            for call in (pairs, pairs[:1], pairs[5:7]):
                with_d = [tuple(x.copy() for x in t) for t in matcher.match_pairs(call, 0.8, cross)]
                cnt, lhs, rhs, _ = matcher.match_pairs_async(call, 0.8, cross, stride=2048, want_dist=False)
                matcher.sync()
                for k, (l, r, d) in enumerate(with_d):
                    c = int(cnt[k])
                    assert lhs[k, :c].tolist() == l.tolist() and rhs[k, :c].tolist() == r.tolist(), (call[k], cross)
        for a, b in pairs[:6]:
            exp = oracle.match_pair(imgs[a - 900], imgs[b - 900], 0.8, True, threads=4)
            cnt, lhs, rhs, _ = matcher.match_pairs_async([(a, b)], 0.8, True, stride=2048, want_dist=False)
            matcher.sync()
            c = int(cnt[0])
            assert lhs[0, :c].tolist() == exp[0].tolist() and rhs[0, :c].tolist() == exp[1].tolist()
    finally:
        matcher.sync()
        for i in range(12):
            matcher.drop(900 + i)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_crowds_of_near_equal_distances_around_the_ratio_threshold(matcher, seed):
    """Every query has a nearest column at distance r, a second at r / 0.8 * (1 + delta) with delta within
    a few 1e-4 .. 1e-3 of the threshold either way, and six more columns a few 1e-4 apart right behind
    it: closer together than a candidate key resolves (a key keeps eight bits of the distance's
    mantissa), so the lists cannot rank them and the decision cannot come from keys.  The answers must
    still be the oracle's, bit for bit -- through the exact distances of the candidates or the exact
    full scan."""
    rng = np.random.default_rng(seed)
    na, per = 300, 8
    A = synth.rootsift_rows(na, 700 + seed)
    B = np.empty((na * per, 128), np.float32)
    deltas = np.array([-2e-3, -5e-4, -2e-4, 2e-4, 5e-4, 2e-3])
    for i in range(na):
        r1 = 0.04 + 0.02 * rng.random()
        r2 = r1 / 0.8 * (1.0 + deltas[i % len(deltas)])
        radii = [r1, r2] + [r2 * (1.0 + 3e-4 * (k + 1)) for k in range(per - 2)]
        for k, r in enumerate(radii):
            e = rng.standard_normal(128)
            e *= r / np.linalg.norm(e)
            B[i * per + k] = (A[i].astype(np.float64) + e).astype(np.float32)
    B = B[rng.permutation(len(B))]
    for cross in (True, False):
        got = run_pair(matcher, A, B, 0.8, cross)
        exp = oracle.match_pair(A, B, 0.8, cross, threads=8)
        assert_same(got, exp, "seed %d cross=%s" % (seed, cross))
    # and the other way round (many queries, few columns each a near-tie for several of them)
    got = run_pair(matcher, B, A, 0.8, True, ids=(2, 3))
    assert_same(got, oracle.match_pair(B, A, 0.8, True, threads=8), "swapped")


def test_four_rows_in_one_column_list_with_tied_keys(matcher):
    """Found by tests/parity_stress.py: a first image of four rows, three of them within 3e-5 of each
    other in distance from a column -- one value to an 8-bit key -- so that the lane's two-entry column
    list drops one of them by index, and it was the nearest.  A small image is no reason to skip the
    completeness proof (tests/golden/near_tie_4x3.npz holds the inputs as they were drawn)."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "near_tie_4x3.npz"))
    A, B = z["A"], z["B"]
    for ratio in (1.0, 0.8):
        for cross in (True, False):
            assert_same(run_pair(matcher, A, B, ratio, cross), oracle.match_pair(A, B, ratio, cross), "A, B %g %s" % (ratio, cross))
            assert_same(run_pair(matcher, B, A, ratio, cross, ids=(2, 3)), oracle.match_pair(B, A, ratio, cross), "B, A")
    # the same situation at every small size: near-duplicates of a few base rows on both sides
    rng = np.random.default_rng(5)
    for na in range(1, 9):
        for nb in range(1, 9):
            base = synth.rootsift_rows(3, 100 * na + nb)
            A = (base[rng.integers(0, 3, na)] + rng.standard_normal((na, 128)) * 1e-5).astype(np.float32)
            B = (base[rng.integers(0, 3, nb)] + rng.standard_normal((nb, 128)) * 1e-5).astype(np.float32)
            assert_same(run_pair(matcher, A, B, 1.0, True), oracle.match_pair(A, B, 1.0, True), "%d x %d" % (na, nb))
