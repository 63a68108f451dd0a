"""BASELINE.json configs[2] on the GPU: exhaustive matching of 100 synthetic images x 8192 features
(4,950 pairs) through 64-pair calls of the C ABI, the way bench.py's whole_job leg and the C++
matcher class run it.  Every pair is checked against the planted truth and against properties any
correct result has; a sample of 20 pairs is compared with the CPU oracle bit for bit."""
import numpy as np
import pytest

from sfm_amd import synth
from sfm_amd.distributed import exhaustive_pairs

pytestmark = pytest.mark.gpu

N_IMAGES, N_FEATURES, PLANTED, SEED = 100, 8192, 4096, 4242


def test_config2_all_pairs(matcher):
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    pool = synth.scene_pool(PLANTED, SEED)
    with ThreadPoolExecutor(max_workers=8) as ex:
        images = list(ex.map(lambda i: synth.scene_image(i, N_FEATURES, pool, SEED), range(N_IMAGES)))
    pos = [synth.scene_positions(i, N_FEATURES, PLANTED, SEED) for i in range(N_IMAGES)]
    ids = [3000 + i for i in range(N_IMAGES)]
    for i, im in zip(ids, images):
        matcher.upload(i, im)
    pairs = exhaustive_pairs(range(N_IMAGES))
    assert len(pairs) == 4950
    sample = set(np.linspace(0, len(pairs) - 1, 20).astype(np.int64).tolist())
    kept = {}
    try:
        prev = None

        def check(lo, out):
            cnt, lhs, rhs, dist = out
            for k in range(len(cnt)):
                q = lo + k
                a, b = pairs[q].tolist()
                c = int(cnt[k])
                l, r, d = lhs[k, :c], rhs[k, :c], dist[k, :c]
                # properties: ascending lhs, one match per query, indices in range, and with cross-check a
                # column is matched at most once -- except by row 0, which the reference's map_[j] default
                # also lets through (brute_force_matcher.cc:75-78)
                assert np.all(np.diff(l) > 0) and c <= N_FEATURES
                assert l.min() >= 0 and l.max() < N_FEATURES and r.min() >= 0 and r.max() < N_FEATURES
                rr = r[l != 0]
                assert len(np.unique(rr)) == len(rr)
                assert np.all(d >= 0) and np.all(np.isfinite(d))
                # planted truth: every shared scene row is found, and nothing else passes the ratio test
                truth = np.full(N_FEATURES, -1, np.int64)
                truth[pos[a]] = pos[b]
                assert c == PLANTED and np.array_equal(truth[l], r), (a, b)
                if q in sample:
                    kept[q] = (l.copy(), r.copy(), d.copy())

        for lo in range(0, len(pairs), 64):
            p = pairs[lo:lo + 64] + 3000
            matcher.sync()
            out = matcher.match_pairs_async(p, 0.8, True, stride=N_FEATURES)
            if prev is not None:
                check(*prev)
            prev = (lo, tuple(x[:len(p)] for x in out))
        matcher.sync()
        check(*prev)
    finally:
        matcher.sync()
        for i in ids:
            matcher.drop(i)
    assert len(kept) == 20
    for q, (l, r, d) in kept.items():
        a, b = pairs[q].tolist()
        el, er, ed = oracle.match_pair(images[a], images[b], 0.8, True, threads=16)
        assert l.tolist() == el.tolist() and r.tolist() == er.tolist()
        assert d.view(np.uint32).tolist() == ed.view(np.uint32).tolist()
