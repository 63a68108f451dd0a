"""BASELINE.json configs[3] (500 images x 8192, 124,750 pairs over 2/4/8 GPUs) and configs[4]
(1,000 images x 16384, 499,500 pairs over 8 GPUs) as far as one GPU can take them: the real pair
lists, the library's real 8-way (and 2-, 4-way) partition, and from EVERY shard a sample of its pairs
matched on this GPU the way a rank would (only the shard's own images uploaded), checked against the
planted truth.  What eight GPUs add to this is eight processes doing it at once."""
import numpy as np
import pytest

from sfm_amd import synth
from sfm_amd.distributed import exhaustive_pairs, images_of, partition

pytestmark = pytest.mark.gpu

SEED = 4242


def _run_samples(matcher, n_images, n_features, world, per_shard):
    planted = n_features // 2
    rows = {i: n_features for i in range(n_images)}
    pairs = exhaustive_pairs(range(n_images))
    assert len(pairs) == n_images * (n_images - 1) // 2
    shards = partition(pairs, rows, world)
    assert sorted(np.concatenate(shards).tolist()) == list(range(len(pairs)))
    sizes = [len(s) for s in shards]
    assert max(sizes) - min(sizes) <= 1                        # equal images: equal cost means equal counts
    need = [len(images_of(pairs, s)) for s in shards]
    if world >= 4:
        assert max(need) <= 2 * n_images / np.sqrt(world)
    pool = synth.scene_pool(planted, SEED)
    for r, shard in enumerate(shards):
        take = np.concatenate([shard[:per_shard], shard[-per_shard:]])
        sample = pairs[take]
        ids = np.unique(sample).tolist()
        assert set(ids) <= set(images_of(pairs, shard))
        for i in ids:                                          # a rank uploads only its own block's images
            matcher.upload(5000 + i, synth.scene_image(i, n_features, pool, SEED))
        try:
            got = matcher.match_pairs(sample + 5000, 0.8, True)
        finally:
            for i in ids:
                matcher.drop(5000 + i)
        for (a, b), (l, r_idx, d) in zip(sample.tolist(), got):
            truth = np.full(n_features, -1, np.int64)
            truth[synth.scene_positions(a, n_features, planted, SEED)] = synth.scene_positions(b, n_features, planted, SEED)
            assert len(l) == planted and np.array_equal(truth[l], r_idx), (r, a, b)
            assert np.all(np.diff(l) > 0) and np.all(np.isfinite(d))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_config3_every_shard_sampled(matcher, world):
    _run_samples(matcher, 500, 8192, world, per_shard=max(1, 8 // world))


def test_config4_every_shard_sampled(matcher):
    _run_samples(matcher, 1000, 16384, 8, per_shard=1)
