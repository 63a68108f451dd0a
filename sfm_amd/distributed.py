"""Static sharding of an image-pair list over the ranks of one node.

Pairs are independent (each result depends on its two images only), so the data path
needs no collective: every rank holds all descriptors, matches a contiguous slice of the
lexicographically ordered pair list (the order of the reference's std::set<Pair>,
src/feature_matcher/exhaustive_pair_builder.cc:11-16) and the per-pair match lists are
gathered to rank 0.  The gather goes through torch.distributed (RCCL on GPUs, gloo on CPU).
"""
import numpy as np


def shard_bounds(n_items, rank, world):
    """[lo, hi) of rank's contiguous share; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world: %r/%r" % (rank, world))
    base, extra = divmod(int(n_items), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def exhaustive_pairs(image_ids):
    """All (min, max) pairs over the ids, lexicographic: what ExhaustivePairBuilder::BuildPair
    inserts into its std::set (src/feature_matcher/exhaustive_pair_builder.cc:3-19)."""
    ids = sorted(int(i) for i in image_ids)
    return np.array([(a, b) for k, a in enumerate(ids) for b in ids[k + 1:]], dtype=np.int64).reshape(-1, 2)


def match_pairs_sharded(matcher, pairs, ratio=0.8, cross_check=True, rank=0, world=1, group=None):
    """Match this rank's slice with `matcher` (anything with match_pairs(pairs, ratio, cross_check))
    and gather.  Returns the full, ordered list of (lhs, rhs, dist) on rank 0 and None elsewhere."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
    lo, hi = shard_bounds(len(pairs), rank, world)
    local = matcher.match_pairs(pairs[lo:hi], ratio, cross_check) if hi > lo else []
    if world == 1:
        return local
    import torch.distributed as dist
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((lo, local), gathered, dst=0, group=group)
    if rank != 0:
        return None
    out = []
    for lo_r, res in sorted(gathered, key=lambda t: t[0]):
        assert lo_r == len(out), "shards do not tile the pair list"
        out.extend(res)
    return out
