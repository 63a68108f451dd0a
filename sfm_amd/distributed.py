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


def cascade_match_pairs_sharded(matcher, pairs, cross_check=True, rank=0, world=1, group=None):
    """The cascade-hash matcher over ranks.  Hashing depends on the mean vector of ALL images the
    job's pair list names (cascade_hash_matcher.cc:148-165), so every rank prepares with the whole
    list (matcher.cascade_prepare(pairs): every rank holds every image, as above) and only the
    matching itself is sliced.  matcher: anything with cascade_prepare(pairs) and
    cascade_match_pairs(pairs, cross_check).  Returns the full, ordered list of
    (n_default, lhs, rhs, dist) on rank 0 and None elsewhere."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
    matcher.cascade_prepare(pairs)
    lo, hi = shard_bounds(len(pairs), rank, world)
    local = matcher.cascade_match_pairs(pairs[lo:hi], cross_check) if hi > lo else []
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


def replicate_images(matcher, images, rank=0, world=1, group=None, device=None, method="collective"):
    """Give every rank's matcher every image.

    images: {image_id: float32 array [n, 128]} -- with method "host" every rank passes the whole
    set and uploads it from its own host memory; with method "collective" rank r passes (at least)
    the images of its share, shard_bounds(len(ids), r, world) of the sorted ids, uploads only
    those to its device, and an all-gather moves every share to every other device (one
    collective, equal-sized shares: on xGMI each link then carries only its peer's share), from
    where the matcher takes them without another trip through the host.
    `device`: torch device of this rank's GPU; None keeps the gathered set on the host (gloo),
    which exercises the same bookkeeping on machines without GPUs.
    Returns the number of descriptor rows registered.
    """
    import torch
    import torch.distributed as dist
    if method not in ("host", "collective"):
        raise ValueError("method must be 'host' or 'collective'")
    if method == "host" or not (dist.is_available() and dist.is_initialized()):
        ids = sorted(int(i) for i in images)
        for i in ids:
            matcher.upload(i, images[i])
        return sum(len(images[i]) for i in ids)
    # every rank must know every image's owner and size: a small object all-gather of (id, n)
    mine = sorted(int(i) for i in images)
    sizes = [None] * world
    dist.all_gather_object(sizes, [(i, int(len(images[i]))) for i in mine], group=group)
    all_ids = sorted({i for part in sizes for i, _ in part})
    n_of = {i: n for part in sizes for i, n in part}
    owner_rows = []
    layout = {}                       # id -> (owner, first row inside the owner's share)
    for r in range(world):
        lo, hi = shard_bounds(len(all_ids), r, world)
        at = 0
        for i in all_ids[lo:hi]:
            layout[i] = (r, at)
            at += n_of[i]
        owner_rows.append(at)
    share_rows = max(max(owner_rows), 1)
    lo, hi = shard_bounds(len(all_ids), rank, world)
    share = torch.zeros((share_rows, 128), dtype=torch.float32)
    for i in all_ids[lo:hi]:
        if i not in images:
            raise KeyError("rank %d owns image %d but was not given it" % (rank, i))
        at = layout[i][1]
        share[at:at + n_of[i]] = torch.from_numpy(np.ascontiguousarray(images[i], dtype=np.float32))
    if device is not None:
        share = share.to(device, non_blocking=False)
    everything = torch.empty((world * share_rows, 128), dtype=torch.float32, device=share.device)
    dist.all_gather_into_tensor(everything, share, group=group)
    if device is not None:
        torch.cuda.synchronize(device)
    for i in all_ids:
        r, at = layout[i]
        rows = everything[r * share_rows + at: r * share_rows + at + n_of[i]]
        if device is not None:
            matcher.upload_device(i, rows.data_ptr() if n_of[i] else 0, n_of[i])
        else:
            matcher.upload(i, rows.numpy())
    return sum(n_of.values())
