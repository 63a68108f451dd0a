"""Static sharding of an image-pair list over the ranks of one node.

Pairs are independent (each result depends on its two images only; the reference runs them as
separate tasks, src/feature_matcher/brute_force_matcher.cc:46-61), so the data path needs no
collective.  The list -- in the order of the reference's std::set<Pair>,
src/feature_matcher/exhaustive_pair_builder.cc:11-16 -- is partitioned by sfm_hip_partition_pairs
(include/sfm_hip.h): every rank gets a compact 2-D block of the (first, second) triangle of equal
cost, a pair costing rows(first) x rows(second), and needs only the images of its block.  The
per-pair match lists are gathered to rank 0 as fixed-width int32 tensors through torch.distributed
(RCCL on GPUs, gloo on CPU).
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
    ids = np.array(sorted(int(i) for i in image_ids), dtype=np.int64)
    a, b = np.triu_indices(len(ids), k=1)
    return np.stack([ids[a], ids[b]], axis=1).astype(np.int64).reshape(-1, 2)


def partition(pairs, image_rows, world):
    """[indices into `pairs` of rank r's shard, for r in range(world)], each in the order the rank
    works through them.  image_rows: {image id: descriptor rows} for every image the list names, or
    None when all images have the same size."""
    from .binding import partition_pairs
    pairs = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
    if image_rows is None:
        image_rows = {int(i): 1 for i in np.unique(pairs)}
    order, begin = partition_pairs(pairs, image_rows, world)
    return [order[begin[r]:begin[r + 1]].astype(np.int64) for r in range(world)]


def images_of(pairs, indices):
    """Sorted ids of the images the pairs at `indices` name: what a rank has to upload."""
    pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    return np.unique(pairs[np.asarray(indices, dtype=np.int64)]).tolist()


def gather_flat(per_pair, columns, shards, rank, world, group=None, device=None, chunk_elems=1 << 24):
    """Gather every rank's results to rank 0 as int32 tensors (no pickling).

    per_pair: int32 [n_local, k] -- k numbers per pair of this rank's shard, in shard order; the LAST
    one is the number of entries the pair owns in `columns`.  columns: list of c 4-byte arrays
    (int32 / float32) of sum(per_pair[:, -1]) entries each, the pairs' entries one after the other.
    shards: partition(...)'s result (every rank computes the same one).
    Returns on rank 0 (per_pair_all int32 [n_pairs, k], offsets int64 [n_pairs + 1], columns_all) in
    the order of the pair list, None elsewhere.  Payload travels in chunks of at most chunk_elems
    entries per column, so rank 0 never holds more than world x c x chunk_elems x 4 bytes of staging.
    """
    import torch
    import torch.distributed as dist
    per_pair = np.ascontiguousarray(per_pair, dtype=np.int32)
    if per_pair.ndim != 2 or per_pair.shape[0] != len(shards[rank]):
        raise ValueError("per_pair must be [pairs of this rank's shard, k]")
    k, c = per_pair.shape[1], len(columns)
    total = int(per_pair[:, -1].sum()) if len(per_pair) else 0
    cols = np.empty((c, total), np.int32)
    for j, col in enumerate(columns):
        col = np.ascontiguousarray(col)
        assert col.dtype.itemsize == 4 and col.size == total, "columns must be 4-byte arrays matching the counts"
        cols[j] = col.view(np.int32)
    dev = torch.device("cpu") if device is None else device
    n_pairs = sum(len(x) for x in shards)
    max_local = max(max(len(x) for x in shards), 1)

    def gather(t):
        t = t.to(dev)
        parts = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
        dist.gather(t, parts, dst=0, group=group)
        return [x.cpu().numpy() for x in parts] if rank == 0 else None

    padded = np.zeros((max_local, k), np.int32)
    padded[:len(per_pair)] = per_pair
    got = gather(torch.from_numpy(padded))
    totals = torch.tensor([total], dtype=torch.int64, device=dev)
    dist.all_reduce(totals, op=dist.ReduceOp.MAX, group=group)
    max_total = int(totals.item())
    per_rank = [np.empty((c, int(got[r][:len(shards[r]), -1].sum())), np.int32) for r in range(world)] if rank == 0 else None
    for at in range(0, max_total, chunk_elems):
        width = min(chunk_elems, max_total - at)
        piece = np.zeros((c, width), np.int32)
        mine = cols[:, at:at + width]
        piece[:, :mine.shape[1]] = mine
        parts = gather(torch.from_numpy(piece))
        if rank == 0:
            for r in range(world):
                w = max(0, min(width, per_rank[r].shape[1] - at))
                per_rank[r][:, at:at + w] = parts[r][:, :w]
    if rank != 0:
        return None
    # back into list order
    per_pair_all = np.zeros((n_pairs, k), np.int32)
    for r in range(world):
        per_pair_all[shards[r]] = got[r][:len(shards[r])]
    offsets = np.zeros(n_pairs + 1, np.int64)
    np.cumsum(per_pair_all[:, -1], out=offsets[1:])
    columns_all = np.empty((c, int(offsets[-1])), np.int32)
    for r in range(world):
        cnt = got[r][:len(shards[r]), -1].astype(np.int64)
        src = np.zeros(len(cnt) + 1, np.int64)
        np.cumsum(cnt, out=src[1:])
        idx = np.asarray(shards[r], np.int64)
        dst = offsets[idx]
        # run by run: a shard is a 2-D block of the triangle, so its pairs lie scattered over the
        # list, but neighbours in the shard are often neighbours in the list too
        starts = np.concatenate(([0], np.nonzero(np.diff(idx) != 1)[0] + 1, [len(idx)])) if len(idx) else np.zeros(1, np.int64)
        for a, b in zip(starts[:-1].tolist(), starts[1:].tolist()):
            n = int(src[b] - src[a])
            columns_all[:, dst[a]:dst[a] + n] = per_rank[r][:, src[a]:src[b]]
    return per_pair_all, offsets, [columns_all[j] for j in range(c)]


def _flatten(results, n_head):
    """List of per-pair tuples (n_head leading ints, then equally long arrays) -> (per_pair, columns)."""
    per_pair = np.array([[int(x) for x in t[:n_head]] + [len(t[n_head])] for t in results], np.int32).reshape(len(results), n_head + 1)
    n_cols = 3
    columns = []
    for j in range(n_cols):
        parts = [np.ascontiguousarray(t[n_head + j]) for t in results]
        want = np.float32 if j == 2 else np.int32
        columns.append(np.concatenate(parts).astype(want) if parts else np.zeros(0, want))
    return per_pair, columns


def match_pairs_sharded(matcher, pairs, ratio=0.8, cross_check=True, rank=0, world=1, group=None, image_rows=None,
                        device=None):
    """Match this rank's shard with `matcher` (anything with match_pairs(pairs, ratio, cross_check);
    only the images of images_of(pairs, partition(...)[rank]) need to be uploaded to it) and gather.
    Returns the full list of (lhs int32, rhs int32, dist float32) in the order of `pairs` on rank 0
    and None elsewhere.  device: torch device the collective needs its tensors on (None: CPU, gloo)."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
    shards = partition(pairs, image_rows, world)
    mine = shards[rank]
    local = matcher.match_pairs(pairs[mine], ratio, cross_check) if len(mine) else []
    if world == 1:
        out = [None] * len(pairs)
        for q, res in zip(mine.tolist(), local):
            out[q] = res
        return out
    per_pair, columns = _flatten(local, 0)
    got = gather_flat(per_pair, columns, shards, rank, world, group, device)
    if got is None:
        return None
    _, off, (lhs, rhs, dist_bits) = got
    dist = dist_bits.view(np.float32)
    return [(lhs[off[q]:off[q + 1]], rhs[off[q]:off[q + 1]], dist[off[q]:off[q + 1]]) for q in range(len(pairs))]


def cascade_match_pairs_sharded(matcher, pairs, cross_check=True, rank=0, world=1, group=None, device=None):
    """The cascade-hash matcher over ranks.  Hashing depends on the mean vector of ALL images the
    job's pair list names (cascade_hash_matcher.cc:148-165), so every rank holds every image and
    prepares with the whole list (matcher.cascade_prepare(pairs)); only the matching itself is
    shared out, in contiguous slices of the list.  matcher: anything with cascade_prepare(pairs) and
    cascade_match_pairs(pairs, cross_check).  Returns the full, ordered list of
    (n_default, lhs, rhs, dist) on rank 0 and None elsewhere."""
    pairs = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
    matcher.cascade_prepare(pairs)
    shards = [np.arange(*shard_bounds(len(pairs), r, world), dtype=np.int64) for r in range(world)]
    mine = shards[rank]
    local = matcher.cascade_match_pairs(pairs[mine], cross_check) if len(mine) else []
    if world == 1:
        return local
    per_pair, columns = _flatten(local, 1)
    got = gather_flat(per_pair, columns, shards, rank, world, group, device)
    if got is None:
        return None
    head, off, (lhs, rhs, dist_bits) = got
    dist = dist_bits.view(np.float32)
    return [(int(head[q, 0]), lhs[off[q]:off[q + 1]], rhs[off[q]:off[q + 1]], dist[off[q]:off[q + 1]])
            for q in range(len(pairs))]


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
