#!/usr/bin/env python3
"""Benchmark of the brute-force matching hot path on MI355X.

A step = one pass of the hot path over one image pair of the headline
configuration (8192 x 8192 descriptors of 128 floats, ratio 0.8, cross-check):
distance/top-k kernel, exact re-rank, cross-check, compaction, copy of the
compact match list to pinned host memory.  Descriptors are resident in HBM
before the timed region starts.

With --gpus N > 1 the ranks run the SHARDED JOB of BASELINE.json configs[3]
(exhaustive matching of 500 synthetic images x 8192 features): the pair list of
ExhaustivePairBuilder is partitioned by the library (sfm_hip_partition_pairs:
2-D blocks of the pair triangle of equal cost), every rank uploads only its
block's images, and a step is one 64-pair call of sfm_hip_match_pairs on the
next pairs of the rank's shard -- the way the C++ matcher class drives the
library.  Pairs are independent, so there is no collective on the data path and
per-GPU work per step is fixed ("weak" scaling); value = pairs all ranks matched
/ the slowest rank's time.  The ranks come from a launcher (torch.distributed.run
sets WORLD_SIZE) or, when there is none, from this program itself: `python
bench.py --gpus N` starts N rank processes and relays rank 0's line.  The line
carries the same step measured on rank 0's GPU alone (n1_same_step_pairs_per_s)
and the scaling efficiency against it.  After the timed region the same ranks run the whole
job once, uploads, matching and the gather of all match lists to rank 0 timed
separately ("whole_job"), and time the two ways of replicating a descriptor set
("replication").

Prints ONE JSON line on rank 0.
"""
import argparse
import json

import numpy as np
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FEATURES = 8192
PLANTED = 4096
PEAK_FP32_MFMA_TFLOPS = 157.3   # 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_FP16_MFMA_TFLOPS = 2516.6  # 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz (dense)
PEAK_HBM_GBPS = 8000.0          # HBM3E
# fp64 vector instructions: a SIMD issues one 64-lane instruction in four cycles (16 lanes per clock);
# 256 CU x 4 SIMD x 16 lanes x 2.4 GHz lane-operations per second.  (The 78.6 TFLOP/s fp64 vector
# figure is twice this: it counts a fused multiply-add as two, and nothing here may be fused.)
PEAK_FP64_VALU_TOPS = 39.3
PEAK_VALU_TOPS = 39.3           # 32-bit vector instructions issue at the same rate: 16 lanes per clock and SIMD
PEAK_GATHER_GBPS = 7400.0       # random whole rows from a table beyond the L2s (MI355X_MICROARCH.md, "Indexed rows")
PEAK_PCIE_GBPS = 63.0           # PCIe Gen5 x16
WALK_OPS_PER_CANDIDATE = 5      # 2 x v_xor, 2 x v_bcnt (the second accumulates), 1 x v_cmp
# epipolar_line_error (src/solver/fundamental_solver.hpp:59-72) as the reference writes it: five
# rows of a 3x3 product (2 mul + 2 add each), x'.l (2 mul + 2 add), two sums of two squares (3 each),
# two divisions, their sum, num^2, the product, the comparison with 3.84
GEO_OPS_PER_MATCH_ROUND = 5 * 4 + 4 + 3 + 3 + 2 + 1 + 1 + 1 + 1
# the eight-point fit of one round (fundamental_solver.cc:13-53) in this build's convention: the
# 8 x 9 design matrix (32 products), the upper triangle of A^T A (45 entries x 16), twelve cyclic
# Jacobi sweeps of the 9 x 9 problem (36 rotations: 19 operations for the angle, 16 element pairs x
# 6), eight sweeps of the 3 x 3 problem of the rank-two step (3 rotations: 19 + 4 pairs x 6), the
# final 3 x 3 update.  Divisions and square roots count as one operation each; a rotation the
# solver skips because its entry has vanished still counts (the algorithmic figure is the
# convention's, not what a converged wavefront executes).
GEO_OPS_PER_FIT = 32 + 45 * 16 + 12 * 36 * (19 + 16 * 6) + 8 * 3 * (19 + 4 * 6) + 30 + 33


def host_cores():
    """(threads this process may use, CPUs the machine has).  The first is what a CPU leg runs on:
    the scheduler affinity of the process, capped by OMP_NUM_THREADS where the box sets a share."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except AttributeError:
        allowed = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and int(share) > 0:
        allowed = min(allowed, int(share))
    return max(1, allowed), os.cpu_count() or 1


def cpu_baseline(A, B, expected):
    """Time the CPU oracle (a port of the reference matcher) on this host.

    Sample: one full 8192x8192 pair with cross-check on two threads, which is
    how the reference runs a pair (its ThreadPool has two workers, one per
    direction); then the same pair on every core this process may use.
    """
    from oracle import oracle
    cores, machine = host_cores()
    t0 = time.perf_counter()
    l, r, d = oracle.match_pair(A, B, 0.8, True, threads=2)
    t2 = time.perf_counter() - t0
    same = (l.tolist() == expected[0].tolist() and r.tolist() == expected[1].tolist())
    t0 = time.perf_counter()
    oracle.match_pair(A, B, 0.8, True, threads=cores)
    tall = time.perf_counter() - t0
    return {"value": 1.0 / t2, "unit": "pairs/s", "cores": 2, "kind": "port",
            "sample": "one full 8192x8192 pair, ratio 0.8 + cross-check, 2 threads (the reference's pool size)",
            "seconds_per_pair": t2, "matches_equal_gpu": bool(same),
            "all_cores": {"value": 1.0 / tall, "cores": cores, "seconds_per_pair": tall, "cpus_of_the_machine": machine}}


def cascade_cpu_baseline(images, pairs, got):
    """The CPU oracle of the cascade-hash matcher (single-threaded, as the reference runs it) on a
    bounded sample: the first four images of the set, their six pairs."""
    from oracle import cascade
    whole, hash_s, match_s = cascade.match_set(images, pairs, True, threads=1)
    same = all((nd, l.tolist(), r.tolist()) == (whole[p][0], whole[p][1].tolist(), whole[p][2].tolist())
               for p, (nd, l, r, _) in zip(pairs, got))
    return {"value": len(pairs) / match_s, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": "%d images x %d features, %d pairs, cross-check; hashing the images timed apart"
                      % (len(images), N_FEATURES, len(pairs)),
            "seconds_per_pair": match_s / len(pairs), "hash_seconds_per_image": hash_s / len(images),
            "matches_equal_gpu": bool(same)}


def cascade_leg(m, with_cpu, n_images=32, reps=40):
    """The cascade-hash matcher (the one the reference's driver runs) on the same GPU: all pairs of
    a synthetic set with repeated structure, descriptors and hashes resident.  value counts
    back-to-back batches; the roofline figure comes from batches run one at a time between HIP
    events (bytes: the library's count of what a pair has to read at the least)."""
    from sfm_amd import synth
    imgs = synth.make_repetitive_set(n_images, N_FEATURES, 0.5, 2, seed=17, sigma=0.002)
    ids = list(range(100, 100 + n_images))
    for i, d in zip(ids, imgs):
        m.upload(i, d)
    pairs = [(a, b) for k, a in enumerate(ids) for b in ids[k + 1:]]
    m.cascade_prepare(pairs)
    t0 = time.perf_counter()
    m.cascade_prepare(pairs)
    prepare_s = time.perf_counter() - t0
    for _ in range(20):   # ~12 ms: the first batches after a change of workload run slower
        out = m.cascade_match_pairs_async(pairs, True, stride=N_FEATURES)
    m.sync()
    n_matches, n_defaults = int(out[1].sum()), int(out[0].sum())
    t0 = time.perf_counter()
    for _ in range(reps):
        m.cascade_match_pairs_async(pairs, True, stride=N_FEATURES)
    m.sync()
    value = reps * len(pairs) / (time.perf_counter() - t0)
    m.set_profiling(True)
    m.cascade_reset_stats()
    for _ in range(reps):
        m.cascade_match_pairs_async(pairs, True, stride=N_FEATURES)
        m.sync()
    st = m.cascade_stats()
    gbps = st["match_bytes"] / (st["match_ms"] * 1e-3) / 1e9
    # the same once more with an event between every two kernels and the kernels counting their work
    m.set_profiling(2)
    m.cascade_reset_stats()
    for _ in range(reps):
        m.cascade_match_pairs_async(pairs, True, stride=N_FEATURES)
        m.sync()
    sk = m.cascade_stats()
    m.set_profiling(False)
    nb = max(1, sk["match_launches"])

    def kernel_roof(kernel, bound, achieved, peak, unit, ms, **more):
        r = {"kernel": kernel, "bound": bound, "achieved": achieved, "peak": peak, "unit": unit,
             "frac": achieved / peak, "ms_per_batch": ms}
        r.update(more)
        return r

    # One roofline per kernel of a batch, each against the limit that binds it (an event between every
    # two kernels of the batches above).  Walk: instruction issue -- per candidate two XORs, two
    # population counts (the second adds the first), one comparison, on 64-bit halves of the hash
    # codes staged in LDS.  Nearest: random 512-byte rows gathered from the images' descriptors (the
    # set's 128 MiB; MI355X_MICROARCH.md "Indexed rows", 151 MB table: 7.4-7.9 TB/s).  Compaction: its
    # stores go straight into the caller's pinned host memory, over PCIe (Gen5 x16, 63 GB/s).
    walk_ms, near_ms, comp_ms = sk["walk_ms"] / nb, sk["nearest_ms"] / nb, sk["compact_ms"] / nb
    cand, close, exact = sk["walk_candidates"] / nb, sk["close_pairs"] / nb, sk["exact_distances"] / nb
    out_bytes = n_matches * 12.0 + len(pairs) * 8.0
    per_kernel = [
        kernel_roof("k_ch_walk_count", "vector instruction issue", cand * WALK_OPS_PER_CANDIDATE / (walk_ms * 1e-3) / 1e12,
                    PEAK_VALU_TOPS, "Tops/s (32-bit lane-operations)", walk_ms, candidates_per_batch=cand,
                    ops_per_candidate=WALK_OPS_PER_CANDIDATE),
        kernel_roof("k_ch_nearest", "random gather of 512-byte rows", exact * 1024.0 / (near_ms * 1e-3) / 1e9,
                    PEAK_GATHER_GBPS, "GB/s", near_ms, close_pairs_per_batch=close, exact_distances_per_batch=exact,
                    bytes_per_exact_distance=1024),
        kernel_roof("k_ch_compact", "pcie", out_bytes / (comp_ms * 1e-3) / 1e9, PEAK_PCIE_GBPS, "GB/s", comp_ms,
                    bytes_to_host_per_batch=out_bytes),
    ]
    # HBM-side bytes of one such batch from a separate hardware-counter run of this workload
    # (tools/collect_cascade_profiles.sh), kept beside the profiles
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "cascade_traffic.json")
    if os.path.exists(tpath) and n_images == 32:
        with open(tpath) as fh:
            traffic = json.load(fh).get("cascade_batch", {}).get("traffic_bytes_per_batch")
    leg = {"matcher": "cascade hash", "value": value, "unit": "pairs/s",
           "workload": "%d synthetic images x %d features with repeated structure, all %d pairs, cross-check"
                       % (n_images, N_FEATURES, len(pairs)),
           "matches": n_matches, "default_records": n_defaults,
           "prepare_seconds_all_images": prepare_s, "list_overflows": st["list_overflows"],
           "roofline": {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                        "traffic": traffic, "kernel": "one batch of %d pairs: k_ch_walk_count .. k_ch_compact" % len(pairs),
                        "bytes_per_batch": st["match_bytes"] / st["match_launches"],
                        "traffic_over_algorithmic": (traffic / (st["match_bytes"] / st["match_launches"])) if traffic else None,
                        "avg_batch_ms": st["match_ms"] / st["match_launches"],
                        "bytes_per_pair": st["match_bytes"] / (reps * len(pairs))},
           "roofline_per_kernel": per_kernel, "fallback_kernels_ms_per_batch": sk["fallback_ms"] / nb,
           "batch_ms_with_per_kernel_events": sk["match_ms"] / nb}
    if with_cpu:
        sub = {i: d for i, d in zip(ids[:4], imgs[:4])}
        sp = [(a, b) for k, a in enumerate(ids[:4]) for b in ids[k + 1:4]]
        m.cascade_prepare(sp)       # the mean vector is taken over the images that take part
        leg["cpu_baseline"] = cascade_cpu_baseline(sub, sp, m.cascade_match_pairs(sp, True))
    for i in ids:
        m.drop(i)
    return leg


def geometry_leg(m, with_cpu, n_pairs=496, n_matches=1000, reps=3, cpu_pairs=(200, 50)):
    """Geometric verification of matches (the step that consumes the matcher's output:
    GeometryFilter, src/sfm_main.cc:259-282; Ransac<...>::Inference, src/ransac.hpp:94-150) on the
    same GPU: 496 pairs x 1,000 matches per C-ABI call, synthetic two-view geometry with +-0.3 px
    noise, at 70 % planted inliers (the stopping rule ends a pair after a few hundred rounds) and at
    40 % (every pair runs all 4,096 rounds).  value = pairs per second of the whole blocking call,
    upload of the positions and download of the flags included.  The rooflines are per kernel,
    against the rate at which the SIMDs can issue fp64 vector instructions; the work is counted by
    the library (rounds fitted, (match, round) pairs measured) and priced at the operation counts
    of the reference's own formulas."""
    from sfm_amd import synth
    leg = {"step": "geometric verification (RANSAC over the eight-point fundamental-matrix fit)", "cases": []}
    for inliers, n_cpu in zip((0.7, 0.4), cpu_pairs):
        sets = [synth.make_two_view_matches(n_matches, inliers, seed=500 + k, noise=0.3)[:2] for k in range(n_pairs)]
        offsets = np.zeros(n_pairs + 1, np.int64)
        offsets[1:] = np.cumsum([len(l) for l, _ in sets])
        lhs = np.concatenate([l for l, _ in sets])
        rhs = np.concatenate([r for _, r in sets])
        F, flags, counts, rounds = m.geometry_verify_flat(offsets, lhs, rhs, seed=1)   # warm-up, and the results
        t0 = time.perf_counter()
        for _ in range(reps):
            m.geometry_verify_flat(offsets, lhs, rhs, seed=1)
        dt = (time.perf_counter() - t0) / reps
        m.set_profiling(True)
        m.geometry_reset_stats()
        m.geometry_verify_flat(offsets, lhs, rhs, seed=1)
        st = m.geometry_stats()
        m.set_profiling(False)
        fit_ops, count_ops = st["rounds"] * GEO_OPS_PER_FIT, st["match_rounds"] * GEO_OPS_PER_MATCH_ROUND

        def roof(kernel, ops, ms):
            a = ops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            return {"bound": "fp64 vector instruction issue", "kernel": kernel, "achieved": a, "peak": PEAK_FP64_VALU_TOPS,
                    "unit": "Tops/s (fp64 lane-operations, unfused)", "frac": a / PEAK_FP64_VALU_TOPS, "traffic": None,
                    "ms_per_call": ms, "ops_per_call": ops}

        case = {"workload": "%d pairs x %d matches, %.0f %% planted inliers, one call" % (n_pairs, n_matches, 100 * inliers),
                "value": n_pairs / dt, "unit": "pairs/s", "seconds_per_call": dt,
                "accepted_pairs": int((counts > 30).sum()), "ransac_rounds_mean": float(rounds.mean()),
                "ransac_rounds_max": int(rounds.max()), "inliers_mean": float(counts.mean()),
                "rounds_worked_on": st["rounds"], "timed_launches": st["timed_launches"],
                "ops_per_fit": GEO_OPS_PER_FIT, "ops_per_match_and_round": GEO_OPS_PER_MATCH_ROUND,
                "roofline": [roof("k_geo_fit", fit_ops, st["fit_ms"]), roof("k_geo_count", count_ops, st["count_ms"])]}
        if with_cpu and n_cpu > 0:
            from oracle import geometry   # the checker, timed as the CPU baseline on a bounded sample
            t0 = time.perf_counter()
            same = True
            for k in range(n_cpu):
                ok, oF, oflags, r = geometry.compute_fundamental(sets[k][0], sets[k][1], seed=1, pair=k)
                mine = flags[offsets[k]:offsets[k + 1]]
                same = same and ok == (int(counts[k]) > 30) and r == int(rounds[k]) and oflags.tolist() == mine.tolist() \
                    and oF.reshape(-1).view(np.uint64).tolist() == F[k].view(np.uint64).tolist()
            dc = time.perf_counter() - t0
            case["cpu_baseline"] = {"value": n_cpu / dc, "unit": "pairs/s", "cores": 1, "kind": "port",
                                    "sample": "the first %d pairs of the same call" % n_cpu, "equal_to_gpu": bool(same)}
        leg["cases"].append(case)
    return leg


def replication_check(m, rank, world, device, barrier, n_images=64):
    """Outside the timed region, N > 1 only: how long it takes to give every GPU a descriptor set
    of n_images x 8192 rows (i) by uploading everything from each rank's host memory and (ii) by
    uploading each rank's share and all-gathering over RCCL/xGMI.  Reported, never fatal."""
    import torch
    import torch.distributed as dist
    from sfm_amd.distributed import replicate_images, shard_bounds
    try:
        rng = np.random.default_rng(7)
        ids = list(range(1000, 1000 + n_images))
        images = {i: rng.random((N_FEATURES, 128), dtype=np.float32) for i in ids}
        lo, hi = shard_bounds(len(ids), rank, world)
        mine = {i: images[i] for i in ids[lo:hi]}
        out = {"images": n_images, "rows_per_image": N_FEATURES, "bytes": n_images * N_FEATURES * 512}
        for name, method, src in (("host_uploads_s", "host", images), ("collective_s", "collective", mine)):
            barrier()
            t0 = time.perf_counter()
            replicate_images(m, src, rank, world, device=device if device.type == "cuda" else None, method=method)
            barrier()
            tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            out[name] = float(tt.item())
            for i in ids:
                m.drop(i)
        return out
    except Exception as e:   # a report, not a gate
        return {"error": "%s: %s" % (type(e).__name__, e)}


def with_deadline(fn, seconds, what, device=None):
    """An optional multi-rank leg, in a thread of its own: a leg that has not returned after `seconds`
    (a rank stuck in a collective that another rank never entered) is given up, so that the line still
    goes out.  Returns (result, gave_up); after gave_up the caller starts nothing more on the GPU --
    the thread may still be inside the library -- and ends the process with os._exit once the line is
    printed."""
    import threading
    box = {}

    def run():
        try:
            if device is not None and device.type == "cuda":
                import torch
                torch.cuda.set_device(device)      # the current device is a per-thread setting
            box["result"] = fn()
        except Exception as e:   # a report, not a gate
            box["result"] = {"error": "%s: %s" % (type(e).__name__, e)}

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        return {"error": "%s did not finish within %d s on rank %s: given up" % (what, seconds, os.environ.get("RANK", "0"))}, True
    return box.get("result"), False


SET_SEED = 4242
PAIRS_PER_CALL = 64


class SetJob:
    """An exhaustive matching job over a synthetic scene set (every two images share PLANTED true
    correspondences), cut for `world` ranks by the library's partition.  Images are generated per
    rank, only those its shard names."""

    def __init__(self, n_images, rank, world, replication="host"):
        """replication "host": this rank generates and uploads the images its shard names;
        "collective": it generates its 1/world share of ALL images, uploads that, and one RCCL
        all-gather over xGMI hands every share to every GPU (BASELINE.json configs[4])."""
        from sfm_amd import synth
        from sfm_amd.distributed import exhaustive_pairs, images_of, partition, shard_bounds
        self.n_images, self.rank, self.world, self.replication = n_images, rank, world, replication
        self.ids = list(range(n_images))
        self.rows = {i: N_FEATURES for i in self.ids}
        self.pairs = exhaustive_pairs(self.ids)
        self.shards = partition(self.pairs, self.rows, world)
        self.mine = self.shards[rank]
        if replication == "collective":
            lo, hi = shard_bounds(n_images, rank, world)
            self.need = self.ids[lo:hi]
            self.images_per_rank = [n_images] * world
        else:
            self.need = images_of(self.pairs, self.mine) if len(self.mine) else []
            self.images_per_rank = [len(images_of(self.pairs, s)) if len(s) else 0 for s in self.shards]
        pool = synth.scene_pool(PLANTED, SET_SEED)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(8, host_cores()[0])) as ex:
            made = list(ex.map(lambda i: synth.scene_image(i, N_FEATURES, pool, SET_SEED), self.need))
        self.images = dict(zip(self.need, made))

    def upload(self, m, coll_device=None):
        if self.replication == "collective":
            from sfm_amd.distributed import replicate_images
            replicate_images(m, self.images, self.rank, self.world,
                             device=coll_device if coll_device is not None and coll_device.type == "cuda" else None,
                             method="collective")
            return
        for i in self.need:
            m.upload(i, self.images[i])

    def batch(self, k):
        """The k-th 64-pair call of this rank's shard (wrapping round at its end)."""
        n = len(self.mine)
        idx = self.mine[(np.arange(PAIRS_PER_CALL) + k * PAIRS_PER_CALL) % n]
        return self.pairs[idx]

    def run(self, m, upload_as_needed=False):
        """The whole shard through 64-pair calls, two sets of pinned outputs in turn: the GPU matches
        call b, with call b + 1 queued behind it, while the host copies the compact lists of call
        b - 1 out.  upload_as_needed: nothing is resident at the start; an image is uploaded right
        before the first call that names it, while the GPU works on the calls queued before (what
        the C++ matcher class does; the shard follows a space-filling curve over the pair triangle,
        so a call brings few new images).  Returns (counts, lhs, rhs); self.upload_seconds is the
        time spent inside those upload calls."""
        counts = np.zeros(len(self.mine), np.int32)
        lhs_parts, rhs_parts = [], []
        resident = set()
        self.upload_seconds = 0.0

        def collect(lo, out):
            cnt, lhs, rhs, _ = out
            counts[lo:lo + len(cnt)] = cnt
            for k in range(len(cnt)):
                lhs_parts.append(lhs[k, :cnt[k]].copy())
                rhs_parts.append(rhs[k, :cnt[k]].copy())

        prev = None
        for lo in range(0, len(self.mine), PAIRS_PER_CALL):
            p = self.pairs[self.mine[lo:lo + PAIRS_PER_CALL]]
            if upload_as_needed:
                t_up = time.perf_counter()
                for i in np.unique(p).tolist():
                    if i not in resident:
                        m.upload_async(i, self.images[i])   # the call below waits for them; they overlap each other
                        resident.add(i)
                self.upload_seconds += time.perf_counter() - t_up
            # queued behind call b - 1; no distances, as the C++ matcher class calls it (a Match record has none)
            out = m.match_pairs_async(p, 0.8, True, stride=N_FEATURES, want_dist=False)
            if prev is not None:
                m.wait_outputs(prev[1][0])    # call b - 1 alone is complete; call b runs on
                collect(*prev)
            prev = (lo, out)
        m.sync()
        if prev is not None:
            collect(*prev)
        cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0, np.int32)
        return counts, cat(lhs_parts), cat(rhs_parts)

    def check(self, counts, offsets, lhs, rhs, sample=48):
        """Planted truth on a sample of pairs spread over the list; totals over all of it."""
        from sfm_amd import synth
        n = len(self.pairs)
        pos = {}
        found = expected = 0
        for q in np.linspace(0, n - 1, min(sample, n)).astype(np.int64).tolist():
            a, b = self.pairs[q].tolist()
            for i in (a, b):
                if i not in pos:
                    pos[i] = synth.scene_positions(i, N_FEATURES, PLANTED, SET_SEED)
            truth = set(zip(pos[a].tolist(), pos[b].tolist()))
            got = set(zip(lhs[offsets[q]:offsets[q + 1]].tolist(), rhs[offsets[q]:offsets[q + 1]].tolist()))
            found += len(got & truth)
            expected += len(truth)
        return {"matches": int(counts.sum()), "matches_per_pair_min": int(counts.min()) if n else 0,
                "matches_per_pair_max": int(counts.max()) if n else 0,
                "sample_pairs": min(sample, n), "planted_in_sample": expected, "planted_found_in_sample": found}


def whole_job_leg(m, job, barrier, max_over_ranks, device, engine):
    """The whole sharded job once, outside the headline's timed region: uploads of each rank's
    images, matching of its shard, gather of every pair's (lhs, rhs) lists to rank 0 -- each timed
    between barriers, the slowest rank counting."""
    from sfm_amd.distributed import gather_flat
    m.drop_all()
    barrier()
    # Host uploads ride along with the matching (an image goes up right before the first call that
    # names it); a collective replication is a phase of its own, between barriers.
    overlapped = job.replication != "collective"
    t_up = 0.0
    if not overlapped:
        t0 = time.perf_counter()
        job.upload(m, device)
        m.sync()
        barrier()
        t_up = max_over_ranks(time.perf_counter() - t0)
    m.set_profiling(True)
    m.reset_stats()
    t0 = time.perf_counter()
    counts, lhs, rhs = job.run(m, upload_as_needed=overlapped)
    t_local = time.perf_counter() - t0
    barrier()
    t_match = max_over_ranks(time.perf_counter() - t0)
    t_up_inside = max_over_ranks(job.upload_seconds) if overlapped else 0.0
    st = m.stats()
    m.set_profiling(False)
    t0 = time.perf_counter()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        got = gather_flat(counts.reshape(-1, 1), [lhs, rhs], job.shards, job.rank, job.world, device=device)
    else:
        off = np.zeros(len(counts) + 1, np.int64)
        np.cumsum(counts, out=off[1:])
        assert np.array_equal(job.mine, np.arange(len(job.mine))), "one shard is the list itself"
        got = (counts.reshape(-1, 1), off, [lhs, rhs])
    barrier()
    t_gather = max_over_ranks(time.perf_counter() - t0)
    if job.rank != 0:
        return None
    head, off, (g_lhs, g_rhs) = got
    n = len(job.pairs)
    flop = 2.0 * N_FEATURES * N_FEATURES * 128
    leg = {"workload": "exhaustive matching, %d synthetic images x %d features, %d pairs, ratio 0.8 + cross-check, "
                       "%d-pair calls of sfm_hip_match_pairs" % (job.n_images, N_FEATURES, n, PAIRS_PER_CALL),
           "n_gpus": job.world, "pairs": n, "replication": job.replication,
           "images_resident_per_rank": job.images_per_rank,
           "uploads": "overlapped with matching: each image right before the first call that names it (inside match_s)"
                      if overlapped else "a phase of its own before the matching",
           "uploads_s": t_up_inside if overlapped else t_up, "match_s": t_match, "gather_s": t_gather,
           "pairs_per_s_matching": n / t_match, "pairs_per_s_with_uploads_and_gather": n / (t_up + t_match + t_gather),
           "value": n / (t_up + t_match + t_gather), "unit": "pairs/s",
           "frac_of_fp32_mfma_peak_matching": n / t_match * flop / (job.world * PEAK_FP32_MFMA_TFLOPS * 1e12),
           "rank0_match_s": t_local, "roofline": roofline(st, engine, None),
           "rescanned_queries": st["rescanned_queries"]}
    leg.update(job.check(head[:, 0], off, g_lhs, g_rhs))
    return leg


def batched_leg(m, engine, n_pairs=64, calls=24):
    """The same pair handed to the library 64 at a time (how the C++ matcher class calls it on a whole
    image set): chunks then span all columns, the per-workgroup set-up is paid once per eight times
    the work, and the re-rank and compaction of one call run beside the next call's distance kernel."""
    m.set_engine(engine)
    pairs = np.array([[0, 1]] * n_pairs, np.int64)
    for _ in range(4):
        m.match_pairs_async(pairs, 0.8, True, stride=N_FEATURES)
    m.sync()
    m.set_profiling(True)
    m.reset_stats()
    t0 = time.perf_counter()
    for _ in range(calls):
        m.match_pairs_async(pairs, 0.8, True, stride=N_FEATURES)
    m.sync()
    elapsed = time.perf_counter() - t0
    st = m.stats()
    m.set_profiling(False)
    value = calls * n_pairs / elapsed
    return {"pairs_per_call": n_pairs, "calls": calls, "engine": engine, "value": value, "unit": "pairs/s",
            "ms_per_pair": elapsed / (calls * n_pairs) * 1e3,
            "frac_of_fp32_mfma_peak_end_to_end": value * 2.0 * N_FEATURES * N_FEATURES * 128 / (PEAK_FP32_MFMA_TFLOPS * 1e12),
            "roofline": roofline(st, engine, None)}


def large_pair_leg(m, engine, n=16384, steps=60):
    """One pair at the image size of BASELINE.json configs[4] (16384 x 16384 descriptors), one pair per call:
    four times the work of the headline pair per launch, chunks of 2,048 columns."""
    from sfm_amd import synth
    m.set_engine(engine)
    A, B, truth = synth.make_pair(n, n, n // 2, seed=4096)
    m.upload(20, A)
    m.upload(21, B)
    pair = np.array([[20, 21]], np.int64)
    for _ in range(20):
        out = m.match_pairs_async(pair, 0.8, True, stride=n)
    m.sync()
    cnt = int(out[0][0])
    found = len(set(zip(out[1][0, :cnt].tolist(), out[2][0, :cnt].tolist())) & set(map(tuple, truth.tolist())))
    m.set_profiling(True)
    m.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        m.match_pairs_async(pair, 0.8, True, stride=n)
    m.sync()
    elapsed = time.perf_counter() - t0
    st = m.stats()
    m.set_profiling(False)
    m.drop(20)
    m.drop(21)
    value = steps / elapsed
    return {"workload": "single synthetic pair, %dx%d 128-d float descriptors, ratio 0.8 + cross-check" % (n, n),
            "engine": engine, "value": value, "unit": "pairs/s", "ms_per_step": elapsed / steps * 1e3,
            "matches_found": cnt, "planted_found": found, "rescanned_queries_per_pair": st["rescanned_queries"] / max(1, st["pairs"]),
            "frac_of_fp32_mfma_peak_end_to_end": value * 2.0 * n * n * 128 / (PEAK_FP32_MFMA_TFLOPS * 1e12),
            "roofline": roofline(st, engine, None)}


def roofline(st, engine, traffic):
    """Distance kernel against the matrix-instruction peak of the type it multiplies in.

    achieved counts the algorithmic work only: 2*Na*Nb*128 FLOP per pair, once, however many
    partial products the engine actually issues (the split-fp16 engine issues three per term).
    """
    if not st["distance_launches"]:
        return None
    avg_ms = st["distance_ms"] / st["distance_launches"]
    per_launch = st["distance_flop"] / st["distance_launches"]
    achieved = per_launch / (avg_ms * 1e-3) / 1e12
    peak = PEAK_FP32_MFMA_TFLOPS if engine == "f32" else PEAK_FP16_MFMA_TFLOPS
    r = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
         "traffic": traffic, "kernel": "k_distance_topk<%s>" % engine, "avg_launch_ms": avg_ms,
         "flop_per_launch": per_launch,
         # the library brackets one launch in eight with HIP events (a start event is a packet of its
         # own in front of the kernel and costs a few microseconds)
         "timed_launches": st["distance_launches"]}
    if engine == "f16x3":
        # what the matrix units execute: 24 fp16 instructions of 32x32x16 and one fp32 32x32x2 per tile
        r["executed_over_algorithmic"] = (24 * 16 + 2) / 128.0
        r["executed_frac_of_peak"] = r["frac"] * r["executed_over_algorithmic"]
    return r


def launch_ranks(n, script=None, argv=None):
    """`python bench.py --gpus N` without a launcher around it: this process becomes the parent of N
    rank processes (the environment torch.distributed.run would give them), relays what they print
    -- rank 0 prints the JSON line -- and exits non-zero if any of them does.  It makes no GPU call
    itself (counting devices is none) and replaces no process image: children are started with
    subprocess before anything here could touch the GPU."""
    import socket
    import subprocess
    import torch

    backend = os.environ.get("SFM_BENCH_BACKEND", "nccl")
    visible = torch.cuda.device_count()
    if n > visible and backend != "gloo":
        sys.stderr.write("bench.py: --gpus %d but %d GPU(s) are visible; one rank per GPU over RCCL needs %d "
                         "(SFM_BENCH_BACKEND=gloo lets ranks share GPUs for a rehearsal)\n" % (n, visible, n))
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    children = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SFM_BENCH_LAUNCHED_BY="bench.py")
        # rank 0 owns this process's stdout (the one JSON line); whatever another rank prints is commentary
        children.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)]
                                         + (sys.argv[1:] if argv is None else list(argv)), env=env,
                                         stdout=None if r == 0 else sys.stderr))
    rc = 0
    kill_at = None
    pending = dict(enumerate(children))
    while pending:
        for r, c in list(pending.items()):
            code = c.poll()
            if code is None:
                continue
            del pending[r]
            if code != 0 and rc == 0:
                rc = code if 0 < code < 256 else 1
                sys.stderr.write("bench.py: rank %d exited with status %d; stopping the other ranks\n" % (r, code))
                for o in pending.values():     # exactly the processes started above
                    o.terminate()
                kill_at = time.monotonic() + 15.0
        if kill_at is not None and time.monotonic() > kill_at:
            for o in pending.values():
                o.kill()
            kill_at = None
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pairs-per-step", type=int, default=1,
                    help="N = 1: pairs handed to one sfm_hip_match_pairs call (default 1: the headline configuration)")
    ap.add_argument("--engine", choices=["f32", "f16x3"], default="f32",
                    help="how the distance kernel forms its products (include/sfm_hip.h)")
    ap.add_argument("--no-other-engine", action="store_true",
                    help="do not time the library's other distance engine after the main run")
    ap.add_argument("--no-replication-check", action="store_true",
                    help="N > 1: do not time host uploads against the RCCL all-gather of a descriptor set "
                         "(outside the timed region)")
    ap.add_argument("--set-images", type=int, default=0,
                    help="images of the exhaustive job: N > 1 default 500 (BASELINE.json configs[3]), "
                         "N = 1 whole_job leg default 100 (configs[2])")
    ap.add_argument("--set-features", type=int, default=8192,
                    help="N > 1: features per image of the exhaustive job (16384 with --set-images 1000: configs[4])")
    ap.add_argument("--replication", choices=["host", "collective"], default="host",
                    help="N > 1: how the job's descriptors reach the GPUs: every rank uploads its shard's images "
                         "from its host memory, or every rank uploads a 1/N share of all images and an RCCL "
                         "all-gather over xGMI replicates them (configs[4]'s descriptor broadcast)")
    ap.add_argument("--no-whole-job", action="store_true",
                    help="do not run the whole exhaustive job (uploads + matching + gather) after the timed region")
    ap.add_argument("--no-batched", action="store_true",
                    help="do not time the 64-pairs-per-call leg after the main run")
    ap.add_argument("--no-cascade", action="store_true",
                    help="do not time the cascade-hash matcher after the main run")
    ap.add_argument("--no-geometry", action="store_true",
                    help="do not time the geometric verification of matches after the main run")
    ap.add_argument("--untimed-pass", action="store_true",
                    help="after the timed region, run the same K steps once more without per-launch kernel "
                         "timing and report that rate as an extra field")
    ap.add_argument("--preheat-ms", type=float, default=100.0,
                    help="before the W warm-up steps, run the step for this long: the GPU needs ~25 ms of "
                         "load to reach its working clocks (the first 50 steps take 175-195 us each, "
                         "steps after the 150th 150 us), whatever W is")
    ap.add_argument("--extra-leg-seconds", type=float, default=150.0,
                    help="N > 1: the replication check is given up after this many seconds, the whole-job leg after "
                         "twice as many (a rank stuck in a collective must not cost the line)")
    ap.add_argument("--no-profile-events", action="store_true",
                    help="do not bracket the distance kernel with HIP events in the timed region")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not (args.gpus == 1 and world == 1):
        # a launcher's world size and --gpus must say the same thing: a line whose n_gpus differs from
        # what was asked for is worse than no line
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SFM_BENCH_FORCE_DIST=1 takes the multi-rank code path (sharded job, RCCL group, barriers, max
    # over ranks, gather) even with one rank: a rehearsal for machines with a single GPU
    dist_on = world > 1 or os.environ.get("SFM_BENCH_FORCE_DIST") == "1"
    # SFM_BENCH_BACKEND=gloo: collectives over CPU tensors, ranks share the visible GPUs round robin --
    # lets several ranks rehearse the multi-rank path on a machine with one GPU (RCCL refuses two
    # ranks on one device).  The default, nccl, is RCCL: one rank per GPU.
    backend = os.environ.get("SFM_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank if dist_on else 0)
    device = torch.device("cuda", local_rank if dist_on else 0)
    coll_device = device if backend == "nccl" else torch.device("cpu")

    import sfm_amd
    from sfm_amd import synth

    # The matcher's context first, the process group after it: a process has four hardware queues and
    # streams are given them in order of creation.  The context's four streams must not share one
    # (a tail kernel queued in front of a distance kernel delays it); RCCL's streams, idle during the
    # timed region, may share as they like.
    global N_FEATURES, PLANTED
    m = sfm_amd.Matcher(local_rank if dist_on else 0)
    m.set_engine(args.engine)
    t0 = time.perf_counter()
    if dist_on:
        if args.set_features % 256 or args.set_features <= 0:
            raise SystemExit("--set-features must be a positive multiple of 256")
        N_FEATURES, PLANTED = args.set_features, args.set_features // 2
        job = SetJob(args.set_images or 500, rank, world, args.replication)
        if not len(job.mine):
            raise SystemExit("rank %d has no pairs: fewer pairs than ranks" % rank)
        setup_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        if args.replication == "host":
            job.upload(m)
        else:   # the collective needs the process group, which comes after the context's streams are in use
            for i in (0, 1):
                m.upload(10**9 + i, np.zeros((256, 128), np.float32))
            for _ in range(4):
                m.match_pairs_async(np.array([[10**9, 10**9 + 1]], np.int64), 0.8, True, stride=256)
            m.sync()
            m.drop_all()
        upload_s = time.perf_counter() - t0
        per_step = PAIRS_PER_CALL
        step_no = [0]

        def step():
            k = step_no[0]
            step_no[0] += 1
            return m.match_pairs_async(job.batch(k), 0.8, True, stride=N_FEATURES)
    else:
        A, B, truth = synth.make_pair(N_FEATURES, N_FEATURES, PLANTED, seed=2024 + rank)
        m.upload(0, A)
        m.upload(1, B)
        upload_s = time.perf_counter() - t0
        per_step = args.pairs_per_step
        pairs = np.array([[0, 1]] * per_step, np.int64)

        def step():
            return m.match_pairs_async(pairs, 0.8, True, stride=N_FEATURES)
    if not (dist_on and args.replication == "collective"):
        for _ in range(4):   # every stream of the context has been used once (queues are bound at first use)
            step()
        m.sync()
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:   # the one-rank rehearsal, started by hand
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        if args.replication == "collective":
            t0 = time.perf_counter()
            job.upload(m, coll_device)
            upload_s = time.perf_counter() - t0

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not dist_on:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed_region(engine, step, steps, alone=False):
        """W untimed steps, then K timed steps between barriers; kernel timing on (one launch in eight).
        alone: this rank only, no barrier with the others (they are parked at one)."""
        dist_on_here = dist_on and not alone
        m.set_engine(engine)
        # Kernel timing is already on during the preheat and the warm-up: the second timed launch of
        # a process with an RCCL communicator takes 5 ms longer than any other (a one-time effect of
        # the timing marker, seen in a kernel trace), and that belongs to the warm-up.
        m.set_profiling(not args.no_profile_events)
        t_heat = time.perf_counter()
        while (time.perf_counter() - t_heat) * 1e3 < args.preheat_ms:
            for _ in range(20 if per_step == 1 else 2):
                step()
            m.sync()
        # The W warm-up steps are still running while the ranks meet at the barrier (with a few more
        # behind them to cover it): a barrier on an idle GPU lets its clocks drop again.
        for _ in range(args.warmup + (40 if dist_on_here and per_step == 1 else 0)):
            step()
        if dist_on_here:
            dist.barrier()
        m.sync()
        m.set_profiling(not args.no_profile_events)
        m.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        m.sync()
        if dist_on_here:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        st = m.stats()
        m.set_profiling(False)
        return out, elapsed, st

    # The like-for-like denominator of a scaling figure, measured in this run: rank 0 does the same
    # steps (64-pair calls on its own shard) while every other rank waits at a barrier, its GPU idle.
    solo = None
    if dist_on:
        if rank == 0:
            _, solo_elapsed, solo_st = timed_region(args.engine, step, args.steps, alone=True)
            solo = {"value": args.steps * per_step / solo_elapsed, "ms_per_step": solo_elapsed / args.steps * 1e3,
                    "roofline": roofline(solo_st, args.engine, None)}
            step_no[0] = 0
        barrier()

    out, elapsed, st = timed_region(args.engine, step, args.steps)
    out = tuple(a.copy() for a in out)    # the pinned buffers are the binding's and go when the call shape changes
    elapsed = max_over_ranks(elapsed)
    # extra information outside the timed region: the same K steps without per-launch kernel timing
    free_rate = None
    if args.untimed_pass:
        m.sync()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        m.sync()
        free_rate = args.steps * per_step / (time.perf_counter() - t1)

    flop_per_pair = 2.0 * N_FEATURES * N_FEATURES * 128
    line = None
    gave_up = False   # an optional multi-rank leg ran out of time (with_deadline)
    if dist_on:
        # secondary: each rank's single-pair rate (BASELINE.json configs[1]) on the first pair of its shard
        one = job.pairs[job.mine[:1]]
        for _ in range(50):
            m.match_pairs_async(one, 0.8, True, stride=N_FEATURES)
        m.sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            m.match_pairs_async(one, 0.8, True, stride=N_FEATURES)
        m.sync()
        rate = args.steps / (time.perf_counter() - t1)
        rates = torch.zeros(world, dtype=torch.float64, device=coll_device)
        rates[rank] = rate
        dist.all_reduce(rates)
        ordinals = torch.zeros(world, dtype=torch.float64, device=coll_device)
        ordinals[rank] = device.index
        dist.all_reduce(ordinals)
        # The two legs below are extras with collectives in them; neither may cost the line above them.
        replication, whole = None, None
        if not args.no_replication_check:
            replication, gave_up = with_deadline(lambda: replication_check(m, rank, world, coll_device, barrier),
                                                 args.extra_leg_seconds, "the replication check", device)
        if not args.no_whole_job and not gave_up:
            whole, gave_up = with_deadline(lambda: whole_job_leg(m, job, barrier, max_over_ranks, coll_device, args.engine),
                                           2 * args.extra_leg_seconds, "the whole-job leg", device)
        elif not args.no_whole_job:
            whole = {"error": "not started: an earlier leg was given up"}
        if rank == 0:
            value = args.steps * per_step * world / elapsed
            cnt = out[0]
            line = {
                "metric": "image pairs/sec, 8k x 8k 128-d descriptors",
                "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                # what actually ran: ranks of the process group, its backend ("nccl" is RCCL), each rank's device
                "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                "device_ordinal_per_rank": [int(x) for x in ordinals.tolist()],
                "launched_by": os.environ.get("SFM_BENCH_LAUNCHED_BY", "an outside launcher (torch.distributed.run)"),
                # the same step on rank 0's GPU alone, the other ranks waiting at a barrier: the
                # denominator a scaling figure of THIS line needs (not the N = 1 line's headline value,
                # which is one pair per call)
                "n1_same_step_pairs_per_s": solo["value"], "n1_same_step_ms_per_step": solo["ms_per_step"],
                "n1_same_step_roofline": solo["roofline"],
                "scaling_efficiency": value / (world * solo["value"]),
                "config": {"workload": "exhaustive matching of %d synthetic images x %d features (%d pairs, BASELINE.json "
                                       "configs[%s]%s), ratio 0.8 + cross-check, pair list cut into %d shards of equal cost "
                                       "(2-D blocks of the pair triangle); a step = one %d-pair call of sfm_hip_match_pairs "
                                       "per GPU on the next pairs of its shard"
                                       % (job.n_images, N_FEATURES, len(job.pairs),
                                          "4" if N_FEATURES == 16384 else "3",
                                          "" if (job.n_images, N_FEATURES) in ((500, 8192), (1000, 16384)) else ": a subset",
                                          world, PAIRS_PER_CALL),
                           "engine": args.engine, "pairs_per_step_per_gpu": per_step, "planted_matches": PLANTED,
                           "matches_last_call_min_max": [int(cnt.min()), int(cnt.max())],
                           "images_resident_per_gpu": job.images_per_rank, "replication": job.replication,
                           "parallelism": "pairs sharded over ranks, one process per GPU, no collective on the data path"},
                "frac_of_fp32_mfma_peak_end_to_end": value / world * flop_per_pair / (PEAK_FP32_MFMA_TFLOPS * 1e12),
                "rescanned_queries_per_pair": st["rescanned_queries"] / max(1, st["pairs"]),
                "setup_seconds_generating_images": setup_s, "upload_seconds_this_ranks_images": upload_s,
                "preheat_ms": args.preheat_ms, "pairs_per_s_this_gpu_without_kernel_timing": free_rate,
                "roofline": roofline(st, args.engine, None),
                "single_pair_rate_per_gpu": {"unit": "pairs/s", "workload": "BASELINE.json configs[1], one pair per call",
                                             "per_rank": [float(x) for x in rates.tolist()]},
            }
            if replication is not None:
                line["replication"] = replication
            if whole is not None:
                line["whole_job"] = whole
    elif rank == 0:
        cnt = int(out[0][0])
        got = (out[1][0, :cnt].astype(np.int64), out[2][0, :cnt].astype(np.int64))
        value = args.steps * per_step / elapsed
        # HBM-side bytes per launch of the distance kernel come from a separate hardware-counter
        # run of this same workload (tools/collect_profiles.sh), kept beside the profiles.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "distance_kernel_traffic.json")
        if os.path.exists(tpath) and per_step == 1:
            with open(tpath) as fh:
                traffic = json.load(fh).get(args.engine, {}).get("traffic_bytes_per_launch")
        planted_found = len(set(zip(got[0].tolist(), got[1].tolist())) & set(map(tuple, truth.tolist())))
        line = {
            "metric": "image pairs/sec, 8k x 8k 128-d descriptors",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "single synthetic pair, 8192x8192 128-d float descriptors, "
                                   "ratio 0.8 + cross-check (BASELINE.json configs[1])",
                       "engine": args.engine, "pairs_per_step_per_gpu": per_step, "planted_matches": PLANTED, "matches_found": cnt,
                       "planted_found": planted_found, "parallelism": "one GPU"},
            "frac_of_fp32_mfma_peak_end_to_end": value * flop_per_pair / (PEAK_FP32_MFMA_TFLOPS * 1e12),
            "rescanned_queries_per_pair": st["rescanned_queries"] / max(1, st["pairs"]),
            "upload_seconds_both_images": upload_s, "preheat_ms": args.preheat_ms,
            "pairs_per_s_this_gpu_without_kernel_timing": free_rate,
            "roofline": roofline(st, args.engine, traffic),
        }
        if traffic is not None and line["roofline"] is not None:
            line["roofline"]["traffic_source"] = "profiles/distance_kernel_traffic.json (separate counter run)"
        if not args.no_other_engine:
            # the library's other distance engine on the same steps (its own timed region)
            other = "f16x3" if args.engine == "f32" else "f32"
            o_out, o_elapsed, o_st = timed_region(other, step, args.steps)
            o_cnt = int(o_out[0][0])
            same = (o_cnt == cnt and o_out[1][0, :cnt].tolist() == got[0].tolist()
                    and o_out[2][0, :cnt].tolist() == got[1].tolist())
            o_value = args.steps * per_step / o_elapsed
            line["other_engine"] = {
                "engine": other, "value": o_value, "unit": "pairs/s", "ms_per_step": o_elapsed / args.steps * 1e3,
                "frac_of_fp32_mfma_peak_end_to_end": o_value * flop_per_pair / (PEAK_FP32_MFMA_TFLOPS * 1e12),
                "rescanned_queries_per_pair": o_st["rescanned_queries"] / max(1, o_st["pairs"]),
                "matches_equal_main_engine": bool(same), "roofline": roofline(o_st, other, None)}
            m.set_engine(args.engine)
        if not args.no_batched and per_step == 1:
            line["batched"] = batched_leg(m, args.engine)
            # the step of the N > 1 lines (64-pair calls) on this one GPU: what their value divides by
            line["value_same_step_as_multi_gpu"] = line["batched"]["value"]
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(A, B, got)
        if not args.no_batched and per_step == 1:
            try:
                line["pair_16k"] = large_pair_leg(m, args.engine)
            except Exception as e:   # an extra, never the reason the headline line is lost
                line["pair_16k"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_whole_job:
            try:
                m.set_engine(args.engine)
                job = SetJob(args.set_images or 100, 0, 1)
                line["whole_job"] = whole_job_leg(m, job, barrier, max_over_ranks, device, args.engine)
                cb = line.get("cpu_baseline")
                if cb:
                    # the reference cannot hold this job in memory (it queues deep copies for every task up
                    # front); its cost is linear in pairs, so the one-pair timings are scaled, not measured
                    n = line["whole_job"]["pairs"]
                    line["whole_job"]["cpu_baseline"] = {
                        "extrapolated": True, "kind": "port", "unit": "pairs/s",
                        "sample": "one measured 8192x8192 pair x %d pairs (cost is linear in pairs)" % n,
                        "value": cb["value"], "cores": cb["cores"], "seconds_whole_job": n * cb["seconds_per_pair"],
                        "all_cores": {"value": cb["all_cores"]["value"], "cores": cb["all_cores"]["cores"],
                                      "seconds_whole_job": n * cb["all_cores"]["seconds_per_pair"]}}
                m.drop_all()
            except Exception as e:   # an extra, never the reason the headline line is lost
                line["whole_job"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_cascade:
            try:
                line["cascade_hash"] = cascade_leg(m, not args.no_cpu_baseline)
            except Exception as e:   # an extra, never the reason the headline line is lost
                line["cascade_hash"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if not args.no_geometry:
            try:
                line["geometry"] = geometry_leg(m, not args.no_cpu_baseline)
            except Exception as e:   # an extra, never the reason the headline line is lost
                line["geometry"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if line is not None:
        print(json.dumps(line), flush=True)
    if gave_up:
        # a thread of this process is still inside a collective or the library: no orderly way out
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)

    m.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
