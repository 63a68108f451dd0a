#!/usr/bin/env python3
"""Benchmark of the brute-force matching hot path on MI355X.

A step = one pass of the hot path over one image pair of the headline
configuration (8192 x 8192 descriptors of 128 floats, ratio 0.8, cross-check):
distance/top-k kernel, exact re-rank, cross-check, compaction, copy of the
compact match list to pinned host memory.  Descriptors are resident in HBM
before the timed region starts.  With --gpus N every rank matches its own
pairs (pairs are independent: no collective on the data path), so the value is
the sum over ranks ("weak" scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
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


def cpu_baseline(A, B, expected):
    """Time the CPU oracle (a port of the reference matcher) on this host.

    Sample: one full 8192x8192 pair with cross-check on two threads, which is
    how the reference runs a pair (its ThreadPool has two workers, one per
    direction); then the same pair on 16 threads.
    """
    from oracle import oracle
    t0 = time.perf_counter()
    l, r, d = oracle.match_pair(A, B, 0.8, True, threads=2)
    t2 = time.perf_counter() - t0
    same = (l.tolist() == expected[0].tolist() and r.tolist() == expected[1].tolist())
    t0 = time.perf_counter()
    oracle.match_pair(A, B, 0.8, True, threads=16)
    t16 = time.perf_counter() - t0
    return {"value": 1.0 / t2, "unit": "pairs/s", "cores": 2, "kind": "port",
            "sample": "one full 8192x8192 pair, ratio 0.8 + cross-check, 2 threads (the reference's pool size)",
            "seconds_per_pair": t2, "matches_equal_gpu": bool(same),
            "all_cores": {"value": 1.0 / t16, "cores": 16, "seconds_per_pair": t16}}


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
    import numpy as np
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
    m.set_profiling(False)
    gbps = st["match_bytes"] / (st["match_ms"] * 1e-3) / 1e9
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
                        "avg_batch_ms": st["match_ms"] / st["match_launches"],
                        "bytes_per_pair": st["match_bytes"] / (reps * len(pairs))}}
    if with_cpu:
        sub = {i: d for i, d in zip(ids[:4], imgs[:4])}
        sp = [(a, b) for k, a in enumerate(ids[:4]) for b in ids[k + 1:4]]
        m.cascade_prepare(sp)       # the mean vector is taken over the images that take part
        leg["cpu_baseline"] = cascade_cpu_baseline(sub, sp, m.cascade_match_pairs(sp, True))
    for i in ids:
        m.drop(i)
    return leg


def replication_check(m, rank, world, local_rank, barrier, n_images=64):
    """Outside the timed region, N > 1 only: how long it takes to give every GPU a descriptor set
    of n_images x 8192 rows (i) by uploading everything from each rank's host memory and (ii) by
    uploading each rank's share and all-gathering over RCCL/xGMI.  Reported, never fatal."""
    import numpy as np
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
            replicate_images(m, src, rank, world, device=torch.device("cuda", local_rank), method=method)
            barrier()
            tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            out[name] = float(tt.item())
            for i in ids:
                m.drop(i)
        return out
    except Exception as e:   # a report, not a gate
        return {"error": "%s: %s" % (type(e).__name__, e)}


def batched_leg(m, engine, n_pairs=64, calls=24):
    """The same pair handed to the library 64 at a time (how the C++ matcher class calls it on a whole
    image set): chunks then span all columns, the per-workgroup set-up is paid once per eight times
    the work, and the re-rank and compaction of one call run beside the next call's distance kernel."""
    import numpy as np
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pairs-per-step", type=int, default=1,
                    help="pairs handed to one sfm_hip_match_pairs call (default 1: the headline configuration)")
    ap.add_argument("--engine", choices=["f32", "f16x3"], default="f32",
                    help="how the distance kernel forms its products (include/sfm_hip.h)")
    ap.add_argument("--no-other-engine", action="store_true",
                    help="do not time the library's other distance engine after the main run")
    ap.add_argument("--replication-check", action="store_true",
                    help="N > 1: also time host uploads against the RCCL all-gather of a descriptor set "
                         "(outside the timed region; adds collectives to the run, hence opt-in)")
    ap.add_argument("--no-batched", action="store_true",
                    help="do not time the 64-pairs-per-call leg after the main run")
    ap.add_argument("--no-cascade", action="store_true",
                    help="do not time the cascade-hash matcher after the main run")
    ap.add_argument("--untimed-pass", action="store_true",
                    help="after the timed region, run the same K steps once more without per-launch kernel "
                         "timing and report that rate as an extra field")
    ap.add_argument("--preheat-ms", type=float, default=100.0,
                    help="before the W warm-up steps, run the step for this long: the GPU needs ~25 ms of "
                         "load to reach its working clocks (the first 50 steps take 175-195 us each, "
                         "steps after the 150th 150 us), whatever W is")
    ap.add_argument("--no-profile-events", action="store_true",
                    help="do not bracket the distance kernel with HIP events in the timed region")
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SFM_BENCH_FORCE_DIST=1 takes the multi-rank code path (RCCL group, barriers, max over ranks)
    # even with one rank: a rehearsal for machines with a single GPU
    dist_on = world > 1 or os.environ.get("SFM_BENCH_FORCE_DIST") == "1"
    torch.cuda.set_device(local_rank if dist_on else 0)

    import sfm_amd
    from sfm_amd import synth

    # The matcher's context first, the process group after it: a process has four hardware queues and
    # streams are given them in order of creation.  The context's four streams must not share one
    # (a tail kernel queued in front of a distance kernel delays it); RCCL's streams, idle during the
    # timed region, may share as they like.
    m = sfm_amd.Matcher(local_rank if dist_on else 0)
    m.set_engine(args.engine)
    A, B, truth = synth.make_pair(N_FEATURES, N_FEATURES, PLANTED, seed=2024 + rank)
    t0 = time.perf_counter()
    m.upload(0, A)
    m.upload(1, B)
    upload_s = time.perf_counter() - t0
    pairs = np.array([[0, 1]] * args.pairs_per_step, np.int64)
    for _ in range(4):   # every stream of the context has been used once (queues are bound at first use)
        m.match_pairs_async(pairs, 0.8, True, stride=N_FEATURES)
    m.sync()
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(engine):
        """W untimed steps, then K timed steps between barriers; kernel timing on (one launch in eight)."""
        m.set_engine(engine)
        t_heat = time.perf_counter()
        while (time.perf_counter() - t_heat) * 1e3 < args.preheat_ms:
            for _ in range(20):
                m.match_pairs_async(pairs, 0.8, True, stride=N_FEATURES)
            m.sync()
        # The W warm-up steps are still running while the ranks meet at the barrier (with a few more
        # behind them to cover it): a barrier on an idle GPU lets its clocks drop again.
        for _ in range(args.warmup + (40 if dist_on else 0)):
            m.match_pairs_async(pairs, 0.8, True, stride=N_FEATURES)
        if dist_on:
            dist.barrier()
        m.sync()
        m.set_profiling(not args.no_profile_events)
        m.reset_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = m.match_pairs_async(pairs, 0.8, True, stride=N_FEATURES)
        m.sync()
        barrier()
        elapsed = time.perf_counter() - t0
        st = m.stats()
        m.set_profiling(False)
        return out, elapsed, st

    out, elapsed, st = timed_region(args.engine)
    # extra information outside the timed region: the same K steps without per-launch kernel timing
    # (the library then lets consecutive distance kernels overlap at their edges)
    free_rate = None
    if args.untimed_pass:
        m.sync()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            m.match_pairs_async(pairs, 0.8, True, stride=N_FEATURES)
        m.sync()
        free_rate = args.steps * args.pairs_per_step / (time.perf_counter() - t1)
    replication = None
    if dist_on and args.replication_check:
        replication = replication_check(m, rank, world, local_rank, barrier)

    cnt = int(out[0][0])
    got = (out[1][0, :cnt].astype(np.int64), out[2][0, :cnt].astype(np.int64))

    if dist_on:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        total_pairs = args.steps * world * args.pairs_per_step
        value = total_pairs / elapsed
        flop_per_launch = 2.0 * N_FEATURES * N_FEATURES * 128
        # HBM-side bytes per launch of the distance kernel come from a separate hardware-counter
        # run of this same workload (tools/collect_profiles.sh), kept beside the profiles.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "distance_kernel_traffic.json")
        if os.path.exists(tpath) and args.pairs_per_step == 1:
            with open(tpath) as fh:
                traffic = json.load(fh).get(args.engine, {}).get("traffic_bytes_per_launch")
        roof = roofline(st, args.engine, traffic)
        planted_found = len(set(zip(got[0].tolist(), got[1].tolist())) & set(map(tuple, truth.tolist())))
        line = {
            "metric": "image pairs/sec, 8k x 8k 128-d descriptors",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "single synthetic pair, 8192x8192 128-d float descriptors, "
                                   "ratio 0.8 + cross-check (BASELINE.json configs[1])",
                       "engine": args.engine, "pairs_per_step_per_gpu": args.pairs_per_step, "planted_matches": PLANTED, "matches_found": cnt,
                       "planted_found": planted_found, "parallelism": "pairs sharded over ranks, no collective"},
            "frac_of_fp32_mfma_peak_end_to_end": value / world * flop_per_launch / (PEAK_FP32_MFMA_TFLOPS * 1e12),
            "rescanned_queries_per_pair": st["rescanned_queries"] / max(1, st["pairs"]),
            "upload_seconds_both_images": upload_s, "preheat_ms": args.preheat_ms,
            "pairs_per_s_this_gpu_without_kernel_timing": free_rate,
            "roofline": roof,
        }
        if replication is not None:
            line["replication"] = replication
        if world == 1 and not args.no_other_engine:
            # the library's other distance engine on the same steps (its own timed region)
            other = "f16x3" if args.engine == "f32" else "f32"
            o_out, o_elapsed, o_st = timed_region(other)
            o_cnt = int(o_out[0][0])
            same = (o_cnt == cnt and o_out[1][0, :cnt].tolist() == got[0].tolist()
                    and o_out[2][0, :cnt].tolist() == got[1].tolist())
            o_value = args.steps * args.pairs_per_step / o_elapsed
            line["other_engine"] = {
                "engine": other, "value": o_value, "unit": "pairs/s", "ms_per_step": o_elapsed / args.steps * 1e3,
                "frac_of_fp32_mfma_peak_end_to_end": o_value * flop_per_launch / (PEAK_FP32_MFMA_TFLOPS * 1e12),
                "rescanned_queries_per_pair": o_st["rescanned_queries"] / max(1, o_st["pairs"]),
                "matches_equal_main_engine": bool(same), "roofline": roofline(o_st, other, None)}
        if world == 1 and not args.no_batched and args.pairs_per_step == 1:
            line["batched"] = batched_leg(m, args.engine)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(A, B, got)
        if world == 1 and not args.no_cascade:
            try:
                line["cascade_hash"] = cascade_leg(m, not args.no_cpu_baseline)
            except Exception as e:   # an extra, never the reason the headline line is lost
                line["cascade_hash"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(line), flush=True)

    m.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
