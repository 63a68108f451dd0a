"""Throughput of the cascade-hash path on one GPU: upload, prepare (projection, mean, hashes),
then all pairs of a synthetic image set, with the CPU oracle timed beside it on a bounded sample.

    python tools/cascade_bench.py [--images 32] [--features 8192] [--reps 5] [--cpu-pairs 6]

Prints one JSON line.  The bytes figure is the library's own count of what a batch has to move at
the least (include/sfm_hip.h, sfm_hip_cascade_stats.match_bytes).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import sfm_amd  # noqa: E402
from sfm_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=32)
    ap.add_argument("--features", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cpu-pairs", type=int, default=6)
    ap.add_argument("--no-cross-check", action="store_true")
    a = ap.parse_args()
    cross = not a.no_cross_check
    imgs = synth.make_repetitive_set(a.images, a.features, 0.5, 2, seed=17, sigma=0.002)
    images = dict(enumerate(imgs))
    pairs = [(i, j) for i in range(a.images) for j in range(i + 1, a.images)]
    m = sfm_amd.Matcher(0)
    t0 = time.perf_counter()
    for i, d in images.items():
        m.upload(i, d)
    t1 = time.perf_counter()
    m.cascade_prepare(pairs)
    t2 = time.perf_counter()
    m.cascade_prepare(pairs)          # second time: allocations warm
    t3 = time.perf_counter()
    out = m.cascade_match_pairs_async(pairs, cross)
    m.sync()
    n_matches = int(out[1].sum())
    n_defaults = int(out[0].sum())
    m.set_profiling(True)
    m.cascade_reset_stats()
    t4 = time.perf_counter()
    for _ in range(a.reps):
        m.cascade_match_pairs_async(pairs, cross)
    m.sync()
    t5 = time.perf_counter()
    st = m.cascade_stats()
    m.set_profiling(False)
    t6 = time.perf_counter()
    for _ in range(a.reps):
        m.cascade_match_pairs_async(pairs, cross)
    m.sync()
    t7 = time.perf_counter()
    res = {
        "workload": "%d images x %d features, %d pairs, cross-check %s" % (a.images, a.features, len(pairs), cross),
        "upload_s": t1 - t0, "prepare_first_s": t2 - t1, "prepare_s": t3 - t2,
        "pairs_per_s": a.reps * len(pairs) / (t7 - t6),
        "pairs_per_s_timed": a.reps * len(pairs) / (t5 - t4),
        "gpu_ms_per_pair": st["match_ms"] / (a.reps * len(pairs)),
        "bytes_per_pair": st["match_bytes"] / (a.reps * len(pairs)),
        "achieved_GBps": st["match_bytes"] / (st["match_ms"] * 1e-3) / 1e9,
        "matches": n_matches, "default_records": n_defaults,
    }
    if a.cpu_pairs > 0:
        from oracle import cascade  # the checker, timed as the CPU baseline
        k = 2
        while k * (k - 1) // 2 < a.cpu_pairs and k < a.images:
            k += 1
        sub = {i: images[i] for i in range(k)}
        sp = [(i, j) for i in range(k) for j in range(i + 1, k)]
        whole, th, tm = cascade.match_set(sub, sp, cross, threads=1)
        got = m.cascade_match_pairs(sp, cross)
        # same mean only when the same images take part: re-prepare for the subset
        m.cascade_prepare(sp)
        got = m.cascade_match_pairs(sp, cross)
        same = all((nd, l.tolist(), r.tolist()) == (whole[p][0], whole[p][1].tolist(), whole[p][2].tolist())
                   for p, (nd, l, r, _) in zip(sp, got))
        res["cpu"] = {"images": k, "pairs": len(sp), "threads": 1, "hash_s_per_image": th / k,
                      "match_s_per_pair": tm / len(sp), "pairs_per_s": len(sp) / tm, "equal_to_gpu": bool(same)}
    m.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
