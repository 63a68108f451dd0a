"""Fixed cost of a timed region of the single-pair step: region time for K = 10..160 steps after an
idle sync, least-squares line through them (slope = steady step, intercept = what a region pays once)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
import sfm_amd
from sfm_amd import synth
m = sfm_amd.Matcher(0)
A, B, _ = synth.make_pair(8192, 8192, 4096, seed=2024)
m.upload(0, A); m.upload(1, B)
pairs = np.array([[0, 1]], np.int64)
for _ in range(800): m.match_pairs_async(pairs, 0.8, True, stride=8192)
m.sync()
Ks = [10, 20, 40, 80, 160]
res = {k: [] for k in Ks}
for rep in range(7):
    for K in Ks:
        for _ in range(45): m.match_pairs_async(pairs, 0.8, True, stride=8192)
        m.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): m.match_pairs_async(pairs, 0.8, True, stride=8192)
        t1 = time.perf_counter()
        m.sync(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        res[K].append(((t2 - t0) * 1e6, (t1 - t0) * 1e6))
med = [np.median([r[0] for r in res[K]]) for K in Ks]
slope, icpt = np.polyfit(Ks, med, 1)
for K, v in zip(Ks, med):
    print("K=%3d: region %8.1f us = %.1f per step; host loop done after %.1f us" % (K, v, v / K, np.median([r[1] for r in res[K]])))
print("steady step %.2f us, fixed cost of a region %.1f us" % (slope, icpt))
