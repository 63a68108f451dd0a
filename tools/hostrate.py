import sys, time, numpy as np
sys.path.insert(0, '.')
import sfm_amd
from sfm_amd import synth
m = sfm_amd.Matcher(0)
A, B, truth = synth.make_pair(8192, 8192, 4096, seed=2024)
m.upload(0, A); m.upload(1, B)
pairs = np.array([[0, 1]], np.int64)
for _ in range(30): m.match_pairs_async(pairs, 0.8, True, stride=8192)
m.sync()
for n in (300, 300):
    t0 = time.perf_counter()
    for _ in range(n): m.match_pairs_async(pairs, 0.8, True, stride=8192)
    t1 = time.perf_counter()
    m.sync()
    t2 = time.perf_counter()
print(sys.argv[1:], "enqueue us/call %.1f" % ((t1 - t0) / n * 1e6), "total us/step %.1f" % ((t2 - t0) / n * 1e6))
