"""Step time over the first second of load: per block of 50 single-pair steps (sync after each block)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import sfm_amd
from sfm_amd import synth
m = sfm_amd.Matcher(0)
A, B, truth = synth.make_pair(8192, 8192, 4096, seed=2024)
m.upload(0, A); m.upload(1, B)
pairs = np.array([[0, 1]], np.int64)
m.match_pairs_async(pairs, 0.8, True, stride=8192); m.sync()
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
out = []
for blk in range(24):
    t0 = time.perf_counter()
    for _ in range(50): m.match_pairs_async(pairs, 0.8, True, stride=8192)
    m.sync()
    out.append((time.perf_counter() - t0) / 50 * 1e6)
print(" ".join("%.1f" % v for v in out))
