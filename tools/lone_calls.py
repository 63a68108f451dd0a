"""Fifty lone steps (call, then sync), for an API trace of one call's host path."""
import sys, numpy as np
sys.path.insert(0, '.')
import sfm_amd
from sfm_amd import synth
m = sfm_amd.Matcher(0)
A, B, _ = synth.make_pair(8192, 8192, 4096, seed=2024)
m.upload(0, A); m.upload(1, B)
pairs = np.array([[0, 1]], np.int64)
for _ in range(20): m.match_pairs_async(pairs, 0.8, True, stride=8192)
m.sync()
for _ in range(50):
    m.match_pairs_async(pairs, 0.8, True, stride=8192)
    m.sync()
