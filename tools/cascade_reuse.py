"""How often does a descriptor row recur among the close pairs (Hamming distance <= 10 in a shared
bucket) of an image pair of the bench's cascade-hash set?  CPU only (hashing by the oracle).
That decides what the exact-distance step must read at the least: a row per appearance, or less."""
import sys, numpy as np
sys.path.insert(0, '.')
from sfm_amd import synth
from oracle import cascade as oc

N = 8192
imgs = synth.make_repetitive_set(4, N, 0.5, 2, seed=17, sigma=0.002)
mean = oc.mean(imgs)
hashed = [oc.hash_image(im, mean) for im in imgs]

def popcount32(x):
    x = x - ((x >> 1) & 0x55555555)
    x = (x & 0x33333333) + ((x >> 2) & 0x33333333)
    x = (x + (x >> 4)) & 0x0F0F0F0F
    return (x * 0x01010101 & 0xFFFFFFFF) >> 24

for a, b in [(0, 1), (0, 2), (1, 3), (2, 3)]:
    (ca, ba), (cb, bb) = hashed[a], hashed[b]
    close = set()
    for g in range(6):
        order = np.argsort(bb[:, g], kind="stable")
        keys = bb[order, g]
        lo = np.searchsorted(keys, ba[:, g], "left")
        hi = np.searchsorted(keys, ba[:, g], "right")
        for i in np.nonzero(hi > lo)[0]:
            js = order[lo[i]:hi[i]]
            d = popcount32((ca[i][None, :] ^ cb[js]).astype(np.uint64) & 0xFFFFFFFF).sum(axis=1)
            for j in js[d <= 10]:
                close.add((int(i), int(j)))
    rows_a = {i for i, _ in close}
    rows_b = {j for _, j in close}
    print("pair (%d, %d): %d close pairs, %d distinct first-image rows, %d distinct second-image rows: "
          "%.2f row reads per distinct row" % (a, b, len(close), len(rows_a), len(rows_b),
                                               2 * len(close) / max(1, len(rows_a) + len(rows_b))))
