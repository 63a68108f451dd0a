#!/usr/bin/env python3
"""Regenerate tests/golden/oracle_cases.json.

The expected values are outputs of the CPU oracle (oracle/bf_oracle.c) on inputs from
sfm_amd/synth.py.  The oracle itself is pinned separately: its L2 metric against the
reference's metric.hpp built from the reference checkout (oracle/_ref), its matching rule
against the reference's recorded known answers (kat.json).  This file pins the pair
(generator, oracle) so that any machine can tell whether it reproduces the same bits.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from sfm_amd import synth  # noqa: E402

CASES = [
    # name, n_a, n_b, planted, seed, ratio, cross
    ("tiny", 5, 7, 3, 1, 0.8, True),
    ("ragged", 127, 129, 60, 2, 0.8, True),
    ("ragged_nocross", 127, 129, 60, 2, 0.8, False),
    ("wide", 200, 1000, 100, 3, 0.8, True),
    ("tall", 1000, 200, 100, 4, 0.8, True),
    ("k1", 1024, 1024, 512, 5, 0.8, True),
    ("k1_loose", 1024, 1024, 512, 5, 1.5, False),
    ("k2", 2048, 1536, 700, 6, 0.8, True),
    # the sizes the benchmark configurations use
    ("k4", 4096, 4096, 2048, 7, 0.8, True),
    ("k8", 8192, 8192, 4096, 8, 0.8, True),
    ("k8_nocross", 8192, 8192, 4096, 8, 0.8, False),
    ("k16", 16384, 16384, 8192, 9, 0.8, True),
]

# image sets: (name, images, features, seed); every two images share features/2 scene rows
# (synth.scene_image); all pairs in the order of the reference's pair builder
SET_CASES = [
    ("set6", 6, 2048, 10),
]


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def run_case(c):
    name, na, nb, planted, seed, ratio, cross = c
    A, B, truth = synth.make_pair(na, nb, planted, seed)
    l, r, d = oracle.match_pair(A, B, ratio, cross, threads=8)
    return {
        "name": name, "n_a": na, "n_b": nb, "planted": planted, "seed": seed, "ratio": ratio, "cross": cross,
        "input_sha256": digest(A, B),
        "count": int(len(l)),
        "lhs_rhs_sha256": digest(l.astype(np.int64), r.astype(np.int64)),
        "dist_bits_sha256": digest(d.view(np.uint32)),
        "first": [[int(a), int(b)] for a, b in zip(l[:4], r[:4])],
        "last": [[int(a), int(b)] for a, b in zip(l[-4:], r[-4:])],
    }


def set_images(c):
    name, n_images, n_features, seed = c
    pool = synth.scene_pool(n_features // 2, seed)
    return [synth.scene_image(i, n_features, pool, seed) for i in range(n_images)]


def set_digest(per_pair):
    """per_pair: [(lhs, rhs, dist)] in pair-list order -> what the fixture stores."""
    return {
        "counts": [int(len(l)) for l, _, _ in per_pair],
        "lhs_rhs_sha256": digest(*[np.asarray(x).astype(np.int64) for l, r, _ in per_pair for x in (l, r)]),
        "dist_bits_sha256": digest(*[np.asarray(d, dtype=np.float32).view(np.uint32) for _, _, d in per_pair]),
    }


def run_set_case(c):
    name, n_images, n_features, seed = c
    imgs = set_images(c)
    pairs = [(a, b) for a in range(n_images) for b in range(a + 1, n_images)]
    got = [oracle.match_pair(imgs[a], imgs[b], 0.8, True, threads=8) for a, b in pairs]
    out = {"name": name, "n_images": n_images, "n_features": n_features, "seed": seed, "ratio": 0.8, "cross": True,
           "pairs": [list(p) for p in pairs], "input_sha256": digest(*imgs)}
    out.update(set_digest(got))
    return out


if __name__ == "__main__":
    out = {"_provenance": __doc__.strip(), "cases": [run_case(c) for c in CASES],
           "set_cases": [run_set_case(c) for c in SET_CASES]}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_cases.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote %d cases" % len(out["cases"]))
