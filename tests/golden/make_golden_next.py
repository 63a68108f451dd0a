#!/usr/bin/env python3
"""Regenerate tests/golden/cascade_cases.json and tests/golden/geometry_cases.json.

The expected values are outputs of the CPU oracles (oracle/cascade_oracle.cc,
oracle/geometry_oracle.cc) on inputs from sfm_amd/synth.py.  Neither oracle is pinned against the
reference running (their headers say why and what stands in for that); these files pin the pair
(generator, oracle), so that a change of either, or a machine whose C++ or math library behaves
differently (the projection matrices come from std::mt19937 + std::normal_distribution, the
RANSAC round limit from std::log and std::pow), shows up as a changed digest.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import cascade, geometry  # noqa: E402
from sfm_amd import synth  # noqa: E402

CASCADE_CASES = [
    # name, images, features, shared fraction, copies, sigma, seed, cross
    ("pair", 2, 600, 0.5, 2, 0.002, 1, True),
    ("pair_nocross", 2, 600, 0.5, 2, 0.002, 1, False),
    ("triple", 3, 900, 0.6, 3, 0.001, 2, True),
    ("noisy", 2, 1500, 0.5, 2, 0.004, 3, True),
]
GEOMETRY_CASES = [
    # name, matches, inlier fraction, noise, seed, ransac seed, pair index
    ("clean", 50, 1.0, 0.0, 3, 4, 7),
    ("mostly_inliers", 1000, 0.9, 0.5, 4, 99, 5),
    ("half", 400, 0.5, 0.3, 5, 1, 0),
    ("hopeless", 60, 0.1, 0.5, 6, 2, 1),
    ("too_few", 7, 1.0, 0.5, 7, 3, 2),
]


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def cascade_case(c):
    name, n_img, n_feat, frac, copies, sigma, seed, cross = c
    imgs = synth.make_repetitive_set(n_img, n_feat, frac, copies, seed=seed, sigma=sigma)
    images = {3 * k + 1: d[: n_feat - 11 * k] for k, d in enumerate(imgs)}
    ids = sorted(images)
    pairs = [(a, b) for x, a in enumerate(ids) for b in ids[x + 1:]]
    mean = cascade.mean([images[int(i)] for i in cascade.image_order(pairs)])
    codes, bids = cascade.hash_image(images[ids[0]], mean)
    whole, _, _ = cascade.match_set(images, pairs, cross)
    return {"name": name, "images": n_img, "features": n_feat, "shared": frac, "copies": copies, "sigma": sigma,
            "seed": seed, "cross": cross, "input_sha256": digest(*[images[i] for i in ids]),
            "mean_bits_sha256": digest(mean.view(np.uint32)),
            "first_image_hashes_sha256": digest(codes, bids),
            "pairs": [{"pair": list(p), "defaults": whole[p][0], "count": int(len(whole[p][1])),
                       "lhs_rhs_sha256": digest(whole[p][1], whole[p][2])} for p in pairs]}


def geometry_case(c):
    name, n, frac, noise, seed, rseed, pair = c
    l, r, _ = synth.make_two_view_matches(n, frac, seed=seed, noise=noise)
    ok, F, flags, rounds = geometry.compute_fundamental(l, r, seed=rseed, pair=pair)
    return {"name": name, "matches": n, "inliers": frac, "noise": noise, "seed": seed, "ransac_seed": rseed, "pair": pair,
            "input_sha256": digest(l, r), "ok": bool(ok), "rounds": rounds, "n_inliers": int(flags.sum()),
            "F_bits_sha256": digest(F.view(np.uint64)), "flags_sha256": digest(flags)}


if __name__ == "__main__":
    p, s = cascade.projection()
    out = {"_provenance": __doc__.strip(), "projection_bits_sha256": digest(p.view(np.uint32), s.view(np.uint32)),
           "cases": [cascade_case(c) for c in CASCADE_CASES]}
    with open(os.path.join(HERE, "cascade_cases.json"), "w") as f:
        json.dump(out, f, indent=1)
    out = {"_provenance": __doc__.strip(), "cases": [geometry_case(c) for c in GEOMETRY_CASES]}
    with open(os.path.join(HERE, "geometry_cases.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(CASCADE_CASES), "cascade and", len(GEOMETRY_CASES), "geometry cases")
