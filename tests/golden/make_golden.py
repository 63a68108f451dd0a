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


if __name__ == "__main__":
    out = {"_provenance": __doc__.strip(), "cases": [run_case(c) for c in CASES]}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_cases.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote %d cases" % len(out["cases"]))
