"""GPU output against the committed fixtures of tests/golden/oracle_cases.json: generator inputs,
match indices and distance bits as sha256.  No oracle library takes part here -- a machine whose
oracle build were stale or different would still be held to the numbers in the file."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from sfm_amd import synth

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden  # noqa: E402  (digest helpers and the set recipe; its oracle import is not used here)

pytestmark = pytest.mark.gpu


def _fixtures():
    with open(os.path.join(ROOT, "tests", "golden", "oracle_cases.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", _fixtures()["cases"], ids=lambda c: c["name"])
def test_pair_fixture(matcher, case):
    A, B, _ = synth.make_pair(case["n_a"], case["n_b"], case["planted"], case["seed"])
    assert make_golden.digest(A, B) == case["input_sha256"], "the generator gives other bits on this machine"
    matcher.upload(700, A)
    matcher.upload(701, B)
    try:
        (l, r, d), = matcher.match_pairs([(700, 701)], case["ratio"], case["cross"])
    finally:
        matcher.drop(700)
        matcher.drop(701)
    assert len(l) == case["count"]
    assert [[int(a), int(b)] for a, b in zip(l[:4], r[:4])] == case["first"]
    assert [[int(a), int(b)] for a, b in zip(l[-4:], r[-4:])] == case["last"]
    assert make_golden.digest(l.astype(np.int64), r.astype(np.int64)) == case["lhs_rhs_sha256"]
    assert make_golden.digest(d.view(np.uint32)) == case["dist_bits_sha256"]


@pytest.mark.parametrize("case", _fixtures()["set_cases"], ids=lambda c: c["name"])
def test_set_fixture(matcher, case):
    imgs = make_golden.set_images((case["name"], case["n_images"], case["n_features"], case["seed"]))
    assert make_golden.digest(*imgs) == case["input_sha256"]
    ids = [720 + i for i in range(len(imgs))]
    for i, im in zip(ids, imgs):
        matcher.upload(i, im)
    try:
        got = matcher.match_pairs([(720 + a, 720 + b) for a, b in case["pairs"]], case["ratio"], case["cross"])
    finally:
        for i in ids:
            matcher.drop(i)
    assert make_golden.set_digest(got) == {k: case[k] for k in ("counts", "lhs_rhs_sha256", "dist_bits_sha256")}
