import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def kat_cases():
    with open(os.path.join(ROOT, "tests", "golden", "kat.json")) as f:
        return json.load(f)["cases"]


def kat_derived_cases():
    """Cases worked out by hand from the reference's text (NOT outputs of the reference)."""
    with open(os.path.join(ROOT, "tests", "golden", "kat_derived.json")) as f:
        return json.load(f)["cases"]


def kat_descriptors(rows):
    a = np.zeros((len(rows), 128), np.float32)
    for i, r in enumerate(rows):
        a[i, :3] = r
    return a


@pytest.fixture(scope="session", params=["f32", "f16x3"])
def matcher(request):
    """One GPU context per distance engine for the whole session (GPU tests only): every parity
    test runs once with the fp32 matrix path and once with the split-fp16 path."""
    import sfm_amd
    m = sfm_amd.Matcher(0)
    m.set_engine(request.param)
    yield m
    m.close()
