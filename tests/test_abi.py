"""CPU-only: the shared library exists, loads, and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "sfm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sfm_hip_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    from sfm_amd.binding import EXPORTS
    assert sorted(EXPORTS) == header_symbols()


def test_library_exports_every_declared_symbol():
    import sfm_amd
    lib = sfm_amd.library()
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_no_silent_cpu_path_without_a_device():
    import sfm_amd
    lib = sfm_amd.library()
    n = ctypes.c_int(0)
    lib.sfm_hip_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(sfm_amd.SfmHipError, match="no HIP device"):
        sfm_amd.Matcher(0)


def test_null_arguments_are_rejected_not_dereferenced():
    import sfm_amd
    lib = sfm_amd.library()
    assert lib.sfm_hip_ctx_create(0, None) != 0
    assert lib.sfm_hip_sync(None) != 0
    assert lib.sfm_hip_upload_image(None, 0, None, 0) != 0
    assert lib.sfm_hip_match_pairs(None, None, 0, 0.8, 1, None, None, None, None, 0) != 0
    assert lib.sfm_hip_device_count(None) != 0
    assert lib.sfm_hip_set_engine(None, 0) != 0
    assert b"null" in lib.sfm_hip_last_error(None)
    # the cascade-hash and geometry entry points
    assert lib.sfm_hip_cascade_set_projection(None, None, None) != 0
    assert lib.sfm_hip_cascade_column_sums(None, None, 0, None) != 0
    assert lib.sfm_hip_cascade_hash_images(None, None, 0, None) != 0
    assert lib.sfm_hip_cascade_get_hashes(None, 0, None, None) != 0
    assert lib.sfm_hip_cascade_match_pairs(None, None, 0, 1, None, None, None, None, None, 0) != 0
    assert lib.sfm_hip_cascade_prepare(None, None, 0, None) != 0
    assert lib.sfm_hip_cascade_get_stats(None, None) != 0
    assert lib.sfm_hip_cascade_reference_projection(None, None) != 0
    assert lib.sfm_hip_cascade_image_order(None, 1, None, None) != 0
    assert lib.sfm_hip_geometry_verify_pairs(None, 0, None, None, None, 0, 0, None, None, None, None) != 0


def test_product_code_never_touches_the_oracle():
    """The CPU oracle is test infrastructure: nothing under sfm_amd/ (Python, C++, HIP, build
    script) may import, include, link or name it, and bench.py may use it only in its CPU-baseline leg."""
    import os
    from conftest import ROOT
    offenders = []
    for base, _, files in os.walk(os.path.join(ROOT, "sfm_amd")):
        if "__pycache__" in base or base.endswith(os.sep + "lib"):
            continue
        for f in files:
            if f.endswith((".py", ".hpp", ".h", ".cc", ".hip", ".cpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                if "oracle" in text.lower():
                    offenders.append(os.path.join(base, f))
    assert offenders == []
    bench = open(os.path.join(ROOT, "bench.py")).read()
    uses = [i for i, line in enumerate(bench.split("\n")) if "oracle" in line and "import" in line]
    body = bench[bench.index("def cpu_baseline"):bench.index("def replication_check")]
    # the three CPU-baseline legs: brute force, cascade hash, geometric verification
    assert len(uses) == 3 and "from oracle import oracle" in body and "from oracle import cascade" in body \
        and "from oracle import geometry" in body
