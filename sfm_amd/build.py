"""Compile the gfx950 shared library (and the C++ host programs) in-tree.

hipcc cross-compiles without a GPU.  Outputs land in sfm_amd/lib/ and are
git-ignored; they travel to the GPU machine with the source tree.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sfm_amd", "csrc")
HOST = os.path.join(ROOT, "sfm_amd", "host")
LIBDIR = os.path.join(ROOT, "sfm_amd", "lib")
LIB = os.path.join(LIBDIR, "libsfm_hip.so")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# -amdgpu-mfma-vgpr-form: MFMA results land in the vector half of the register file, where the
# epilogue's vector instructions read them without v_accvgpr_read copies.
# -amdgpu-kernarg-preload-count: a kernel's first arguments arrive in scalar registers with the wave
# instead of through a load from the argument buffer: the distance kernel reaches its block table
# one trip to memory earlier (about 500 cycles per workgroup, 0.5 us per single-pair step).
HIP_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
             "-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-amdgpu-kernarg-preload-count=2"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build step failed: " + " ".join(cmd))
    return r.stdout


def build_library(force=False):
    """libsfm_hip.so: HIP kernels + C ABI (include/sfm_hip.h)."""
    os.makedirs(LIBDIR, exist_ok=True)
    shared = [os.path.join(CSRC, "ctx.hpp"), os.path.join(ROOT, "include", "sfm_hip.h")]
    units = {"sfm_hip_capi.hip": ["bf_kernels.hpp"], "sfm_hip_cascade.hip": ["cascade_kernels.hpp"],
             "sfm_hip_geometry.hip": ["geometry_kernels.hpp"], "sfm_hip_replicate.hip": []}
    objs = []
    for unit, headers in units.items():
        src = os.path.join(CSRC, unit)
        obj = os.path.join(LIBDIR, unit.replace(".hip", ".o"))
        if force or _newer(obj, [src] + [os.path.join(CSRC, h) for h in headers] + shared):
            _run([HIPCC] + HIP_FLAGS + ["-c", "-o", obj, src])
        objs.append(obj)
    for unit in ("sfm_hip_partition.cc",):   # host-only parts of the C ABI
        src = os.path.join(CSRC, unit)
        obj = os.path.join(LIBDIR, unit.replace(".cc", ".o"))
        if force or _newer(obj, [src, shared[1]]):
            _run(["g++", "-O2", "-std=c++17", "-fPIC", "-c", "-o", obj, src])
        objs.append(obj)
    if force or _newer(LIB, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


def build_host(force=False):
    """C++ host layer: the matcher class behind the reference's interface, the
    driver program and the host-side test programs.  Returns {name: path}."""
    out = {}
    if not os.path.isdir(HOST):
        return out
    build_library()
    common = [os.path.join(HOST, f) for f in sorted(os.listdir(HOST)) if f.endswith((".hpp", ".cc"))]
    inc = ["-I", HOST, "-I", os.path.join(ROOT, "include")]
    link = ["-L", LIBDIR, "-lsfm_hip", "-Wl,-rpath,$ORIGIN", "-pthread"]
    lib_srcs = [os.path.join(HOST, f) for f in ("hip_brute_force_matcher.cc", "hip_cascade_hash_matcher.cc", "hip_geometry_filter.cc", "exhaustive_pair_builder.cc", "sfm_data_io.cc", "sfm_data_json.cc")
                if os.path.exists(os.path.join(HOST, f))]
    for name in ("feature_match_main", "match_set_bench", "geometry_filter_main"):
        src = os.path.join(HOST, name + ".cc")
        if not os.path.exists(src):
            continue
        exe = os.path.join(LIBDIR, name)
        if force or _newer(exe, common + [LIB]):
            _run(["g++", "-O2", "-std=c++17"] + inc + ["-o", exe, src] + lib_srcs + link)
        out[name] = exe
    return out


def build_all(force=False):
    build_library(force)
    return build_host(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print(LIB)
