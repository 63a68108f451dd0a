"""sfm_amd: MI355X-native brute-force descriptor matching.

The product is the shared library sfm_amd/lib/libsfm_hip.so (C ABI in
include/sfm_hip.h) and the C++ host layer in sfm_amd/host/.  This Python
package is a thin ctypes binding over the same C ABI, used by the tests and
the benchmark.  There is no CPU fallback: if the library is missing or no
gfx950 device is present, calls raise.
"""
from .binding import Matcher, SfmHipError, library, library_path  # noqa: F401
