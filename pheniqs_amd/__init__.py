"""pheniqs_amd — MI355X (gfx950) implementation of Pheniqs' barcode classification path.

The product is the C ABI library `libpheniqs_amd.so` (include/pheniqs_amd.h): C++ host code driving
hand written HIP kernels.  This package only carries the ctypes binding used by tests and bench.py,
the job specification helpers, and the synthetic read generator.  There is no CPU fallback: without
the built library every call fails loudly.
"""
from .capi import DecoderSet, PheniqsError, RESULT_DTYPE, OPTION_DTYPE, library_path, load_library  # noqa: F401
from .build import build  # noqa: F401
