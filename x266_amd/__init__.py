"""x266_amd -- thin Python binding (ctypes) over libx266hip.so, the MI355X
implementation of x266's DCT32 / SATD hot path.

The product is the C-ABI shared library (include/x266hip.h); this module only
exists so that Python hosts (tests, bench.py) can call the same entry points a
C/C++ host would.  It contains no arithmetic and no fallback: if the library
or a gfx950 device is missing, calls raise.
"""
from ._lib import (X266Error, Codec, lib_path, load_library, build_library,  # noqa: F401
                   pack_diff_rows, pack_dct_word)

__all__ = ["X266Error", "Codec", "lib_path", "load_library", "build_library",
           "pack_diff_rows", "pack_dct_word"]
