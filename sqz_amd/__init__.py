"""sqz_amd -- MI355X-native LZ77 + adaptive-Huffman encode/decode path of sqz.

The product is libsqz_amd.so (hand-written HIP for gfx950 behind the C ABI of
include/sqz/sqz.h).  This package is the thin host side: ctypes binding,
a mirror of the reference's codec interface, the device-resident batch path and
the multi-GPU block sharding.
"""
from .codec import (SqzError, bound, compress, decompress, device_info,  # noqa: F401
                    file_words, MIN_WIN_BITS, MAX_WIN_BITS)

__all__ = ["SqzError", "bound", "compress", "decompress", "device_info", "file_words",
           "MIN_WIN_BITS", "MAX_WIN_BITS"]
