"""ctypes binding of libsqz_amd.so -- exactly the entry points include/sqz/*.h declare.

The product path has no CPU fallback: a missing library is a hard error.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SQZ_AMD_LIB: developer override (an instrumented build of the same sources)
LIB_PATH = os.environ.get("SQZ_AMD_LIB") or os.path.join(_HERE, "lib", "libsqz_amd.so")


class Bitstream(C.Structure):
    """struct bitstream of include/sqz/sqz.h (bitstream.h:7-18, fields in the reference's order)."""


WORD_CALLBACK = C.CFUNCTYPE(C.c_int, C.POINTER(Bitstream))
Bitstream._fields_ = [("stream", C.c_void_p), ("data", C.POINTER(C.c_uint8)),
                      ("capacity", C.c_uint64), ("bytes", C.c_uint64), ("read", C.c_uint64),
                      ("b64", C.c_uint64), ("bits", C.c_int32), ("error", C.c_int32),
                      ("output", WORD_CALLBACK), ("input", WORD_CALLBACK)]


class Sqz(C.Structure):
    """struct sqz / sqz_type of include/sqz/sqz.h."""
    _fields_ = [("error", C.c_int32), ("device", C.c_int32), ("tokens", C.c_uint64),
                ("bs", C.POINTER(Bitstream)), ("stream", C.c_void_p), ("reserved", C.c_uint64 * 3)]


KERNEL_NAMES = ["lz77_scan_kernel", "huffman_emit_kernel", "entropy_decode_kernel",
                "index_sort_kernel", "index_match_kernel", "index_parse_kernel",
                "lz_expand_kernel", "rc_encode_kernel", "rc_decode_kernel", "reserved", "reserved", "reserved"]


class BlockStats(C.Structure):
    """sqz_block_stats: the reference's counters per block (huffman.h:29-33, squeeze.h:397-403)."""
    _fields_ = [(n, C.c_uint32) for n in ("lit_updates lit_swaps lit_moves pos_updates pos_swaps pos_moves "
                                          "literal_bytes backref_bytes lit_depth pos_depth tokens reserved").split()] + \
               [("lit_freq", C.c_uint32 * 288), ("pos_freq", C.c_uint32 * 32)]


class RangeCoder(C.Structure):
    """struct sqz_rc_range_coder of include/sqz/sqz_rc.h (struct range_coder, inc/sqz/sqz.h:45-53)."""


RC_WRITE = C.CFUNCTYPE(None, C.POINTER(RangeCoder), C.c_uint8)
RC_READ = C.CFUNCTYPE(C.c_uint8, C.POINTER(RangeCoder))
RangeCoder._fields_ = [("low", C.c_uint64), ("range", C.c_uint64), ("code", C.c_uint64), ("write", RC_WRITE),
                       ("read", RC_READ), ("error", C.c_int32), ("padding", C.c_int32)]


class SqzRc(C.Structure):
    """struct sqz_rc of include/sqz/sqz_rc.h (struct sqz at HEAD, inc/sqz/sqz.h:69-79)."""
    _fields_ = [("rc", RangeCoder), ("that", C.c_void_p), ("reserved", C.c_uint64 * 8)]


class Timing(C.Structure):
    """sqz_hip_timing: per-kernel summed durations (HIP events on the launch stream)."""
    _fields_ = [("ms", C.c_float * 12), ("launches", C.c_uint32 * 12)]


class SqueezeInterface(C.Structure):
    """squeeze_interface vtable (squeeze.h:109-125)."""
    _fields_ = [
        ("alloc", C.CFUNCTYPE(C.POINTER(Sqz), C.c_uint8)),
        ("init_with", C.CFUNCTYPE(C.c_int, C.POINTER(Sqz), C.c_void_p, C.c_size_t, C.c_uint8)),
        ("write_header", C.CFUNCTYPE(None, C.POINTER(Bitstream), C.c_uint64, C.c_uint8)),
        ("compress", C.CFUNCTYPE(None, C.POINTER(Sqz), C.POINTER(Bitstream), C.c_void_p,
                                 C.c_size_t, C.c_uint16)),
        ("read_header", C.CFUNCTYPE(None, C.POINTER(Bitstream), C.POINTER(C.c_uint64),
                                    C.POINTER(C.c_uint8))),
        ("decompress", C.CFUNCTYPE(None, C.POINTER(Sqz), C.POINTER(Bitstream), C.c_void_p,
                                   C.c_size_t)),
        ("free", C.CFUNCTYPE(None, C.POINTER(Sqz))),
    ]


# name -> (restype, argtypes); kept in sync with include/sqz/sqz.h + sqz_workload.h
_u8p, _u32p, _u64p, _i32p, _vp = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.c_void_p)
PROTOTYPES = {
    "sqz_version": (C.c_char_p, []),
    "sqz_hip_device_info": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), _u64p]),
    "sqz_bound": (C.c_uint64, [C.c_uint64]),
    "sqz_file_words": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "sqz_init": (None, [C.POINTER(Sqz)]),
    "sqz_write_header": (None, [C.POINTER(Bitstream), C.c_uint64]),
    "sqz_read_header": (None, [C.POINTER(Bitstream), _u64p]),
    "sqz_write_header_h0": (None, [C.POINTER(Bitstream), C.c_uint64, C.c_uint8]),
    "sqz_read_header_h0": (None, [C.POINTER(Bitstream), _u64p, _u8p]),
    "sqz_compress": (None, [C.POINTER(Sqz), C.POINTER(Bitstream), _vp, C.c_size_t, C.c_uint32]),
    "sqz_decompress": (None, [C.POINTER(Sqz), C.POINTER(Bitstream), _vp, C.c_size_t]),
    "sqz_encode_blocks": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp]),
    "sqz_decode_blocks": (C.c_int, [_vp, _vp, C.c_uint32, _vp, _vp, _vp]),
    "sqz_hip_encode_scratch_bytes": (C.c_uint64, [C.c_uint32, C.c_uint64]),
    "sqz_hip_encode_blocks": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp,
                                        _vp, C.c_uint64, _vp]),
    "sqz_hip_encode_blocks_stats": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp, _vp,
                                              _vp, C.c_uint64, _vp, _vp]),
    "sqz_stats_entropy": (C.c_double, [_vp, C.c_uint32]),
    "sqz_hip_decode_scratch_bytes": (C.c_uint64, [C.c_uint32, C.c_uint64]),
    "sqz_hip_decode_blocks": (C.c_int, [_vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, C.c_uint64, _vp]),
    "sqz_hip_lz77_blocks": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "sqz_hip_lz77_blocks_ex": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp, _vp, C.c_int, _vp,
                                         C.c_uint64, _vp]),
    "sqz_hip_huffman_blocks": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp]),
    "sqz_hip_debug_tree": (C.c_int, [_vp, C.c_uint32, C.c_int, C.c_int, _vp, _vp]),
    "sqz_hip_pack_blocks": (C.c_int, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, C.c_uint64, _vp]),
    "sqz_rc_init": (None, [C.POINTER(SqzRc), _vp, C.c_size_t]),
    "sqz_rc_compress": (None, [C.POINTER(SqzRc), _vp, C.c_size_t, C.c_uint32]),
    "sqz_rc_decompress": (C.c_uint64, [C.POINTER(SqzRc), _vp, C.c_size_t]),
    "sqz_rc_bound": (C.c_uint64, [C.c_uint64]),
    "sqz_hip_rc_encode_blocks": (C.c_int, [_vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp]),
    "sqz_hip_rc_decode_blocks": (C.c_int, [_vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sqz_hip_set_finder": (None, [C.c_int]),
    "sqz_hip_get_finder": (C.c_int, []),
    "sqz_hip_set_timing": (None, [C.c_int]),
    "sqz_hip_get_timing": (C.c_int, [C.POINTER(Timing), C.c_int]),
    "sqz_hip_zipf_blocks": (C.c_int, [_vp, C.c_uint64, C.c_uint64, C.c_uint64, _vp]),
    "sqz_zipf_cdf_table": (C.POINTER(C.c_uint32), []),
}
DATA_SYMBOLS = ["squeeze"]

_lib = None


def lib():
    """The loaded library; raises loudly when it is missing (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m sqz_amd.build` "
                "(hipcc --offload-arch=gfx950). sqz_amd has no CPU fallback.")
        # torch owns device memory and streams in this package, so the library
        # must bind to the SAME HIP runtime instance: load torch's first (its
        # libamdhip64 has the same SONAME the linker recorded for ours).
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def squeeze_vtable():
    return SqueezeInterface.in_dll(lib(), "squeeze")
