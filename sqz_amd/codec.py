"""Host-side mirror of the reference's codec interface, over the C ABI.

Same names and argument meaning as the reference:
  H1 spelling  shl/README.md:31-62   sqz_init / sqz_write_header / sqz_compress /
                                     sqz_read_header / sqz_decompress, s.error
  H0 spelling  squeeze.h:109-125     squeeze.alloc / write_header / compress /
                                     read_header / decompress / free
Every call goes through libsqz_amd.so (HIP kernels); nothing is computed here.
"""
import ctypes as C
import errno

from . import _native as N

MIN_WIN_BITS, MAX_WIN_BITS = 10, 15


class SqzError(OSError):
    pass


def _raise(code, what):
    if code != 0:
        raise SqzError(code, f"{what}: {errno.errorcode.get(code, code)}")


def bound(nbytes: int) -> int:
    return int(N.lib().sqz_bound(nbytes))


def file_words(stream) -> bytes:
    """Memory-mode stream <-> the byte image of a file written by the reference's file mode
    (attic test.c:39-42: fwrite(&b64, 8, 1, f) in host byte order).  Its own inverse."""
    stream = bytes(stream)
    out = C.create_string_buffer(len(stream))
    rc = N.lib().sqz_file_words(stream, len(stream), out)
    if rc != 0:
        _raise(rc, "sqz_file_words")
    return out.raw


def compress(data, win_bits: int = 15, header: bool = True, capacity: int = None,
             window: int = None) -> bytes:
    """attic test.c:44-96 in memory: [write_header] + compress -> bytes.

    `header=True` writes the pinned 72-bit H0 header (squeeze.h:255-265)."""
    L = N.lib()
    data = bytes(data)
    cap = bound(len(data)) + 16 if capacity is None else capacity
    out = (C.c_uint8 * max(cap, 1))()
    bs = N.Bitstream(data=C.cast(out, C.POINTER(C.c_uint8)), capacity=cap)
    if header:
        L.sqz_write_header_h0(C.byref(bs), len(data), win_bits)
        _raise(bs.error, "sqz_write_header")
    s = N.Sqz()
    L.sqz_init(C.byref(s))
    src = C.create_string_buffer(data, len(data)) if data else None
    L.sqz_compress(C.byref(s), C.byref(bs), src, len(data),
                   window if window is not None else (1 << win_bits))
    _raise(s.error, "sqz_compress")
    return bytes(bytearray(out)[:bs.bytes])


def decompress(comp, header: bool = True, nbytes: int = None) -> bytes:
    """attic test.c:103-162 in memory: [read_header] + decompress -> bytes."""
    L = N.lib()
    comp = bytes(comp)
    buf = (C.c_uint8 * max(len(comp), 1)).from_buffer_copy(comp or b"\0")
    bs = N.Bitstream(data=C.cast(buf, C.POINTER(C.c_uint8)), bytes=len(comp))
    if header:
        n = C.c_uint64(0)
        wb = C.c_uint8(0)
        L.sqz_read_header_h0(C.byref(bs), C.byref(n), C.byref(wb))
        _raise(bs.error, "sqz_read_header")
        nbytes = n.value
    elif nbytes is None:
        raise ValueError("nbytes is required for payload-only streams")
    out = (C.c_uint8 * max(nbytes, 1))()
    s = N.Sqz()
    L.sqz_init(C.byref(s))
    L.sqz_decompress(C.byref(s), C.byref(bs), out, nbytes)
    _raise(s.error, "sqz_decompress")
    return bytes(bytearray(out)[:nbytes])


def device_info():
    L = N.lib()
    name = C.create_string_buffer(256)
    cus = C.c_int(0)
    lds = C.c_uint64(0)
    _raise(L.sqz_hip_device_info(name, 256, C.byref(cus), C.byref(lds)), "sqz_hip_device_info")
    return {"name": name.value.decode(), "compute_units": cus.value, "lds_bytes_per_cu": lds.value}
