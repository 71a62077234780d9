"""CPU: the C-ABI library loads and exports every symbol include/sqz/*.h declares.
No compute call is made (there is no GPU here); the no-device behaviour is checked."""
import ctypes as C
import errno
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("sqz.h", "sqz_workload.h"):
        text = open(os.path.join(ROOT, "include", "sqz", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))
        for m in re.finditer(r"SQZ_API\s+[^;{]*?\b(\w+)\s*(?:\(|;)", text):
            names.add(m.group(1))
    return names


@pytest.fixture(scope="module")
def native():
    from sqz_amd import build, _native
    build.build_native()
    return _native


def test_header_symbols_are_exported(native):
    syms = declared_symbols()
    assert {"sqz_init", "sqz_compress", "sqz_decompress", "sqz_write_header", "sqz_read_header",
            "sqz_encode_blocks", "sqz_hip_encode_blocks", "sqz_hip_decode_blocks",
            "sqz_hip_lz77_blocks", "sqz_hip_huffman_blocks", "squeeze",
            "sqz_hip_zipf_blocks"} <= syms
    handle = C.CDLL(native.LIB_PATH)
    for s in sorted(syms):
        assert hasattr(handle, s) or C.c_void_p.in_dll(handle, s) is not None, s
    # and the binding table covers exactly the declared functions
    assert set(native.PROTOTYPES) | set(native.DATA_SYMBOLS) == syms


def test_struct_layout(native):
    # struct bitstream / struct sqz as laid out in include/sqz/sqz.h
    assert C.sizeof(native.Bitstream) == 48
    assert C.sizeof(native.Sqz) == 56
    assert native.Bitstream.error.offset == 44


def test_version_and_bound(native):
    L = native.lib()
    assert b"gfx950" in L.sqz_version()
    assert L.sqz_bound(0) % 8 == 0 and L.sqz_bound(262144) >= 262144 + 262144 // 8


def test_no_device_is_loud(native):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import sqz_amd
    with pytest.raises(sqz_amd.SqzError) as ei:
        sqz_amd.compress(b"hello hello hello", win_bits=10)
    assert ei.value.errno == errno.ENODEV          # no CPU fallback behind the ABI
    L = native.lib()
    assert L.sqz_hip_device_info(None, 0, None, None) == errno.ENODEV


def test_header_roundtrip_is_host_only(native):
    # header framing (squeeze.h:255-265) runs on the host and needs no device
    L = native.lib()
    buf = (C.c_uint8 * 64)()
    bs = native.Bitstream(data=C.cast(buf, C.POINTER(C.c_uint8)), capacity=64)
    L.sqz_write_header_h0(C.byref(bs), 35, 10)
    assert bs.error == 0 and bs.bytes == 8 and bs.bits == 8
    assert bytes(buf[:8]).hex() == "c400000000000000"      # golden 'hello' header prefix
    bs2 = native.Bitstream(data=C.cast(buf, C.POINTER(C.c_uint8)), capacity=64)
    L.sqz_write_header_h0(C.byref(bs2), 35, 9)
    assert bs2.error == errno.EINVAL                       # squeeze.h:257-258
    rd = native.Bitstream(data=C.cast(buf, C.POINTER(C.c_uint8)), bytes=16)
    buf[8] = 0x50                                          # win_bits=10, LSB first
    n, wb = C.c_uint64(0), C.c_uint8(0)
    L.sqz_read_header_h0(C.byref(rd), C.byref(n), C.byref(wb))
    assert rd.error == 0 and n.value == 35 and wb.value == 10 and rd.read == 16 and rd.bits == 56
