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
    for h in ("sqz.h", "sqz_workload.h", "sqz_rc.h"):
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
    # the reference's field order (bitstream.h:7-18): stream first, the two callbacks last
    assert C.sizeof(native.Bitstream) == 72
    assert [f[0] for f in native.Bitstream._fields_] == [
        "stream", "data", "capacity", "bytes", "read", "b64", "bits", "error", "output", "input"]
    assert native.Bitstream.stream.offset == 0 and native.Bitstream.data.offset == 8
    assert native.Bitstream.error.offset == 52 and native.Bitstream.output.offset == 56
    # squeeze_type: error first (squeeze.h:82), `bs` present (squeeze.h:89)
    assert C.sizeof(native.Sqz) == 56
    assert native.Sqz.error.offset == 0 and native.Sqz.bs.offset == 16


def test_struct_layout_seen_by_a_c_compiler(native, tmp_path):
    """the same numbers from gcc's view of include/sqz/sqz.h, and the reference's designated
    initialisers (attic/map_experiment/test.c:53,110; `s->bs->bytes` :84) compile against it"""
    import subprocess
    src = tmp_path / "layout.c"
    src.write_text(r"""
#include <stddef.h>
#include <stdio.h>
#include <sqz/sqz.h>
static int write_file(bitstream* bs) { (void)bs; return 0; }
static int read_file(bitstream* bs) { (void)bs; return 0; }
int main(void) {
    bitstream w = { .stream = stdout, .output = write_file };
    bitstream r = { .stream = stdin, .input = read_file };
    squeeze_type s = { .bs = &w };
    printf("%zu %zu %zu %zu %zu %zu %zu %llu\n", sizeof(bitstream), offsetof(bitstream, data),
           offsetof(bitstream, error), offsetof(bitstream, output), offsetof(bitstream, input),
           sizeof(squeeze_type), offsetof(squeeze_type, bs), (unsigned long long)s.bs->bytes);
    return (w.output == write_file && r.input == read_file) ? 0 : 1;
}
""")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split()
    assert out == ["72", "8", "52", "56", "64", "56", "16", "0"]


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


def test_header_through_callbacks_is_host_only(native):
    """callback-mode streams (bitstream.h:44-48,81-85): the header's first full word goes to
    .output with the reference's bookkeeping; .input feeds the reader one word per call"""
    L = native.lib()
    words = []

    @native.WORD_CALLBACK
    def out_cb(bs):
        words.append(bs.contents.b64)
        return 0

    w = native.Bitstream(output=out_cb)
    L.sqz_write_header_h0(C.byref(w), 35, 10)
    assert w.error == 0 and words == [0xC400000000000000] and w.bytes == 8 and w.bits == 8
    assert w.b64 == 0x50 >> 0 and w.bits == 8            # win_bits=10 LSB first = 0101 0000

    @native.WORD_CALLBACK
    def failing(bs):
        return errno.ENOSPC

    w2 = native.Bitstream(output=failing)
    L.sqz_write_header_h0(C.byref(w2), 35, 10)
    assert w2.error == errno.ENOSPC and w2.bytes == 0     # bitstream.h:46-47

    feed = [0xC400000000000000, 0x5000000000000000]
    calls = []

    @native.WORD_CALLBACK
    def in_cb(bs):
        if not feed:
            return errno.EIO
        bs.contents.b64 = feed.pop(0)
        calls.append(1)
        return 0

    r = native.Bitstream(input=in_cb)
    n, wb = C.c_uint64(0), C.c_uint8(0)
    L.sqz_read_header_h0(C.byref(r), C.byref(n), C.byref(wb))
    assert r.error == 0 and (n.value, wb.value) == (35, 10) and r.read == 16 and len(calls) == 2
    r2 = native.Bitstream(input=in_cb)                    # nothing left: the callback's error sticks
    L.sqz_read_header_h0(C.byref(r2), C.byref(n), C.byref(wb))
    assert r2.error == errno.EIO
    none = native.Bitstream()                             # neither memory nor callbacks
    L.sqz_write_header_h0(C.byref(none), 1 << 40, 10)
    assert none.error == errno.EINVAL
