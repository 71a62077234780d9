"""File-mode bit streams (attic/map_experiment/test.c:39-42,98-101).

The reference's harness does not write the memory-mode stream to its files: the bit stream
hands every 64-bit word to a callback that fwrite()s `b64` in host byte order.  The golden
images tests/golden/*.file were written by the COMPILED REFERENCE through exactly that
callback path (oracle/ref_wrap.c: sqz_ref_compress_file, oracle/gen_golden.py), so the format
is pinned by the reference itself.  The oracle is the checker here, never the product."""
import ctypes as C
import json
import os

import pytest

from oracle_lib import GOLD, CORPUS, REF

with open(os.path.join(GOLD, "golden.json")) as fh:
    FILE_MODE = json.load(fh)["file_mode"]


def _read(name, folder=GOLD):
    with open(os.path.join(folder, name), "rb") as fh:
        return fh.read()


def _lib():
    from sqz_amd import _native as N
    return N.lib()


@pytest.mark.parametrize("v", FILE_MODE, ids=lambda v: f"w{v['win_bits']}")
def test_file_words_matches_reference_file_image(v):
    """host code only: the C ABI helper turns the golden stream into the golden file image"""
    stream, image = _read(v["stream"]), _read(v["image"])
    assert len(stream) == len(image) == v["bytes"] and stream != image
    out = C.create_string_buffer(len(stream))
    assert _lib().sqz_file_words(stream, len(stream), out) == 0
    assert out.raw == image
    back = C.create_string_buffer(len(image))                       # its own inverse
    assert _lib().sqz_file_words(image, len(image), back) == 0
    assert back.raw == stream
    buf = C.create_string_buffer(stream, len(stream))               # in place
    assert _lib().sqz_file_words(buf, len(stream), buf) == 0
    assert buf.raw == image


def test_file_words_rejects_ragged_and_null():
    import errno
    L = _lib()
    out = C.create_string_buffer(16)
    assert L.sqz_file_words(b"\0" * 12, 12, out) == errno.EINVAL     # streams are whole words
    assert L.sqz_file_words(None, 8, out) == errno.EINVAL
    assert L.sqz_file_words(b"\0" * 8, 8, None) == errno.EINVAL
    assert L.sqz_file_words(None, 0, None) == 0                      # an empty stream is fine


@pytest.mark.skipif(REF is None, reason="oracle/_ref not built (reference not mounted)")
def test_reference_file_mode_live(tmp_path):
    """the reference's own .output / .input callbacks, run here: image == helper(stream)"""
    REF.sqz_ref_compress_file.restype = C.c_int64
    REF.sqz_ref_compress_file.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_char_p]
    REF.sqz_ref_decompress_file.restype = C.c_int
    REF.sqz_ref_decompress_file.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64,
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    data = _read("confucius.txt", CORPUS)
    path = str(tmp_path / "confucius.w12.file").encode()
    n = REF.sqz_ref_compress_file(data, len(data), 12, path)
    assert n > 0
    image = open(path, "rb").read()
    stream = _read("confucius.txt.w12.sqz")
    assert len(image) == n == len(stream)
    out = C.create_string_buffer(len(stream))
    assert _lib().sqz_file_words(stream, len(stream), out) == 0
    assert out.raw == image
    back = C.create_string_buffer(len(data))
    nb, wb = C.c_uint64(0), C.c_int(0)
    assert REF.sqz_ref_decompress_file(path, back, len(data), C.byref(nb), C.byref(wb)) == 0
    assert (nb.value, wb.value) == (len(data), 12) and back.raw == data


@pytest.mark.gpu
@pytest.mark.parametrize("v", FILE_MODE, ids=lambda v: f"w{v['win_bits']}")
def test_device_writes_and_reads_reference_files(v, tmp_path):
    """HIP path -> file image identical to the reference harness's file, and back"""
    import torch
    assert torch.cuda.is_available()
    import sqz_amd
    data = _read(v["file"], CORPUS)
    stream = sqz_amd.compress(data, win_bits=v["win_bits"], header=True)
    image = sqz_amd.file_words(stream)
    assert image == _read(v["image"])
    path = tmp_path / v["image"]
    path.write_bytes(image)
    assert sqz_amd.decompress(sqz_amd.file_words(path.read_bytes()), header=True) == data
    # a file written by the reference itself decodes on the device
    assert sqz_amd.decompress(sqz_amd.file_words(_read(v["image"])), header=True) == data


@pytest.mark.gpu
@pytest.mark.parametrize("v", FILE_MODE, ids=lambda v: f"w{v['win_bits']}")
def test_callback_mode_streams(v):
    """`.output` / `.input` bit streams (bitstream.h:16-17,44-48,81-85) through the shim: the
    words handed to .output are the reference file's words, one call per word with the
    reference's bookkeeping; a reader fed through .input decodes them, consumes what the
    reference's reader would and never more than twice that from the source"""
    import struct
    from sqz_amd import _native as N
    L = N.lib()
    data = _read(v["file"], CORPUS)
    image = _read(v["image"])                             # host-order words, as fwrite(&b64) left them
    want = list(struct.unpack(f"={len(image) // 8}Q", image))
    got, seen_bytes = [], []

    @N.WORD_CALLBACK
    def out_cb(bs):
        got.append(bs.contents.b64)
        seen_bytes.append(bs.contents.bytes)
        return 0

    w = N.Bitstream(output=out_cb)
    L.sqz_write_header_h0(C.byref(w), len(data), v["win_bits"])
    s = N.Sqz()
    L.sqz_init(C.byref(s))
    L.sqz_compress(C.byref(s), C.byref(w), data, len(data), 1 << v["win_bits"])
    assert s.error == 0 and w.error == 0
    assert got == want
    assert seen_bytes == [8 * k for k in range(len(want))]        # bytes grows after each call
    assert w.bytes == len(image) and s.bs.contents.bytes == len(image)   # attic test.c:84
    assert (w.bits, w.b64) == (0, 0)

    feed = list(want) + [0xDEADBEEFDEADBEEF] * (2 * len(want))   # junk behind the stream
    pulled = []

    @N.WORD_CALLBACK
    def in_cb(bs):
        bs.contents.b64 = feed[len(pulled)]
        pulled.append(1)
        return 0

    r = N.Bitstream(input=in_cb)
    n, wb = C.c_uint64(0), C.c_uint8(0)
    L.sqz_read_header_h0(C.byref(r), C.byref(n), C.byref(wb))
    assert (r.error, n.value, wb.value) == (0, len(data), v["win_bits"])
    back = C.create_string_buffer(len(data))
    d = N.Sqz()
    L.sqz_init(C.byref(d))
    L.sqz_decompress(C.byref(d), C.byref(r), back, len(data))
    assert d.error == 0 and back.raw == data
    assert r.read == len(image)                           # what the reference's reader fetches
    assert len(want) <= len(pulled) <= max(6, 2 * len(want))   # include/sqz/sqz.h: the over-read bound

    # a source that ends too early: the callback's error is the stream's error (bitstream.h:83)
    short = want[:len(want) // 2]
    count = [0]

    @N.WORD_CALLBACK
    def dry_cb(bs):
        if count[0] >= len(short):
            return 5                                      # EIO
        bs.contents.b64 = short[count[0]]
        count[0] += 1
        return 0

    r2 = N.Bitstream(input=dry_cb)
    L.sqz_read_header_h0(C.byref(r2), C.byref(n), C.byref(wb))
    d2 = N.Sqz()
    L.sqz_init(C.byref(d2))
    L.sqz_decompress(C.byref(d2), C.byref(r2), back, len(data))
    assert d2.error == 5 and r2.error == 5

    # a sink that fails at the third word: two words accepted, the error sticks (bitstream.h:46-47)
    acc = []

    @N.WORD_CALLBACK
    def full_cb(bs):
        if len(acc) == 2:
            return 28                                     # ENOSPC
        acc.append(bs.contents.b64)
        return 0

    w3 = N.Bitstream(output=full_cb)
    L.sqz_write_header_h0(C.byref(w3), len(data), v["win_bits"])
    s3 = N.Sqz()
    L.sqz_init(C.byref(s3))
    L.sqz_compress(C.byref(s3), C.byref(w3), data, len(data), 1 << v["win_bits"])
    assert (s3.error, w3.error, w3.bytes) == (28, 28, 16) and acc == want[:2]


@pytest.mark.gpu
def test_callback_reader_overread_is_bounded_by_the_stream_not_the_output():
    """a stream that compresses far better than 8:1 (256 KB of zeros -> about 1.3 KB): the words pulled
    through `.input` stay within max(6, 2 x the stream's words), however large the OUTPUT is"""
    import struct
    import sqz_amd
    from sqz_amd import _native as N
    L = N.lib()
    data = bytes(262144)
    stream = sqz_amd.compress(data, win_bits=15, header=True)
    words = list(struct.unpack(f">{len(stream) // 8}Q", stream))      # memory mode: most significant byte first
    assert len(words) < 400
    feed = words + [0x0123456789ABCDEF] * 4096
    pulled = []

    @N.WORD_CALLBACK
    def in_cb(bs):
        bs.contents.b64 = feed[len(pulled)]
        pulled.append(1)
        return 0

    r = N.Bitstream(input=in_cb)
    n, wb = C.c_uint64(0), C.c_uint8(0)
    L.sqz_read_header_h0(C.byref(r), C.byref(n), C.byref(wb))
    assert (r.error, n.value, wb.value) == (0, len(data), 15)
    back = C.create_string_buffer(len(data))
    d = N.Sqz()
    L.sqz_init(C.byref(d))
    L.sqz_decompress(C.byref(d), C.byref(r), back, len(data))
    assert d.error == 0 and back.raw == data
    assert r.read == len(stream)
    assert len(words) <= len(pulled) <= max(6, 2 * len(words))
