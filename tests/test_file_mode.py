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
