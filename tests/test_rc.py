"""The reference's HEAD ("R-era") codec, SURVEY.md section 8f-1: adaptive order-0 range coder behind
struct sqz / rc.write / rc.read (include/sqz/sqz_rc.h).

CPU: the oracle restatement (oracle/sqz_rc_oracle.c) against the fixtures the compiled reference
produced (tests/golden/golden_rc.json, which also carry the fingerprints SURVEY.md quotes) and, when
oracle/_ref/libsqz_ref_rc.so is present, against the reference itself on fuzzed inputs and corrupted
streams; the kernels of sqz_amd/csrc/range_coder.hip on the CPU wave emulator; HEAD's own caller
(shl.c:23-68) compiling against the header under SQZ_RC_REFERENCE_NAMES.
GPU (-m gpu): the same through the C ABI on the MI355X -- callbacks and device-resident batches."""
import ctypes as C
import errno
import json
import os
import random
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(O.GOLD, "golden_rc.json")) as fh:
    G = json.load(fh)
_rc_ref = os.path.join(O.ODIR, "_ref", "libsqz_ref_rc.so")
REF = C.CDLL(_rc_ref) if os.path.exists(_rc_ref) else None


def ora_encode(d, cap=None):
    cap = 2 * len(d) + 64 if cap is None else cap
    out = C.create_string_buffer(max(cap, 1))
    n = C.c_uint64()
    e = O.ORACLE.sqzo_rc_encode(d, C.c_uint64(len(d)), out, C.c_uint64(cap), C.byref(n))
    return e, out.raw[:min(n.value, cap)], n.value


def ora_decode(s, cap):
    out = C.create_string_buffer(max(cap, 1))
    n, cons = C.c_uint64(), C.c_uint64()
    e = O.ORACLE.sqzo_rc_decode(s, C.c_uint64(len(s)), out, C.c_uint64(cap), C.byref(n), C.byref(cons))
    return e, out.raw[:n.value], cons.value


# ------------------------------------------------------------------------------------------- CPU
def test_oracle_against_reference_fixtures():
    for v in G["vectors"]:
        d = bytes.fromhex(v["input_hex"])
        e, s, n = ora_encode(d)
        assert e == 0 and s.hex() == v["out_hex"], v["name"]
        e, back, cons = ora_decode(s, len(d))
        assert e == 0 and back == d and cons == len(s), v["name"]
    for c in G["corpus"]:
        d = O.corpus(c["file"])
        assert O.fnv(d) == c["in_fnv"]
        e, s, n = ora_encode(d)
        assert (e, len(s), O.fnv(s)) == (0, c["out_bytes"], c["out_fnv"]), c["file"]
        e, back, cons = ora_decode(s, len(d))
        assert e == 0 and back == d
    z = {x["block"]: x for x in G["zipf"] if x["in_bytes"] == 16384}
    e, s, n = ora_encode(O.zipf_block(5, 16384))
    assert (len(s), O.fnv(s)) == (z[5]["out_bytes"], z[5]["out_fnv"])
    # the figures SURVEY.md section 8f-1 quotes
    by = {c["file"]: c for c in G["corpus"]}
    assert (by["laozi.txt"]["out_bytes"], by["laozi.txt"]["out_fnv"]) == (14735, "99bd04202966bc15")
    assert (by["confucius.txt"]["out_bytes"], by["confucius.txt"]["out_fnv"]) == (47914, "2a8534293b6b47c2")


def test_oracle_errors():
    d = O.corpus("laozi.txt")[:3000]
    e, s, n = ora_encode(d, cap=100)
    assert e == errno.ENOBUFS and n > 100                           # does not fit: the size is still reported
    e, s, n = ora_encode(d)
    assert ora_decode(s, len(d) - 1)[0] == errno.ENOBUFS            # src/sqz.c:806
    assert ora_decode(s, len(d) + 50)[1] == d                       # stops at the end-of-stream symbol


@pytest.mark.skipif(REF is None, reason="oracle/_ref/libsqz_ref_rc.so not built (reference not mounted)")
def test_oracle_against_reference_live():
    rng = random.Random(3)
    for _ in range(40):
        n = rng.randint(0, 3000)
        d = bytes(rng.randrange(rng.choice([2, 7, 256])) for _ in range(n))
        out = C.create_string_buffer(2 * n + 64)
        nb = C.c_uint64()
        assert REF.sqz_ref_rc_compress(d, C.c_uint64(n), C.c_uint32(1 << 12), out, C.c_uint64(len(out)), C.byref(nb)) == 0
        e, s, _ = ora_encode(d)
        assert e == 0 and s == out.raw[:nb.value]
        # corrupted streams: same errno, same bytes, same number of stream bytes asked for
        for _ in range(4):
            bad = bytearray(s)
            if len(bad) > 9:
                bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
            back = C.create_string_buffer(n + 300)
            got, cons = C.c_uint64(), C.c_uint64()
            er = REF.sqz_ref_rc_decompress(bytes(bad), C.c_uint64(len(bad)), back, C.c_uint64(n + 300), C.byref(got), C.byref(cons))
            eo, bo, co = ora_decode(bytes(bad), n + 300)
            assert (er, back.raw[:got.value], cons.value) == (eo, bo, co)


def test_the_kernels_division_is_exact():
    """rc_div (range_coder.hip: a double-precision estimate made low, the exact remainder, a second estimate, two
    steps) against 64-bit integer division: corners and two million random pairs of every magnitude"""
    import test_emu as TE
    E = TE._build("default", "rc")
    E.emu_rc_div_mismatches.restype = C.c_uint64
    rng = np.random.default_rng(5)
    edge = np.array([0, 1, 2, 3, 255, 256, 2**31 - 1, 2**31, 2**32 - 1, 2**32, 2**32 + 1, 2**52, 2**53 - 1, 2**53,
                     2**53 + 1, 2**56, 2**63 - 1, 2**63, 2**63 + 1, 2**64 - 2, 2**64 - 1], dtype=np.uint64)
    a = [np.repeat(edge, len(edge)), ]
    d = [np.tile(edge, len(edge)), ]
    n = 1 << 20
    for shift_a, shift_d in ((0, 0), (0, 32), (0, 45), (8, 40), (0, 56), (20, 20)):
        a.append(rng.integers(0, 2**64, n, dtype=np.uint64) >> np.uint64(shift_a))
        d.append((rng.integers(0, 2**64, n, dtype=np.uint64) >> np.uint64(shift_d)) | np.uint64(1))
    # multiples of the divisor and their neighbours: the quotient changes exactly there
    dd = (rng.integers(0, 2**64, n, dtype=np.uint64) >> np.uint64(40)) | np.uint64(1)
    qq = rng.integers(0, 2**39, n, dtype=np.uint64)
    for off in (0, 1, -1):
        a.append((dd * qq + np.uint64(off & (2**64 - 1))).astype(np.uint64))
        d.append(dd)
    a, d = np.ascontiguousarray(np.concatenate(a)), np.ascontiguousarray(np.concatenate(d))
    bad = C.c_uint64(0)
    wrong = E.emu_rc_div_mismatches(a.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), C.c_uint64(len(a)), C.byref(bad))
    assert wrong == 0, (wrong, int(a[bad.value]), int(d[bad.value]))


def test_kernels_on_the_wave_emulator():
    import test_emu as TE
    TE.VARIANTS.setdefault("default", [])
    E = TE._build("default", "rc")
    rng = random.Random(5)
    cases = [O.corpus("laozi.txt")[:5000], b"", b"a", b"Lorem ipsum dolor sit amet. " * 3, bytes(2000),
             bytes(range(256)) * 4, O.zipf_block(2, 4000), bytes(rng.randrange(256) for _ in range(1500))]
    n = len(cases)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    in_off = np.zeros(n + 1, np.uint64)
    in_off[1:] = np.cumsum([len(c) for c in cases])
    data = np.frombuffer(b"".join(cases) + b"\0" * 8, np.uint8).copy()
    out_off = np.zeros(n + 1, np.uint64)
    out_off[1:] = np.cumsum([2 * len(c) + 64 for c in cases])
    out = np.zeros(int(out_off[-1]), np.uint8)
    ob, err = np.zeros(n, np.uint64), np.zeros(n, np.int32)
    E.emu_rc_encode(p(data), p(in_off), n, p(out), p(out_off), p(ob), p(err))
    want = [ora_encode(c)[1] for c in cases]
    assert err.tolist() == [0] * n
    assert [out[int(out_off[b]):int(out_off[b]) + int(ob[b])].tobytes() for b in range(n)] == want
    # decode: good streams, then damaged ones (errno, output and consumption equal the oracle's)
    streams = want + [bytes(bytearray(w[:len(w) // 2]) + bytearray(b"\x55" * 9)) for w in want[:3]]
    caps = [len(c) for c in cases] + [len(c) + 40 for c in cases[:3]]
    m = len(streams)
    s_off = np.zeros(m + 1, np.uint64)
    s_off[1:] = np.cumsum([len(s) for s in streams])
    sd = np.frombuffer(b"".join(streams) + b"\0" * 8, np.uint8).copy()
    d_off = np.zeros(m + 1, np.uint64)
    d_off[1:] = np.cumsum(caps)
    dec = np.zeros(int(d_off[-1]) + 8, np.uint8)
    db, cons, derr = np.zeros(m, np.uint64), np.zeros(m, np.uint64), np.zeros(m, np.int32)
    E.emu_rc_decode(p(sd), p(s_off), m, p(dec), p(d_off), p(db), p(cons), p(derr))
    for b in range(m):
        eo, bo, co = ora_decode(streams[b], caps[b])
        assert (int(derr[b]), dec[int(d_off[b]):int(d_off[b]) + int(db[b])].tobytes(), int(cons[b])) == (eo, bo, co), b



# ---- the decoder's back-reference branch (src/sqz.c:809-833): streams no encoder of the reference writes ----
def _token_cases():
    """(name, tokens, capacity): token sequences written by tests/rc_token_encoder.py (test-only encoder with
    HEAD's models and arithmetic).  Covers: plain and overlapping copies (dist < size: RLE), the longest size,
    distances through many bit models, a copy reaching before the start (ERANGE, :824), a copy running past
    the caller's buffer (ENOBUFS, :832), sizes outside 2..254 (ERANGE, :810), a literal into a full buffer."""
    import rc_token_encoder as R
    rng = random.Random(11)
    text = [("lit", b) for b in b"the quick brown fox jumps over the lazy dog. "]
    cases = []
    cases.append(("copies", text + [("match", 9, 5, 12), ("match", 5, 2, 1), ("lit", 33), ("match", 254, 1, 0),
                                     ("match", 2, 4, 7), ("lit", 10)], None))
    cases.append(("rle", [("lit", 65), ("lit", 66), ("match", 200, 1, 0), ("match", 17, 1, 0), ("lit", 67)], None))
    long = [("lit", rng.randrange(256)) for _ in range(5000)]
    for _ in range(300):
        bits = rng.randint(1, 12)
        low = rng.randrange(1 << (bits - 1)) if bits > 1 else 0
        long.append(("match", rng.randint(2, 254), bits, low) if rng.random() < 0.5 else ("lit", rng.randrange(256)))
    cases.append(("long", long, None))
    cases.append(("before_start", text[:5] + [("match", 4, 3, 2), ("lit", 1)], None))            # dist 10 > i = 5
    cases.append(("past_buffer", text + [("match", 100, 2, 0)], len(text) + 50))                  # n > capacity
    cases.append(("size_0", text[:9] + [("match", 0, 0, 0), ("lit", 2)], None))
    cases.append(("size_1", text[:9] + [("match", 1, 0, 0), ("lit", 2)], None))
    cases.append(("full_literal", text, len(text) - 1))
    out = []
    for name, toks, cap in cases:
        good = []
        for t in toks:                                   # what a decoder writes before the first refused token
            if t[0] == "match" and (t[1] < 2 or t[1] > 254 or R.distance(t[2], t[3]) > len(R.expand(good))):
                break
            good.append(t)
        out.append((name, R.encode_tokens(toks), cap if cap is not None else len(R.expand(good)) + 40))
    return out


def _ref_decode_dry(stream, cap, dry):
    REF.sqz_ref_rc_decompress_dry.restype = C.c_int
    back = C.create_string_buffer(max(cap, 1))
    got, cons = C.c_uint64(), C.c_uint64()
    e = REF.sqz_ref_rc_decompress_dry(stream, C.c_uint64(len(stream)), back, C.c_uint64(cap), C.byref(got), C.byref(cons),
                                      C.c_int(dry))
    return e, back.raw[:got.value], cons.value


def _ora_decode_dry(stream, cap, dry):
    out = C.create_string_buffer(max(cap, 1))
    n, cons = C.c_uint64(), C.c_uint64()
    e = O.ORACLE.sqzo_rc_decode_dry(stream, C.c_uint64(len(stream)), out, C.c_uint64(cap), C.byref(n), C.byref(cons), C.c_int(dry))
    return e, out.raw[:n.value], cons.value


def _dry_cases():
    """valid and token streams cut short, read from a source that FAILS at its end (errno 5): the byte count
    is what the reference returns when its read callback reports the failure (test.c:112-121)"""
    cases = _token_cases()
    d = O.corpus("laozi.txt")[:3000]
    s = ora_encode(d)[1]
    cuts = [(s[:k], len(d) + 10) for k in (0, 5, 8, 9, 100, 1500, len(s) - 9, len(s) - 1, len(s))]
    cuts += [(st[:len(st) * 2 // 3], cap) for _, st, cap in cases[:3]]
    return cuts


def test_back_reference_branch_on_the_oracle_and_the_reference():
    import rc_token_encoder as R
    for name, stream, cap in _token_cases():
        eo, bo, co = ora_decode(stream, cap)
        if name in ("copies", "rle", "long"):
            assert eo == 0 and co == len(stream), name
        elif name in ("before_start", "size_0", "size_1"):
            assert eo == errno.ERANGE, name
        else:
            assert eo == errno.ENOBUFS, name
        if REF is not None:
            back = C.create_string_buffer(max(cap, 1))
            got, cons = C.c_uint64(), C.c_uint64()
            er = REF.sqz_ref_rc_decompress(stream, C.c_uint64(len(stream)), back, C.c_uint64(cap), C.byref(got), C.byref(cons))
            assert (er, back.raw[:got.value], cons.value) == (eo, bo, co), name
    # the test encoder against itself: literal-only streams are the oracle's encoder's
    d = O.zipf_block(3, 2000)
    assert R.encode_tokens([("lit", b) for b in d]) == ora_encode(d)[1]
    toks = [("lit", 7), ("lit", 8), ("match", 10, 1, 0)]
    assert ora_decode(R.encode_tokens(toks), 12)[1] == R.expand(toks) == bytes([7, 8] * 6)
    for stream, cap in _dry_cases():
        eo, bo, co = _ora_decode_dry(stream, cap, 5)
        if REF is not None:
            assert _ref_decode_dry(stream, cap, 5) == (eo, bo, co)


def test_back_reference_branch_on_the_wave_emulator():
    import test_emu as TE
    TE.VARIANTS.setdefault("default", [])
    E = TE._build("default", "rc")
    p = lambda a: a.ctypes.data_as(C.c_void_p)

    def run(streams, caps, dry):
        m = len(streams)
        s_off = np.zeros(m + 1, np.uint64)
        s_off[1:] = np.cumsum([len(s) for s in streams])
        sd = np.frombuffer(b"".join(streams) + b"\0" * 8, np.uint8).copy()
        d_off = np.zeros(m + 1, np.uint64)
        d_off[1:] = np.cumsum(caps)
        dec = np.zeros(int(d_off[-1]) + 8, np.uint8)
        db, cons, derr = np.zeros(m, np.uint64), np.zeros(m, np.uint64), np.zeros(m, np.int32)
        if dry:
            E.emu_rc_decode_dry(p(sd), p(s_off), m, p(dec), p(d_off), p(db), p(cons), p(derr), dry)
        else:
            E.emu_rc_decode(p(sd), p(s_off), m, p(dec), p(d_off), p(db), p(cons), p(derr))
        return [(int(derr[b]), dec[int(d_off[b]):int(d_off[b]) + int(db[b])].tobytes(), int(cons[b])) for b in range(m)]

    cases = _token_cases()
    got = run([c[1] for c in cases], [c[2] for c in cases], 0)
    for (name, stream, cap), g in zip(cases, got):
        assert g == ora_decode(stream, cap), name
    dry = _dry_cases()
    got = run([c[0] for c in dry], [c[1] for c in dry], 5)
    for k, ((stream, cap), g) in enumerate(zip(dry, got)):
        assert g == _ora_decode_dry(stream, cap, 5), k


def test_heads_caller_compiles_against_the_header(tmp_path):
    """shl.c:13-68's flow (put / get over a static buffer, sqz_init, rc.write / rc.read, sqz_compress,
    sqz_decompress) written against <sqz/sqz_rc.h> with the reference's own names"""
    from sqz_amd import build
    build.build_native()
    src = tmp_path / "shl_like.c"
    src.write_text(r"""
#define SQZ_RC_REFERENCE_NAMES
#include <sqz/sqz_rc.h>
#include <stdio.h>
#include <string.h>
static struct { uint8_t data[1024]; size_t bytes; size_t written; } io;
static void put(struct range_coder* rc, uint8_t b) { (void)rc; io.data[io.written++] = b; }
static uint8_t get(struct range_coder* rc) { (void)rc; return io.data[io.bytes++]; }
int main(void) {
    const char* text = "Lorem ipsum dolor sit amet. Lorem ipsum dolor sit amet. Lorem ipsum dolor sit amet. ";
    size_t input_size = strlen(text);
    static struct sqz compress;
    compress.that = 0;
    sqz_init(&compress, NULL, 0);
    compress.rc.write = put;
    sqz_compress(&compress, text, input_size, 1u << 11);
    if (compress.rc.error != 0) { printf("Compression error: %d\n", compress.rc.error); return compress.rc.error; }
    printf("%d into %d bytes\n", (int)input_size, (int)io.written);
    static char decompressed_data[1024];
    static struct sqz decompress;
    sqz_init(&decompress, NULL, 0);
    decompress.rc.read = get;
    uint64_t decompressed = sqz_decompress(&decompress, decompressed_data, input_size);
    if (decompress.rc.error != 0) { printf("Decompression error: %d\n", decompress.rc.error); return decompress.rc.error; }
    if (decompressed != input_size || memcmp(decompressed_data, text, input_size) != 0) { return 1; }
    for (size_t k = 0; k < io.written; k++) { printf("%02x", io.data[k]); }
    printf("\nDecompression successful.\n");
    return 0;
}
""")
    exe = tmp_path / "shl_like"
    libdir = os.path.join(ROOT, "sqz_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src),
                           "-L" + libdir, "-lsqz_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    import torch
    p = subprocess.run([str(exe)], capture_output=True, text=True)
    if not torch.cuda.is_available():
        assert p.returncode == errno.ENODEV and "Compression error" in p.stdout      # no CPU fallback behind the ABI
        return
    assert p.returncode == 0 and "Decompression successful." in p.stdout, p.stdout + p.stderr
    want = [v for v in G["vectors"] if v["name"] == "lorem3"][0]["out_hex"]
    assert p.stdout.splitlines()[1] == want


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_callbacks_and_fixtures():
    import torch
    assert torch.cuda.is_available()
    from sqz_amd import _native as N
    L = N.lib()

    def gpu_encode(d):
        got = bytearray()

        @N.RC_WRITE
        def put(rc, b):
            got.append(b)

        s = N.SqzRc()
        L.sqz_rc_init(C.byref(s), None, 0)
        s.rc.write = put
        L.sqz_rc_compress(C.byref(s), d, len(d), 1 << 15)
        return s.rc.error, bytes(got)

    def gpu_decode(stream, cap):
        pos = [0]

        @N.RC_READ
        def get(rc):
            v = stream[pos[0]] if pos[0] < len(stream) else 0
            pos[0] += 1
            return v

        s = N.SqzRc()
        L.sqz_rc_init(C.byref(s), None, 0)
        s.rc.read = get
        out = C.create_string_buffer(max(cap, 1))
        n = L.sqz_rc_decompress(C.byref(s), out, cap)
        return s.rc.error, out.raw[:n], pos[0]

    for v in G["vectors"]:
        d = bytes.fromhex(v["input_hex"])
        e, s = gpu_encode(d)
        assert e == 0 and s.hex() == v["out_hex"], v["name"]
        e, back, pulled = gpu_decode(s, len(d))
        assert e == 0 and back == d and len(s) <= pulled <= 2 * len(s) + 64, v["name"]
    for c in G["corpus"][:2]:
        d = O.corpus(c["file"])
        e, s = gpu_encode(d)
        assert (e, len(s), O.fnv(s)) == (0, c["out_bytes"], c["out_fnv"]), c["file"]
        e, back, _ = gpu_decode(s, len(d))
        assert e == 0 and back == d
    # errors: the caller's buffer too small (ENOBUFS, src/sqz.c:806), a damaged stream (the oracle's errno)
    d = O.corpus("laozi.txt")[:4000]
    e, s = gpu_encode(d)
    assert gpu_decode(s, len(d) - 1)[0] == errno.ENOBUFS
    bad = bytearray(s)
    bad[50] ^= 0x10
    eo, bo, _ = ora_decode(bytes(bad), len(d) + 100)
    eg, bg, _ = gpu_decode(bytes(bad), len(d) + 100)
    assert (eg, bg) == (eo, bo)

    # a sink that fails: the error sticks and the writes stop
    calls = []

    @N.RC_WRITE
    def failing(rc, b):
        calls.append(b)
        if len(calls) == 5:
            rc.contents.error = errno.EIO

    s2 = N.SqzRc()
    L.sqz_rc_init(C.byref(s2), None, 0)
    s2.rc.write = failing
    L.sqz_rc_compress(C.byref(s2), d, len(d), 1 << 15)
    assert s2.rc.error == errno.EIO and len(calls) == 5


@pytest.mark.gpu
def test_gpu_batch_device_resident():
    import torch
    from sqz_amd import _native as N
    from sqz_amd import batch
    L = N.lib()
    n, bb = 64, 16384
    d_in = batch.zipf_blocks(n, bb)
    off = batch.uniform_offsets(n, bb)
    cap = int(L.sqz_rc_bound(bb))
    out_off = batch.uniform_offsets(n, cap)
    out = torch.zeros(n * cap, dtype=torch.uint8, device="cuda")
    ob = torch.zeros(n, dtype=torch.int64, device="cuda")
    err = torch.zeros(n, dtype=torch.int32, device="cuda")
    P = lambda t: C.c_void_p(t.data_ptr())
    assert L.sqz_hip_rc_encode_blocks(P(d_in), P(off), n, P(out), P(out_off), P(ob), P(err), None) == 0
    back = torch.zeros_like(d_in)
    db = torch.zeros(n, dtype=torch.int64, device="cuda")
    cons = torch.zeros(n, dtype=torch.int64, device="cuda")
    derr = torch.zeros(n, dtype=torch.int32, device="cuda")
    assert L.sqz_hip_rc_decode_blocks(P(out), P(out_off), n, P(back), P(off), P(db), P(cons), P(derr), None) == 0
    torch.cuda.synchronize()
    assert err.tolist() == [0] * n and derr.tolist() == [0] * n
    assert torch.equal(back, d_in) and db.tolist() == [bb] * n
    assert torch.equal(cons, ob)                                    # the decoder asks for exactly the stream
    h = out.cpu().numpy()
    z = {x["block"]: x for x in G["zipf"] if x["in_bytes"] == bb}
    for b in (5, 17, 63):
        got = h[b * cap:b * cap + int(ob[b])].tobytes()
        assert got == ora_encode(O.zipf_block(b, bb))[1]
        if b in z:
            assert (len(got), O.fnv(got)) == (z[b]["out_bytes"], z[b]["out_fnv"])


@pytest.mark.gpu
def test_gpu_back_reference_branch_and_dry_sources():
    """the decoder's back-reference branch (src/sqz.c:809-833) on the device, on streams written by the test-only
    token encoder: errno, bytes and consumption equal the oracle's (which equals the compiled reference's, CPU
    tests); then through rc.read with a source that fails at its end: the byte count sqz_rc_decompress returns"""
    import torch
    from sqz_amd import _native as N
    from sqz_amd import batch
    L = N.lib()
    cases = _token_cases()
    m = len(cases)
    streams, caps = [c[1] for c in cases], [c[2] for c in cases]
    s_off = np.zeros(m + 1, np.int64)
    s_off[1:] = np.cumsum([len(s) for s in streams])
    d_off = np.zeros(m + 1, np.int64)
    d_off[1:] = np.cumsum(caps)
    dev = "cuda"
    sd = torch.tensor(np.frombuffer(b"".join(streams) + b"\0" * 8, np.uint8).copy(), device=dev)
    dec = torch.zeros(int(d_off[-1]) + 8, dtype=torch.uint8, device=dev)
    t_s, t_d = torch.tensor(s_off, device=dev), torch.tensor(d_off, device=dev)
    db = torch.zeros(m, dtype=torch.int64, device=dev)
    cons = torch.zeros(m, dtype=torch.int64, device=dev)
    derr = torch.zeros(m, dtype=torch.int32, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    assert L.sqz_hip_rc_decode_blocks(P(sd), P(t_s), m, P(dec), P(t_d), P(db), P(cons), P(derr), None) == 0
    torch.cuda.synchronize()
    h = dec.cpu().numpy()
    for b, (name, stream, cap) in enumerate(cases):
        got = (int(derr[b]), h[int(d_off[b]):int(d_off[b]) + int(db[b])].tobytes(), int(cons[b]))
        assert got == ora_decode(stream, cap), name

    def gpu_decode_failing_source(stream, cap):
        pos = [0]

        @N.RC_READ
        def get(rc):
            if pos[0] >= len(stream):
                rc.contents.error = 5                 # the source failed (EIO): test.c:112-121
                return 0
            v = stream[pos[0]]
            pos[0] += 1
            return v

        s = N.SqzRc()
        L.sqz_rc_init(C.byref(s), None, 0)
        s.rc.read = get
        out = C.create_string_buffer(max(cap, 1))
        n = L.sqz_rc_decompress(C.byref(s), out, cap)
        return s.rc.error, out.raw[:n]

    for k, (stream, cap) in enumerate(_dry_cases()):
        eo, bo, _ = _ora_decode_dry(stream, cap, 5)
        assert gpu_decode_failing_source(stream, cap) == (eo, bo), k
