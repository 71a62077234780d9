"""CPU: the oracle (restatement) against the committed golden vectors, which were
generated from the reference itself (oracle/gen_golden.py), and -- when the
reference build oracle/_ref exists (build container only) -- against it directly."""
import os
import random

import numpy as np
import pytest

import oracle_lib as O

G = O.golden()


@pytest.mark.parametrize("v", G["vectors"], ids=lambda v: f"{v['name']}-w{v['win_bits']}-h{int(v['header'])}")
def test_small_vectors(v):
    data = bytes.fromhex(v["input_hex"])
    out = O.encode(data, v["win_bits"], header=v["header"])
    assert out.hex() == v["out_hex"]
    assert len(out) % 8 == 0
    e, back, wb = O.decode(out, header=v["header"], nbytes=len(data))
    assert e == 0 and back == data
    if v["header"]:
        assert wb == v["win_bits"]


def test_survey_known_answers():
    # SURVEY.md section 8c table, hand-verified there
    assert O.encode(b"a", 10, header=False).hex() == "c300000000000000"
    assert O.encode(b"aaaa", 10, header=False).hex() == "c330180000000000"
    assert O.encode(b"Hello World Hello.World Hello World", 10, header=True).hex() == \
        "c400000000000000508934c9b1fd9047a809c39305c0d9ee8e4279062e000000"


@pytest.mark.parametrize("c", [c for c in G["corpus"] if c["in_bytes"] < 100000],
                         ids=lambda c: f"{c['file']}-w{c['win_bits']}")
def test_corpus_small_files(c):
    data = O.corpus(c["file"])
    assert len(data) == c["in_bytes"] and O.fnv(data) == c["in_fnv"]
    out = O.encode(data, c["win_bits"], header=True)
    assert len(out) == c["out_bytes"] and O.fnv(out) == c["out_fnv"]
    with open(os.path.join(O.GOLD, f"{c['file']}.w{c['win_bits']}.sqz"), "rb") as fh:
        assert out == fh.read()
    e, back, wb = O.decode(out, header=True)
    assert e == 0 and back == data and wb == c["win_bits"]


@pytest.mark.parametrize("c", [c for c in G["corpus"] if c["in_bytes"] >= 100000 and c["win_bits"] == 12
                               and c["file"] in ("arm64.elf",)],
                         ids=lambda c: f"{c['file']}-w{c['win_bits']}")
def test_corpus_large_file_fingerprint(c):
    # one large file at the 4 KB window keeps the CPU suite short (~2 s)
    data = O.corpus(c["file"])
    out = O.encode(data, c["win_bits"], header=True)
    assert len(out) == c["out_bytes"] and O.fnv(out) == c["out_fnv"]


@pytest.mark.parametrize("z", [z for z in G["zipf"] if z["in_bytes"] <= 40000],
                         ids=lambda z: f"blk{z['block']}-{z['in_bytes']}-w{z['win_bits']}")
def test_zipf_blocks(z):
    data = O.zipf_block(z["block"], z["in_bytes"])
    assert O.fnv(data) == z["in_fnv"]
    out = O.encode(data, z["win_bits"], header=False)
    assert len(out) == z["out_bytes"] and O.fnv(out) == z["out_fnv"]


def test_zipf_generator_pins():
    cdf = O.zipf_cdf()
    assert cdf[0] == 701294150 and cdf[1] == 1051941225 and cdf[255] == 0xFFFFFFFF
    big = [z for z in G["zipf"] if z["in_bytes"] == 262144 and z["block"] == 0][0]
    assert big["in_fnv"] == "99208ba81eda50a7"
    assert O.fnv(O.zipf_block(0, 262144)) == big["in_fnv"]


def test_tree_dumps():
    z = np.load(os.path.join(O.GOLD, "trees.npz"))
    names = sorted({k.split(".")[0] for k in z.files})
    assert len(names) >= 6
    for name in names:
        n = int(z[name + ".n"])
        arrs, info = O.tree_run(O.ORACLE, "sqzo_tree_run", n, z[name + ".symbols"])
        for key, a in zip(("freq", "path", "bits", "pix", "lix", "rix"), arrs):
            assert (a == z[f"{name}.{key}"]).all(), (name, key)
        assert (info == z[name + ".info"]).all(), name


def test_token_dump():
    want = np.load(os.path.join(O.GOLD, "laozi_tokens_w15.npy"))
    got = O.tokens(O.corpus("laozi.txt"), 1 << 15)
    assert (got == want).all()
    # the parse reproduces the input
    data = O.corpus("laozi.txt")
    out = bytearray()
    for t in got:
        t = int(t)
        if t & 0x80000000:
            ln, ds = (t >> 16) & 0x1FF, t & 0x7FFF
            for _ in range(ln):
                out.append(out[-ds])
        else:
            out.append(t)
    assert bytes(out) == data


def test_match_finder_differential():
    """bst.c:254-308 pattern: a second finder must equal brute force at every position."""
    rng = random.Random(5)
    data = bytes(rng.choice(b"abcab") for _ in range(600)) + b"x" * 300 + bytes(range(40)) * 3
    for window in (8, 64, 1024):
        for i in range(len(data)):
            ln, ds = O.match_at(data, i, window)
            best, where = 0, 0
            for d in range(1, min(i, window - 1) + 1):
                k = 0
                while k < min(len(data) - i, 257) and data[i - d + k] == data[i + k]:
                    k += 1
                if k >= 3 and k > best:
                    best, where = k, d
            assert (ln, ds) == (best, where), (window, i)


def test_errors():
    import errno
    data = O.zipf_block(3, 2000)
    e, part = O.encode_err(data, 10, True, 64)          # bitstream.h:38
    assert e == errno.E2BIG and len(part) == 64
    full = O.encode(data, 10, header=True)
    assert part == full[:64]
    e, _ = O.encode_err(data, 9, True, 4096)             # squeeze.h:257-258
    assert e == errno.EINVAL
    e, _, _ = O.decode(full[:len(full) - 8], header=True)  # bitstream.h:74
    assert e == errno.E2BIG
    e, _, _ = O.decode(full[:8] + bytes([0xFF]) + full[9:], header=True)  # win_bits 255
    assert e == errno.EINVAL                                                # squeeze.h:449-450


@pytest.mark.skipif(O.REF is None, reason="reference build (oracle/_ref) only exists in the build container")
def test_against_reference_build():
    rng = random.Random(11)
    cases = [b"", b"a", b"ab", b"abc" * 50, bytes(300), bytes(range(256)) * 2]
    for _ in range(40):
        n = rng.randint(0, 2500)
        alpha = rng.choice([2, 3, 8, 64, 256])
        cases.append(bytes(rng.randrange(alpha) for _ in range(n)))
    for data in cases:
        for wb in (10, 13, 15):
            for hdr in (False, True):
                assert O.encode(data, wb, header=hdr) == O.ref_compress(data, wb, hdr)
    # trees, fuzzed
    for trial in range(60):
        n = rng.choice([8, 32, 512])
        syms = [min(n - 1, int(rng.expovariate(rng.choice([0.05, 0.3, 1.0])))) for _ in range(rng.randint(1, 3000))]
        a, ia = O.tree_run(O.REF, "sqz_ref_tree_run", n, syms)
        b, ib = O.tree_run(O.ORACLE, "sqzo_tree_run", n, syms)
        assert all((x == y).all() for x, y in zip(a, b)) and (ia == ib).all()


def test_tree_invariants_the_kernels_rely_on():
    """The batched update in the HIP kernels (sqz_tree.h: bump_batch) drops two tests
    because of two properties of the tree at rest: every sibling pair is ordered (lo <= hi) and
    every node below the root has a sibling once the tree holds two leaves.  The oracle built
    with -DSQZO_CHECK_INVARIANTS asserts both after EVERY update (it aborts otherwise)."""
    import ctypes as C
    import random
    import subprocess
    subprocess.check_call(["make", "-C", O.ODIR, "-s", "check-invariants"])
    L = C.CDLL(os.path.join(O.ODIR, "liboracle_check.so"))
    L.sqzo_encode.restype = C.c_int
    L.sqzo_encode.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_uint64,
                              C.POINTER(C.c_uint64)]

    def enc(data, window):
        out = C.create_string_buffer(2 * len(data) + 4096)
        nb = C.c_uint64(0)
        assert L.sqzo_encode(data, len(data), window, 0, out, len(out), C.byref(nb)) == 0

    for name in ("laozi.txt", "confucius.txt", "x64.elf", "mandrill.bmp"):
        enc(O.corpus(name)[:40000], 1 << 12)
    enc(O.zipf_block(0, 32768), 1 << 12)
    rng = random.Random(5)
    for _ in range(12):
        n = rng.choice([1, 2, 3, 50, 999, 5000])
        alpha = rng.choice([2, 4, 16, 256])
        enc(bytes(rng.randrange(alpha) for _ in range(n)), rng.choice([32, 1 << 10]))
    enc(bytes(20000), 1 << 12)
    enc(b"abc" * 7000, 1 << 12)


def test_restatement_under_sanitizers():
    """the CPU restatement under ASan + UBSan over real inputs (encode, decode,
    200 corrupted streams, the token-driven encoder with tight buffers, the tree alone): the golden
    sizes / fingerprints must come out and the sanitizers must stay silent."""
    import subprocess
    subprocess.check_call(["make", "-C", O.ODIR, "-s", "asan-driver"])
    exe = os.path.join(O.ODIR, "asan_driver")
    gold = {(c["file"], c["win_bits"]): c for c in O.golden()["corpus"]}
    for name, wb in (("laozi.txt", 10), ("laozi.txt", 15), ("confucius.txt", 12)):
        c = gold[(name, wb)]
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
        p = subprocess.run([exe, os.path.join(O.CORPUS, name), str(wb), str(c["out_bytes"]), c["out_fnv"]],
                           capture_output=True, text=True, env=env, timeout=300)
        assert p.returncode == 0 and p.stdout.startswith("ok "), p.stdout + p.stderr
        assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr
