#!/usr/bin/env python3
"""oracle/gen_golden.py -- TEST INFRASTRUCTURE ONLY.

Regenerates tests/golden/ from the REFERENCE ITSELF (oracle/_ref/libsqz_ref.so,
built by `make -C oracle ref` from /root/reference/attic/map_experiment, see
oracle/Makefile).  Runs only in the build container; the GPU box gets the
committed fixtures.

What is written (all plain data -- inputs and expected outputs):
  golden.json        small known-answer vectors (inputs are the reference's own
                     test strings: attic test.c:199-210, shl.c:24-26,
                     test.c:547), corpus fingerprints (size + FNV-1a-64 of the
                     compressed bytes), Zipf block
                     fingerprints (section 8d)
  <file>.w<bits>.sqz full reference outputs for laozi/confucius (header included)
  trees.npz          reference huffman.h tree dumps for fixed symbol sequences
  laozi_tokens_w15.npy  LZ77 parse of laozi.txt -- the reference exposes no token
                     interface, so this one comes from the restatement AFTER its
                     compressed bytes were checked equal to the reference's.
"""
import ctypes
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
CORPUS_SRC = "/root/reference/test"
CORPUS_DST = os.path.join(ROOT, "tests", "corpus")
CORPUS = ["laozi.txt", "confucius.txt", "mandrill.bmp", "arm64.elf", "x64.elf",
          "mandrill.png"]

REF = ctypes.CDLL(os.path.join(HERE, "_ref", "libsqz_ref.so"))
ORA = ctypes.CDLL(os.path.join(HERE, "liboracle.so"))
REF.sqz_ref_compress.restype = ctypes.c_int64
REF.sqz_ref_compress_file.restype = ctypes.c_int64
REF.sqz_ref_compress_file.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_char_p]
REF.sqz_ref_compress.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64]
ORA.sqzo_fnv1a64.restype = ctypes.c_uint64
ORA.sqzo_fnv1a64.argtypes = [ctypes.c_char_p, ctypes.c_uint64]


def ref_compress(data: bytes, win_bits: int, header: bool) -> bytes:
    out = ctypes.create_string_buffer(2 * len(data) + 64)
    n = REF.sqz_ref_compress(data, len(data), win_bits, int(header), out, len(out))
    if n < 0:
        raise RuntimeError(f"reference error {-n}")
    return out.raw[:n]


def fnv(b: bytes) -> str:
    return f"{ORA.sqzo_fnv1a64(b, len(b)):016x}"


def zipf_block(index: int, nbytes: int) -> bytes:
    buf = ctypes.create_string_buffer(nbytes)
    ORA.sqzo_zipf_block(ctypes.c_uint64(index), buf, ctypes.c_uint64(nbytes))
    return buf.raw


def ref_tree(n, syms):
    m = 2 * n - 1
    s = np.asarray(syms, dtype=np.int32)
    arrs = [np.zeros(m, np.uint64), np.zeros(m, np.uint64)] + \
           [np.zeros(m, np.int32) for _ in range(4)]
    info = np.zeros(4, np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    r = REF.sqz_ref_tree_run(ctypes.c_int32(n), p(s), ctypes.c_uint64(len(s)),
                             *[p(a) for a in arrs], p(info))
    assert r == 0
    return arrs, info


def main():
    os.makedirs(GOLD, exist_ok=True)
    os.makedirs(CORPUS_DST, exist_ok=True)
    gold = {"format": 1,
            "source": "reference H0 (attic/map_experiment) compiled into oracle/_ref",
            "vectors": [], "corpus": [], "zipf": []}

    lorem = b"Lorem ipsum dolor sit amet. " * 3
    small = [
        ("empty", b"", 10), ("a", b"a", 10), ("aaaa", b"aaaa", 10),
        ("hello", b"Hello World Hello.World Hello World", 10),
        ("lorem3", lorem, 11),
        ("zeros4k", bytes(4096), 10),
        ("rle1234", b"\x01\x02\x03\x04" * 1024, 10),
        ("abc_ladder", b"abcabcdabcdeabcdefabcdefgabcdefabcdeabcd", 10),
        ("bytes0_255", bytes(range(256)), 10),
        ("bytes0_255x3", bytes(range(256)) * 3, 12),
        ("two", b"ab", 10), ("three", b"aaa", 10),
        ("len257", b"x" * 258, 10), ("len258", b"x" * 259, 10),
        ("len259", b"x" * 260, 10), ("len600", b"q" * 600, 15),
    ]
    for name, data, wb in small:
        for header in (False, True):
            out = ref_compress(data, wb, header)
            gold["vectors"].append({"name": name, "input_hex": data.hex(),
                                    "win_bits": wb, "header": header,
                                    "out_hex": out.hex()})

    for f in CORPUS:
        src = os.path.join(CORPUS_SRC, f)
        data = open(src, "rb").read()
        shutil.copyfile(src, os.path.join(CORPUS_DST, f))
        wins = (10, 12, 15) if f == "laozi.txt" else (12, 15)
        for wb in wins:
            out = ref_compress(data, wb, True)
            gold["corpus"].append({"file": f, "in_bytes": len(data),
                                   "in_fnv": fnv(data), "win_bits": wb,
                                   "out_bytes": len(out), "out_fnv": fnv(out)})
            if f in ("laozi.txt", "confucius.txt"):
                with open(os.path.join(GOLD, f"{f}.w{wb}.sqz"), "wb") as fh:
                    fh.write(out)
            if f == "laozi.txt":
                # the same stream as the reference's FILE mode writes it (attic test.c:39-42):
                # produced by the reference's own .output callback path, not by swapping here
                path = os.path.join(GOLD, f"{f}.w{wb}.file")
                n = REF.sqz_ref_compress_file(data, len(data), wb, path.encode())
                assert n == len(out), (n, len(out))
                gold.setdefault("file_mode", []).append(
                    {"file": f, "win_bits": wb, "image": os.path.basename(path),
                     "stream": f"{f}.w{wb}.sqz", "bytes": int(n),
                     "image_fnv": fnv(open(path, "rb").read())})
            print(f, wb, len(out), fnv(out), flush=True)

    # Zipf blocks: full size at 2^15 (slow: ~8 s each) and small ones
    for idx, nbytes, wb in [(0, 262144, 15), (1, 262144, 15),
                            (0, 16384, 12), (5, 16384, 12), (7, 40000, 15),
                            (4095, 8192, 10)]:
        data = zipf_block(idx, nbytes)
        out = ref_compress(data, wb, False)
        gold["zipf"].append({"block": idx, "in_bytes": nbytes, "in_fnv": fnv(data),
                             "win_bits": wb, "header": False,
                             "out_bytes": len(out), "out_fnv": fnv(out)})
        print("zipf", idx, nbytes, wb, len(out), fnv(out), flush=True)

    with open(os.path.join(GOLD, "golden.json"), "w") as fh:
        json.dump(gold, fh, indent=1)

    # tree dumps
    rng = np.random.default_rng(20241024)
    trees = {}
    seqs = {
        "lit_uniform": (512, rng.integers(0, 286, 5000)),
        "lit_skew": (512, np.minimum(rng.geometric(0.05, 8000) - 1, 285)),
        "lit_sequential": (512, np.arange(286)),
        "pos_uniform": (32, rng.integers(0, 31, 3000)),
        "pos_skew": (32, np.minimum(rng.geometric(0.3, 3000) - 1, 30)),
        "pos_fib": (32, np.concatenate([np.full(int(1.5 ** k) + 1, k) for k in range(20)])),
        "tiny8": (8, rng.integers(0, 8, 500)),
    }
    for name, (n, syms) in seqs.items():
        arrs, info = ref_tree(n, syms)
        trees[name + ".symbols"] = np.asarray(syms, np.int32)
        trees[name + ".n"] = np.int32(n)
        for k, a in zip(("freq", "path", "bits", "pix", "lix", "rix"), arrs):
            trees[f"{name}.{k}"] = a
        trees[name + ".info"] = info
    np.savez_compressed(os.path.join(GOLD, "trees.npz"), **trees)

    # token dump (restatement, after equality of compressed bytes was checked)
    ORA.sqzo_encode.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint32,
                                ctypes.c_int, ctypes.c_char_p, ctypes.c_uint64,
                                ctypes.POINTER(ctypes.c_uint64)]
    data = open(os.path.join(CORPUS_SRC, "laozi.txt"), "rb").read()
    out = ctypes.create_string_buffer(2 * len(data) + 64)
    nb = ctypes.c_uint64()
    assert ORA.sqzo_encode(data, len(data), 1 << 15, 15, out, len(out), ctypes.byref(nb)) == 0
    assert out.raw[:nb.value] == ref_compress(data, 15, True)
    toks = np.zeros(len(data), np.uint32)
    cnt = ctypes.c_uint64()
    assert ORA.sqzo_tokens(data, ctypes.c_uint64(len(data)), ctypes.c_uint32(1 << 15),
                           toks.ctypes.data_as(ctypes.c_void_p),
                           ctypes.c_uint64(len(toks)), ctypes.byref(cnt)) == 0
    np.save(os.path.join(GOLD, "laozi_tokens_w15.npy"), toks[:cnt.value])
    print("tokens", cnt.value)


if __name__ == "__main__":
    sys.exit(main())
