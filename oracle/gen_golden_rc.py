#!/usr/bin/env python3
"""oracle/gen_golden_rc.py -- TEST INFRASTRUCTURE ONLY.  Writes tests/golden/golden_rc.json from the
reference's HEAD codec compiled where it lies (oracle/_ref/libsqz_ref_rc.so, `make -C oracle ref-rc`):
known-answer vectors (the reference's own test strings: shl.c:24-26, attic test.c:199-210) with the
complete stream in hex, and size + FNV-1a-64 fingerprints of corpus files and Zipf blocks
(SURVEY.md section 8f-1 quotes laozi 14,735 B 99bd04202966bc15 and confucius 47,914 B 2a8534293b6b47c2).
Runs only in the build container."""
import ctypes as C
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O

R = C.CDLL(os.path.join(HERE, "_ref", "libsqz_ref_rc.so"))


def ref_encode(d):
    out = C.create_string_buffer(2 * len(d) + 64)
    n = C.c_uint64()
    e = R.sqz_ref_rc_compress(d, C.c_uint64(len(d)), C.c_uint32(1 << 15), out, C.c_uint64(len(out)), C.byref(n))
    assert e == 0
    return out.raw[:n.value]


def main():
    gold = {"format": 1, "source": "reference HEAD (inc/sqz/sqz.h + src/sqz.c) compiled into oracle/_ref/libsqz_ref_rc.so",
            "vectors": [], "corpus": [], "zipf": []}
    small = [("empty", b""), ("a", b"a"), ("lorem3", b"Lorem ipsum dolor sit amet. " * 3),
             ("hello", b"Hello World Hello.World Hello World"), ("zeros4k", bytes(4096)),
             ("rle1234", b"\x01\x02\x03\x04" * 1024), ("bytes0_255", bytes(range(256))), ("ff300", b"\xff" * 300)]
    for name, d in small:
        gold["vectors"].append({"name": name, "input_hex": d.hex(), "out_hex": ref_encode(d).hex()})
    for f in ("laozi.txt", "confucius.txt", "x64.elf", "mandrill.png"):
        d = O.corpus(f)
        out = ref_encode(d)
        gold["corpus"].append({"file": f, "in_bytes": len(d), "in_fnv": O.fnv(d), "out_bytes": len(out), "out_fnv": O.fnv(out)})
        print(f, len(out), O.fnv(out), flush=True)
    for idx, nb in ((0, 262144), (1, 262144), (5, 16384)):
        d = O.zipf_block(idx, nb)
        out = ref_encode(d)
        gold["zipf"].append({"block": idx, "in_bytes": nb, "in_fnv": O.fnv(d), "out_bytes": len(out), "out_fnv": O.fnv(out)})
    with open(os.path.join(ROOT, "tests", "golden", "golden_rc.json"), "w") as fh:
        json.dump(gold, fh, indent=1)
    print("wrote golden_rc.json")


if __name__ == "__main__":
    main()
