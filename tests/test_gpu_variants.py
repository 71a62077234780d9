"""GPU: device builds of the kernels with LOWERED thresholds (sqz_amd/build.py VARIANTS), so that the
code paths no real stream reaches run on the MI355X and are held against the oracle:

  wide     the token limit where a stream gives up its leaf intervals (2^24 in production, 1500 here):
           full 32-bit counts, depths worked out on demand, one symbol at a time
  shallow  trees "too deep" for the interval machinery (26 -> 7) and for one-lane-per-level chains
           (60 -> 12): the reference sequence on one lane, codes by a walk
  freeze   huffman.h:228-234, the tree stops taking updates (depth mark 63 -> 9), against the oracle
           built with the same threshold (make -C oracle freeze9)

Each build is loaded by a child process through SQZ_AMD_LIB (one library per process)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, json, os, sys
sys.path[:0] = [os.environ["SQZ_ROOT"], os.path.join(os.environ["SQZ_ROOT"], "tests")]
import oracle_lib as O
import sqz_amd
from sqz_amd import batch
name = os.environ["SQZ_VARIANT"]
enc_oracle = lambda d, w: O.encode(d, 15, header=False, window=w)
if name == "freeze":
    F = C.CDLL(os.path.join(O.ODIR, "liboracle_freeze9.so"))
    F.sqzo_encode.restype = C.c_int
    F.sqzo_encode.argtypes = O.ORACLE.sqzo_encode.argtypes
    def enc_oracle(d, w):
        out = C.create_string_buffer(2 * len(d) + 1088); n = C.c_uint64()
        assert F.sqzo_encode(d, len(d), w, 0, out, len(out), C.byref(n)) == 0
        return out.raw[:n.value]
cases = [O.corpus("laozi.txt"), O.zipf_block(3, 40000), O.corpus("confucius.txt")[:30000], b"", b"ab",
         bytes(range(256)) * 20, O.corpus("x64.elf")[100000:140000], bytes(9000)]
res = {"lib": os.environ["SQZ_AMD_LIB"], "windows": {}}
for w in (1 << 12, 1 << 15):
    outs, err = batch.encode_blocks_host(cases, w)
    want = [enc_oracle(c, w) for c in cases]
    back, derr = batch.decode_blocks_host(want, [len(c) for c in cases])
    res["windows"][str(w)] = {"err": [int(e) for e in err], "derr": [int(e) for e in derr],
                              "enc_ok": [a == b for a, b in zip(outs, want)],
                              "dec_ok": [a == bytes(c) for a, c in zip(back, cases)]}
print("RESULT " + json.dumps(res))
"""


@pytest.mark.parametrize("name", ["wide", "shallow", "freeze"])
def test_variant_build_equals_the_oracle(name):
    from sqz_amd import build
    lib = build.variant_path(name)
    assert os.path.exists(lib), f"{lib} missing: python -m sqz_amd.build --variants"
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "all", "freeze9"])
    env = dict(os.environ, SQZ_AMD_LIB=lib, SQZ_VARIANT=name, SQZ_ROOT=ROOT)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["lib"] == lib
    for w, r in res["windows"].items():
        assert not any(r["err"]) and not any(r["derr"]), (name, w, r)
        assert all(r["enc_ok"]) and all(r["dec_ok"]), (name, w, r)
