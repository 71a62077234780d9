"""CPU: the one-wave-per-stream kernels themselves (sqz_amd/csrc/huffman_emit.hip, decode.hip with
sqz_tree.h), compiled by g++ against tests/emu/hip/hip_runtime.h and run lane by lane on the CPU wave
emulator (64 cooperative fibers, every cross-lane operation a rendezvous), held against the oracle:

  * default build: streams equal the oracle's, decode gives the input back, the device tree equals
    the reference's tree fixtures (tests/golden/trees.npz) node for node;
  * builds with LOWERED thresholds, so that the code paths real streams cannot reach are executed
    and still agree with the oracle:
      wide     SQZ_BATCH_TOKENS=1500: the token limit where a stream gives up its intervals and
               goes on with full 32-bit counts and depths worked out on demand (2^24 in production)
      shallow  SQZ_AUX_DEPTH=7 SQZ_MAX_FAST_DEPTH=12: trees "too deep" for the interval machinery
               and for one-lane-per-level chains: the reference sequence on one lane, codes by a walk
      freeze   SQZ_FREEZE_DEPTH=9 against an oracle built with the same threshold: huffman.h:228-234
               (the tree stops taking updates; 63 in production, unreachable by any real stream)

This pins the kernels' LOGIC without a GPU; it says nothing about gfx950 code generation, LDS
ordering or timing -- the -m gpu tests do (and run the same variants as device builds)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
CSRC = os.path.join(ROOT, "sqz_amd", "csrc")
VARIANTS = {
    "default": [],
    "wide": ["-DSQZ_BATCH_TOKENS=1500"],
    "shallow": ["-DSQZ_AUX_DEPTH=7", "-DSQZ_MAX_FAST_DEPTH=12"],
    "freeze": ["-DSQZ_FREEZE_DEPTH=9"],
}


def _target(name, unit):
    out = os.path.join(EMU, f"libsqz_emu_{unit}_{name}.so")
    deps = [os.path.join(EMU, f) for f in ("emu_runtime.cpp", f"emu_{unit}.cpp", "hip/hip_runtime.h")] + \
           [os.path.join(CSRC, f) for f in ("sqz_tree.h", "sqz_device.h", "huffman_emit.hip", "decode.hip", "range_coder.hip",
                                            "sqz_kernels.h")]
    stale = not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", *VARIANTS[name], "-I" + EMU,
           "-I" + os.path.join(ROOT, "include"), "-Wno-unused-function", "-Wno-unused-variable",
           "-Wno-attributes", os.path.join(EMU, "emu_runtime.cpp"), os.path.join(EMU, f"emu_{unit}.cpp"), "-o", out]
    return out, stale, cmd


_all_built = False


def _build(name, unit):
    """the first call compiles every stale emulator library, all at once (eight g++ runs of ~30 s)"""
    global _all_built
    if not _all_built:
        jobs = []
        for v in VARIANTS:
            for u in ("emit", "decode"):
                out, stale, cmd = _target(v, u)
                if stale:
                    jobs.append((out, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
        for out, p in jobs:
            err = p.communicate()[1]
            assert p.returncode == 0, f"{out}:\n{err[-3000:]}"
        _all_built = True
    out, stale, cmd = _target(name, unit)
    if stale:                                 # a unit outside the batch above (the range coder's)
        subprocess.check_call(cmd)
    return C.CDLL(out)


@pytest.fixture(scope="module", params=list(VARIANTS))
def emu(request):
    name = request.param
    enc_oracle = lambda data, w: O.encode(data, 15, header=False, window=w)
    if name == "freeze":                      # the oracle with the same lowered threshold
        subprocess.check_call(["make", "-C", O.ODIR, "-s", "freeze9"])
        F = C.CDLL(os.path.join(O.ODIR, "liboracle_freeze9.so"))
        F.sqzo_encode.restype = C.c_int
        F.sqzo_encode.argtypes = O.ORACLE.sqzo_encode.argtypes

        def enc_oracle(data, w):
            out = C.create_string_buffer(2 * len(data) + 1088)
            n = C.c_uint64()
            assert F.sqzo_encode(data, len(data), w, 0, out, len(out), C.byref(n)) == 0
            return out.raw[:n.value]
    return name, _build(name, "emit"), _build(name, "decode"), enc_oracle


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def emit(E, blocks_tokens, caps):
    n = len(blocks_tokens)
    tok_off = np.zeros(n + 1, np.uint64)
    tok_off[1:] = np.cumsum(caps)
    toks = np.zeros(int(tok_off[-1]) + 64, np.uint32)
    cnt = np.zeros(n, np.uint32)
    for b, t in enumerate(blocks_tokens):
        toks[int(tok_off[b]):int(tok_off[b]) + len(t)] = t
        cnt[b] = len(t)
    out_off = np.zeros(n + 1, np.uint64)
    out_off[1:] = np.cumsum([2 * c + 1024 for c in caps])
    out = np.zeros(int(out_off[-1]), np.uint8)
    ob, err = np.zeros(n, np.uint64), np.zeros(n, np.int32)
    E.emu_huffman_emit(_p(toks), _p(tok_off), _p(cnt), n, _p(out), _p(out_off), _p(ob), _p(err), None)
    return [out[int(out_off[b]):int(out_off[b]) + int(ob[b])].tobytes() for b in range(n)], err


def decode(D, streams, sizes, waves=1):
    n = len(streams)
    in_off = np.zeros(n + 1, np.uint64)
    in_off[1:] = np.cumsum([len(s) for s in streams])
    data = np.frombuffer(b"".join(streams) + b"\0" * 8, np.uint8).copy()
    out_off = np.zeros(n + 1, np.uint64)
    out_off[1:] = np.cumsum(sizes)
    out = np.zeros(int(out_off[-1]) + 8, np.uint8)
    toks = np.zeros(int(out_off[-1]) + 64, np.uint32)
    cnt, err = np.zeros(n, np.uint32), np.zeros(n, np.int32)
    if waves == 1:
        D.emu_decode(_p(data), _p(in_off), n, _p(out), _p(out_off), _p(toks), _p(cnt), _p(err))
    else:                                     # the multi-wave decoder: `waves` wavefronts per stream
        D.emu_decode_waves(_p(data), _p(in_off), n, _p(out), _p(out_off), _p(toks), _p(cnt), _p(err), waves)
    return [out[int(out_off[b]):int(out_off[b + 1])].tobytes() for b in range(n)], err


def _moving(size, seed=9):
    """statistics that keep moving (small alphabets, drifting values, runs, repeats): the trees keep
    restructuring, which is what the decoder's kept read-ahead has to survive"""
    import random
    rng = random.Random(seed)
    out = bytearray()
    while len(out) < size:
        kind, n = rng.randrange(4), rng.randrange(30, 600)
        if kind == 0:
            alpha = [rng.randrange(256) for _ in range(rng.randrange(2, 10))]
            out += bytes(rng.choice(alpha) for _ in range(n))
        elif kind == 1:
            base = rng.randrange(256)
            out += bytes((base + int(rng.gauss(0, 12))) & 0xFF for _ in range(n))
        elif kind == 2 and len(out) > 60:
            a = rng.randrange(len(out) - 40)
            out += out[a:a + min(n, len(out) - a)]
        else:
            out += bytes([rng.randrange(256)]) * rng.randrange(3, 80)
    return bytes(out[:size])


def test_streams_equal_the_oracle(emu):
    name, E, D, enc_oracle = emu
    cases = [O.corpus("laozi.txt")[:9000], O.zipf_block(3, 7000), b"", b"a", b"abcabcabc" * 40,
             bytes(range(256)) * 6, bytes(3000), O.corpus("confucius.txt")[20000:26000],
             _moving(6000)]
    w = 1 << 12
    outs, err = emit(E, [O.tokens(c, w) for c in cases], [max(len(c), 1) for c in cases])
    want = [enc_oracle(c, w) for c in cases]
    assert err.tolist() == [0] * len(cases)
    assert outs == want
    back, derr = decode(D, want, [len(c) for c in cases])
    assert derr.tolist() == [0] * len(cases)
    assert back == [bytes(c) for c in cases]
    # the same streams through the decoder with 2 and 4 wavefronts per stream (read-ahead over 128 / 256
    # bit offsets per round, workgroup barriers between the waves: tests/emu runs the waves as they are)
    # (every wave count in the default build; the lowered-threshold builds, whose deep-tree paths belong to the
    # update side that wave 0 runs alone, with four waves on the streams that reach those paths)
    for waves in ((2, 4, 8) if name == "default" else (4,)):
        pick = list(range(len(cases))) if name == "default" else [0, 1, 8]
        back, derr = decode(D, [want[k] for k in pick], [len(cases[k]) for k in pick], waves=waves)
        assert derr.tolist() == [0] * len(pick), waves
        assert back == [bytes(cases[k]) for k in pick], waves


def slow_random_block(nbytes, block=38):
    """uniform random bytes, generator seed 1, 256 KB blocks: in block 38 (and 47) the first fifty symbols -- each
    seen once -- build a chain deeper than 26 levels down the literal tree's left edge, so the tree gives up its
    leaf positions; it settles at depth 10 soon after and has to take them up again (sqz_tree.h: regain_aux)"""
    bb = 262144
    return np.random.default_rng(1).integers(0, 256, 64 * bb, dtype=np.uint8)[block * bb:block * bb + nbytes].tobytes()


def test_a_tree_that_was_too_deep_takes_its_positions_up_again():
    """give_up_aux at depth 26, regain_aux once the tree is shallow again: same streams as the oracle, and the
    encoder is back on batches (the emulated kernel would need minutes for this input otherwise)"""
    import time
    E, D = _build("default", "emit"), _build("default", "decode")
    data = slow_random_block(60000)
    w = 1 << 15
    t0 = time.time()
    outs, err = emit(E, [O.tokens(data, w)], [len(data)])
    took = time.time() - t0
    want = O.encode(data, 15, header=False, window=w)
    assert err.tolist() == [0] and outs == [want]
    assert took < 20, f"the emulated emit kernel took {took:.0f} s: still one symbol at a time?"
    for waves in (1, 4):
        back, derr = decode(D, [want], [len(data)], waves=waves)
        assert derr.tolist() == [0] and back == [data], waves


def test_multi_wave_decoder_on_damaged_streams():
    """errors are the one-wave decoder's (and therefore the hardened oracle's): same errno for truncated and
    bit-flipped streams, with 2, 4 and 8 waves per stream"""
    import random
    D = _build("default", "decode")
    rng = random.Random(21)
    data = O.corpus("laozi.txt")[:3000]
    good = O.encode(data, 12, header=False)
    streams = [good]
    for _ in range(6):
        bad = bytearray(good)
        bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
        streams.append(bytes(bad))
    streams += [good[:len(good) // 2 // 8 * 8], good[:8], b""]
    sizes = [len(data)] * len(streams)
    base, e1 = decode(D, streams, sizes)
    for waves in (2, 4, 8):
        back, e = decode(D, streams, sizes, waves=waves)
        assert e.tolist() == e1.tolist(), waves
        for b in range(len(streams)):
            if e1[b] == 0:
                assert back[b] == base[b], (waves, b)


def test_reference_tree_fixtures_through_the_emulated_kernel():
    import test_gpu_tree as T
    E = _build("default", "emit")
    z = np.load(os.path.join(O.GOLD, "trees.npz"))
    for name in sorted({k.split(".")[0] for k in z.files}):
        n = int(z[name + ".n"])
        syms = z[name + ".symbols"][:2500]                # (the emulator makes ~5,000 updates a second)
        if n == 8:
            n = 32
        arrs, info = O.tree_run(O.ORACLE, "sqzo_tree_run", n, syms)
        which = 0 if n == 512 else 1
        nodes = 576 if which == 0 else 64
        for batch in (64, 1):
            s = np.ascontiguousarray(syms, dtype=np.int32)
            dump = np.zeros(8 + 4 * nodes, np.uint32)
            E.emu_tree_debug(_p(s), len(s), which, batch, _p(dump))
            T.compare(which, dump[:8], dump[8:].reshape(nodes, 4), n, arrs, info, T.oracle_counters(n, syms))
