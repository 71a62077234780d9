"""CPU: the interval ("range") form of the batched adaptive-Huffman update behind the entropy
kernels (sqz_amd/csrc/sqz_tree.h keeps the same intervals as first/last leaf + leaf positions),
restated in C with flat per-node passes
(tests/model/range_model.c) and held against the oracle's tree in lockstep: same links,
counts, depths, depth mark, codes and huffman.h:29-33 counters after every batch and after
every exact step, on the reference's own tree fixtures, on the symbol streams of real
inputs, and on random ones.  No GPU involved: this pins the ALGORITHM; the GPU parity tests
pin the kernels."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "model", "range_model.c")
LIB = os.path.join(ROOT, "tests", "model", "librange_model.so")


@pytest.fixture(scope="module")
def model():
    deps = [SRC, os.path.join(ROOT, "oracle", "sqz_oracle.c")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["gcc", "-std=c99", "-O2", "-fPIC", "-shared", "-Wall", "-Wextra",
                               "-Wno-unused-function", SRC, "-o", LIB, "-lm"])
    L = C.CDLL(LIB)
    L.range_model_run.restype = C.c_int
    L.range_model_run.argtypes = [C.c_int32, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
    return L


def run(L, n_ref, syms, batch=64):
    s = np.ascontiguousarray(syms, dtype=np.int32)
    st = np.zeros(4, np.uint64)
    rc = L.range_model_run(n_ref, s.ctypes.data_as(C.c_void_p), len(s), batch, st.ctypes.data_as(C.c_void_p))
    return rc, st


def lit_pos_symbols(data: bytes, window: int):
    """the two symbol streams the codec feeds its trees for `data` (squeeze.h:278-315)"""
    toks = O.tokens(data, window)
    lens = (toks >> 16) & 0x1FF
    is_m = (toks >> 31) != 0
    len_base = np.array([3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99,
                         115, 131, 163, 195, 227, 258])
    pos_base = np.array([1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025,
                         1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577])
    lit = np.where(is_m, 257 + np.searchsorted(len_base, lens, side="right") - 1, toks & 0xFF)
    pos = np.searchsorted(pos_base, toks[is_m] & 0x7FFF, side="right") - 1
    return np.concatenate([[285], lit]).astype(np.int32), np.concatenate([[30], pos]).astype(np.int32)


def test_reference_tree_fixtures(model):
    z = np.load(os.path.join(O.GOLD, "trees.npz"))
    names = sorted({k.split(".")[0] for k in z.files})
    assert len(names) == 7
    for name in names:
        n = int(z[name + ".n"])
        if n not in (32, 512):
            continue                                    # the model is sized for the codec's two trees
        for batch in (1, 7, 64):
            rc, st = run(model, n, z[name + ".symbols"], batch)
            assert rc == 0, (name, batch)


@pytest.mark.parametrize("name,window", [("laozi.txt", 1 << 15), ("confucius.txt", 1 << 12)])
def test_corpus_symbol_streams(model, name, window):
    lit, pos = lit_pos_symbols(O.corpus(name), window)
    for n, syms in ((512, lit), (32, pos)):
        rc, st = run(model, n, syms)
        assert rc == 0
        assert st[1] + st[2] == len(syms)
        assert st[1] > 0.8 * len(syms)                  # most symbols go through batches


def test_zipf_block_symbols(model):
    data = O.zipf_block(1, 60000)
    lit, pos = lit_pos_symbols(data, 1 << 15)
    rc, st = run(model, 512, lit)
    assert rc == 0
    rc, st2 = run(model, 32, pos)
    assert rc == 0


def test_random_and_degenerate_streams(model):
    rng = np.random.default_rng(7)
    cases = [
        (512, rng.integers(0, 286, 20000)),
        (512, np.minimum(rng.geometric(0.02, 30000) - 1, 285)),
        (512, np.concatenate([[285], np.full(5000, 65), rng.integers(0, 256, 3000)])),
        (512, np.concatenate([np.full(1 << k, k) for k in range(15)])),       # doubling counts: a deep chain
        (512, np.concatenate([np.full(1 << k, k) for k in range(14, -1, -1)])),
        (32, rng.integers(0, 31, 20000)),
        (32, np.minimum(rng.geometric(0.3, 20000) - 1, 30)),
        (32, np.concatenate([np.full(int(1.6 ** k) + 1, k) for k in range(24)])),
        (32, np.arange(31)),
    ]
    for n, syms in cases:
        for batch in (64, 5):
            rc, st = run(model, n, syms, batch)
            assert rc == 0, (n, len(syms), batch)
