"""GPU: the adaptive Huffman tree as the kernels keep it in LDS, dumped after fixed symbol
sequences (sqz_hip_debug_tree) and compared node for node with

  * the seven dumps of the REFERENCE's huffman.h tree in tests/golden/trees.npz
    (freq / path / bits / pix / lix / rix of every node + n, next, depth, complete), and
  * the oracle's tree on random and adversarial sequences, including its huffman.h:29-33 counters.

Both ways through the device code are driven: batches of 64 (the batched interval update) and
one symbol at a time (the exact path only), plus an odd batch size.  The comparison walks both
trees from their roots (node numbering differs by design: leaves keep the symbol value, the
device counts internal nodes up from the root, the reference down from 2n-2) and checks shape,
leaf identity, counts, depths, codes, and that the device's leaf intervals and cached test
partners agree with the shape."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

NIL = 0x3FF


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    import sqz_amd
    assert "gfx950" in sqz_amd.device_info()["name"]
    from sqz_amd import _native as N
    return torch, N.lib()


def device_tree(dev, n_ref, syms, batch):
    torch, L = dev
    which = 0 if n_ref == 512 else 1
    nodes = 576 if which == 0 else 64
    s = torch.tensor(np.ascontiguousarray(syms, dtype=np.int32), device="cuda")
    dump = torch.zeros(8 + 4 * nodes, dtype=torch.int32, device="cuda")
    rc = L.sqz_hip_debug_tree(C.c_void_p(s.data_ptr()), len(syms), which, batch, C.c_void_p(dump.data_ptr()), None)
    torch.cuda.synchronize()
    assert rc == 0
    d = dump.cpu().numpy().view(np.uint32)
    return which, d[:8], d[8:].reshape(nodes, 4)


def first_bad_prefix(dev, n, syms, batch, limit=400):
    """on a mismatch: the shortest prefix of the sequence whose device tree differs from the oracle's"""
    for k in range(1, min(len(syms), limit) + 1):
        arrs, info = O.tree_run(O.ORACLE, "sqzo_tree_run", n, syms[:k])
        which, head, nodes = device_tree(dev, n, syms[:k], batch)
        try:
            compare(which, head, nodes, n, arrs, info)
        except AssertionError as e:
            return k, [int(x) for x in syms[max(0, k - 6):k]], str(e), [int(h) for h in head]
    return None


def rev_bits(path, bits):
    c = 0
    for k in range(bits):
        c = (c << 1) | ((int(path) >> k) & 1)
    return c


def compare(which, head, nodes, n_ref, ref, ref_info, counters=None):
    freq, path, bits, pix, lix, rix = ref
    base, leaves, pos0 = (0, 288, 0) if which == 0 else (576, 32, 288)
    root_d, root_r = base + leaves, 2 * n_ref - 2
    assert int(head[2]) == int(ref_info[3]), "complete"
    assert int(head[1]) == int(ref_info[2]), "depth mark"
    assert int(head[3]) == 1 and int(head[4]) == 0          # intervals kept, no fault
    # internal nodes allocated: the reference counts down from m-1, the device up from the root
    assert int(head[0]) - root_d == (2 * n_ref - 2) - int(ref_info[1]) + 1
    if counters is not None:
        assert [int(head[5]), int(head[6]), int(head[7])] == [int(c) for c in counters]
    lnk = lambda v: (int(nodes[v - base][0]) & NIL, (int(nodes[v - base][0]) >> 10) & NIL, (int(nodes[v - base][0]) >> 20) & NIL)
    # rng word: a leaf's position (low 9 bits), an internal node's first | last << 10 leaf; partner << 20
    rng = lambda v: int(nodes[v - base][1])
    cnt = lambda v: (int(nodes[v - base][2]) & 0xFFFFFF, (int(nodes[v - base][2]) >> 24) & 0x3F)
    stack, pos = [(root_d, root_r, 0)], pos0
    seen = 0
    ends = {}                                                  # node -> (leftmost leaf, rightmost leaf) by the walk
    order = []
    while stack:
        v, w, depth = stack.pop()
        up, lo, hi = lnk(v)
        f, d = cnt(v)
        pa = (rng(v) >> 20) & NIL
        seen += 1
        order.append(v)
        if v != root_d:
            assert f == int(freq[w]), ("freq", v, w)
        assert depth == int(bits[w]), ("depth", v, w)
        # the cached test partner: sibling for a lo child, uncle for a hi child below the root's children
        if v != root_d:
            pu, plo, phi = lnk(up)
            want = NIL
            if phi != v:
                want = phi
            elif pu != NIL:
                g = lnk(pu)
                want = g[2] if g[1] == up else g[1]
            assert pa == want, ("partner", v, pa, want)
        if v < root_d:                                        # a leaf: same symbol, same code, its place in leaf order
            assert v - base == w, ("leaf id", v, w)
            assert lo == NIL and hi == NIL and int(lix[w]) == -1 and int(rix[w]) == -1
            assert d == depth, ("stored leaf depth", v, d, depth)
            assert int(nodes[v - base][3]) == rev_bits(path[w], int(bits[w])), ("code", v)
            assert rng(v) & 0x1FF == pos, ("leaf position", v, rng(v) & 0x1FF, pos)
            ends[v] = (v, v)
            pos += 1
            continue
        assert (lo == NIL) == (int(lix[w]) == -1) and (hi == NIL) == (int(rix[w]) == -1), ("children", v, w)
        if hi != NIL:
            assert lnk(hi)[0] == v
            stack.append((hi, int(rix[w]), depth + 1))
        if lo != NIL:
            assert lnk(lo)[0] == v
            stack.append((lo, int(lix[w]), depth + 1))
    # every internal node names its leftmost and rightmost leaf
    for v in reversed(order):
        if v < root_d:
            continue
        up, lo, hi = lnk(v)
        kids = [k for k in (lo, hi) if k != NIL]
        if not kids:
            continue
        ends[v] = (ends[kids[0]][0], ends[kids[-1]][1])
        assert (rng(v) & NIL, (rng(v) >> 10) & NIL) == ends[v], ("ends", v, rng(v) & NIL, (rng(v) >> 10) & NIL, ends[v])
    attached = sum(1 for i in range(2 * n_ref - 1) if i == root_r or int(pix[i]) != -1)
    assert seen == attached


@pytest.mark.parametrize("batch", [64, 1, 7])
def test_reference_tree_dumps(dev, batch):
    z = np.load(os.path.join(O.GOLD, "trees.npz"))
    names = sorted({k.split(".")[0] for k in z.files})
    assert len(names) == 7
    for name in names:
        n = int(z[name + ".n"])
        syms = z[name + ".symbols"]
        ref = [z[f"{name}.{k}"] for k in ("freq", "path", "bits", "pix", "lix", "rix")]
        info = z[name + ".info"]
        if n == 8:
            # an 8-leaf tree through the 32-leaf device tree: same shape (no pool is exhausted),
            # the reference arrays re-indexed for n = 32 by the oracle
            arrs, info = O.tree_run(O.ORACLE, "sqzo_tree_run", 32, syms)
            ref, n = arrs, 32
        which, head, nodes = device_tree(dev, n, syms, batch)
        try:
            compare(which, head, nodes, n, ref, info)
        except AssertionError as e:
            raise AssertionError(f"{name} batch {batch}: {e}; first bad prefix: {first_bad_prefix(dev, n, syms, batch)}")


def oracle_counters(n, syms):
    out = (C.c_uint64 * 3)()
    s = np.ascontiguousarray(syms, dtype=np.int32)
    assert O.ORACLE.sqzo_tree_counters(C.c_int32(n), s.ctypes.data_as(C.c_void_p), C.c_uint64(len(s)), out) == 0
    return list(out)


@pytest.mark.parametrize("batch", [64, 1])
def test_device_tree_vs_oracle(dev, batch):
    rng = np.random.default_rng(11)
    cases = [
        (512, rng.integers(0, 286, 30000)),
        (512, np.minimum(rng.geometric(0.03, 60000) - 1, 285)),
        (512, np.concatenate([np.full(1 << k, k) for k in range(16)])),          # doubling counts: depth 16
        (512, np.concatenate([np.full(1 << k, k) for k in range(15, -1, -1)])),
        (512, np.concatenate([[285], np.full(9000, 65), rng.integers(0, 256, 5000)])),
        (32, rng.integers(0, 31, 30000)),
        (32, np.minimum(rng.geometric(0.25, 40000) - 1, 30)),
        (32, np.concatenate([np.full(int(1.6 ** k) + 1, k) for k in range(26)])),
        (32, np.arange(31)),
        (32, np.array([30])),
        (32, np.array([30, 4])),
    ]
    for n, syms in cases:
        arrs, info = O.tree_run(O.ORACLE, "sqzo_tree_run", n, syms)
        which, head, nodes = device_tree(dev, n, syms, batch)
        try:
            compare(which, head, nodes, n, arrs, info, oracle_counters(n, syms))
        except AssertionError as e:
            raise AssertionError(f"n={n} len={len(syms)} batch {batch}: {e}; first bad prefix: "
                                 f"{first_bad_prefix(dev, n, syms, batch)}")
