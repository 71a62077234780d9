"""tools/debug_tree.py NAME K [BATCH] -- print the device tree and the oracle's tree after the first
K symbols of a tests/golden/trees.npz sequence (needs an MI355X).  Development aid."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import oracle_lib as O
import test_gpu_tree as T
import torch, sqz_amd
from sqz_amd import _native as N

name, K = sys.argv[1], int(sys.argv[2])
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
z = np.load(os.path.join(O.GOLD, "trees.npz"))
n = int(z[name + ".n"]); syms = z[name + ".symbols"][:K]
dev = (torch, N.lib())
print("symbols", [int(s) for s in syms])
which, head, nodes = T.device_tree(dev, n, syms, batch)
base, leaves = (0, 288) if which == 0 else (576, 32)
print("head next/mark/complete/aux/fault/updates/swaps/moves", [int(h) for h in head])
def show(v, ind=0):
    w, r, c, code = [int(x) for x in nodes[v - base]]
    up, lo, hi = w & 0x3FF, (w >> 10) & 0x3FF, (w >> 20) & 0x3FF
    print("  " * ind + f"{v}: f={c & 0xFFFFFF} d={c >> 24} [{r & 0x1FF},{(r >> 9) & 0x1FF}) pa={(r >> 18) & 0x3FF} up={up}" + (f" code={code:b}" if v < base + leaves else ""))
    for ch in (lo, hi):
        if ch != 0x3FF and ind < 40: show(ch, ind + 1)
show(base + leaves)
arrs, info = O.tree_run(O.ORACLE, "sqzo_tree_run", n, syms)
freq, path, bits, pix, lix, rix = arrs
print("oracle info n/next/depth/complete", [int(i) for i in info])
def osh(w, ind=0):
    print("  " * ind + f"{w}: f={int(freq[w])} d={int(bits[w])}")
    for ch in (int(lix[w]), int(rix[w])):
        if ch != -1: osh(ch, ind + 1)
osh(2 * n - 2)
