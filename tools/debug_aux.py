"""tools/debug_aux.py NAME [BATCH]: first prefix of a trees.npz sequence after which the device tree has
given up its intervals (aux == 0) or differs from the oracle -- development aid (needs an MI355X)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import oracle_lib as O
import test_gpu_tree as T
import torch, sqz_amd
from sqz_amd import _native as N
name = sys.argv[1]; batch = (int(sys.argv[2]) if len(sys.argv) > 2 else 64) | 0x100
z = np.load(os.path.join(O.GOLD, "trees.npz"))
n = int(z[name + ".n"]); syms = z[name + ".symbols"]
dev = (torch, N.lib())
def bad(k):
    which, head, nodes = T.device_tree(dev, n, syms[:k], batch)
    if int(head[3]) != 1 or int(head[4]) != 0: return "aux/fault %s" % [int(h) for h in head]
    arrs, info = O.tree_run(O.ORACLE, "sqzo_tree_run", n, syms[:k])
    try:
        T.compare(which, head, nodes, n, arrs, info)
    except AssertionError as e:
        return str(e)[:200]
    return None
import time
lo, hi = 0, len(syms)
t0 = time.time(); r = bad(hi); print("full", r, time.time() - t0, flush=True)
assert r is not None
while hi - lo > 1:
    mid = (lo + hi) // 2
    t0 = time.time(); r = bad(mid); print(mid, r, round(time.time() - t0, 2), flush=True)
    if r is None: lo = mid
    else: hi = mid
print("first bad prefix", hi, "symbol", int(syms[hi - 1]), "->", bad(hi))
print("previous symbols", [int(s) for s in syms[max(0, hi - 70):hi]])
