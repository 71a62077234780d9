"""tools/debug_scan.py NAME START STOP STEP [BATCH]: run prefixes of a trees.npz sequence on the device until the
first one that differs from the oracle; prints timing per run (development aid, needs an MI355X)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import oracle_lib as O
import test_gpu_tree as T
import torch, sqz_amd
from sqz_amd import _native as N
name, a, b, step = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
batch = (int(sys.argv[5]) if len(sys.argv) > 5 else 64) | 0x100
z = np.load(os.path.join(O.GOLD, "trees.npz"))
n = int(z[name + ".n"]); syms = z[name + ".symbols"]
dev = (torch, N.lib())
def bad(k):
    t0 = time.time()
    which, head, nodes = T.device_tree(dev, n, syms[:k], batch)
    dt = time.time() - t0
    msg = None
    if int(head[3]) != 1 or int(head[4]) != 0: msg = "aux/fault"
    else:
        arrs, info = O.tree_run(O.ORACLE, "sqzo_tree_run", n, syms[:k])
        try: T.compare(which, head, nodes, n, arrs, info)
        except AssertionError as e: msg = str(e)[:150].replace("\n", " ")
    print(k, "head", [int(h) for h in head], "time %.2f" % dt, msg, flush=True)
    return msg
k = a
while k <= b:
    if bad(k) is not None:
        lo = max(a, k - step)
        for j in range(lo, k + 1):
            if bad(j) is not None:
                print("FIRST BAD", j, "symbols", [int(s) for s in syms[max(0, j - 8):j]]); break
        break
    k += step
