"""tools/debug_steps.py NAME K [BATCH]: development aid (needs an MI355X).  Runs the first K symbols of a trees.npz
sequence on the device, stopping after N driver steps for growing N, and reports the first step after which the
device tree differs from the oracle's tree for the symbols consumed so far."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import ctypes as C
import numpy as np
import oracle_lib as O
import test_gpu_tree as T
import torch, sqz_amd
from sqz_amd import _native as N
name, K = sys.argv[1], int(sys.argv[2])
batch = (int(sys.argv[3]) if len(sys.argv) > 3 else 64) | 0x100
z = np.load(os.path.join(O.GOLD, "trees.npz"))
n = int(z[name + ".n"]); syms = np.ascontiguousarray(z[name + ".symbols"][:K], dtype=np.int32)
L = N.lib()
which0 = 0 if n == 512 else 1
nodes_n = 576 if which0 == 0 else 64
s = torch.tensor(syms, device="cuda")
def run(steps):
    dump = torch.zeros(8 + 4 * nodes_n, dtype=torch.int32, device="cuda")
    rc = L.sqz_hip_debug_tree(C.c_void_p(s.data_ptr()), K, which0 | (steps << 8), batch, C.c_void_p(dump.data_ptr()), None)
    torch.cuda.synchronize(); assert rc == 0
    d = dump.cpu().numpy().view(np.uint32)
    head = d[:8].copy(); consumed = int(head[4]) >> 8; head[4] &= 0xFF
    return head, d[8:].reshape(nodes_n, 4), consumed
def bad(steps):
    t0 = time.time(); head, nodes, consumed = run(steps); dt = time.time() - t0
    msg = None
    if int(head[3]) != 1 or int(head[4]) != 0: msg = "aux/fault"
    else:
        arrs, info = O.tree_run(O.ORACLE, "sqzo_tree_run", n, syms[:consumed])
        try: T.compare(which0, head, nodes, n, arrs, info)
        except AssertionError as e: msg = str(e)[:150].replace("\n", " ")
    print("steps", steps, "consumed", consumed, "head", [int(h) for h in head], "time %.2f" % dt, msg, flush=True)
    return msg, consumed
lo, hi = 1, 1
while True:
    m, c = bad(hi)
    if m is not None or c >= K: break
    lo = hi; hi *= 2
if m is not None:
    while hi - lo > 1:
        mid = (lo + hi) // 2
        m2, _ = bad(mid)
        if m2 is None: lo = mid
        else: hi = mid
    _, c0 = bad(lo); _, c1 = bad(hi)
    print("FIRST BAD STEP", hi, "consumes symbols", c0, "..", c1, [int(x) for x in syms[c0:c1]])
