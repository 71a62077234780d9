"""debug helper (GPU box): first token mismatch between the HIP scan and the oracle."""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib as O
import sqz_amd
from sqz_amd import batch

def show(t):
    t = int(t)
    return f"M(len={(t>>16)&0x1ff},dist={t&0x7fff})" if t & 0x80000000 else f"L({t})"

name, wb = sys.argv[1], int(sys.argv[2])
data = O.corpus(name)
if len(sys.argv) > 3:
    data = data[:int(sys.argv[3])]
d_in = torch.tensor(np.frombuffer(data, np.uint8), device="cuda")
off = torch.tensor([0, len(data)], dtype=torch.int64, device="cuda")
enc = batch.Encoder(1, len(data), sqz_amd.bound(len(data)))
toks, counts = enc.tokens(d_in, off, 1 << wb)
torch.cuda.synchronize()
got = toks.cpu().numpy().view(np.uint32)[:int(counts[0])]
want = O.tokens(data, 1 << wb)
print("counts", len(got), len(want))
pos = 0
for k in range(min(len(got), len(want))):
    if got[k] != want[k]:
        print("first mismatch at token", k, "pos", pos, "got", show(got[k]), "want", show(want[k]))
        print("context", data[pos:pos+24])
        break
    t = int(want[k]); pos += ((t >> 16) & 0x1ff) if t & 0x80000000 else 1
else:
    print("tokens equal")
out, out_off, out_bytes, err = enc.encode(d_in, off, 1 << wb)
torch.cuda.synchronize()
g = out[:int(out_bytes[0])].cpu().numpy().tobytes()
w = O.encode(data, wb, header=False)
print("bytes", len(g), len(w), "equal" if g == w else "DIFFER")
if g != w:
    for k in range(min(len(g), len(w))):
        if g[k] != w[k]:
            print("first byte diff at", k); break
