"""development aid: the counters kernel on two small blocks (needs an MI355X)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import oracle_lib as O
import sqz_amd
from sqz_amd import batch
datas = [O.zipf_block(0, 4096), O.zipf_block(1, 4096)]
n = 2
off = torch.tensor([0, 4096, 8192], dtype=torch.int64, device="cuda")
d_in = torch.tensor(np.frombuffer(b"".join(datas), np.uint8).copy(), device="cuda")
enc = batch.Encoder(n, 8192, sqz_amd.bound(4096))
print("plain", enc.encode(d_in, off, 1 << 12)[3].tolist(), flush=True)
torch.cuda.synchronize()
out, out_off, out_bytes, err, stats = enc.encode_stats(d_in, off, 1 << 12)
print("stats", err.tolist(), stats[0], flush=True)
