import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch, sqz_amd
from sqz_amd import batch
BB, N = 262144, 64
d_in = torch.tensor(np.random.default_rng(1).integers(0, 256, N * BB, dtype=np.uint8), device="cuda")
off = batch.uniform_offsets(N, BB)
enc = batch.Encoder(N, N * BB, sqz_amd.bound(BB))
out, out_off, out_bytes, err, st = enc.encode_stats(d_in, off, 1 << 15)
import collections
for b, s in enumerate(st):
    print(b, s["tokens"], "lit depth", s["lit_depth"], "pos depth", s["pos_depth"], "lit swaps", s["lit_swaps"], "moves", s["lit_moves"], "updates", s["lit_updates"], "pos swaps", s["pos_swaps"], s["pos_moves"], s["pos_updates"])
