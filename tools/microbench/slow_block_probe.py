"""tools/microbench/slow_block_probe.py -- which streams of a uniform-random batch are slow to decode?
Encodes 64 blocks, then decodes them one at a time (python tools/microbench/slow_block_probe.py, needs an MI355X)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, sqz_amd
from sqz_amd import batch
BB, N = 262144, 64
arr = np.random.default_rng(1).integers(0, 256, N * BB, dtype=np.uint8)
one_off = batch.uniform_offsets(1, BB)
enc1 = batch.Encoder(1, BB, sqz_amd.bound(BB))
for b in range(N):
    d = torch.tensor(arr[b * BB:(b + 1) * BB], device="cuda")
    batch.set_timing(True); batch.get_timing(reset=True)
    out, out_off, out_bytes, err, st = enc1.encode_stats(d, one_off, 1 << 15)
    o2, oo2, ob2, e2 = enc1.encode(d, one_off, 1 << 15)
    back = torch.empty_like(d)
    derr = torch.zeros(1, dtype=torch.int32, device="cuda")
    batch.decode_blocks(o2, oo2, 1, back, one_off, derr); torch.cuda.synchronize()
    tim = batch.get_timing(reset=True); batch.set_timing(False)
    s = st[0]
    print(b, "emit", round(tim["huffman_emit_kernel"][0] / tim["huffman_emit_kernel"][1], 1), "decode", round(tim["entropy_decode_kernel"][0], 1),
          "tokens", s["tokens"], "depth", s["lit_depth"], s["pos_depth"], "swaps", s["lit_swaps"], "moves", s["lit_moves"], "upd", s["lit_updates"],
          "bytes", int(ob2[0]), flush=True)
