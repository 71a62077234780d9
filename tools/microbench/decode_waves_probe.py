"""tools/microbench/decode_waves_probe.py -- entropy decode time of N equal-length streams of one kind
(uniform random bytes or Zipf blocks) for N = 1, 8, 64, 256: tells a per-stream cost from a placement /
co-residency effect when comparing SQZ_DECODE_WAVES settings.  (needs an MI355X)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, sqz_amd
from sqz_amd import batch
BB = 262144
kind = sys.argv[1] if len(sys.argv) > 1 else "random"
for N in [int(x) for x in os.environ.get("PROBE_N", "1 8 64 256").split()]:
    if kind == "random":
        d_in = torch.tensor(np.random.default_rng(1).integers(0, 256, N * BB, dtype=np.uint8), device="cuda")
    else:
        d_in = batch.zipf_blocks(N, BB)
    off = batch.uniform_offsets(N, BB)
    enc = batch.Encoder(N, N * BB, sqz_amd.bound(BB))
    out, out_off, out_bytes, err = enc.encode(d_in, off, 1 << 15)
    back = torch.empty_like(d_in)
    derr = torch.zeros(N, dtype=torch.int32, device="cuda")
    batch.decode_blocks(out, out_off, N, back, off, derr); torch.cuda.synchronize()
    batch.set_timing(True); batch.get_timing(reset=True)
    batch.decode_blocks(out, out_off, N, back, off, derr); torch.cuda.synchronize()
    tim = batch.get_timing(reset=True); batch.set_timing(False)
    assert int(derr.abs().sum()) == 0 and bool((back == d_in).all())
    print(kind, "streams", N, "waves", os.environ.get("SQZ_DECODE_WAVES", "auto"),
          " ".join(f"{k.replace('_kernel','')}={v[0]:.2f}" for k, v in tim.items() if v[1]), flush=True)
