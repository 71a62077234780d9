"""tools/microbench/time_stage1.py -- time the stage-1 kernels alone (sort / match / parse) on the
bench batch through sqz_hip_lz77_blocks_ex (python tools/microbench/time_stage1.py, needs an MI355X)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, sqz_amd
from sqz_amd import batch
n, bb = 4096, 262144
d_in = batch.zipf_blocks(n, bb)
off = batch.uniform_offsets(n, bb)
enc = batch.Encoder(n, n * bb, sqz_amd.bound(bb))
from sqz_amd import _native as N
L = N.lib()
total = n * bb
toks = torch.zeros(total + 64, dtype=torch.int32, device="cuda")
counts = torch.zeros(n, dtype=torch.int32, device="cuda")
work = torch.zeros(8 * (total + 64), dtype=torch.uint8, device="cuda")   # zeroed: a valid all-literal match table
for rep in range(3):
    work.zero_()
    torch.cuda.synchronize()
    batch.set_timing(True); batch.get_timing(reset=True)
    rc = L.sqz_hip_lz77_blocks_ex(d_in.data_ptr(), off.data_ptr(), n, 1 << 15, toks.data_ptr(), counts.data_ptr(),
                                  1, work.data_ptr(), work.numel(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == 0, rc
    t = batch.get_timing(reset=True); batch.set_timing(False)
    print({k: round(v[0] / max(v[1], 1), 2) for k, v in t.items() if v[1]}, "tokens", int(counts.to(torch.int64).sum()))
