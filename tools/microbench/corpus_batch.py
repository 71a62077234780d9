"""tools/microbench/corpus_batch.py -- sanity timing on non-synthetic data: blocks cut from the corpus
files (text, executables, a bitmap), plus degenerate ones (zeros, a short period), through the
device-resident encode / decode.  Not the benchmark; looks for pathological slow-downs.
(python tools/microbench/corpus_batch.py, needs an MI355X)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, sqz_amd
from sqz_amd import batch

BB, N = 262144, int(os.environ.get("CORPUS_BLOCKS", "1024"))
ONLY = os.environ.get("CORPUS_ONLY")          # comma-separated case names
def blocks_from(data, n):
    data = np.frombuffer(data, np.uint8)
    reps = (n * BB + len(data) - 1) // len(data)
    return np.tile(data, reps)[:n * BB]

cases = {}
for f in ("confucius.txt", "laozi.txt", "x64.elf", "arm64.elf", "mandrill.bmp"):
    cases[f] = blocks_from(open(os.path.join(ROOT, "tests", "corpus", f), "rb").read(), N)
cases["zeros"] = np.zeros(N * BB, np.uint8)
cases["period 3"] = blocks_from(b"abc", N)
rng = np.random.default_rng(1)
cases["uniform random"] = rng.integers(0, 256, N * BB, dtype=np.uint8)
off = batch.uniform_offsets(N, BB)
enc = batch.Encoder(N, N * BB, sqz_amd.bound(BB))
for name, arr in cases.items():
    if ONLY and name not in ONLY.split(","):
        continue
    d_in = torch.tensor(arr, device="cuda")
    back = torch.empty_like(d_in)
    derr = torch.zeros(N, dtype=torch.int32, device="cuda")
    enc.encode(d_in, off, 1 << 15); torch.cuda.synchronize()
    batch.set_timing(True); batch.get_timing(reset=True)
    t0 = time.perf_counter()
    out, out_off, out_bytes, err = enc.encode(d_in, off, 1 << 15); torch.cuda.synchronize()
    t1 = time.perf_counter()
    batch.decode_blocks(out, out_off, N, back, off, derr); torch.cuda.synchronize()
    t2 = time.perf_counter()
    tim = batch.get_timing(reset=True); batch.set_timing(False)
    ok = int(err.abs().sum()) == 0 and int(derr.abs().sum()) == 0 and bool((back == d_in).all())
    ratio = float(out_bytes.sum()) / (N * BB)
    print(f"{name:16s} ratio {ratio:6.3f} encode {N*BB/(t1-t0)/1e6:8.0f} MB/s decode {N*BB/(t2-t1)/1e6:8.0f} MB/s round trip {'ok' if ok else 'FAILED'}  "
          + " ".join(f"{k.replace('_kernel','')}={v[0]:.1f}" for k, v in tim.items() if v[1]), flush=True)
