"""GPU, backend `nccl` (= RCCL on ROCm), world_size 1 on the one device of the box: every shard.py call of
BASELINE.json configs[3]'s transfer leg on DEVICE uint8 tensors -- scatter_blocks, gather_dense (the
sizes gather, the agreement on the longest dense image, the padded gather and the root's placement), gather_slabs,
max / min / sum_over_ranks -- around a HIP encode, the gathered streams compared with the oracle.
What two ranks add (real peer transfers) needs two GPUs; what this pins is that the calls, dtypes and
device tensors bench.py hands to RCCL are accepted by it (rccl.h:700-767).
The rank is a fresh interpreter (spawn) with a timeout, so a refused or hung collective fails the test
with a diagnosis instead of blocking the run."""
import os
import socket
import time

import pytest
import torch
import torch.multiprocessing as mp

import oracle_lib as O

pytestmark = pytest.mark.gpu

N_BLOCKS, BLOCK, WB = 9, 16384, 12


def _worker(port, q):
    import datetime
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev,
                            timeout=datetime.timedelta(seconds=120))
    import sqz_amd
    from sqz_amd import batch, shard
    assert "gfx950" in sqz_amd.device_info()["name"]
    root = batch.zipf_blocks(N_BLOCKS, BLOCK, device=dev)                       # device uint8, as in bench.py
    mine, (lo, hi) = shard.scatter_blocks(root, N_BLOCKS, BLOCK, dev)
    assert (lo, hi) == (0, N_BLOCKS) and mine.is_cuda and torch.equal(mine, root)
    n = hi - lo
    enc = batch.Encoder(n, n * BLOCK, sqz_amd.bound(BLOCK), device=dev)
    out, out_off, out_bytes, err = enc.encode(mine, batch.uniform_offsets(n, BLOCK, device=dev), 1 << WB)
    dense, off = batch.pack_blocks(out, out_off, out_bytes)
    torch.cuda.synchronize()
    assert err.tolist() == [0] * n
    g_dense, g_sizes, g_off = shard.gather_dense(dense, out_bytes, N_BLOCKS, dev)
    assert g_dense.is_cuda and g_sizes.is_cuda
    slabs, sizes = shard.gather_slabs(out, out_bytes, N_BLOCKS, out.numel() // n, dev)
    assert torch.equal(sizes, out_bytes) and torch.equal(slabs, out)
    total = shard.sum_over_ranks(float(out_bytes.sum()), dev)
    assert shard.max_over_ranks(1.25, dev) == 1.25 and shard.min_over_ranks(0.0, dev) == 0.0
    dist.barrier()
    torch.cuda.synchronize()
    img = g_dense.cpu().numpy()
    h_off, h_sizes = g_off.cpu(), g_sizes.cpu()
    streams = [img[int(h_off[b]):int(h_off[b]) + int(h_sizes[b])].tobytes() for b in range(N_BLOCKS)]
    back = torch.empty(N_BLOCKS * BLOCK, dtype=torch.uint8, device=dev)
    derr = batch.decode_blocks(g_dense, g_off, N_BLOCKS, back, batch.uniform_offsets(N_BLOCKS, BLOCK, device=dev))
    torch.cuda.synchronize()
    q.put((streams, total, derr.tolist(), bool(torch.equal(back, root)), dist.get_backend()))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_shard_calls_over_rccl_world1():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    p = ctx.Process(target=_worker, args=(port, q))
    p.start()
    deadline = time.time() + 420
    while q.empty():
        assert p.is_alive() or not q.empty(), f"the nccl worker exited without a result (exit code {p.exitcode})"
        if time.time() > deadline:
            p.kill()
            pytest.fail("no result from the nccl worker within 420 s (a collective hung?)")
        time.sleep(0.2)
    streams, total, derr, same, backend = q.get()
    p.join(120)
    assert p.exitcode == 0
    assert backend == "nccl"
    want = [O.encode(O.zipf_block(b, BLOCK), WB, header=False) for b in range(N_BLOCKS)]
    assert streams == want
    assert total == float(sum(len(w) for w in want))
    assert derr == [0] * N_BLOCKS and same
