"""GPU, world_size 2 on the one device of the box: BASELINE.json configs[3]'s data path with
the HIP codec in the middle -- scatter of the root batch's block ranges, sqz_hip_encode_blocks
on every rank, sqz_hip_pack_blocks, gather of sizes + dense streams back to rank 0 in block
order -- and every gathered stream compared with the oracle; rank 0 then decodes the gathered
image on the device.  Two ranks cannot share a GPU under RCCL, so the collectives run over
gloo on host tensors (the rehearsal pattern of tools/rehearse_n2.sh); bench.py runs the same
shard.py calls over nccl on the 8-GPU node.  The ranks are fresh interpreters (spawn) that
initialise the GPU themselves."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import oracle_lib as O

pytestmark = pytest.mark.gpu

N_BLOCKS, BLOCK, WB = 13, 16384, 12          # 13 over 2 ranks: 6 + 7, the padded-range path


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sqz_amd
    from sqz_amd import batch, shard
    assert "gfx950" in sqz_amd.device_info()["name"]          # the HIP library, loudly
    dev, cpu = torch.device("cuda", 0), torch.device("cpu")
    root = None
    if rank == 0:                                              # the batch lives on rank 0
        root = batch.zipf_blocks(N_BLOCKS, BLOCK, device=dev).cpu()
        assert root.numpy().tobytes() == b"".join(O.zipf_block(b, BLOCK) for b in range(N_BLOCKS))
    mine, (lo, hi) = shard.scatter_blocks(root, N_BLOCKS, BLOCK, cpu)
    n = hi - lo
    d_in = mine.to(dev)
    enc = batch.Encoder(n, n * BLOCK, sqz_amd.bound(BLOCK), device=dev)
    out, out_off, out_bytes, err = enc.encode(d_in, batch.uniform_offsets(n, BLOCK, device=dev), 1 << WB)
    dense, off = batch.pack_blocks(out, out_off, out_bytes)
    torch.cuda.synchronize()
    assert err.tolist() == [0] * n
    g_dense, g_sizes, g_off = shard.gather_dense(dense, out_bytes, N_BLOCKS, cpu)
    total = shard.sum_over_ranks(float(out_bytes.sum()), cpu)
    if rank == 0:
        img = g_dense.numpy()
        streams = [img[int(g_off[b]):int(g_off[b]) + int(g_sizes[b])].tobytes() for b in range(N_BLOCKS)]
        # and back: the gathered image is what the decoder takes
        back = torch.empty(N_BLOCKS * BLOCK, dtype=torch.uint8, device=dev)
        derr = batch.decode_blocks(g_dense.to(dev), g_off.to(dev), N_BLOCKS, back,
                                   batch.uniform_offsets(N_BLOCKS, BLOCK, device=dev))
        torch.cuda.synchronize()
        q.put((streams, total, derr.tolist(), bool(torch.equal(back.cpu(), root))))
    else:
        assert g_dense is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_scatter_hip_encode_gather_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    # rank 0's result is read BEFORE the joins (a full pipe would block its put) -- but by polling: a
    # worker that dies before its put must fail the test with its exit code, not block it for ever
    import time
    deadline = time.time() + 540
    while q.empty():
        dead = [(r, p.exitcode) for r, p in enumerate(procs) if not p.is_alive() and p.exitcode != 0]
        assert not dead, f"worker(s) exited without a result: (rank, exit code) {dead}"
        assert any(p.is_alive() for p in procs) or not q.empty(), "workers exited cleanly without a result"
        assert time.time() < deadline, "no result from rank 0 within 540 s"
        time.sleep(0.2)
    streams, total, derr, same = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    want = [O.encode(O.zipf_block(b, BLOCK), WB, header=False) for b in range(N_BLOCKS)]
    assert streams == want                      # every gathered stream is the oracle's, in block order
    assert total == float(sum(len(w) for w in want))
    assert derr == [0] * N_BLOCKS and same
