"""CPU, world_size 2, gloo: the N>1 path of bench.py / sqz_amd.shard -- block ranges,
scatter of the input batch, per-rank encode of the owned range, gather of the
fixed-stride slabs in block order, max-over-ranks timing.  The codec passed to the
sharding layer here is the oracle (the HIP path needs a GPU); shard.py itself is
codec-agnostic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from sqz_amd import shard

N_BLOCKS, BLOCK, WB, SLAB = 7, 2048, 10, 4096 + 1024


def test_block_range_partition():
    for n in (1, 7, 8, 4096):
        for world in (1, 2, 3, 8):
            spans = [shard.block_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard.block_range(4096, 3, 8) == (1536, 2048)      # 512 per GPU


def _worker(rank, world, port, q, N_BLOCKS=N_BLOCKS):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    root = None
    if rank == 0:
        root = torch.tensor(np.frombuffer(b"".join(O.zipf_block(b, BLOCK) for b in range(N_BLOCKS)),
                                          np.uint8).copy())
    mine, (lo, hi) = shard.scatter_blocks(root, N_BLOCKS, BLOCK, dev)
    assert (lo, hi) == shard.block_range(N_BLOCKS, rank, world)
    raw = mine.numpy().tobytes()
    slabs = torch.zeros((hi - lo) * SLAB, dtype=torch.uint8)
    sizes = torch.zeros(hi - lo, dtype=torch.int64)
    for k in range(hi - lo):
        blk = raw[k * BLOCK:(k + 1) * BLOCK]
        assert blk == O.zipf_block(lo + k, BLOCK)
        comp = O.encode(blk, WB, header=False)
        slabs[k * SLAB:k * SLAB + len(comp)] = torch.tensor(np.frombuffer(comp, np.uint8).copy())
        sizes[k] = len(comp)
    all_slabs, all_sizes = shard.gather_slabs(slabs, sizes, N_BLOCKS, SLAB, dev)
    # the dense flavour (what bench.py ships): streams back to back, 8-byte aligned starts
    off = shard.dense_offsets(sizes)
    dense = torch.zeros(int(off[-1]), dtype=torch.uint8)
    for k in range(hi - lo):
        dense[int(off[k]):int(off[k]) + int(sizes[k])] = slabs[k * SLAB:k * SLAB + int(sizes[k])]
    g_dense, g_sizes, g_off = shard.gather_dense(dense, sizes, N_BLOCKS, dev)
    slowest = shard.max_over_ranks(1.0 + rank, dev)
    total = shard.sum_over_ranks(float(sizes.sum()), dev)
    assert slowest == float(world)
    if rank == 0:
        got = []
        for b in range(N_BLOCKS):
            n = int(all_sizes[b])
            got.append(all_slabs[b * SLAB:b * SLAB + n].numpy().tobytes())
        assert torch.equal(g_sizes, all_sizes) and int(g_off[-1]) == g_dense.numel()
        got2 = [g_dense[int(g_off[b]):int(g_off[b]) + int(g_sizes[b])].numpy().tobytes() for b in range(N_BLOCKS)]
        assert got2 == got
        q.put((got, total))
    else:
        assert all_slabs is None and g_dense is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world,n_blocks", [(2, N_BLOCKS), (3, 2)])      # (3, 2): the root itself owns no block
def test_scatter_encode_gather_world2(world, n_blocks):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n_blocks)) for r in range(world)]
    for p in procs:
        p.start()
    got, total = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    want = [O.encode(O.zipf_block(b, BLOCK), WB, header=False) for b in range(n_blocks)]
    assert got == want                       # gathered results concatenate in block order
    assert total == float(sum(len(w) for w in want))
