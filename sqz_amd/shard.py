"""Multi-GPU: independent blocks shard across ranks.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo"
in the CPU tests).  Blocks are self-contained streams, so the data path needs no
collective: rank r owns the contiguous block range block_range(n, r, world).
The optional scatter / gather pair moves a root-resident batch out to the ranks and
the compressed streams back (ncclScatter / ncclGather shapes: one transfer per peer link,
no ring).

The codec is passed in as a callable, this module never touches kernels itself."""
import torch
import torch.distributed as dist


def block_range(n_blocks: int, rank: int, world: int):
    """[lo, hi) of the blocks rank `rank` owns; contiguous, sizes differ by <= 1."""
    return (n_blocks * rank) // world, (n_blocks * (rank + 1)) // world


def scatter_blocks(root_blocks, n_blocks: int, block_bytes: int, device, group=None, src=0):
    """root holds uint8[n_blocks*block_bytes]; every rank gets its own range.

    Ranges are padded to the largest range so one `scatter` (ncclScatter shape:
    equal slabs, rccl.h:767) serves all ranks."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    widest = max(block_range(n_blocks, r, world)[1] - block_range(n_blocks, r, world)[0]
                 for r in range(world))
    mine = torch.empty(widest * block_bytes, dtype=torch.uint8, device=device)
    chunks = None
    if rank == src:
        chunks = []
        for r in range(world):
            lo, hi = block_range(n_blocks, r, world)
            if hi - lo == widest:                    # the usual case (4096 / 1,2,4,8): no copy
                chunks.append(root_blocks[lo * block_bytes:hi * block_bytes])
                continue
            c = torch.zeros(widest * block_bytes, dtype=torch.uint8, device=device)
            c[:(hi - lo) * block_bytes] = root_blocks[lo * block_bytes:hi * block_bytes]
            chunks.append(c)
    dist.scatter(mine, chunks, src=src, group=group)
    lo, hi = block_range(n_blocks, rank, world)
    return mine[:(hi - lo) * block_bytes], (lo, hi)


def gather_slabs(local_slabs, local_sizes, n_blocks: int, slab_bytes: int, device, group=None, dst=0):
    """Fixed-stride compressed slabs + their sizes back to `dst` in block order.

    Returns (slabs uint8[n_blocks*slab_bytes], sizes int64[n_blocks]) on dst, (None, None)
    elsewhere."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    widest = max(block_range(n_blocks, r, world)[1] - block_range(n_blocks, r, world)[0]
                 for r in range(world))
    pad_slabs = torch.zeros(widest * slab_bytes, dtype=torch.uint8, device=device)
    pad_slabs[:local_slabs.numel()] = local_slabs
    pad_sizes = torch.zeros(widest, dtype=torch.int64, device=device)
    pad_sizes[:local_sizes.numel()] = local_sizes
    slabs_list = sizes_list = None
    if rank == dst:
        slabs_list = [torch.empty_like(pad_slabs) for _ in range(world)]
        sizes_list = [torch.empty_like(pad_sizes) for _ in range(world)]
    dist.gather(pad_slabs, slabs_list, dst=dst, group=group)
    dist.gather(pad_sizes, sizes_list, dst=dst, group=group)
    if rank != dst:
        return None, None
    slabs, sizes = [], []
    for r in range(world):
        lo, hi = block_range(n_blocks, r, world)
        slabs.append(slabs_list[r][:(hi - lo) * slab_bytes])
        sizes.append(sizes_list[r][:hi - lo])
    return torch.cat(slabs), torch.cat(sizes)


def dense_offsets(sizes):
    """int64[n] stream sizes -> int64[n+1] offsets of the dense image (every stream starts on an
    8-byte boundary: sizes are multiples of 8 unless a stream was cut short by E2BIG)."""
    aligned = (sizes + 7) // 8 * 8
    off = torch.zeros(sizes.numel() + 1, dtype=torch.int64, device=sizes.device)
    torch.cumsum(aligned, 0, out=off[1:])
    return off


def gather_dense(local_dense, local_sizes, n_blocks: int, device, group=None, dst=0):
    """Variable-sized compressed streams back to `dst` in block order, without shipping the
    worst-case slabs: every rank has packed its streams back to back (sqz_hip_pack_blocks);
    the ranks gather the sizes (ncclGather shape, rccl.h:745), agree on the largest dense
    image of any rank (all_reduce MAX) and gather the dense images padded to that length --
    independent streams of one batch compress to within a few percent of one another, so the
    padding is small -- and `dst` moves every rank's part to its place in the root image.
    Collectives only, the same call on every rank: a lone `send` on the peers against grouped
    receives on the root would meet different communicators under the nccl backend (single
    point-to-point calls use a two-rank communicator, grouped ones the group's own).
    local_dense: uint8[>= dense_offsets(local_sizes)[-1]].

    Returns (dense uint8[total], sizes int64[n_blocks], offsets int64[n_blocks+1]) on dst and
    (None, None, None) elsewhere; the triple is what sqz_hip_decode_blocks takes."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    widest = max(block_range(n_blocks, r, world)[1] - block_range(n_blocks, r, world)[0]
                 for r in range(world))
    pad_sizes = torch.zeros(widest, dtype=torch.int64, device=device)
    pad_sizes[:local_sizes.numel()] = local_sizes.to(device)
    sizes_list = [torch.empty_like(pad_sizes) for _ in range(world)] if rank == dst else None
    dist.gather(pad_sizes, sizes_list, dst=dst, group=group)
    total = int(dense_offsets(local_sizes)[-1])
    longest = int(max_over_ranks(float(total), device, group))       # (exact: far below 2^53)
    parts = None
    if longest > 0:                                                  # (every rank sees the same `longest`)
        mine = torch.empty(longest, dtype=torch.uint8, device=device)
        mine[:total] = local_dense[:total].to(device)
        parts = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
        dist.gather(mine, parts, dst=dst, group=group)
    if rank != dst:
        return None, None, None
    sizes = torch.cat([sizes_list[r][:block_range(n_blocks, r, world)[1] - block_range(n_blocks, r, world)[0]]
                       for r in range(world)])
    off = dense_offsets(sizes)
    host_off = off.cpu()
    dense = torch.empty(int(host_off[-1]), dtype=torch.uint8, device=device)
    for r in range(world):
        lo, hi = block_range(n_blocks, r, world)
        a, b = int(host_off[lo]), int(host_off[hi])
        if b > a:
            dense[a:b] = parts[r][:b - a]
    return dense, sizes, off


def max_over_ranks(value: float, device, group=None) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def min_over_ranks(value: float, device, group=None) -> float:
    """all ranks agree on the smallest value: an ok flag (1.0 / 0.0) after a phase that may fail on one
    rank alone, so that every rank takes or skips the next collective TOGETHER"""
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return float(t.item())


def sum_over_ranks(value: float, device, group=None) -> float:
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item())
