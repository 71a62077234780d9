"""Batch path: n independent blocks, device resident (torch tensors hold the HBM).

torch is plumbing here -- it owns device memory and streams; every byte of codec
work happens in libsqz_amd.so (sqz_hip_* entry points of include/sqz/sqz.h).
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N
from .codec import SqzError, _raise


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def uniform_offsets(n_blocks: int, block_bytes: int, device="cuda"):
    return torch.arange(0, (n_blocks + 1) * block_bytes, block_bytes, dtype=torch.int64,
                        device=device)


def zipf_blocks(n_blocks: int, block_bytes: int, first_block: int = 0, device="cuda"):
    """BASELINE.json configs[2] workload, generated straight into HBM."""
    out = torch.empty(n_blocks * block_bytes, dtype=torch.uint8, device=device)
    _raise(N.lib().sqz_hip_zipf_blocks(_ptr(out), first_block, n_blocks, block_bytes, _stream()),
           "sqz_hip_zipf_blocks")
    return out


class Encoder:
    """Reusable device buffers for encode of up to (n_blocks, total_bytes)."""

    def __init__(self, n_blocks: int, total_bytes: int, out_capacity_per_block: int,
                 device="cuda"):
        L = N.lib()
        self.n = n_blocks
        self.device = device
        cap = (out_capacity_per_block + 7) & ~7
        self.out_off = uniform_offsets(n_blocks, cap, device)
        self.out = torch.empty(n_blocks * cap, dtype=torch.uint8, device=device)
        self.out_bytes = torch.zeros(n_blocks, dtype=torch.int64, device=device)
        self.err = torch.zeros(n_blocks, dtype=torch.int32, device=device)
        self.scratch_bytes = int(L.sqz_hip_encode_scratch_bytes(n_blocks, total_bytes))
        self.scratch = torch.empty(self.scratch_bytes, dtype=torch.uint8, device=device)

    def encode(self, d_in, in_off, window: int):
        _raise(N.lib().sqz_hip_encode_blocks(
            _ptr(d_in), _ptr(in_off), self.n, window, _ptr(self.out), _ptr(self.out_off),
            _ptr(self.out_bytes), _ptr(self.err), _ptr(self.scratch), self.scratch_bytes,
            _stream()), "sqz_hip_encode_blocks")
        return self.out, self.out_off, self.out_bytes, self.err

    def encode_stats(self, d_in, in_off, window: int):
        """encode + the reference's counters per block (sqz_block_stats):
        returns (out, out_off, out_bytes, err, list of dicts)."""
        st = torch.zeros(self.n * C.sizeof(N.BlockStats), dtype=torch.uint8, device=self.device)
        _raise(N.lib().sqz_hip_encode_blocks_stats(
            _ptr(d_in), _ptr(in_off), self.n, window, _ptr(self.out), _ptr(self.out_off),
            _ptr(self.out_bytes), _ptr(self.err), _ptr(self.scratch), self.scratch_bytes, _ptr(st),
            _stream()), "sqz_hip_encode_blocks_stats")
        torch.cuda.synchronize()
        raw = st.cpu().numpy().tobytes()
        res = []
        for b in range(self.n):
            s = N.BlockStats.from_buffer_copy(raw, b * C.sizeof(N.BlockStats))
            d = {k: int(getattr(s, k)) for k, _ in N.BlockStats._fields_[:12]}
            d["lit_entropy"] = float(N.lib().sqz_stats_entropy(s.lit_freq, 288))
            d["pos_entropy"] = float(N.lib().sqz_stats_entropy(s.pos_freq, 32))
            res.append(d)
        return self.out, self.out_off, self.out_bytes, self.err, res

    def tokens(self, d_in, in_off, window: int, finder: str = "index"):
        """stage 1 alone: (tokens int32[total], counts int32[n]).

        finder: "scan" = brute force as the reference writes it, "index" = same
        tokens through the sorted 3-byte-prefix index."""
        total = int(in_off[-1].item())
        toks = torch.zeros(total + 64, dtype=torch.int32, device=self.device)
        counts = torch.zeros(self.n, dtype=torch.int32, device=self.device)
        if finder == "scan":
            _raise(N.lib().sqz_hip_lz77_blocks(_ptr(d_in), _ptr(in_off), self.n, window,
                                                _ptr(toks), _ptr(counts), _stream()),
                   "sqz_hip_lz77_blocks")
        else:
            work = torch.empty(8 * (total + 64), dtype=torch.uint8, device=self.device)
            _raise(N.lib().sqz_hip_lz77_blocks_ex(_ptr(d_in), _ptr(in_off), self.n, window,
                                                   _ptr(toks), _ptr(counts), 1, _ptr(work),
                                                   work.numel(), _stream()),
                   "sqz_hip_lz77_blocks_ex")
            torch.cuda.synchronize()
        return toks, counts


def pack_blocks(slabs, slab_off, sizes, dense=None):
    """streams back to back (sqz_hip_pack_blocks): returns (dense uint8, offsets int64[n+1]).
    The pair (dense, offsets) is a valid input of decode_blocks."""
    from .shard import dense_offsets
    n = sizes.numel()
    off = dense_offsets(sizes)
    if dense is None:
        dense = torch.empty(int(off[-1]), dtype=torch.uint8, device=slabs.device)
    avg = max(slabs.numel() // max(n, 1) // 2, 1)
    _raise(N.lib().sqz_hip_pack_blocks(_ptr(slabs), _ptr(slab_off), _ptr(sizes), n, _ptr(dense),
                                       _ptr(off), avg, _stream()), "sqz_hip_pack_blocks")
    return dense, off


_decode_scratch = {}


def decode_blocks(d_comp, comp_off, n_blocks, d_out, out_off, err=None, scratch=None):
    """n streams -> d_out.  scratch: uint8 tensor of sqz_hip_decode_scratch_bytes()
    (kept per device between calls when not given)."""
    L = N.lib()
    if err is None:
        err = torch.zeros(n_blocks, dtype=torch.int32, device=d_out.device)
    need = int(L.sqz_hip_decode_scratch_bytes(n_blocks, d_out.numel()))
    if scratch is None:
        key = str(d_out.device)
        scratch = _decode_scratch.get(key)
        if scratch is None or scratch.numel() < need:
            scratch = torch.empty(need, dtype=torch.uint8, device=d_out.device)
            _decode_scratch[key] = scratch
    _raise(L.sqz_hip_decode_blocks(_ptr(d_comp), _ptr(comp_off), n_blocks, _ptr(d_out),
                                   _ptr(out_off), _ptr(err), _ptr(scratch), scratch.numel(),
                                   _stream()), "sqz_hip_decode_blocks")
    return err


def set_finder(name: str):
    """"index" (default) or "scan" (brute force, the reference's loop as written)."""
    N.lib().sqz_hip_set_finder(0 if name == "scan" else 1)


def set_timing(on: bool):
    N.lib().sqz_hip_set_timing(1 if on else 0)


def get_timing(reset=True):
    """{kernel name: (total ms, launches)} measured with HIP events on the launch stream."""
    t = N.Timing()
    N.lib().sqz_hip_get_timing(C.byref(t), 1 if reset else 0)
    return {N.KERNEL_NAMES[k]: (float(t.ms[k]), int(t.launches[k]))
            for k in range(len(N.KERNEL_NAMES)) if t.launches[k] > 0}


# ---- host-buffer flavour (numpy in / numpy out) ------------------------------
def encode_blocks_host(blocks, window: int, capacity=None):
    """blocks: list of bytes-like.  Returns (list of compressed bytes, err array)."""
    L = N.lib()
    n = len(blocks)
    sizes = [len(b) for b in blocks]
    in_off = np.zeros(n + 1, np.uint64)
    in_off[1:] = np.cumsum(sizes, dtype=np.uint64)
    data = np.frombuffer(b"".join(bytes(b) for b in blocks) or b"\0", np.uint8).copy()
    caps = [int(L.sqz_bound(s)) if capacity is None else (capacity + 7) & ~7 for s in sizes]
    out_off = np.zeros(n + 1, np.uint64)
    out_off[1:] = np.cumsum(caps, dtype=np.uint64)
    out = np.zeros(max(int(out_off[-1]), 1), np.uint8)
    out_bytes = np.zeros(n, np.uint64)
    err = np.zeros(n, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _raise(L.sqz_encode_blocks(p(data), p(in_off), n, window, p(out), p(out_off), p(out_bytes),
                               p(err)), "sqz_encode_blocks")
    res = [out[int(out_off[b]):int(out_off[b]) + int(out_bytes[b])].tobytes() for b in range(n)]
    return res, err


def decode_blocks_host(comps, sizes):
    L = N.lib()
    n = len(comps)
    in_off = np.zeros(n + 1, np.uint64)
    in_off[1:] = np.cumsum([len(c) for c in comps], dtype=np.uint64)
    data = np.frombuffer(b"".join(comps) or b"\0", np.uint8).copy()
    out_off = np.zeros(n + 1, np.uint64)
    out_off[1:] = np.cumsum(sizes, dtype=np.uint64)
    out = np.zeros(max(int(out_off[-1]), 1), np.uint8)
    err = np.zeros(n, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    _raise(L.sqz_decode_blocks(p(data), p(in_off), n, p(out), p(out_off), p(err)),
           "sqz_decode_blocks")
    return [out[int(out_off[b]):int(out_off[b + 1])].tobytes() for b in range(n)], err
