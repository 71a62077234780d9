#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one encode pass (LZ77 match finding + greedy parse + adaptive-Huffman emit) of
the hot path over one batch of synthetic input that is already resident in HBM.

  N = 1   BASELINE.json configs[2]: 4096 x 256 KB Zipf(s=1) blocks, window 32 KB, one
          independent stream per block.
  N > 1   BASELINE.json configs[3]: THE SAME 4096-block batch (global block ids 0..4095),
          rank r owning the contiguous range shard.block_range(4096, r, N) -- 512 blocks per
          GPU at N = 8.  Strong scaling: total work is fixed, `value` = the batch's
          uncompressed MB (10^6 B) / the slowest rank's time.  Blocks are self-contained
          streams, so the timed data path has no collective.  `with_transfer` repeats the
          measurement with the batch starting and ending on rank 0: RCCL scatter of the
          input ranges, encode, pack, gather of sizes + dense streams back in block order
          (SURVEY.md section 8d config 4 asks for both figures).
  --scaling weak  keeps the round-1 measurement (every rank its own --blocks batch).

  --codec rc   measures SURVEY.md section 8f-1 instead: the reference's HEAD ("R-era") adaptive range
               coder (src/sqz.c:506-548,717-743; literal-only, the finders are compiled out at HEAD)
               on the same batch, with its own roofline (rc_encode_kernel) and CPU baseline
               (oracle/_ref/libsqz_ref_rc.so).  Not BASELINE.json's metric: a secondary line.

The decode pass over the produced streams is timed right after, with the same K and W, and
reported in the same line (`decode_MBps`), together with
  roofline      the dominant kernel of the encode step (longest average launch, HIP
                events on the launch stream, measured live in this process) vs the HBM roof
  cpu_baseline  the reference itself (oracle/_ref, built in the build container and
                shipped as a .so) or, if absent, the oracle restatement, timed
                single-thread on a bounded sample of the same workload (rank 0, N=1)
"""
import argparse
import datetime
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(n_sample_blocks, block_bytes, win_bits):
    """single-thread CPU encode/decode MB/s on blocks [0, n_sample_blocks)."""
    import ctypes as C
    import oracle_lib as O       # checker only: TEST INFRASTRUCTURE
    kind = "reference" if O.REF is not None else "port"
    enc_s = dec_s = 0.0
    total = 0
    for b in range(n_sample_blocks):
        data = O.zipf_block(b, block_bytes)
        t0 = time.perf_counter()
        comp = O.ref_compress(data, win_bits, False) if kind == "reference" \
            else O.encode(data, win_bits, header=False)
        enc_s += time.perf_counter() - t0
        t0 = time.perf_counter()
        if kind == "reference":
            out = C.create_string_buffer(block_bytes)
            n = C.c_uint64(block_bytes)
            e = O.REF.sqz_ref_decompress(comp, len(comp), 0, out, block_bytes, C.byref(n), None)
            back = out.raw
        else:
            e, back, _ = O.decode(comp, header=False, nbytes=block_bytes)
        dec_s += time.perf_counter() - t0
        assert e == 0 and back == data
        total += block_bytes
    return {"value": round(total / enc_s / 1e6, 5), "unit": "MB/s", "cores": 1, "kind": kind,
            "decode_value": round(total / dec_s / 1e6, 3),
            "sample": f"Zipf blocks 0..{n_sample_blocks - 1} of the same workload "
                      f"({n_sample_blocks} x {block_bytes} B, window 2^{win_bits}), "
                      f"encode {enc_s:.1f} s + decode {dec_s:.2f} s on one host core "
                      f"of {os.cpu_count()}"}


def kernel_build_id():
    """identifies the kernels a traffic figure was measured on: sha256 of the device sources"""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "sqz_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        with open(os.path.join(csrc, name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


# Kernels whose reads are ONE coalesced pass over a stream of known length, and whose raw FETCH_SIZE is half of that
# length on gfx950 (emit: 1.55 GB counted for 3.07 GB of token words; entropy decode: 0.48 GB for 0.92 GB of
# streams; the R-era kernels carry the doubling in their own file, tools/r03_final.sh): their FETCH_SIZE is doubled.
HALVED_FETCH = ("huffman_emit_kernel", "entropy_decode_kernel")


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the last committed PMC run (profiles/traffic.json,
    written by tools/profile_round.sh) -- but only if it was measured on THESE kernels."""
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tp):
        return None, "no profiles/traffic.json"
    with open(tp) as fh:
        t = json.load(fh)
    cur = t.get("current", {})
    if cur.get("kernel_build_id") != kernel_build_id():
        return None, (f"profiles/traffic.json was measured on build {cur.get('kernel_build_id')}, "
                      f"this is {kernel_build_id()}: not reported")
    ks = cur.get("kernels", {})
    # rocprofv3 prints template instances as "void name<args>"
    k = ks.get(kernel) or next((v for n, v in ks.items() if n.split("<")[0].split()[-1] == kernel), None)
    if not k:
        return None, "kernel not in profiles/traffic.json"
    if kernel in HALVED_FETCH and "hbm_bytes_per_launch_if_fetch_doubled" in k:
        return int(k["hbm_bytes_per_launch_if_fetch_doubled"]), (
            "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (--kernel-trace only), KiB -> bytes, "
            "per launch; FETCH_SIZE DOUBLED: this kernel's reads are one coalesced pass over a stream of known length "
            "and the raw counter is half of it (MI355X_MICROARCH.md, HBM section: gfx950 reports half the bytes of "
            "coalesced streaming reads); WRITE_SIZE exact")
    return int(k["hbm_bytes_per_launch"]), cur.get("method")


def cpu_baseline_rc(n_sample_blocks, block_bytes):
    """single-thread CPU MB/s of the R-era coder on blocks [0, n_sample_blocks): the reference's own
    src/sqz.c (oracle/_ref/libsqz_ref_rc.so) when it was built, else the oracle restatement."""
    import ctypes as C
    import oracle_lib as O       # checker only: TEST INFRASTRUCTURE
    path = os.path.join(O.ODIR, "_ref", "libsqz_ref_rc.so")
    ref = C.CDLL(path) if os.path.exists(path) else None
    kind = "reference" if ref is not None else "port"
    enc_s = dec_s = 0.0
    total = 0
    for b in range(n_sample_blocks):
        data = O.zipf_block(b, block_bytes)
        cap = 2 * block_bytes + 64
        out = C.create_string_buffer(cap)
        n = C.c_uint64()
        t0 = time.perf_counter()
        if ref is not None:
            e = ref.sqz_ref_rc_compress(data, C.c_uint64(block_bytes), C.c_uint32(1 << 15), out, C.c_uint64(cap), C.byref(n))
        else:
            e = O.ORACLE.sqzo_rc_encode(data, C.c_uint64(block_bytes), out, C.c_uint64(cap), C.byref(n))
        enc_s += time.perf_counter() - t0
        assert e == 0
        back = C.create_string_buffer(block_bytes)
        nb, cons = C.c_uint64(), C.c_uint64()
        t0 = time.perf_counter()
        if ref is not None:
            e = ref.sqz_ref_rc_decompress(out, C.c_uint64(n.value), back, C.c_uint64(block_bytes), C.byref(nb), C.byref(cons))
        else:
            e = O.ORACLE.sqzo_rc_decode(out, C.c_uint64(n.value), back, C.c_uint64(block_bytes), C.byref(nb), C.byref(cons))
        dec_s += time.perf_counter() - t0
        assert e == 0 and back.raw == data
        total += block_bytes
    return {"value": round(total / enc_s / 1e6, 4), "unit": "MB/s", "cores": 1, "kind": kind,
            "decode_value": round(total / dec_s / 1e6, 4),
            "sample": f"Zipf blocks 0..{n_sample_blocks - 1} of the same workload ({n_sample_blocks} x {block_bytes} B), "
                      f"encode {enc_s:.2f} s + decode {dec_s:.2f} s on one host core of {os.cpu_count()}"}


def run_rc(args, torch, dist, batch, shard, info, dev, comm_dev, rank, world, lo, n, total_blocks, strong):
    """the R-era coder (src/sqz.c:506-548,717-743) on the same batch.  One wave per
    stream; encode = rc_encode_kernel, decode = rc_decode_kernel; same K / W / barrier contract."""
    import ctypes as C
    from sqz_amd import _native as N
    L = N.lib()
    bb = args.block_bytes
    d_in = batch.zipf_blocks(n, bb, first_block=lo, device=dev)
    in_off = batch.uniform_offsets(n, bb, device=dev)
    cap = (int(L.sqz_rc_bound(bb)) + 7) // 8 * 8
    out_off = batch.uniform_offsets(n, cap, device=dev)
    out = torch.empty(n * cap, dtype=torch.uint8, device=dev)
    ob = torch.zeros(n, dtype=torch.int64, device=dev)
    err = torch.zeros(n, dtype=torch.int32, device=dev)
    back = torch.empty_like(d_in)
    db = torch.zeros(n, dtype=torch.int64, device=dev)
    cons = torch.zeros(n, dtype=torch.int64, device=dev)
    derr = torch.zeros(n, dtype=torch.int32, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def enc_step():
        assert L.sqz_hip_rc_encode_blocks(P(d_in), P(in_off), n, P(out), P(out_off), P(ob), P(err), st()) == 0

    def dec_step():
        assert L.sqz_hip_rc_decode_blocks(P(out), P(out_off), n, P(back), P(in_off), P(db), P(cons), P(derr), st()) == 0

    for _ in range(args.warmup):
        enc_step()
    barrier()
    batch.set_timing(True)
    batch.get_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        enc_step()
    barrier()
    enc_s = time.perf_counter() - t0
    tim = batch.get_timing(reset=True)
    assert int(err.abs().sum()) == 0, "rc encode reported errors"
    for _ in range(args.warmup):
        dec_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dec_step()
    barrier()
    dec_s = time.perf_counter() - t0
    dtim = batch.get_timing(reset=True)
    batch.set_timing(False)
    assert int(derr.abs().sum()) == 0, "rc decode reported errors"
    if not args.no_verify:
        assert torch.equal(back, d_in), "rc round trip differs"
        assert torch.equal(cons, ob), "rc decoder did not consume exactly the stream"
    comp_bytes = int(ob.sum().item())
    if world > 1:
        enc_s = shard.max_over_ranks(enc_s, comm_dev)
        dec_s = shard.max_over_ranks(dec_s, comm_dev)
        comp_total = shard.sum_over_ranks(float(comp_bytes), comm_dev)
    else:
        comp_total = float(comp_bytes)
    if rank != 0:
        return
    in_total = float(total_blocks) * bb
    enc_ms = tim["rc_encode_kernel"][0] / max(tim["rc_encode_kernel"][1], 1)
    dec_ms = dtim["rc_decode_kernel"][0] / max(dtim["rc_decode_kernel"][1], 1)
    algo = n * bb + comp_bytes                   # every input byte read once, every output byte written once
    achieved = algo / (enc_ms * 1e-3) / 1e9
    rc_traffic, rc_traffic_note = measured_traffic("rc_encode_kernel")   # (tools/r03_final.sh PART=rc adds the R-era kernels)
    line = {
        "metric": "R-era range coder (SURVEY.md 8f-1): encode MB/s + decode MB/s, batched blocks",
        "value": round(in_total / enc_s * args.steps / 1e6, 3), "unit": "MB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(enc_s / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{total_blocks} x {bb} B Zipf(s=1) byte blocks, one independent stream per block, "
                               "adaptive order-0 range coder of the reference's HEAD (literal-only as HEAD's encoder "
                               "runs, src/sqz.c:590-596,717-743) -- NOT BASELINE.json's metric, a secondary line",
                   "total_blocks": total_blocks, "blocks_per_gpu": n, "block_bytes": bb, "codec": "rc",
                   "device": info["name"]},
        "decode_MBps": round(in_total / dec_s * args.steps / 1e6, 3),
        "decode_ms_per_step": round(dec_s / args.steps * 1e3, 3),
        "compressed_ratio": round(comp_total / in_total, 5),
        "kernels_ms": {"rc_encode_kernel": round(enc_ms, 3), "rc_decode_kernel": round(dec_ms, 3)},
        "roofline": {"bound": "hbm", "kernel": "rc_encode_kernel", "achieved": round(achieved, 4),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 7),
                     "traffic": rc_traffic, "traffic_note": rc_traffic_note,
                     "algorithmic_bytes_per_launch": algo,
                     "decode_frac_of_hbm_roof": round(algo / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 7)},
    }
    if world == 1 and args.cpu_blocks > 0:
        line["cpu_baseline"] = cpu_baseline_rc(args.cpu_blocks, bb)
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=4096,
                    help="blocks of the batch (strong: in total, shared by all ranks; weak: per GPU)")
    ap.add_argument("--block-bytes", type=int, default=262144)
    ap.add_argument("--win-bits", type=int, default=15)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N>1: strong = the same --blocks batch sharded over the ranks (configs[3]); "
                         "weak = every rank its own --blocks batch")
    ap.add_argument("--no-transfer", action="store_true",
                    help="N>1, strong: skip the scatter/gather-inclusive measurement")
    ap.add_argument("--cpu-blocks", type=int, default=8, help="CPU baseline sample: blocks 0..7 (SURVEY.md 8d), about 26 s of one host core (0 = skip)")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--codec", choices=["huffman", "rc"], default="huffman",
                    help="huffman = BASELINE.json's path (LZ77 + adaptive Huffman, H0); rc = the R-era range coder "
                         "of HEAD (SURVEY.md section 8f-1), a secondary line")
    ap.add_argument("--finder", choices=["index", "scan"], default="index",
                    help="stage-1 match finder: index (default) or the brute-force scan")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from sqz_amd import batch, shard
    import sqz_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    # one rank per GPU; the modulo only matters when ranks are rehearsed on fewer devices
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    backend = "none"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SQZ_BENCH_BACKEND", "nccl")      # nccl == RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group(backend)
    # collectives run on device tensors over RCCL; a gloo rehearsal stages them through the host
    comm_dev = dev if backend in ("none", "nccl") else torch.device("cpu")

    bb, wb = args.block_bytes, args.win_bits
    strong = args.scaling == "strong"
    total_blocks = args.blocks if strong else args.blocks * world
    lo, hi = shard.block_range(total_blocks, rank, world) if strong \
        else (rank * args.blocks, (rank + 1) * args.blocks)
    n = hi - lo                                    # blocks this rank owns
    info = sqz_amd.device_info()
    if args.codec == "rc":
        run_rc(args, torch, dist, batch, shard, info, dev, comm_dev, rank, world, lo, n, total_blocks, strong)
        if world > 1:
            dist.destroy_process_group()
        return
    batch.set_finder(args.finder)

    # ---- synthetic input, generated straight into HBM ------------------------
    d_in = batch.zipf_blocks(n, bb, first_block=lo, device=dev)
    in_off = batch.uniform_offsets(n, bb, device=dev)
    enc = batch.Encoder(n, n * bb, sqz_amd.bound(bb), device=dev)
    d_back = torch.empty_like(d_in)
    derr = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def encode_step():
        return enc.encode(d_in, in_off, 1 << wb)

    def decode_step(out, out_off):
        batch.decode_blocks(out, out_off, n, d_back, in_off, derr)

    # ---- encode: W warmup + K timed steps -------------------------------------
    for _ in range(args.warmup):
        encode_step()
    barrier()
    batch.set_timing(True)
    batch.get_timing(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, out_off, out_bytes, err = encode_step()
    barrier()
    enc_s = time.perf_counter() - t0
    tim = batch.get_timing(reset=True)
    batch.set_timing(False)
    assert int(err.abs().sum()) == 0, "encode reported errors"
    # ---- secondary figures: what the serial stages actually chew through.
    # tok_count[n] is the head of the encode scratch (include/sqz/sqz.h); block 0's token
    # words follow at the 256-byte-aligned offset.  Not timed.
    tok_counts = enc.scratch[:4 * n].view(torch.int32)
    tokens_rank = int(tok_counts.to(torch.int64).sum())
    head = (4 * n + 255) // 256 * 256
    t0c = int(tok_counts[0])
    tw = enc.scratch[head:head + 4 * t0c].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    tlen = torch.where((tw >> 31) != 0, (tw >> 16) & 0x1FF, torch.ones_like(tw))
    tpos = torch.cumsum(tlen, 0) - tlen                      # start position of every token
    cand0 = int(torch.clamp(tpos, max=(1 << wb) - 1).sum())  # scan: min(i, window-1) candidates per token
    comp_bytes = int(out_bytes.sum().item())

    # ---- decode: same K / W ----------------------------------------------------
    for _ in range(args.warmup):
        decode_step(out, out_off)
    barrier()
    batch.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        decode_step(out, out_off)
    barrier()
    dec_s = time.perf_counter() - t0
    dtim = batch.get_timing(reset=True)
    batch.set_timing(False)
    assert int(derr.abs().sum()) == 0, "decode reported errors"
    if not args.no_verify:
        assert torch.equal(d_back, d_in), "round trip differs"

    if world > 1:
        enc_s = shard.max_over_ranks(enc_s, comm_dev)
        dec_s = shard.max_over_ranks(dec_s, comm_dev)
        comp_total = shard.sum_over_ranks(float(comp_bytes), comm_dev)
        tokens_all = int(shard.sum_over_ranks(float(tokens_rank), comm_dev))
    else:
        comp_total = float(comp_bytes)
        tokens_all = tokens_rank

    # ---- N>1, strong: the same step with the batch starting and ending on rank 0 --------
    # (after every collective the device-local line needs: a failure here cannot take that line down).
    # A rank that raised must not leave its peers blocked inside the next collective, so the leg is cut
    # into PHASES: what can fail on one rank alone (allocation, the local encode, the checks) runs between
    # collectives inside phase(); every phase ends with an all_reduce(MIN) of an ok flag, and once any
    # rank has failed ALL ranks skip the remaining collectives of the leg together.
    xfer = None
    if world > 1 and strong and not args.no_transfer:
        state = {"ok": True, "why": ""}

        def phase(fn):
            """run fn on every rank that is still ok, then agree on the outcome"""
            res = None
            if state["ok"]:
                try:
                    res = fn()
                except Exception as ex:                      # noqa: BLE001 -- reported in the line
                    state["ok"] = False
                    state["why"] = f"{type(ex).__name__}: {ex}"[:400]
            agreed = shard.min_over_ranks(1.0 if state["ok"] else 0.0, comm_dev) > 0.5
            if not agreed and state["ok"]:
                state["ok"] = False
                state["why"] = "another rank failed in this phase"
            return res

        parts = {"scatter": 0.0, "encode": 0.0, "gather": 0.0}
        hold = {}

        def setup():
            hold["root_in"] = batch.zipf_blocks(total_blocks, bb, first_block=0, device=dev) if rank == 0 else None
            if hold["root_in"] is not None and comm_dev.type == "cpu":
                hold["root_in"] = hold["root_in"].cpu()
            hold["dense_buf"] = torch.empty(n * (sqz_amd.bound(bb) // 8 * 8), dtype=torch.uint8, device=dev)

        phase(setup)

        def transfer_step(timed):
            t = [time.perf_counter()]

            def scatter():
                mine, _span = shard.scatter_blocks(hold["root_in"], total_blocks, bb, comm_dev)
                hold["mine"] = mine.to(dev)
                torch.cuda.synchronize()
            phase(scatter)
            t.append(time.perf_counter())

            def local():
                o, o_off, o_bytes, _e = enc.encode(hold["mine"], in_off, 1 << wb)
                hold["dense"], _ = batch.pack_blocks(o, o_off, o_bytes, dense=hold["dense_buf"])
                hold["o_bytes"] = o_bytes
                torch.cuda.synchronize()
            phase(local)
            t.append(time.perf_counter())

            def gather():
                hold["res"] = shard.gather_dense(hold["dense"], hold["o_bytes"], total_blocks, comm_dev)
                torch.cuda.synchronize()
            phase(gather)
            t.append(time.perf_counter())
            if timed:
                parts["scatter"] += t[1] - t[0]
                parts["encode"] += t[2] - t[1]
                parts["gather"] += t[3] - t[2]

        for _ in range(args.warmup):
            transfer_step(False)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            transfer_step(True)
        barrier()
        x_s = shard.max_over_ranks(time.perf_counter() - t0, comm_dev)

        def check():
            if rank != 0:
                return None
            root_dense, _root_sizes, root_off = hold["res"]
            # what came back is the batch's streams in block order: rank 0's own range must
            # equal what it encoded locally, and the sizes must add up to the job's total
            own = root_dense[:int(root_off[n])].to(dev)
            loc_dense, _loc_off = batch.pack_blocks(out, out_off, out_bytes)
            assert torch.equal(own, loc_dense[:own.numel()]), "gathered streams differ from the local encode"
            return {"encode_MBps": round(total_blocks * bb / x_s * args.steps / 1e6, 3),
                    "ms_per_step": round(x_s / args.steps * 1e3, 3),
                    "rank0_ms": {k: round(v / args.steps * 1e3, 3) for k, v in parts.items()},
                    "gathered_bytes": int(root_off[-1]),
                    "path": "rank 0 -> scatter (equal slabs) -> encode -> pack -> gather sizes + "
                            f"dense streams (padded to the longest rank's) -> rank 0, backend {backend}",
                    # what stands behind this leg until a multi-GPU run has been recorded
                    "evidence": "shard.py over RCCL: world-1 nccl test on one MI355X (tests/test_shard_nccl.py); "
                                "world-2: gloo rehearsal only (tests/test_shard_gpu.py, tools/rehearse_n2.sh)"}

        xfer = phase(check)
        if not state["ok"]:
            xfer = {"error": state["why"]}          # on EVERY rank: they all leave with the same exit code

    if rank == 0:
        in_total = float(total_blocks) * bb
        ms_per_step = enc_s / args.steps * 1e3
        enc_k = {k: v[0] / max(v[1], 1) for k, v in tim.items()}      # avg ms per launch
        dec_k = {k: v[0] / max(v[1], 1) for k, v in dtim.items()}
        dominant = max(enc_k, key=enc_k.get)
        algo_bytes = n * bb + comp_bytes          # encode = n + c per block, this rank's launch
        achieved = algo_bytes / (enc_k[dominant] * 1e-3) / 1e9
        traffic, traffic_note = measured_traffic(dominant) if (world == 1 and n == 4096 and bb == 262144) \
            else (None, "traffic is measured on the N=1 default workload only")
        if world == 1:
            scaling = "strong" if strong else "weak"
            workload = (f"{n} x {bb} B Zipf(s=1) byte blocks, window 2^{wb}, one independent stream per "
                        "block (BASELINE.json configs[2])")
            par = "1 rank; the N>1 runs shard this same batch (strong scaling)"
        elif strong:
            scaling = "strong"
            workload = (f"the same {total_blocks} x {bb} B Zipf(s=1) batch (block ids 0..{total_blocks - 1}), "
                        f"window 2^{wb}, sharded {n} blocks per GPU (BASELINE.json configs[3])")
            par = f"contiguous block ranges over {world} ranks, no data-path collective in `value`"
        else:
            scaling = "weak"
            workload = f"{n} x {bb} B Zipf(s=1) byte blocks per GPU, window 2^{wb}"
            par = f"every rank its own batch, {world} ranks, no data-path collective"
        line = {
            "metric": "encode MB/s + decode MB/s, 32KB window, batched blocks, 1/2/4/8 MI355X",
            "value": round(in_total / enc_s * args.steps / 1e6, 3),
            "unit": "MB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload,
                       "total_blocks": total_blocks, "blocks_per_gpu": n, "block_bytes": bb, "win_bits": wb,
                       "parallelism": par,
                       "finder": args.finder,
                       "device": info["name"]},
            "decode_MBps": round(in_total / dec_s * args.steps / 1e6, 3),
            "decode_ms_per_step": round(dec_s / args.steps * 1e3, 3),
            "compressed_ratio": round(comp_total / in_total, 5),
            "kernels_ms": {k: round(v, 3) for k, v in {**enc_k, **dec_k}.items()},
            "roofline": {"bound": "hbm", "kernel": dominant,
                         "achieved": round(achieved, 4), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 7), "traffic": traffic,
                         "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "encode_frac_of_hbm_roof": round(
                             algo_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 7)},
        }
        if xfer is not None:
            line["with_transfer"] = xfer
        serial_ms = sum(enc_k.get(k, 0.0) for k in ("index_parse_kernel", "huffman_emit_kernel"))
        sec = {
            "note": "the path is a serial chain per stream, not HBM-bound: tokens through the "
                    "adaptive-Huffman stage per second",
            "tokens_per_step": tokens_all,
            "tokens_per_s_encode": round(tokens_all / (ms_per_step * 1e-3), 1),
            "tokens_per_s_emit_kernel": round(tokens_rank / (enc_k["huffman_emit_kernel"] * 1e-3), 1),
            "tokens_per_s_entropy_decode_kernel": round(tokens_rank / (dec_k["entropy_decode_kernel"] * 1e-3), 1),
            "serial_stage_ms": round(serial_ms, 3),
        }
        # the candidate tests the reference's O(window) scan makes for the same tokens (block 0 of
        # this rank x its blocks).  Only --finder scan PERFORMS them; the indexed finder visits the
        # equal-prefix candidates only, so for it the figure is what it avoids, not a rate it achieves.
        cand = cand0 * total_blocks                 # block 0 of rank 0 stands for every block of the job
        if args.finder == "scan":
            sec["scan_candidate_tests_per_step"] = cand
            sec["scan_candidate_tests_per_s"] = round(cand / (ms_per_step * 1e-3), 1)
        else:
            sec["reference_scan_candidate_tests_per_step_not_performed"] = cand
        line["secondary"] = sec
        if world == 1 and args.cpu_blocks > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_blocks, bb, wb)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()
    if isinstance(xfer, dict) and "error" in xfer:
        # the device-local line is out; the failed transfer leg must not pass for a clean run
        print(f"bench.py: with_transfer leg failed on rank {rank}: {xfer['error']}", file=sys.stderr, flush=True)
        sys.exit(3)


if __name__ == "__main__":
    main()
