#!/bin/bash
# single-GPU lines for the per-GPU work of configs[3] at 8 / 4 / 2 GPUs (512 / 1024 / 2048 blocks of
# the same batch) next to the full 4096: the only strong-scaling evidence obtainable on a one-GPU
# box.  Also the two-rank rehearsal of the strong mode (gloo, collectives staged through the host)
# and one --finder scan line.  Output: gpurun_out/scale_$TAG/*.json
set -o pipefail
TAG=${TAG:-r02}
O=gpurun_out/scale_$TAG
mkdir -p $O
for n in 512 1024 2048 4096; do
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --blocks $n --cpu-blocks 0 > $O/b$n.log 2>&1 || { tail -5 $O/b$n.log; exit 1; }
  grep '^{' $O/b$n.log > $O/blocks_$n.json
  echo "blocks $n done"
done
SQZ_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --blocks 1024 > $O/n2.log 2>&1 || { tail -20 $O/n2.log; exit 1; }
grep '^{' $O/n2.log > $O/n2_gloo_rehearsal.json
timeout -k 10 300 python bench.py --steps 1 --warmup 0 --blocks 256 --cpu-blocks 0 --finder scan > $O/scan.log 2>&1 || { tail -5 $O/scan.log; exit 1; }
grep '^{' $O/scan.log > $O/finder_scan_256.json
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.json")):
    d = json.loads(open(f).read())
    print(f.split("/")[-1], d["n_gpus"], d["config"]["blocks_per_gpu"], d["value"], d["ms_per_step"], d["decode_MBps"], d["decode_ms_per_step"], d["kernels_ms"], d.get("with_transfer"))
PY
