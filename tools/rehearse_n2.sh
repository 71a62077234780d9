#!/bin/bash
# two ranks of bench.py on ONE GPU over gloo: exercises the N>1 code path (barriers,
# max/sum over ranks, rank-0 line).  Not a performance number.
set -o pipefail
mkdir -p gpurun_out
SQZ_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 1 --blocks 512 > gpurun_out/n2.log 2>&1 || { tail -20 gpurun_out/n2.log; exit 1; }
python - <<'PY'
import json
for l in open("gpurun_out/n2.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("N=2 line ok:", d["n_gpus"], d["value"], d["unit"], d["scaling"], d["secondary"]["tokens_per_step"], "cpu_baseline" in d)
PY
