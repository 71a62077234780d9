#!/bin/bash
# the R-era kernels alone: their GPU tests, then the bench line at 4096 / 2048 / 1024 blocks (does the time follow the
# block count -- issue bound -- or stay -- one wave's chain?)
set -o pipefail
mkdir -p gpurun_out
PYTHONUNBUFFERED=1 timeout -k 10 600 python -u -m pytest tests/test_rc.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/rc_iter_tests.txt
grep -q " passed" gpurun_out/rc_iter_tests.txt && ! grep -q "failed" gpurun_out/rc_iter_tests.txt || exit 1
for n in 4096 2048 1024; do
  timeout -k 10 200 python bench.py --steps 2 --warmup 1 --codec rc --cpu-blocks 0 --blocks $n > gpurun_out/rc_iter_$n.log 2>&1 || { tail -5 gpurun_out/rc_iter_$n.log; exit 1; }
  python - <<PY
import json
for l in open("gpurun_out/rc_iter_$n.log"):
    if l.startswith("{"):
        d = json.loads(l); print($n, d["value"], d["decode_MBps"], d["kernels_ms"])
PY
done
