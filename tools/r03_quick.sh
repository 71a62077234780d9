#!/bin/bash
# round 3 iteration loop on the GPU box: the parity tests that cover the kernels being changed, stage-1
# timing, one bench line at 4096 and one at 512 blocks.  TAG names the output directory.
set -o pipefail
TAG=${TAG:-r03q}
O=gpurun_out/$TAG
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tree.py tests/test_gpu_variants.py -x -q -m gpu 2>&1 | tail -15 | tee $O/tests.txt
grep -q "passed" $O/tests.txt && ! grep -q "failed" $O/tests.txt || exit 1
timeout -k 10 120 python tools/microbench/time_stage1.py 2>&1 | tail -2 | tee $O/stage1.txt
for n in 4096 512; do
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --blocks $n --cpu-blocks 0 > $O/b$n.log 2>&1 || { tail -5 $O/b$n.log; exit 1; }
  grep '^{' $O/b$n.log > $O/blocks_$n.json
done
python - <<PY
import json
for n in (4096, 512):
    d = json.loads(open("$O/blocks_%d.json" % n).read())
    print(n, "enc", d["value"], d["ms_per_step"], "dec", d["decode_MBps"], d["decode_ms_per_step"], d["kernels_ms"])
PY
