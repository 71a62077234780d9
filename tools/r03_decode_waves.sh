#!/bin/bash
# decoder with 1 / 2 / 4 wavefronts per stream: the GPU parity tests under each setting, then the bench
# line at 512 / 1024 / 2048 blocks per GPU for each
set -o pipefail
TAG=${TAG:-r03mw}
O=gpurun_out/$TAG
mkdir -p $O
for w in 4 2; do
  SQZ_DECODE_WAVES=$w timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_file_mode.py tests/test_gpu_variants.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests_w$w.txt
  grep -q "passed" $O/tests_w$w.txt && ! grep -q "failed" $O/tests_w$w.txt || exit 1
done
for n in 512 1024 2048; do
  for w in 1 2 4; do
    SQZ_DECODE_WAVES=$w timeout -k 10 200 python bench.py --steps 3 --warmup 1 --blocks $n --cpu-blocks 0 > $O/b${n}_w$w.log 2>&1 || { tail -5 $O/b${n}_w$w.log; exit 1; }
    grep '^{' $O/b${n}_w$w.log > $O/blocks_${n}_w$w.json
    python - <<PY
import json
d = json.loads(open("$O/blocks_${n}_w$w.json").read())
print($n, "waves", $w, "dec", d["decode_MBps"], d["decode_ms_per_step"], {k: v for k, v in d["kernels_ms"].items() if "decode" in k or "expand" in k})
PY
  done
done
