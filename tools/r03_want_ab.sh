#!/bin/bash
# A/B of the emit kernel's offer policy (huffman_emit.hip: want = mean step length x MUL/8 x 4 + ADD): experimental
# libraries sqz_amd/lib/libsqz_amd_exp*.so (built by hand with -DSQZ_WANT_MUL / -DSQZ_WANT_ADD) against the shipping one.
set -o pipefail
mkdir -p gpurun_out
for lib in libsqz_amd.so $(cd sqz_amd/lib && ls libsqz_amd_exp*.so); do
  for n in 4096 512; do
    SQZ_AMD_LIB=$PWD/sqz_amd/lib/$lib timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-blocks 0 --blocks $n > gpurun_out/want_$lib.$n.log 2>&1 || { tail -5 gpurun_out/want_$lib.$n.log; exit 1; }
    python - <<PY
import json
for l in open("gpurun_out/want_$lib.$n.log"):
    if l.startswith("{"):
        d = json.loads(l); print("$lib", $n, "encode ms", d["ms_per_step"], "emit", d["kernels_ms"]["huffman_emit_kernel"])
PY
  done
done
