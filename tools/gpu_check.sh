#!/bin/bash
# tests + short bench on the GPU box; prints the kernel times
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1
rc=$?
tail -3 gpurun_out/t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --steps 2 --warmup 1 "$@" > gpurun_out/b.log 2>&1 || { tail -5 gpurun_out/b.log; exit 1; }
python - <<PY
import json
for l in open("gpurun_out/b.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["value"], d["unit"], "decode", d.get("decode_MBps"), d.get("kernels_ms"), d.get("roofline"))
PY
if [ -f sqz_amd/lib/libsqz_amd_stats.so ]; then
  SQZ_AMD_LIB=$PWD/sqz_amd/lib/libsqz_amd_stats.so timeout -k 10 200 python bench.py --steps 1 --warmup 0 --cpu-blocks 0 2>&1 | grep -E "^block|^cycles|^lit|^bump|^wall|^emit" | cut -c1-300
fi
