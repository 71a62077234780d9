#!/bin/bash
# tests + short bench on the GPU box; prints the kernel times.  Extra arguments go to bench.py.
set -o pipefail
mkdir -p gpurun_out
PYTHONUNBUFFERED=1 timeout -k 10 900 python -u -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tee gpurun_out/t.log
rc=${PIPESTATUS[0]}
tail -5 gpurun_out/t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --steps 2 --warmup 1 "$@" > gpurun_out/b.log 2>&1 || { tail -5 gpurun_out/b.log; exit 1; }
python - <<PY
import json
for l in open("gpurun_out/b.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["value"], d["unit"], "decode", d.get("decode_MBps"), d.get("kernels_ms"), d.get("roofline"))
PY
