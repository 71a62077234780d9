#!/bin/bash
# instruction mix of the entropy kernels (separate PMC passes, kernel trace only)
set -o pipefail
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc
cd /tmp
for set in "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc/$tag -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --cpu-blocks 0 --no-verify > $R/gpurun_out/pmc/$tag.log 2>&1 || { tail -5 $R/gpurun_out/pmc/$tag.log; exit 1; }
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "sqzk" not in k and "kernel" not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in acc.items():
    print(k, {c: f"{x:.4g}" for c, x in sorted(v.items())})
PY
