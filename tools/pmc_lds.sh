#!/bin/bash
# LDS activity / conflict counters of the kernels (separate PMC passes, kernel trace only)
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc2
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_LDS[A-Z_]*|SQ_INST_CYCLES[A-Z_]*|SQ_WAIT_INST_LDS|SQ_INSTS_LDS|SQ_ACTIVE_INST_LDS" | sort -u | tr '\n' ' ' > $R/gpurun_out/pmc2/avail.txt
cat $R/gpurun_out/pmc2/avail.txt; echo
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc2/$tag -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --cpu-blocks 0 --no-verify > $R/gpurun_out/pmc2/$tag.log 2>&1 || { tail -5 $R/gpurun_out/pmc2/$tag.log; }
done
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc2/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "sqzk" not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in acc.items():
    print(k, {c: f"{x:.4g}" for c, x in sorted(v.items())})
PY
