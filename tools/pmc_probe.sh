#!/bin/bash
# one-off counter probes of the bench command: PMC_SETS="A,B C" -> one rocprofv3 pass per
# space-separated set; sums per kernel into gpurun_out/pmc_$TAG/summary.txt
set -o pipefail
export TMPDIR=/tmp
R=$PWD
TAG=${TAG:-probe}
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp
rocprofv3 -L > $O/avail.txt 2>&1 || true
i=0
for set in $PMC_SETS; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc ${set//,/ } --kernel-trace -d $O/pass$i -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --cpu-blocks 0 --no-verify ${BENCH_ARGS} > $O/pass$i.log 2>&1 || { echo "pass $i ($set) failed"; tail -3 $O/pass$i.log; }
done
cd $R
python - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/pmc_" + os.environ.get("TAG", "probe")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(O + "/pass*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("sqzk::", "")
        if "kernel" not in k or "at::" in k or "rocprim" in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
with open(O + "/summary.txt", "w") as fh:
    for k, v in acc.items():
        line = k + "  " + "  ".join(f"{c}={x:.4g}" for c, x in sorted(v.items()))
        print(line); fh.write(line + "\n")
PY
