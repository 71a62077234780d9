#!/bin/bash
# the judged measurement set: bench line, rocprofv3 kernel stats of the same command,
# and HBM traffic (FETCH_SIZE / WRITE_SIZE in their own PMC passes, kernel trace only)
set -o pipefail
export TMPDIR=/tmp
R=$PWD
export PROF_TAG=${PROF_TAG:-v4}
O=$R/gpurun_out/prof_$PROF_TAG
mkdir -p $O
timeout -k 10 300 python bench.py --steps 2 --warmup 1 > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
grep '^{' $O/bench.log > $O/bench_line.json
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/stats -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace -d $O/$c -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --cpu-blocks 0 --no-verify > $O/$c.log 2>&1 || { tail -5 $O/$c.log; exit 1; }
done
cd $R
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python - <<'PY'
import csv, glob, collections, json, os
O = "gpurun_out/prof_" + os.environ["PROF_TAG"]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(O + "/*SIZE/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("sqzk::", "")
        if "kernel" not in k or "at::" in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n[k][row["Counter_Name"]].add(row["Dispatch_Id"])
out = {k: {"fetch_kib_raw": v.get("FETCH_SIZE"), "write_kib": v.get("WRITE_SIZE"),
           "launches": len(n[k].get("FETCH_SIZE", ()))} for k, v in acc.items()}
json.dump(out, open(O + "/pmc_hbm.json", "w"), indent=1)
print(json.dumps(out, indent=1))
print(open(O + "/bench_line.json").read()[:600])
PY
head -8 $O/kernel_stats.csv | cut -c1-160
