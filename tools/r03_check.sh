#!/bin/bash
# round 3, first GPU pass: the whole GPU suite, the default bench line, the R-era bench line, and the
# kernel-stats + counter summaries of the brute-force scan kernel (the kernel north_star describes)
set -o pipefail
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r03a
mkdir -p $O
bash tools/gpu_check.sh || exit 1
cp gpurun_out/b.log $O/bench_default.log
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --codec rc > $O/rc.log 2>&1 || { tail -5 $O/rc.log; exit 1; }
grep '^{' $O/rc.log > $O/rc_bench_line.json
cut -c1-700 $O/rc_bench_line.json
cd /tmp
SCAN="python $R/bench.py --steps 1 --warmup 0 --blocks 512 --cpu-blocks 0 --no-verify --finder scan"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/scan_stats -o p --output-format csv -- $SCAN > $O/scan_stats.log 2>&1 || { tail -5 $O/scan_stats.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/rc_stats -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --codec rc --cpu-blocks 0 > $O/rc_stats.log 2>&1 || { tail -5 $O/rc_stats.log; exit 1; }
for set in "SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE,SQ_LDS_BANK_CONFLICT,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_VALU,SQ_INST_CYCLES_VMEM,SQ_WAVES,GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d, -f1)
  timeout -k 10 300 rocprofv3 --pmc ${set//,/ } --kernel-trace -d $O/scan_pmc_$tag -o p --output-format csv -- $SCAN > $O/scan_pmc_$tag.log 2>&1 || { echo "pmc $tag failed"; tail -3 $O/scan_pmc_$tag.log; }
done
cd $R
find $O/scan_stats -name "*kernel_stats.csv" -exec cp {} $O/scan_kernel_stats.csv \;
find $O/rc_stats -name "*kernel_stats.csv" -exec cp {} $O/rc_kernel_stats.csv \;
python - <<'PY'
import csv, glob, collections
O = "gpurun_out/r03a"
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(O + "/scan_pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("sqzk::", "")
        if "scan" not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
with open(O + "/scan_pmc_summary.txt", "w") as fh:
    for k, v in acc.items():
        line = k + "  " + "  ".join(f"{c}={x:.5g}" for c, x in sorted(v.items()))
        print(line); fh.write(line + "\n")
PY
head -5 $O/scan_kernel_stats.csv | cut -c1-200
head -5 $O/rc_kernel_stats.csv | cut -c1-200
