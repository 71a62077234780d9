#!/bin/bash
# the judged measurement set: bench line, rocprofv3 kernel stats of the same command,
# and HBM traffic (FETCH_SIZE / WRITE_SIZE in their own PMC passes, kernel trace only).
# Writes gpurun_out/prof_$PROF_TAG/{bench_line.json,kernel_stats.csv,pmc_hbm.json,traffic_current.json};
# copy them into profiles/ (traffic_current.json -> the "current" slot of profiles/traffic.json:
# tools/adopt_profile.py does that) and commit.
set -o pipefail
export TMPDIR=/tmp
R=$PWD
export PROF_TAG=${PROF_TAG:-r02}
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
import csv, glob, collections, json, os, sys
sys.path.insert(0, os.getcwd())
import bench
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
# MI355X_MICROARCH.md (HBM / rocprofv3): counters are KiB; on gfx950 FETCH_SIZE reports half the
# bytes of a wide (16 B per lane) coalesced streaming read and is doubled for those; WRITE_SIZE is
# exact.  Round 3 found the halving on 4-byte and 1-byte coalesced loads too (kernels whose read volume is
# known to the byte: emit, entropy decode, the R-era pair), so this file keeps BOTH figures per kernel and
# bench.py (HALVED_FETCH) picks the doubled one for the kernels that read one coalesced stream; gathers
# (match, expand) stay raw.
cur = {"kernel_build_id": bench.kernel_build_id(),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (--kernel-trace only) of "
                 "`python bench.py --steps 1 --warmup 0 --cpu-blocks 0 --no-verify`; KiB -> bytes; per launch; "
                 "FETCH_SIZE not doubled (no 16-B-per-lane streaming reads in these kernels; doubled figure "
                 "listed as hbm_bytes_per_launch_if_fetch_doubled), WRITE_SIZE exact (MI355X_MICROARCH.md, HBM section)",
       "kernels": {}}
for k, v in out.items():
    l = max(v["launches"], 1)
    f, w = (v["fetch_kib_raw"] or 0.0) * 1024 / l, (v["write_kib"] or 0.0) * 1024 / l
    cur["kernels"][k] = {"fetch_bytes": int(f), "write_bytes": int(w), "hbm_bytes_per_launch": int(f + w),
                         "hbm_bytes_per_launch_if_fetch_doubled": int(2 * f + w)}
json.dump(cur, open(O + "/traffic_current.json", "w"), indent=1)
print(json.dumps(cur, indent=1))
print(open(O + "/bench_line.json").read()[:600])
PY
head -8 $O/kernel_stats.csv | cut -c1-160
