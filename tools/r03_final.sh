#!/bin/bash
# round 3 closing set, one GPU call: tools/r03_evidence.sh (GPU suite, the judged measurement set, the per-GPU
# block counts, corpus table, R-era line) and, for the R-era kernels, rocprofv3 kernel stats of the same command
# plus FETCH_SIZE / WRITE_SIZE in their own PMC passes (kernel trace only).
# PART=evidence / PART=rc run one half (each fits a 1200 s call).
# Output: gpurun_out/prof_$TAG/* (tools/adopt_profile.py), gpurun_out/${TAG}_rc_{kernel_stats.csv,pmc_hbm.json}
set -o pipefail
export TMPDIR=/tmp
R=$PWD
TAG=${TAG:-r03b}
sed "s/PROF_TAG=r03 /PROF_TAG=$TAG /; s/TAG=r03 /TAG=$TAG /; s/r03_/${TAG}_/g" tools/r03_evidence.sh > /tmp/evidence_$TAG.sh
if [ "$PART" != "rc" ]; then bash /tmp/evidence_$TAG.sh || exit 1; fi
[ "$PART" = "evidence" ] && exit 0
O=$R/gpurun_out/rc_$TAG
mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o p --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --codec rc --cpu-blocks 0 > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $O/$c -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --codec rc --cpu-blocks 0 > $O/$c.log 2>&1 || { tail -5 $O/$c.log; exit 1; }
done
cd $R
find $O/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_rc_kernel_stats.csv \;
TAG=$TAG python - <<'PY'
import csv, glob, collections, json, os
tag = os.environ["TAG"]
O = "gpurun_out/rc_" + tag
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(O + "/*SIZE/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("sqzk::", "")
        if not k.startswith("rc_"): continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n[k][row["Counter_Name"]].add(row["Dispatch_Id"])
out = {}
for k, v in acc.items():
    l = max(len(n[k].get("FETCH_SIZE", ())), 1)
    f, w = (v.get("FETCH_SIZE") or 0.0) * 1024 / l, (v.get("WRITE_SIZE") or 0.0) * 1024 / l
    # these kernels read their input once, 64 consecutive bytes per wave and load: FETCH_SIZE comes out at HALF of
    # what they must read (encode: 0.54 GB for the 1.07 GB batch; decode: 0.42 GB for 0.84 GB of streams) while
    # WRITE_SIZE equals the bytes written to the byte -- the gfx950 halving of MI355X_MICROARCH.md (HBM section)
    # applies to them, so FETCH_SIZE is doubled here
    out[k] = {"launches": l, "fetch_bytes_raw": int(f), "fetch_bytes": int(2 * f), "write_bytes": int(w),
              "hbm_bytes_per_launch": int(2 * f + w)}
json.dump({"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only; KiB -> bytes; "
                     "per launch; FETCH_SIZE doubled (gfx950 reports half the bytes of these coalesced streaming reads: "
                     "the raw figure is half of what the kernel must read, WRITE_SIZE is exact)", "kernels": out},
          open(f"gpurun_out/{tag}_rc_pmc_hbm.json", "w"), indent=1)
print(json.dumps(out))
PY
head -4 gpurun_out/${TAG}_rc_kernel_stats.csv | cut -c1-160
