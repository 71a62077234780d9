export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/pmc_mw
mkdir -p $O
cd /tmp
for w in ${PMC_WAVES:-1 3}; do
  for set in "SQC_ICACHE_REQ,SQC_ICACHE_MISSES,SQC_ICACHE_HITS,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_SMEM,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INST_CYCLES_SALU,SQ_IFETCH"; do
    tag=w${w}_$(echo $set | cut -d, -f1)
    CORPUS_BLOCKS=256 CORPUS_ONLY="uniform random" SQZ_DECODE_WAVES=$w timeout -k 10 300 rocprofv3 --pmc ${set//,/ } --kernel-trace -d $O/$tag -o p --output-format csv -- python $R/tools/microbench/corpus_batch.py > $O/$tag.log 2>&1 || { echo "$tag failed"; tail -3 $O/$tag.log; }
  done
done
cd $R
python - <<'PY'
import csv, glob, collections
O = "gpurun_out/pmc_mw"
import os
for w in [int(x) for x in os.environ.get("PMC_WAVES", "1 3").split()]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(O + f"/w{w}_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("sqzk::", "")
            if "decode" not in k: continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, v in acc.items():
        print(w, k[:40], "  ".join(f"{c}={x:.4g}" for c, x in sorted(v.items())))
PY
