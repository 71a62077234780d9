#!/usr/bin/env python3
"""tools/adopt_profile.py TAG -- copy gpurun_out/prof_TAG's summaries into profiles/ (tracked):
bench line, rocprofv3 kernel stats, raw PMC counters, and the `current` slot of
profiles/traffic.json (what bench.py reports as roofline.traffic while the kernels are the same)."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
for name, to in (("bench_line.json", f"{tag}_bench_line.json"), ("kernel_stats.csv", f"{tag}_kernel_stats.csv"),
                 ("pmc_hbm.json", f"{tag}_pmc_hbm.json")):
    shutil.copyfile(os.path.join(src, name), os.path.join(dst, to))
with open(os.path.join(dst, "traffic.json")) as fh:
    t = json.load(fh)
with open(os.path.join(src, "traffic_current.json")) as fh:
    t["current"] = json.load(fh)
# the R-era kernels' own passes (tools/r03_final.sh PART=rc), same sources: merged into the same slot
rc = os.path.join(ROOT, "gpurun_out", f"{tag}_rc_pmc_hbm.json")
if os.path.exists(rc):
    with open(rc) as fh:
        t["current"]["kernels"].update(json.load(fh)["kernels"])
    shutil.copyfile(rc, os.path.join(dst, f"{tag}_rc_pmc_hbm.json"))
    shutil.copyfile(os.path.join(ROOT, "gpurun_out", f"{tag}_rc_kernel_stats.csv"), os.path.join(dst, f"{tag}_rc_kernel_stats.csv"))
t.setdefault("history", {})[tag] = t["current"]
with open(os.path.join(dst, "traffic.json"), "w") as fh:
    json.dump(t, fh, indent=1)
print("adopted", tag, "build", t["current"]["kernel_build_id"])
