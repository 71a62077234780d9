#!/bin/bash
# one bench pass with the instrumented build (tools/build_stats.sh): prints the cycle sections
mkdir -p gpurun_out
SQZ_AMD_LIB=$PWD/sqz_amd/lib/libsqz_amd_stats.so timeout -k 10 200 python bench.py --steps 1 --warmup 0 --cpu-blocks 0 "$@" > gpurun_out/stats.log 2>&1
grep -E "^block|^cycles|^lit|^bump|^emit|^sec" gpurun_out/stats.log | cut -c1-300
