#!/bin/bash
# round 3 evidence set, one GPU call: the whole GPU suite, the judged measurement set (bench line, rocprofv3
# kernel stats, FETCH_SIZE / WRITE_SIZE passes), the per-GPU block counts of configs[3] (512 / 1024 / 2048 /
# 4096) with the 2-rank gloo rehearsal, the corpus table and the R-era line.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
PYTHONUNBUFFERED=1 timeout -k 10 900 python -u -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 | tee gpurun_out/r03_tests.txt
grep -q " passed" gpurun_out/r03_tests.txt && ! grep -q "failed" gpurun_out/r03_tests.txt || exit 1
PROF_TAG=r03 bash tools/profile_round.sh > gpurun_out/r03_profile.log 2>&1 || { tail -20 gpurun_out/r03_profile.log; exit 1; }
tail -3 gpurun_out/r03_profile.log | cut -c1-300
TAG=r03 bash tools/scaling_lines.sh > gpurun_out/r03_scaling.log 2>&1 || { tail -20 gpurun_out/r03_scaling.log; exit 1; }
tail -8 gpurun_out/r03_scaling.log | cut -c1-400
timeout -k 10 300 python tools/microbench/corpus_batch.py > gpurun_out/r03_corpus_batch.txt 2>&1 || tail -5 gpurun_out/r03_corpus_batch.txt
tail -9 gpurun_out/r03_corpus_batch.txt | cut -c1-200
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --codec rc > gpurun_out/r03_rc.log 2>&1 && grep '^{' gpurun_out/r03_rc.log > gpurun_out/r03_rc_bench_line.json
