#!/bin/bash
# stage-1 iteration loop: the token parity tests, stage-1 kernel times on the bench batch, the corpus rows
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tokens or full_size or sort_key or large_stream or corpus" 2>&1 | tail -3 || exit 1
timeout -k 10 120 python tools/microbench/time_stage1.py 2>&1 | tail -1 || exit 1
CORPUS_ONLY="${CORPUS_ONLY:-confucius.txt,laozi.txt,x64.elf,arm64.elf,zeros}" timeout -k 10 300 python tools/microbench/corpus_batch.py > gpurun_out/corpus_iter.txt 2>&1
cut -c1-220 gpurun_out/corpus_iter.txt | tail -6
