#!/bin/bash
# instrumented build (cycle sections printed by block 1 of the entropy decode kernel)
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -DSQZ_STATS -Iinclude \
  -o sqz_amd/lib/libsqz_amd_stats.so sqz_amd/csrc/*.hip
