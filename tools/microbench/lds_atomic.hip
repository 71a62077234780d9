// tools/microbench/lds_atomic.hip -- cycles per LDS instruction of a lone wave: atomics (returning or not) and reads, with k lanes on one address.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/lds_atomic tools/microbench/lds_atomic.hip ; run on an MI355X.
// The figures quoted in DESIGN.md section 5 come from this program.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// cycles per LDS instruction for a single wave: atomics (returning / not), reads, with k-way same-address conflicts
template <int MODE>
__global__ void k(uint64_t* out, int ways) {
    __shared__ uint32_t a[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) a[i] = 0;
    __syncthreads();
    // `ways` lanes share an address
    const int idx = (lane / ways) * 33 % 1024;
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 1000; it++) {
        if (MODE == 0) { acc += atomicAdd(&a[idx], 1u); }              // returning, dependent chain? no: independent adds, acc uses
        if (MODE == 1) { atomicAdd(&a[idx], 1u); }
        if (MODE == 2) { acc += a[(idx + acc) & 1023]; }               // dependent read chain
        if (MODE == 3) { const uint32_t o = atomicAdd(&a[(idx + (acc & 1)) & 1023], 1u); acc = o & 0; } // dependent returning atomic chain
        if (MODE == 4) { a[idx] = acc + it; }                          // plain stores
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[0] = t1 - t0; out[1] = acc; }
}
int main() {
    uint64_t* d; hipMalloc(&d, 16);
    uint64_t h[2];
    const char* names[5] = {"atomic rtn (independent)", "atomic no-rtn", "dependent read", "dependent atomic rtn", "plain store"};
    for (int mode = 0; mode < 5; mode++) {
        for (int ways : {1, 2, 4, 8, 16, 32, 64}) {
            for (int rep = 0; rep < 2; rep++) {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, d, ways); break;
                    case 1: hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, d, ways); break;
                    case 2: hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, d, ways); break;
                    case 3: hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, d, ways); break;
                    case 4: hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, d, ways); break;
                }
                hipDeviceSynchronize();
            }
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("%-28s ways %2d: %.1f cycles/instr\n", names[mode], ways, h[0] / 1000.0);
        }
    }
    return 0;
}
