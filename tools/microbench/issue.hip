// tools/microbench/issue.hip -- cycles per instruction of a lone wave: dependent / independent VALU, SALU, v_readlane -> SALU chains, exec-mask regions.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/issue tools/microbench/issue.hip ; run on an MI355X.
// The figures quoted in DESIGN.md section 5 come from this program.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void k(uint64_t* out, int seed) {
    const int lane = threadIdx.x;
    uint32_t a = seed + lane, b = seed * 3 + lane, c = seed * 5 + lane, d = seed * 7 + lane;
    int s = seed;
    const uint64_t t0 = __builtin_readcyclecounter();
    const uint64_t w0 = wall_clock64();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("" :: "s"(t0), "s"(w0));
#pragma unroll 1
    for (int it = 0; it < 1000; it++) {
        if (MODE == 0) {  // 8 dependent VALU
            asm volatile("v_add_u32 %0, %0, %0\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %0\n v_xor_b32 %0, %0, %1\n"
                         "v_add_u32 %0, %0, %0\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %0\n v_xor_b32 %0, %0, %1\n" : "+v"(a) : "v"(b));
        }
        if (MODE == 1) {  // 8 independent VALU (4 chains)
            asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n"
                         "v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        }
        if (MODE == 2) {  // 8 dependent SALU
            asm volatile("s_add_u32 %0, %0, %0\n s_xor_b32 %0, %0, 5\n s_add_u32 %0, %0, %0\n s_xor_b32 %0, %0, 5\n"
                         "s_add_u32 %0, %0, %0\n s_xor_b32 %0, %0, 5\n s_add_u32 %0, %0, %0\n s_xor_b32 %0, %0, 5\n" : "+s"(s) :: "scc");
        }
        if (MODE == 3) {  // 4 x (readlane -> s_add -> s_and) : VALU->SALU->lane select chain
            asm volatile("v_readlane_b32 s20, %1, %0\n s_add_u32 %0, %0, s20\n s_and_b32 %0, %0, 63\n"
                         "v_readlane_b32 s20, %1, %0\n s_add_u32 %0, %0, s20\n s_and_b32 %0, %0, 63\n"
                         "v_readlane_b32 s20, %1, %0\n s_add_u32 %0, %0, s20\n s_and_b32 %0, %0, 63\n"
                         "v_readlane_b32 s20, %1, %0\n s_add_u32 %0, %0, s20\n s_and_b32 %0, %0, 63\n" : "+s"(s) : "v"(a) : "s20", "scc");
        }
        if (MODE == 4) {  // 4 x (v_cmp -> s_and_saveexec -> v_add -> s_or exec)
            asm volatile("v_cmp_lt_u32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, 1\n s_or_b64 exec, exec, s[20:21]\n"
                         "v_cmp_lt_u32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, 1\n s_or_b64 exec, exec, s[20:21]\n"
                         : "+v"(a) : "v"(b) : "vcc", "s20", "s21", "scc");
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t w1 = wall_clock64();
    if (lane == 0) { out[0] = t1 - t0; out[1] = a + b + c + d + s; out[2] = w1 - w0; }
}
int main() {
    uint64_t* d; (void)hipMalloc(&d, 32);
    uint64_t h[3];
    const char* names[5] = {"8 dependent VALU", "8 independent VALU", "8 dependent SALU", "4x readlane->s_add->s_and", "2x cmp/saveexec/add/restore"};
    for (int mode = 0; mode < 5; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, d, 3); break;
                case 1: hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, d, 3); break;
                case 2: hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, d, 3); break;
                case 3: hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, d, 3); break;
                case 4: hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, d, 3); break;
            }
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("%-32s: %.1f cycles/iter, wall_clock %.1f ticks/iter\n", names[mode], h[0] / 1000.0, h[2] / 1000.0);
    }
    int rate = 0; (void)hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate %d kHz, device clock %d kHz\n", rate, clk);
    return 0;
}
