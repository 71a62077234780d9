/* tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
 *
 * A stand-in for <hip/hip_runtime.h> that lets the stream kernels of
 * sqz_amd/csrc (huffman_emit.hip, decode.hip + sqz_tree.h, sqz_device.h) be compiled by g++
 * and RUN ON THE CPU, lane by lane: the 64 lanes of a wavefront are 64 cooperative fibers
 * (ucontext) on one OS thread; every cross-lane operation (__ballot, readlane, readfirstlane,
 * DPP, ds_bpermute, __shfl_xor, LDS fences) is a rendezvous where all lanes OF THE WAVE meet,
 * exchange values and go on.  A workgroup may hold several waves (the multi-wave decoder): a wave
 * runs until it reaches __syncthreads(), where it waits -- the other waves run -- until every wave
 * that has not finished stands at the barrier (what s_barrier does).  Between two rendezvous the lanes run one after the other, so the
 * emulation is exact for code that talks across lanes only through those operations and
 * through LDS around an lds_fence() -- which is what the kernels do.  A lane that arrives at a
 * different operation than the others (divergent control flow around a cross-lane op) aborts
 * the run with a message.
 *
 * Purpose: the kernels' LOGIC can be debugged and tested without a GPU (gdb, printf,
 * sanitizers); tests/test_emu.py holds them against the oracle.  It says nothing about
 * timing, occupancy or the compiler's code for gfx950 -- the GPU parity tests pin those.
 * Only what these kernels use is provided.
 */
#pragma once
#define SQZ_WAVE_EMU 1
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>
#include <execinfo.h>
#include <tuple>
#include <utility>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict

#define __ATOMIC_RELAXED_EMU 0
#define __HIP_MEMORY_SCOPE_AGENT 0

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef void* hipStream_t;
struct uint4 { uint32_t x, y, z, w; };

namespace emu {
constexpr int W = 64;             // lanes of a wave
constexpr int MAXW = 8;           // waves of a workgroup
struct Lane { ucontext_t ctx; char* stack; bool done; };
struct Wave {
    Lane lane[W];
    int cur;                      // lane of this wave running now / to resume
    int live;                     // lanes that have not returned yet
    uint64_t slot[2][W];          // values exchanged at a rendezvous (double-buffered by parity)
    int kind[2][W];               // which operation each lane arrived with
    uint32_t seq[W];              // rendezvous count per lane
    uint32_t barriers;            // workgroup barriers this wave has passed or stands at
    uint32_t lane_barriers[W];    // ... and each of its lanes
};
extern Wave g_wave[MAXW];
extern int g_w;                   // wave running now
extern int g_waves;               // waves of the running workgroup
extern ucontext_t g_main;
extern unsigned g_block, g_grid;
#define g_lane g_wave[emu::g_w].lane
#define g_slot g_wave[emu::g_w].slot
#define g_cur  g_wave[emu::g_w].cur

// all lanes of the running wave meet here; returns when every live lane has arrived at ITS rendezvous number
void rendezvous();
// the running wave stands at a workgroup barrier: the other waves run until all are there
void block_barrier();
// arrive with (kind, value); afterwards slots(par) holds every lane's value
inline int arrive(int kind, uint64_t v) {
    Wave& wv = g_wave[g_w];
    const int me = wv.cur;
    const int par = (int)(wv.seq[me]++ & 1u);
    wv.slot[par][me] = v;
    wv.kind[par][me] = kind;
    rendezvous();
    Wave& w2 = g_wave[g_w];
    for (int l = 0; l < W; l++) {
        if (!w2.lane[l].done && w2.kind[par][l] != kind) {
            fprintf(stderr, "emu: wave %d lane %d arrived at operation %d while lane %d is at %d (divergent cross-lane op)\n",
                    g_w, w2.cur, kind, l, w2.kind[par][l]);
            void* bt[32];
            backtrace_symbols_fd(bt, backtrace(bt, 32), 2);
            abort();
        }
    }
    return par;
}
void run_block(void (*body)(void*), void* arg, unsigned block, unsigned grid, unsigned waves);
}  // namespace emu

struct EmuThreadIdx { struct X { operator unsigned() const { return (unsigned)(emu::g_w * emu::W + emu::g_wave[emu::g_w].cur); } } x; unsigned y = 0, z = 0; };
struct EmuBlockIdx { struct X { operator unsigned() const { return emu::g_block; } } x; unsigned y = 0, z = 0; };
struct EmuGridDim { struct X { operator unsigned() const { return emu::g_grid; } } x; };
struct EmuBlockDim { struct X { operator unsigned() const { return 64u * (unsigned)emu::g_waves; } } x; };
static EmuThreadIdx threadIdx;
static EmuBlockIdx blockIdx;
static EmuGridDim gridDim;
static EmuBlockDim blockDim;

// ---- cross-lane operations -----------------------------------------------------------------
inline uint64_t __ballot(bool p) {
    const int par = emu::arrive(1, p ? 1 : 0);
    uint64_t m = 0;
    for (int l = 0; l < emu::W; l++) { if (!emu::g_lane[l].done && emu::g_slot[par][l]) { m |= 1ull << l; } }
    return m;
}
inline int __builtin_amdgcn_readlane(int v, int l) {
    const int par = emu::arrive(2, (uint32_t)v);
    return (int)(uint32_t)emu::g_slot[par][l & 63];
}
inline int __builtin_amdgcn_readfirstlane(int v) {
    const int par = emu::arrive(3, (uint32_t)v);
    for (int l = 0; l < emu::W; l++) { if (!emu::g_lane[l].done) { return (int)(uint32_t)emu::g_slot[par][l]; } }
    return v;
}
inline int __shfl_xor(int v, int o) {
    const int par = emu::arrive(4, (uint32_t)v);
    return (int)(uint32_t)emu::g_slot[par][(emu::g_cur ^ o) & 63];
}
inline int __builtin_amdgcn_ds_bpermute(int addr, int v) {
    const int par = emu::arrive(5, (uint32_t)v);
    return (int)(uint32_t)emu::g_slot[par][(addr >> 2) & 63];
}
// DPP controls the kernels use: wave_shl:1 0x130, wave_shr:1 0x138, row_shr:n 0x110+n,
// row_bcast:15 0x142, row_bcast:31 0x143; bound_ctrl false: a lane without a source keeps `old`
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound) {
    (void)bank_mask; (void)bound;
    const int par = emu::arrive(6, (uint32_t)src);
    const int l = emu::g_cur, row = l >> 4;
    if (((row_mask >> row) & 1) == 0) { return old; }
    int from = -1;
    if (ctrl == 0x130) { from = l + 1 < 64 ? l + 1 : -1; }
    else if (ctrl == 0x138) { from = l - 1; }
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl - 0x110; from = (l & 15) >= n ? l - n : -1; }
    else if (ctrl == 0x142) { from = (row & 1) ? (row << 4) - 1 : -1; }
    else if (ctrl == 0x143) { from = row >= 2 ? 31 : -1; }
    else { fprintf(stderr, "emu: DPP control %x not modelled\n", ctrl); abort(); }
    return from >= 0 ? (int)(uint32_t)emu::g_slot[par][from] : old;
}
inline uint32_t __builtin_amdgcn_mbcnt_lo(uint32_t mask, uint32_t base) {
    const int l = emu::g_cur;
    const uint32_t below = l >= 32 ? 0xFFFFFFFFu : ((1u << l) - 1u);
    return base + (uint32_t)__builtin_popcount(mask & below);
}
inline uint32_t __builtin_amdgcn_mbcnt_hi(uint32_t mask, uint32_t base) {
    const int l = emu::g_cur;
    const uint32_t below = l <= 32 ? 0u : ((1u << (l - 32)) - 1u);
    return base + (uint32_t)__builtin_popcount(mask & below);
}
// fences and barriers: lanes run one after the other between rendezvous, so every point where
// one lane's LDS write must be seen by another lane has to be one
inline void emu_fence() { (void)emu::arrive(7, 0); }
#define __builtin_amdgcn_fence(order, scope) emu_fence()
// s_barrier: the wave's lanes meet, then the wave waits for the workgroup's other waves
inline void __syncthreads() {
    (void)emu::arrive(8, 0);
    if (emu::g_waves > 1) {
        // every lane of the wave passes through here one after the other; the first one to come back from
        // the rendezvous holds the wave at the barrier, the others find it open already
        emu::block_barrier();
    }
}
inline void __threadfence_block() { emu_fence(); }
inline void __builtin_amdgcn_wave_barrier() { emu_fence(); }
inline void __builtin_amdgcn_s_setprio(int) {}
inline uint64_t __builtin_readcyclecounter() { return 0; }
inline uint64_t wall_clock64() { return 0; }

// ---- the rest -------------------------------------------------------------------------------
inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
inline uint32_t __brev(uint32_t v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(v);
}
template <class T> inline T atomicAdd(T* p, T v) { const T o = *p; *p = (T)(o + v); return o; }
template <class T> inline T atomicMin(T* p, T v) { const T o = *p; *p = v < o ? v : o; return o; }
template <class T> inline T atomicMax(T* p, T v) { const T o = *p; *p = v > o ? v : o; return o; }
inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p = o | v; return o; }
#define __hip_atomic_load(p, order, scope) (*(p))
#define HIP_SYMBOL(x) x
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }

// kernel launch: every block of the grid, one after the other, 64 fibers each
namespace emu {
template <class F, class Tuple, size_t... I> void call_with(F f, Tuple& t, std::index_sequence<I...>) { f(std::get<I>(t)...); }
template <class F, class... A> struct Thunk {
    F f; std::tuple<A...> args;
    static void run(void* self) { Thunk* t = (Thunk*)self; call_with(t->f, t->args, std::index_sequence_for<A...>{}); }
};
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(kernel, grid, block, __VA_ARGS__)
template <class F, class... A> void emu_launch(F f, dim3 grid, dim3 block, A... args) {
    if (block.x % 64 != 0 || block.x / 64 < 1 || block.x / 64 > (unsigned)emu::MAXW) {
        fprintf(stderr, "emu: workgroups of 1..%d whole waves are emulated\n", emu::MAXW); abort();
    }
    emu::Thunk<F, A...> t{f, std::tuple<A...>(args...)};
    for (unsigned b = 0; b < grid.x; b++) { emu::run_block(&emu::Thunk<F, A...>::run, &t, b, grid.x, block.x / 64); }
}
