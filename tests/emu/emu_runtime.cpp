/* tests/emu/emu_runtime.cpp -- TEST INFRASTRUCTURE ONLY: the fiber scheduler of tests/emu/hip/hip_runtime.h */
#include "hip/hip_runtime.h"

namespace emu {
Wave g_wave[MAXW];
int g_w = 0, g_waves = 1;
ucontext_t g_main;
unsigned g_block = 0, g_grid = 1;
static void (*g_body)(void*);
static void* g_arg;
static uint32_t g_released = 0;          // barriers every wave has passed
constexpr size_t kStackBytes = 1u << 20;

static int next_live(const Wave& w, int from) {
    for (int k = 1; k <= W; k++) { const int l = (from + k) % W; if (!w.lane[l].done) { return l; } }
    return -1;
}
static int next_wave(int from) {         // a wave that still has lanes to run, round robin
    for (int k = 1; k <= g_waves; k++) { const int w = (from + k) % g_waves; if (g_wave[w].live > 0) { return w; } }
    return -1;
}

// round robin inside the wave: lane i runs to its next rendezvous, then lane i+1, ...; when control comes
// back to a lane, every other live lane of its wave has run past the same rendezvous number (they all make
// the same sequence of them), so all their values are in the slots
void rendezvous() {
    Wave& w = g_wave[g_w];
    const int me = w.cur, nx = next_live(w, me);
    if (nx < 0 || nx == me) { return; }
    w.cur = nx;
    swapcontext(&w.lane[me].ctx, &w.lane[nx].ctx);
}

// run another wave (the running fiber stays where it is and goes on when its wave is switched to again)
static void switch_wave(int to) {
    const int from = g_w;
    if (to == from) { return; }
    Wave& a = g_wave[from];
    Wave& b = g_wave[to];
    g_w = to;
    swapcontext(&a.lane[a.cur].ctx, &b.lane[b.cur].ctx);
}

void block_barrier() {
    Wave& w = g_wave[g_w];
    const int me_w = g_w;
    // one lane per wave and barrier does the waiting: the first to come back from the wave's rendezvous;
    // the lanes behind it find the barrier passed already
    const uint32_t mine = ++w.lane_barriers[w.cur];
    if (w.barriers >= mine) { return; }
    w.barriers = mine;
    for (;;) {
        bool all = true;
        for (int k = 0; k < g_waves; k++) { if (g_wave[k].live > 0 && g_wave[k].barriers < w.barriers) { all = false; } }
        if (all) { break; }
        const int nx = next_wave(me_w);
        if (nx < 0 || nx == me_w) {
            fprintf(stderr, "emu: wave %d waits at a workgroup barrier no other wave can reach\n", me_w);
            abort();
        }
        switch_wave(nx);
    }
    if (g_released < w.barriers) { g_released = w.barriers; }
}

static void trampoline() {
    g_body(g_arg);
    Wave& w = g_wave[g_w];
    const int me = w.cur;
    w.lane[me].done = true;
    w.live--;
    const int nx = next_live(w, me);
    if (nx >= 0) { w.cur = nx; setcontext(&w.lane[nx].ctx); }
    // this wave is through: go on with one that is not (a wave waiting at a barrier counts the finished ones out)
    const int nw = next_wave(g_w);
    if (nw < 0) { setcontext(&g_main); }
    g_w = nw;
    Wave& o = g_wave[nw];
    setcontext(&o.lane[o.cur].ctx);
}

void run_block(void (*body)(void*), void* arg, unsigned block, unsigned grid, unsigned waves) {
    g_body = body; g_arg = arg; g_block = block; g_grid = grid;
    g_waves = (int)waves;
    g_released = 0;
    for (int k = 0; k < g_waves; k++) {
        Wave& w = g_wave[k];
        w.live = W; w.cur = 0; w.barriers = 0;
        for (int l = 0; l < W; l++) { w.lane_barriers[l] = 0; }
        for (int l = 0; l < W; l++) {
            Lane& L = w.lane[l];
            if (L.stack == nullptr) { L.stack = (char*)malloc(kStackBytes); }
            L.done = false;
            w.seq[l] = 0;
            getcontext(&L.ctx);
            L.ctx.uc_stack.ss_sp = L.stack;
            L.ctx.uc_stack.ss_size = kStackBytes;
            L.ctx.uc_link = nullptr;
            makecontext(&L.ctx, trampoline, 0);
        }
    }
    g_w = 0;
    swapcontext(&g_main, &g_wave[0].lane[0].ctx);
    for (int k = 0; k < g_waves; k++) {
        if (g_wave[k].live != 0) { fprintf(stderr, "emu: wave %d: %d lanes never finished\n", k, g_wave[k].live); abort(); }
    }
}
}  // namespace emu
