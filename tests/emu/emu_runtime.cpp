/* tests/emu/emu_runtime.cpp -- TEST INFRASTRUCTURE ONLY: the fiber scheduler of tests/emu/hip/hip_runtime.h */
#include "hip/hip_runtime.h"

namespace emu {
Lane g_lane[W];
ucontext_t g_main;
int g_cur = 0, g_live = 0;
uint64_t g_slot[2][W];
int g_kind[2][W];
uint32_t g_seq[W];
unsigned g_block = 0, g_grid = 1;
static void (*g_body)(void*);
static void* g_arg;
constexpr size_t kStackBytes = 1u << 20;

static int next_live(int from) {
    for (int k = 1; k <= W; k++) { const int l = (from + k) % W; if (!g_lane[l].done) { return l; } }
    return -1;
}

// round robin: lane i runs to its next rendezvous, then lane i+1, ...; when control comes back to a
// lane, every other live lane has run past the same rendezvous number (they all make the same
// sequence of them), so all their values are in the slots
void rendezvous() {
    const int me = g_cur, nx = next_live(me);
    if (nx < 0 || nx == me) { return; }
    g_cur = nx;
    swapcontext(&g_lane[me].ctx, &g_lane[nx].ctx);
}

static void trampoline() {
    g_body(g_arg);
    const int me = g_cur;
    g_lane[me].done = true;
    g_live--;
    const int nx = next_live(me);
    if (nx < 0) { setcontext(&g_main); }
    g_cur = nx;
    setcontext(&g_lane[nx].ctx);
}

void run_block(void (*body)(void*), void* arg, unsigned block, unsigned grid) {
    g_body = body; g_arg = arg; g_block = block; g_grid = grid;
    g_live = W;
    for (int l = 0; l < W; l++) {
        Lane& L = g_lane[l];
        if (L.stack == nullptr) { L.stack = (char*)malloc(kStackBytes); }
        L.done = false;
        g_seq[l] = 0;
        getcontext(&L.ctx);
        L.ctx.uc_stack.ss_sp = L.stack;
        L.ctx.uc_stack.ss_size = kStackBytes;
        L.ctx.uc_link = nullptr;
        makecontext(&L.ctx, trampoline, 0);
    }
    g_cur = 0;
    swapcontext(&g_main, &g_lane[0].ctx);
    if (g_live != 0) { fprintf(stderr, "emu: %d lanes never finished\n", g_live); abort(); }
}
}  // namespace emu
