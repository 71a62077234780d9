// sqz_amd/csrc/sqz_tree.h -- the adaptive Huffman trees of one stream, LDS resident, and their
// updates (gfx950 only).  Shared by huffman_emit.hip and decode.hip.
//
// Reference semantics restated here (file:line relative to /root/reference/attic/map_experiment):
//   huffman.h:13-34   node / tree           -> TreeLds (three words per node), Tree<> (registers)
//   huffman.h:41-62   huffman_update_paths  -> depths, codes and leaf positions kept by FLAT passes
//                                              (swap_fix / promote_fix / the insert's shift) + the
//                                              depth mark (mark_subtree); relabel() for deep trees
//   huffman.h:64-86   huffman_swap_siblings -> order_only (+ swap_fix), the per-level swaps of climb_wave
//   huffman.h:90-96   huffman_update_freq   -> sum
//   huffman.h:98-147  move_up / frequency_changed -> changed_all (whole wave), changed (one lane, deep trees)
//   huffman.h:149-216 huffman_insert        -> insert_wave
//   huffman.h:218-235 huffman_inc_frequency -> bump_wave (one symbol), bump_batch (up to 64 tokens)
//
// One wavefront owns one stream; both trees live in one set of LDS arrays (absolute node ids:
// the distance tree's ids start at kPosBase, its leaf positions at kPosPos0).
//
// THE BATCHED UPDATE.  The reference updates one symbol at a time; 98 % of the updates change no
// link.  For a batch of symbols the reference would change no link iff, for every node v that
// some symbol's leaf->root chain passes (parent p, sibling s, uncle u; f = counts before the
// batch, n(v) = chains through v):
//     v is lo(p):                 f(v) + n(v) <= f(s)      (never overtakes its sibling)
//     v is hi(p), p not the root: f(v) + n(v) <= f(u)      (never overtakes its uncle)
// (induction over the sequential updates: counts only grow, v's count never exceeds f(v)+n(v),
// s and u never drop below f; a hi child needs no test against its own sibling because every
// pair is ordered lo <= hi at rest and the reference's exchange can only be triggered from the
// lo side -- the oracle asserts both at-rest properties after every update, `make -C oracle
// check-invariants`).  Then the batch leaves f + n on every node and every code is the static
// tree's.  The tests may be conservative: the token a batch stops at goes through the exact
// one-at-a-time path, so the output is always the reference's.
//
// n(v) WITHOUT WALKING ANY CHAIN.  Every leaf has a position in depth-first order (lo before
// hi); the leaves below a node are the positions from its FIRST to its LAST leaf, and a node
// remembers those two leaves (not the numbers: an insert shifts positions, leaf identities stay).
// With P = inclusive prefix sums of the batch's histogram over positions,
// n(v) = P[pos(last(v))] - P[pos(first(v)) - 1] for ALL nodes at once: independent lookups per
// node (eight probes per lane) instead of two dependent walks per token with an LDS atomic on every
// level.  A node's test partner (sibling for a lo child, uncle for a hi child) is cached next to
// its ends.  If node v fails its test, the first token that may not be applied is the
// (f(partner) - f(v) + 1)-th one whose position lies below v; the earliest such token over all
// failing nodes ends the batch.  tests/model/range_model.c is the first form of this algorithm
// (numeric intervals) in plain C, held against the oracle's tree in lockstep
// (tests/test_range_model.py); the kernels themselves run on the CPU in tests/emu.
//
// KEEPING POSITIONS, ENDS AND PARTNERS.  A restructure (2 % of the symbols) moves whole subtrees;
// the exact path does the reference's sequence and repairs positions, depths, per-leaf codes with
// passes over the LEAVES only (one lane per leaf, a few selects), and ends / partners of the few
// internal nodes on the climbed chain from registers (ballots + ds_bpermute):
//     sibling exchange under p:  the two halves of p's leaf range trade places; one code bit flips
//     promotion under g:         three neighbouring ranges rotate; one subtree comes up a level
//                                (a code bit disappears), one goes down (a bit appears)
//     insert:                    every position behind the split leaf moves one to the right
//
// Node words (LDS, 12 B per node):
//     lnk  up | lo << 10 | hi << 20                    (10-bit absolute ids, 0x3FF = none)
//     rng  leaf:     position (9 bits, 0x1FF = not in the tree)  | partner << 20
//          internal: first leaf | last leaf << 10                | partner << 20   (0x3FF = untested)
//     cnt  count (24 bits) | depth << 24 (6 bits; leaves only)     -- "aux" mode, trees < kAuxDepth
//          count (32 bits), depths recomputed on demand            -- wide mode, for good once a
//          tree reaches kAuxDepth or a stream kBatchTokens tokens (give_up_aux)
// and for the encoder code[leaf slot]: the leaf's code in STREAM order (first branch = most
// significant of `depth` bits), valid while the tree is shallower than kAuxDepth.
#pragma once

#include "sqz_device.h"

namespace sqzk {

constexpr uint32_t kNil = 0x3FFu;
constexpr int kLitLeaves = 288;               // symbols 0..285 (+2 pad)
constexpr int kLitNodes  = kLitLeaves + 288;  // root + <=285 splits (+pad)
constexpr int kPosLeaves = 32;
constexpr int kPosNodes  = 64;
constexpr int kPosBase   = kLitNodes;         // first node id of the distance tree
constexpr int kAllNodes  = kLitNodes + kPosNodes;
constexpr int kPosPos0   = kLitLeaves;        // first leaf position of the distance tree
constexpr int kPositions = kLitLeaves + kPosLeaves;     // 320 leaf positions in all
constexpr int kCodeSlots = kPositions;        // code[]: lit leaf s -> s, pos leaf k -> kPosPos0 + k

constexpr uint32_t kCountMask = (1u << 24) - 1u;
constexpr int kDepthShift = 24;

// Trees this deep leave the fast machinery for good (batches, position passes, 32-bit codes):
// every symbol then takes the reference sequence on the chain held by the wave (depth < kMaxFastDepth)
// or by one lane.  Depth grows by one per ~1.7 x more symbols at best (blocks of literals, each new
// symbol 1.7 x as frequent as the last: 1.2e7 literals reach depth 27), so kAuxDepth is within reach
// of long streams (tested in the shipping build) while kMaxFastDepth and kFreezeDepth are not under
// 2^31 bytes; all three can be lowered at build time so that tests reach the deep paths
// (python -m sqz_amd.build --variants).
#ifndef SQZ_AUX_DEPTH
#define SQZ_AUX_DEPTH 26
#endif
#ifndef SQZ_MAX_FAST_DEPTH
#define SQZ_MAX_FAST_DEPTH 60
#endif
#ifndef SQZ_FREEZE_DEPTH
#define SQZ_FREEZE_DEPTH 63                   // huffman.h:228 `t->depth < 63`
#endif
#ifndef SQZ_BATCH_TOKENS
#define SQZ_BATCH_TOKENS ((1u << 24) - 256u)  // a count word holds 24 bits while the positions are kept
#endif
constexpr uint32_t kBatchTokens = SQZ_BATCH_TOKENS;
constexpr int kAuxDepth = SQZ_AUX_DEPTH;
constexpr int kMaxFastDepth = SQZ_MAX_FAST_DEPTH;
constexpr int kFreezeDepth = SQZ_FREEZE_DEPTH;
constexpr int kStack = 128;                   // deepest chain the one-lane path follows (fault beyond)
// most bits one token hands to a pack: two codes shorter than kAuxDepth bits + 5 + 13 extra bits
constexpr int kTokenBits = 2 * (kAuxDepth - 1) + 18;
constexpr int kPackWords = (63 + 58 + kWave * kTokenBits + 63) / 64 + 1;     // carry + pending + 64 tokens
constexpr int kImageWords = kPackWords > kWave ? kPackWords : kWave;         // (the histogram takes 64 words)

#if defined(SQZ_STATS) && defined(SQZ_SEC_TIMERS)
__device__ unsigned long long g_sec[32];
__device__ unsigned int g_secn[32];
#define SEC_BEGIN uint64_t sec_t_ = __builtin_readcyclecounter();
#define SEC(k) { const uint64_t n_ = __builtin_readcyclecounter(); if (blockIdx.x == 1 && threadIdx.x == 0) { g_sec[k] += n_ - sec_t_; g_secn[k]++; } sec_t_ = n_; }
// nested sections of the exact path (each function keeps its own start stamp)
#define XSEC_BEGIN uint64_t xsec_t_ = __builtin_readcyclecounter();
#define XSEC(k) { const uint64_t n_ = __builtin_readcyclecounter(); if (blockIdx.x == 1 && threadIdx.x == 0) { g_sec[k] += n_ - xsec_t_; g_secn[k]++; } xsec_t_ = n_; }
#else
#define SEC_BEGIN
#define SEC(k)
#define XSEC_BEGIN
#define XSEC(k)
#endif

struct TreeLds {
    uint32_t lnk[kAllNodes];
    uint32_t rng[kAllNodes];
    uint32_t cnt[kAllNodes];
    uint32_t pend[kWave];                     // parent << 16 | child, one per pending level
    uint16_t lvl[kStack];                     // the one-lane path's stack
    // one batch: histogram over leaf positions (a byte each), turned into its prefix sums in place
    // (the first 64 words); the encoder's bit image of a step lives here too, after the update
    union { uint64_t P64[kImageWords]; uint32_t P32[2 * kImageWords]; uint8_t P8[8 * kImageWords]; };
};

// huffman.h:29-33 + squeeze.h:397-403, kept only by the kernels' kStats instantiation
struct TreeStats { uint32_t updates, swaps, moves; };

// Lanes of the wave talk to each other through LDS: a store or atomic by one lane, then a load by
// another.  The hardware executes one wave's LDS instructions in issue order, so all that is
// needed is that the COMPILER keeps them in program order -- a release fence alone does not (it
// lets a later load move above an earlier atomic on a "different" type; seen on gfx950: the
// histogram's ds_add landed after the ds_read of the prefix sum).  Full fence, wavefront scope:
// no instruction is emitted for it.
__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ uint32_t l_up(uint32_t w) { return w & 0x3FFu; }
__device__ __forceinline__ uint32_t l_lo(uint32_t w) { return (w >> 10) & 0x3FFu; }
__device__ __forceinline__ uint32_t l_hi(uint32_t w) { return (w >> 20) & 0x3FFu; }
__device__ __forceinline__ uint32_t mk_lnk(uint32_t up, uint32_t lo, uint32_t hi) { return up | (lo << 10) | (hi << 20); }
// rng word: a leaf's position (low 9 bits), an internal node's first | last << 10 leaf; the partner on top
__device__ __forceinline__ uint32_t r_pos(uint32_t w) { return w & 0x1FFu; }
__device__ __forceinline__ uint32_t r_first(uint32_t w) { return w & 0x3FFu; }
__device__ __forceinline__ uint32_t r_last(uint32_t w) { return (w >> 10) & 0x3FFu; }
__device__ __forceinline__ uint32_t r_pa(uint32_t w) { return (w >> 20) & 0x3FFu; }
constexpr uint32_t kEndsMask = 0xFFFFFu;      // everything below the partner
constexpr uint32_t kNoPos = 0x1FFu;           // position field of a leaf that is not in the tree
__device__ __forceinline__ uint32_t c_f(uint32_t w) { return w & kCountMask; }
__device__ __forceinline__ uint32_t c_d(uint32_t w) { return (w >> kDepthShift) & 0x3Fu; }

// maximum over the wave: the same six DPP steps as wave_scan, the total ends up in lane 63
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, kWave - 1);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan(v), kWave - 1);
}

// one lane's share of a root path: its node and that node's parent
struct Chain {
    int mine, par;
    int levels;        // uniform: edges between the start node and the root
    bool holds;        // this lane holds a node of the path
    bool active;       // ... and the node has a parent
    bool has_g;        // ... and that parent is not the root
};

// The slow paths live in real (non-inlined) functions so that the per-step loop of the kernels
// stays small (I-cache).  State crosses the call as plain values.
#ifndef SQZ_INSERT_INLINING
#define SQZ_INSERT_INLINING __noinline__
#endif
#ifndef SQZ_LUT_INLINING
#define SQZ_LUT_INLINING __noinline__
#endif
template <class T> __device__ SQZ_INSERT_INLINING uint32_t slow_insert(TreeLds* lds, uint32_t* code, uint32_t regs, int sym, int lane);
#ifndef SQZ_CHANGED_INLINING
#define SQZ_CHANGED_INLINING __noinline__
#endif
template <class T> __device__ SQZ_CHANGED_INLINING uint32_t slow_changed(TreeLds* lds, uint32_t* code, uint32_t regs, int sym, int mine, int levels, int lane);
template <class T> __device__ SQZ_LUT_INLINING void slow_build_lut(const uint32_t* lnk, uint16_t* lut, int lane);

// BASE: first node id; LEAVES / NODES: id space (leaves keep their symbol value + BASE, internal
// nodes are numbered upwards from the root = BASE + LEAVES; the reference counts down from 2n-2, no
// emitted bit depends on it); REF_LEAVES: the reference's leaf count n (512 / 32, squeeze.h:204-205),
// which only fixes how many leaf splits huffman_insert allows (n - 2, huffman.h:180); POS0: first
// leaf position; LUT_BITS: the decoder's lookup table width.
template <int BASE, int LEAVES, int NODES, int REF_LEAVES, int POS0, int LUT_BITS, bool CODES, bool STATS>
struct Tree {
    TreeLds* lds;
    uint32_t* code;     // encoder only (CODES): kCodeSlots entries
    uint16_t* lut;      // decoder only: 2^kLutBits entries, node | bits used << 10
    // wave-uniform registers
    int next;           // next free internal id
    int depth;          // huffman.h:26 high-water mark
    int complete;       // huffman.h:27
    int fault;          // stack / depth guard (never set for realistic streams)
    int aux;            // leaf positions / ends / partners / codes are kept (tree shallower than kAuxDepth so far)
    int lut_ok;         // decoder: the lookup table matches the tree
    int in_insert;      // between a leaf split and the closing walk of huffman_insert the depth mark may lag behind
    TreeStats stats;

    static constexpr int kBase = BASE;
    static constexpr int kRoot = BASE + LEAVES;
    static constexpr int kLeaves = LEAVES;
    static constexpr int kLutBits = LUT_BITS;
    static constexpr int kChgSlot = BASE == 0 ? 57 : 59;   // pend[]: positions whose codes an exact step changed (decoder)
    static constexpr int kIdEnd = BASE + ((LEAVES + 1 + REF_LEAVES - 2) < NODES ? (LEAVES + 1 + REF_LEAVES - 2) : NODES);
    static constexpr bool kCodes = CODES;
    static constexpr int kLeafRows = (LEAVES + kWave - 1) / kWave;

    __device__ __forceinline__ void init_all(int lane) {      // huffman.h:251-269
        for (int i = BASE + lane; i < BASE + NODES; i += kWave) {
            lds->lnk[i] = 0x3FFFFFFFu;
            lds->rng[i] = kNoPos | (kNil << 20);               // a leaf outside the tree is at no position
            lds->cnt[i] = 0;
        }
        next = kRoot + 1; depth = 0; complete = 0; fault = 0; aux = 1; lut_ok = 0; in_insert = 0;
        stats.updates = stats.swaps = stats.moves = 0;
    }

    __device__ __forceinline__ bool is_leaf(uint32_t v) const { return v < (uint32_t)kRoot; }
    __device__ __forceinline__ uint32_t up_of(int i) const { return l_up(lds->lnk[i]); }
    // While a tree keeps its positions (aux) a count word is count (24 bits) | depth << 24 (the
    // depth of leaves only; internal nodes keep none) and a stream has fewer than 2^24 tokens; once it
    // has given them up (give_up_aux: a very deep tree, or the kernel's token limit) the word is the
    // full 32-bit count and depths are worked out on demand.
    __device__ __forceinline__ uint32_t freq(int i) const { const uint32_t w = lds->cnt[i]; return aux != 0 ? c_f(w) : w; }
    __device__ __forceinline__ void set_freq(int i, uint32_t f) {
        lds->cnt[i] = aux != 0 ? ((lds->cnt[i] & ~kCountMask) | (f & kCountMask)) : f;
    }
    __device__ __forceinline__ int depth_by_walk(int v) const {
        int d = 0;
        for (uint32_t a = l_up(lds->lnk[v]); a != kNil && d < kStack; a = l_up(lds->lnk[a])) { d++; }
        return d;
    }
    // leave the position machinery for good: counts become full words (the depth bits go)
    __device__ __forceinline__ void give_up_aux(int lane) {
        if (aux == 0) { return; }
        aux = 0;
        for (int v = BASE + lane; v < BASE + NODES; v += kWave) { lds->cnt[v] &= kCountMask; }
        lds_fence();
    }
    __device__ __forceinline__ int code_slot(int leaf) const { return leaf - BASE + POS0; }
    // first / last leaf below a node, position of a leaf
    __device__ __forceinline__ uint32_t first_leaf(uint32_t v) const { return is_leaf(v) ? v : r_first(lds->rng[v]); }
    __device__ __forceinline__ uint32_t last_leaf(uint32_t v) const { return is_leaf(v) ? v : r_last(lds->rng[v]); }
    __device__ __forceinline__ uint32_t pos_of(uint32_t leaf) const { return r_pos(lds->rng[leaf]); }
    __device__ __forceinline__ void set_ends(int v, uint32_t first, uint32_t last) {
        lds->rng[v] = (lds->rng[v] & ~kEndsMask) | first | (last << 10);
    }

    // registers <-> one word: next:10 | depth:8 | complete | fault | aux ; bit 31 = the call's own result
    __device__ __forceinline__ uint32_t pack_regs() const {
        return (uint32_t)next | ((uint32_t)(depth & 0xFF) << 10) | ((uint32_t)(complete & 1) << 18) |
               ((uint32_t)(fault & 1) << 19) | ((uint32_t)(aux & 1) << 20);
    }
    __device__ __forceinline__ void unpack_regs(uint32_t r) {
        next = (int)(r & 0x3FFu); depth = (int)((r >> 10) & 0xFFu);
        complete = (int)((r >> 18) & 1u); fault = (int)((r >> 19) & 1u); aux = (int)((r >> 20) & 1u);
    }
    __device__ __forceinline__ void uniform_regs() {
        next = __builtin_amdgcn_readfirstlane(next);
        depth = __builtin_amdgcn_readfirstlane(depth);
        complete = __builtin_amdgcn_readfirstlane(complete);
        fault = __builtin_amdgcn_readfirstlane(fault);
        aux = __builtin_amdgcn_readfirstlane(aux);
    }
    // the reference's high-water mark moved; a tree that is (now) this deep gives up the position machinery.
    // (d is the depth of the deepest leaf the restructure touched -- every change of a depth comes through
    // here with it -- so the machinery follows the tree as it IS; the mark only gates the freeze, huffman.h:228)
    __device__ __forceinline__ void raise_mark(int d, int lane) {
        if (d > depth) { depth = d; }
        if (d >= kAuxDepth) { give_up_aux(lane); }
    }

    // Back to the position machinery.  A burst of symbols seen once each builds a chain down the tree's left
    // edge (huffman_insert always splits the leftmost leaf it reaches, and nothing moves while all counts are
    // equal): uniform random bytes pass depth 26 within their first fifty symbols in one 256 KB block out of
    // thirty -- and that block then took the one-at-a-time path for all its 260,000 tokens, ten times the cost
    // of its neighbours, although its tree had settled at depth 10 after a few hundred symbols.  Everything the
    // machinery keeps is a function of the links: it is rebuilt here, whole wave, once the tree is shallow
    // again (with a margin, so that it does not flap): every leaf's depth and code by a walk to the root, its
    // position as its rank among the codes (depth-first order IS the order of the left-aligned codes), every
    // internal node's first and last leaf by a walk down, every node's test partner.  About 40 k cycles.
    // Returns whether the tree keeps positions afterwards.
    __device__ __forceinline__ bool regain_aux(int lane) {
        if (aux != 0) { return true; }
        if (complete != 0) { return false; }
        constexpr int kInnerRows = (NODES - LEAVES + kWave - 1) / kWave;
        constexpr uint32_t kRegainDepth = kAuxDepth > 8 ? kAuxDepth - 4 : (kAuxDepth > 2 ? kAuxDepth - 2 : 1);
        uint32_t key[kLeafRows], dep[kLeafRows], cod[kLeafRows];
        bool in[kLeafRows];
        bool deep = false;
#pragma unroll
        for (int r = 0; r < kLeafRows; r++) {
            const int v = BASE + r * kWave + lane;
            const int vv = v < kRoot ? v : kRoot - 1;
            in[r] = v < kRoot && l_up(lds->lnk[vv]) != kNil;
            uint32_t a = (uint32_t)vv, c = 0, d = 0;
            for (int it = 0; it <= kAuxDepth; it++) {
                const uint32_t up = l_up(lds->lnk[a]);
                const bool more = in[r] && up != kNil;
                if (__ballot(more) == 0) { break; }
                const uint32_t pw = lds->lnk[more ? up : (uint32_t)kRoot];
                if (more) { c |= (l_hi(pw) == a ? 1u : 0u) << d; d++; a = up; }
            }
            deep |= in[r] && (d > kRegainDepth || l_up(lds->lnk[a]) != kNil);
            dep[r] = d; cod[r] = c;
            key[r] = d != 0 ? c << (32u - d) : 0u;
        }
        if (__ballot(deep) != 0) { return false; }
        // ranks: the keys wait in the leaves' range words (stale while the positions were given up)
#pragma unroll
        for (int r = 0; r < kLeafRows; r++) {
            const int v = BASE + r * kWave + lane;
            if (v < kRoot) { lds->rng[v] = in[r] ? key[r] : 0xFFFFFFFFu; }
        }
        lds_fence();
        uint32_t rank[kLeafRows];
#pragma unroll
        for (int r = 0; r < kLeafRows; r++) { rank[r] = 0; }
        for (int u = 0; u < LEAVES; u += 4) {
            const uint32_t k0 = lds->rng[BASE + u], k1 = lds->rng[BASE + u + 1], k2 = lds->rng[BASE + u + 2], k3 = lds->rng[BASE + u + 3];
#pragma unroll
            for (int r = 0; r < kLeafRows; r++) {
                rank[r] += (k0 < key[r] ? 1u : 0u) + (k1 < key[r] ? 1u : 0u) + (k2 < key[r] ? 1u : 0u) + (k3 < key[r] ? 1u : 0u);
            }
        }
        lds_fence();
        // a node's test partner: its sibling for a lo child, its parent's sibling for a hi child (none at the top)
        auto partner_of = [&](uint32_t v) {
            const uint32_t p = l_up(lds->lnk[v]);
            uint32_t pa = kNil;
            if (p != kNil) {
                const uint32_t pw = lds->lnk[p];
                if (l_hi(pw) != v) { pa = l_hi(pw); }
                else if (l_up(pw) != kNil) {
                    const uint32_t gw = lds->lnk[l_up(pw)];
                    pa = l_lo(gw) == p ? l_hi(gw) : l_lo(gw);
                }
            }
            return pa;
        };
#pragma unroll
        for (int r = 0; r < kLeafRows; r++) {
            const int v = BASE + r * kWave + lane;
            if (v < kRoot) {
                if (in[r]) {
                    lds->rng[v] = ((uint32_t)POS0 + rank[r]) | (partner_of((uint32_t)v) << 20);
                    lds->cnt[v] = (lds->cnt[v] & kCountMask) | (dep[r] << kDepthShift);
                    if (CODES) { code[code_slot(v)] = cod[r]; }
                } else {
                    lds->rng[v] = kNoPos | (kNil << 20);
                }
            }
        }
        // internal nodes: first / last leaf by a walk down the lo / hi edge (a missing lo child: the hi one)
#pragma unroll
        for (int r = 0; r < kInnerRows; r++) {
            const int v = kRoot + r * kWave + lane;
            const bool on = v < next;
            uint32_t f = on ? (uint32_t)v : (uint32_t)kRoot, l = f;
            for (int it = 0; it <= kAuxDepth; it++) {
                const bool mf = f >= (uint32_t)kRoot, ml = l >= (uint32_t)kRoot;
                if (__ballot(mf | ml) == 0) { break; }
                const uint32_t wf = lds->lnk[mf ? f : (uint32_t)kRoot], wl = lds->lnk[ml ? l : (uint32_t)kRoot];
                if (mf) { f = l_lo(wf) != kNil ? l_lo(wf) : l_hi(wf); }
                if (ml) { l = l_hi(wl) != kNil ? l_hi(wl) : l_lo(wl); }
                if (f == kNil) { f = (uint32_t)kRoot - 1u; }           // (an empty tree: nothing to point at)
                if (l == kNil) { l = (uint32_t)kRoot - 1u; }
            }
            if (v < BASE + NODES) {
                lds->rng[v] = on ? (f | (l << 10) | (partner_of((uint32_t)v) << 20)) : (kNoPos | (kNil << 20));
                if (on) { lds->cnt[v] &= kCountMask; }
            }
        }
        lds_fence();
        aux = 1;
        return true;
    }

    // ---------------- passes over the leaves: one lane per leaf --------------------------------
    // The depth mark as huffman_update_paths(top) leaves it (huffman.h:44,61): reset when top is
    // the root, then the deepest node of top's subtree -- always a leaf, one of those at positions
    // [a, b).  The walk visits every node of the subtree once (huffman.h:42): 2 * leaves - 1 of them.
    __device__ __forceinline__ void mark_range(int top, uint32_t a, uint32_t b, uint32_t deepest_known, int lane) {
        // Every restructure comes through here with the positions [a, b) of the leaves below the node
        // whose paths the reference recomputes (huffman_update_paths): those and only those leaves may
        // have a new code.  The decoder keeps the union per exact step (pend[kChgSlot], [kChgSlot + 1];
        // nothing is pending at that level in a tree this shallow): tokens it has read ahead whose
        // leaves lie outside it are still what the new tree decodes at their bit offsets.
        if (!CODES && lane == 0) {
            atomicMin(&lds->pend[kChgSlot], a);
            atomicMax(&lds->pend[kChgSlot + 1], b);
        }
        uint32_t deepest = deepest_known;
        // At rest the mark is at least the depth of every node (a walk from the root sets it to the true
        // maximum, every later walk raises it to what it saw): a walk over unchanged depths cannot move it,
        // except from the root (restart) and inside huffman_insert (the split's new level is not in it yet).
        if (deepest_known == 0 && (top == kRoot || in_insert != 0)) {
#pragma unroll
            for (int r = 0; r < kLeafRows; r++) {
                const int v = BASE + r * kWave + lane;
                if (BASE + r * kWave >= kRoot) { break; }
                const int vv = v < kRoot ? v : kRoot;
                const uint32_t q = r_pos(lds->rng[vv]);
                const uint32_t d = c_d(lds->cnt[vv]);
                const bool in = v < kRoot && q >= a && q < b;
                deepest = (in && d > deepest) ? d : deepest;
            }
            deepest = wave_max(deepest);
        }
        if (top == kRoot) { depth = 0; }
        raise_mark((int)deepest, lane);
        if (STATS) { stats.updates += 2u * (b - a) - ((uint32_t)top == (uint32_t)kRoot && l_lo(lds->lnk[kRoot]) == kNil ? 0u : 1u); }
    }
    __device__ __forceinline__ void mark_subtree(int top, int lane) {
        const uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos_of(first_leaf((uint32_t)top)));
        const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos_of(last_leaf((uint32_t)top))) + 1u;
        mark_range(top, a, b, 0u, lane);
    }

    // test partners of the nodes up to `levels` levels below `top` (they are the only ones whose
    // sibling or uncle a restructure at `top` can have changed): lane j walks down to the j-th
    // of them (children 0..1, grandchildren 2..5, great-grandchildren 6..13), then up twice
    __device__ __forceinline__ void fix_partners(int top, int levels, int lane) {
        const int j = lane + 2;                                // 2..15: the path below `top` in binary, leading 1 dropped
        const int d = 31 - __clz(j);                           // 1..3 levels down
        uint32_t v = (uint32_t)top;
        bool ok = lane < (2 << levels) - 2;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (k < d) {
                const uint32_t w = lds->lnk[ok ? v : (uint32_t)kRoot];
                const uint32_t nx = ((j >> (d - 1 - k)) & 1) ? l_hi(w) : l_lo(w);
                ok = ok && v >= (uint32_t)kRoot && nx != kNil;
                v = ok ? nx : v;
            }
        }
        if (ok) {
            const uint32_t p = l_up(lds->lnk[v]);
            const uint32_t pw = lds->lnk[p];
            uint32_t pa = kNil;
            if (l_hi(pw) != v) { pa = l_hi(pw); }
            else if (l_up(pw) != kNil) {
                const uint32_t gw = lds->lnk[l_up(pw)];
                pa = l_lo(gw) == p ? l_hi(gw) : l_lo(gw);
            }
            lds->rng[v] = (lds->rng[v] & kEndsMask) | (pa << 20);
        }
        lds_fence();
    }

    // The two children of p have traded slots (the links already say so): X = the child now in the
    // lo slot (its leaves, positions [m, b), come first from now on), Y = the child now hi (was
    // [a, m)).  Leaf positions move, one code bit of every leaf below p flips, p's ends and the
    // tests of X and Y follow; the depth mark and the statistics see the walk
    // huffman_swap_siblings makes (huffman.h:76-80).  dp = depth of p; fX..lY = first / last leaf of X
    // and Y; y_partner = the sibling of p (Y's uncle from now on; kNil when p is the root).
    __device__ __forceinline__ void swap_fix(int p, uint32_t dp, uint32_t X, uint32_t Y, uint32_t fX, uint32_t lX,
                                             uint32_t fY, uint32_t lY, uint32_t y_partner, int lane) {
        if (STATS) { stats.swaps += 1; }
        if (aux == 0) {                                        // wide mode: the reference's walk, one lane
            if (lane == 0) { relabel(p); }
            uniform_regs();
            lds_fence();
            return;
        }
        XSEC_BEGIN
        // (read by one lane and broadcast: the pass below rewrites positions)
        const uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos_of(fY));
        const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos_of(fX));
        const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos_of(lX)) + 1u;
        const bool want_depth = p == kRoot || in_insert != 0;
        uint32_t deepest = 0;
        // one lane per leaf, every row's words read FIRST (a wave alone on its SIMD waits ~100 cycles for
        // each LDS round trip it takes one after the other: five rows behind five branches were five to
        // fifteen of them), then the moves
        uint32_t ws[kLeafRows], cs[kLeafRows], ds[kLeafRows];
#pragma unroll
        for (int r = 0; r < kLeafRows; r++) {
            const int v = BASE + r * kWave + lane;
            const int vv = v < kRoot ? v : kRoot - 1;
            ws[r] = lds->rng[vv];
            cs[r] = (CODES | want_depth) ? lds->cnt[vv] : 0u;
            ds[r] = CODES ? code[code_slot(vv)] : 0u;
        }
#pragma unroll
        for (int r = 0; r < kLeafRows; r++) {
            const int v = BASE + r * kWave + lane;
            if (BASE + r * kWave >= kRoot) { continue; }
            const uint32_t w = ws[r];
            const uint32_t q = r_pos(w);
            const bool in_x = v < kRoot && q >= m && q < b, in_y = v < kRoot && q >= a && q < m;
            if (in_x | in_y) {
                lds->rng[v] = (w & ~0x1FFu) | (in_x ? q - (m - a) : q + (b - m));
                if (CODES | want_depth) {
                    const uint32_t d = c_d(cs[r]);
                    if (CODES) { code[code_slot(v)] = ds[r] ^ (1u << (d - 1u - dp)); }
                    deepest = d > deepest ? d : deepest;
                }
            }
        }
        if (lane == 0) {
            set_ends(p, fX, lY);
            lds->rng[X] = (lds->rng[X] & kEndsMask) | (Y << 20);          // a lo child is tested against its sibling
            lds->rng[Y] = (lds->rng[Y] & kEndsMask) | (y_partner << 20);  // a hi child against its uncle
        }
        lds_fence();
        XSEC(13)
        mark_range(p, a, b, want_depth ? wave_max(deepest) : 0u, lane);
        XSEC(14)
    }
    // the same when the caller knows nothing but p (rare: the order checks of a promotion, the root's pair)
    __device__ __forceinline__ void swap_fix_at(int p, uint32_t dp, int lane) {
        const uint32_t pw = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds->lnk[p]);
        const uint32_t X = l_lo(pw), Y = l_hi(pw);
        uint32_t unc = kNil;
        if (l_up(pw) != kNil) {
            const uint32_t gw = lds->lnk[l_up(pw)];
            unc = l_lo(gw) == (uint32_t)p ? l_hi(gw) : l_lo(gw);
        }
        swap_fix(p, dp, X, Y, first_leaf(X), last_leaf(X), first_leaf(Y), last_leaf(Y), unc, lane);
    }

    // c (hi child of p) and its uncle u have traded places under g (the links already say so):
    //   left  (p = lo(g)):  [x][c][u] -> [x][u][c]      c: G01S -> G1S     u: G1S -> G01S
    //   right (p = hi(g)):  [u][x][c] -> [c][x][u]      c: G11S -> G0S     u: G0S -> G11S
    // (G = the code of g, dg bits; S = what follows below the moved node).  c's leaves come up a level.
    __device__ __forceinline__ void promote_fix(int g, uint32_t dg, int p, int c, int u, int left,
                                                uint32_t& ga, uint32_t& gb, uint32_t& deepest_out, int lane) {
        if (STATS) { stats.moves += 1; }
        if (aux == 0) { return; }
        const uint32_t x = l_lo(lds->lnk[p]);
        const uint32_t fc = first_leaf((uint32_t)c), lc = last_leaf((uint32_t)c), fu = first_leaf((uint32_t)u);
        const uint32_t lu = last_leaf((uint32_t)u), fx = first_leaf(x), lx = last_leaf(x);
        const uint32_t ca = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos_of(fc));
        const uint32_t cb = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos_of(lc)) + 1u;
        const uint32_t ua = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos_of(fu));
        const uint32_t ub = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos_of(lu)) + 1u;
        const uint32_t xa = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos_of(fx));
        const uint32_t xb = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos_of(lx)) + 1u;
        const int C = (int)(cb - ca), U = (int)(ub - ua), X = (int)(xb - xa);
        const int dc = left ? U : -(U + X), du = left ? -C : C + X, dx = left ? 0 : C - U;
        ga = left ? xa : ua;                                   // g's leaves: the three runs are neighbours
        gb = ga + (uint32_t)(C + U + X);
        uint32_t deepest = 0;
        uint32_t ws[kLeafRows], cs[kLeafRows], ds[kLeafRows];          // (all rows' words first: see swap_fix)
#pragma unroll
        for (int r = 0; r < kLeafRows; r++) {
            const int v = BASE + r * kWave + lane;
            const int vv = v < kRoot ? v : kRoot - 1;
            ws[r] = lds->rng[vv];
            cs[r] = lds->cnt[vv];
            ds[r] = CODES ? code[code_slot(vv)] : 0u;
        }
#pragma unroll
        for (int r = 0; r < kLeafRows; r++) {
            const int v = BASE + r * kWave + lane;
            if (BASE + r * kWave >= kRoot) { continue; }
            const uint32_t w = ws[r], cw = cs[r];
            const uint32_t q = r_pos(w), d = c_d(cw);
            const bool live = v < kRoot;                         // (a leaf outside the tree is at no position)
            const bool in_c = live && q >= ca && q < cb, in_u = live && q >= ua && q < ub, in_x = live && q >= xa && q < xb;
            if (in_c | in_u | in_x) {
                const int shift = in_c ? dc : in_u ? du : dx;
                lds->rng[v] = (w & ~0x1FFu) | (uint32_t)((int)q + shift);
                const uint32_t nd = in_c ? d - 1u : in_u ? d + 1u : d;
                deepest = nd > deepest ? nd : deepest;
                if (in_c | in_u) {
                    lds->cnt[v] = in_c ? cw - (1u << kDepthShift) : cw + (1u << kDepthShift);
                    if (CODES) {
                        const uint32_t old = ds[r];
                        const uint32_t ls = in_c ? d - dg - 2u : d - dg - 1u;      // bits below the moved node
                        const uint32_t G = dg != 0 ? old >> (d - dg) : 0u;
                        const uint32_t S = old & ((1u << ls) - 1u);
                        const uint32_t head = in_c ? ((G << 1) | (left ? 1u : 0u)) : ((G << 2) | (left ? 1u : 3u));
                        code[code_slot(v)] = (head << ls) | S;
                    }
                }
            }
        }
        if (lane == 0) {                                       // p now holds x and u; g's ends follow its new children
            set_ends(p, fx, lu);
            if (left) { set_ends(g, fx, lc); } else { set_ends(g, fc, lu); }
        }
        lds_fence();
        fix_partners(g, 3, lane);
        deepest_out = wave_max(deepest);
    }

    // ---------------- the reference sequence on one lane (wide mode / deep trees) ----------------
    __device__ __forceinline__ void sum(int i) {               // huffman.h:90-96
        const uint32_t w = lds->lnk[i];
        const uint32_t a = l_lo(w) != kNil ? freq((int)l_lo(w)) : 0u;
        const uint32_t b = l_hi(w) != kNil ? freq((int)l_hi(w)) : 0u;
        set_freq(i, a + b);
    }

    // huffman.h:41-62 by a walk (wide mode only: no stored depths): the mark and the statistics.  A
    // stack entry is node | depth << 10.
    __device__ __forceinline__ void relabel(int top) {
        if (top == kRoot) { depth = 0; }
        int sp = 0;
        lds->lvl[sp++] = (uint16_t)((uint32_t)top | ((uint32_t)depth_by_walk(top) << 10));
        for (int guard = 0; sp > 0; guard++) {
            if (guard >= NODES) { fault = 1; break; }            // more visits than nodes: the links are corrupt
            const uint32_t e = lds->lvl[--sp];
            const int v = (int)(e & 0x3FFu), b = (int)(e >> 10);
            if (STATS) { stats.updates += 1; }
            if (b > depth) { depth = b; }
            if (v < kRoot) { continue; }
            if (b >= 62) { fault = 1; continue; }               // reference asserts bits < 63
            const uint32_t w = lds->lnk[v];
            const uint32_t kids[2] = { l_hi(w), l_lo(w) };
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if (kids[j] == kNil) { continue; }
                if (sp < kStack) { lds->lvl[sp++] = (uint16_t)(kids[j] | ((uint32_t)(b + 1) << 10)); } else { fault = 1; }
            }
        }
    }

    __device__ __forceinline__ int order_pair(int i) {         // huffman.h:64-86, one lane
        const uint32_t p = up_of(i);
        if (p == kNil) { return i; }
        const uint32_t w = lds->lnk[p];
        const uint32_t l = l_lo(w), r = l_hi(w);
        if (l != kNil && r != kNil && freq((int)l) > freq((int)r)) {
            if (STATS) { stats.swaps += 1; }
            lds->lnk[p] = mk_lnk(l_up(w), r, l);
            relabel((int)p);
            return i == (int)l ? (int)r : (int)l;
        }
        return i;
    }

    __device__ __forceinline__ int climb(int i, int sp) {      // huffman.h:132-142, one lane
        for (int guard = 0; ; guard++) {
            if (guard >= kStack) { fault = 1; break; }
            const uint32_t p = up_of(i);
            if (p == kNil) { sum(i); break; }
            sum((int)p);
            i = order_pair(i);
            if (sp < kWave) { lds->pend[sp++] = (p << 16) | (uint32_t)i; }
            else { fault = 1; }
            i = (int)p;
        }
        return sp;
    }

    // huffman.h:130-147 with move_up (:98-128) inlined; LIFO order equals the reference's
    // recursion order because both inner calls are tail calls.  One lane, positions already given up.
    __device__ __forceinline__ void changed(int start) {
        int sp = climb(start, 0);
        for (int guard = 0; sp > 0; guard++) {
            if (guard >= 4 * kStack) { fault = 1; break; }
            const uint32_t e = lds->pend[--sp];
            const int p = (int)(e >> 16), c = (int)(e & 0xFFFFu);
            const uint32_t pw = lds->lnk[p];
            if (l_up(pw) == kNil || l_hi(pw) != (uint32_t)c) { continue; }     // :143
            const int g = (int)l_up(pw);
            const uint32_t gw = lds->lnk[g];
            const bool left = l_lo(gw) == (uint32_t)p;
            const int uncle = (int)(left ? l_hi(gw) : l_lo(gw));
            if (!(freq(c) > freq(uncle))) { continue; }                       // :108
            if (STATS) { stats.moves += 1; }
            lds->lnk[c] = (lds->lnk[c] & ~0x3FFu) | (uint32_t)g;
            lds->lnk[g] = left ? mk_lnk(l_up(gw), l_lo(gw), (uint32_t)c) : mk_lnk(l_up(gw), (uint32_t)c, l_hi(gw));
            lds->lnk[p] = mk_lnk(l_up(pw), l_lo(pw), (uint32_t)uncle);
            lds->lnk[uncle] = (lds->lnk[uncle] & ~0x3FFu) | (uint32_t)p;
            sum(p);
            sum(g);
            (void)order_pair(c);
            (void)order_pair(uncle);
            (void)order_pair(p);
            relabel(g);
            sp = climb(g, sp);                                                // :126
        }
    }

    // ---------------- a node's root path, one level per lane -------------------------------------
    // lane k receives level k (0 = the start node): one dependent read per level.
    __device__ __forceinline__ Chain chain_up(int s, int lane) const {
        XSEC_BEGIN
        int a = s, levels = 0;
        int mine = lane == 0 ? s : (int)kNil;
        for (int k = 1; k <= kMaxFastDepth; k++) {
            a = __builtin_amdgcn_readfirstlane((int)l_up(lds->lnk[a]));
            if (a == (int)kNil) { break; }
            mine = lane == k ? a : mine;
            levels = k;
        }
        if (levels == kMaxFastDepth && l_up(lds->lnk[a == (int)kNil ? s : a]) != kNil) { levels = depth_by_walk(s); }   // longer than the wave covers
        XSEC(8)
        return make_chain(mine, levels, lane);
    }
    __device__ __forceinline__ Chain make_chain(int mine, int levels, int lane) const {
        Chain c;
        c.mine = mine;
        c.par = lane_above(mine);            // lane k+1 holds the parent
        c.levels = levels;
        c.holds = lane <= levels;
        c.active = lane < levels;
        c.has_g = lane + 1 < levels;
        return c;
    }

    // ---------------- restructuring, whole wave -------------------------------------------------
    // sibling order under i's parent (huffman.h:64-86); dp = that parent's depth
    __device__ __forceinline__ void order_only(int i, uint32_t dp, int lane) {
        const uint32_t p = up_of(i);
        if (p == kNil) { return; }
        const uint32_t w = lds->lnk[p];
        const bool swap = __builtin_amdgcn_readfirstlane(
            (l_lo(w) != kNil && l_hi(w) != kNil && freq((int)l_lo(w)) > freq((int)l_hi(w))) ? 1 : 0) != 0;
        if (swap) {
            if (lane == 0) { lds->lnk[p] = mk_lnk(l_up(w), l_hi(w), l_lo(w)); }
            lds_fence();
            swap_fix_at((int)p, dp, lane);
        }
    }

    // What a climb leaves in every lane's registers: enough to pick and carry out the promotion the
    // reference would try next (huffman.h:143-146, :98-109) without reading the tree again.
    struct Climbed {
        uint64_t hits;      // lanes whose pending pair passes both tests of move_up
        int ch;             // the pair's child: my node, or its sibling after an exchange (huffman.h:81)
        uint32_t fch;       // ... its count
        uint32_t fpar;      // my parent's refreshed count
        int sib;            // my node's sibling before the exchange
        uint32_t fs;        // ... its count
        bool ends_hi;       // my node ends up as the hi child
    };

    // The climb of huffman_frequency_changed (huffman.h:132-142) along the chain `c` from its level
    // k0 upwards: lane k >= k0 owns the edge from level k to its parent; new sums by prefix sum,
    // sibling order per level, the parents' first / last leaves, one pending pair per level (bottom
    // first, appended at pend[sp]) -- and, in registers, which of those pairs would be promoted.
    __device__ __forceinline__ int climb_wave(const Chain& c, int k0, int sp, bool ends_moved, Climbed& out, int lane) {
        XSEC_BEGIN
        const int levels = c.levels;
        const int span = levels - k0;
        out.hits = 0;
        if (levels >= kMaxFastDepth || sp + span > kWave) { fault = 1; return 0; }
        const int start = __builtin_amdgcn_readlane(c.mine, k0);
        if (span <= 0) {                                              // the start node is the root
            if (lane == 0) { sum(start); }
            lds_fence();
            return sp;
        }
        const bool act = lane >= k0 && lane < levels;
        const uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)freq(start));
        const int x = act ? c.mine : kRoot;
        const int p = act ? c.par : kRoot;
        const uint32_t pw = lds->lnk[p];
        const bool is_hi = l_hi(pw) == (uint32_t)x;
        const uint32_t sib = is_hi ? l_lo(pw) : l_hi(pw);
        const bool has_sib = act & (sib != kNil);
        const uint32_t fs = has_sib ? freq((int)sib) : 0u;
        const uint32_t incl = wave_scan(act ? fs : 0u);
        const uint32_t fx = f0 + incl - fs;                          // my node's count, refreshed
        const bool swap = has_sib & (is_hi ? (fs > fx) : (fx > fs));  // lo count > hi count
        const uint64_t swaps = __ballot(swap);
        const bool ends_hi = act & (is_hi != swap);
        out.ch = swap ? (int)sib : x;
        out.fch = swap ? fs : fx;
        out.fpar = f0 + incl;
        out.sib = (int)sib;
        out.fs = fs;
        out.ends_hi = ends_hi;
        // move_up's tests for the pair of this level (huffman.h:143, :108): the pair's child is the hi
        // child (after an exchange the OTHER sibling is carried on, and it is hi exactly when my node
        // was), its parent is not the root, and it outweighs its uncle = the sibling one level up
        {
            const uint32_t f_unc = (uint32_t)lane_above((int)fs);
            const bool has_unc = lane_above(has_sib ? 1 : 0) == 1;
            out.hits = __ballot(act & is_hi & (lane + 1 < levels) & has_unc & (out.fch > f_unc));
        }
        // first / last leaf of every parent on the chain: the first leaf of level k+1 is the sibling's
        // where x ends up as the hi child, else what came up from below; the last leaf the other way
        // round.  They only move when a pair was exchanged here or the start node's own ends did
        // (ends_moved: the climb follows a promotion or an insert).
        uint32_t sF = 0, sL = 0, ends = 0;
        const bool redo_ends = aux != 0 && (swaps != 0 || ends_moved);
        if (redo_ends) {
            const uint32_t sw = has_sib ? lds->rng[sib] : 0u;
            sF = has_sib ? (is_leaf(sib) ? sib : r_first(sw)) : (uint32_t)x;
            sL = has_sib ? (is_leaf(sib) ? sib : r_last(sw)) : (uint32_t)x;
            const uint32_t F0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)first_leaf((uint32_t)start));
            const uint32_t L0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)last_leaf((uint32_t)start));
            const uint64_t below = (2ull << lane) - 1ull;             // lanes <= mine
            // (a node without a sibling -- the root's only child -- passes both ends up unchanged)
            const uint64_t mh = __ballot(ends_hi & has_sib) & below, ml = __ballot(act & !ends_hi & has_sib) & below;
            const uint32_t fF = (uint32_t)__builtin_amdgcn_ds_bpermute(mh != 0 ? (63 - __builtin_clzll(mh)) * 4 : 0, (int)sF);
            const uint32_t fL = (uint32_t)__builtin_amdgcn_ds_bpermute(ml != 0 ? (63 - __builtin_clzll(ml)) * 4 : 0, (int)sL);
            ends = (mh != 0 ? fF : F0) | ((ml != 0 ? fL : L0) << 10);
        }
        if (act) {
            lds->cnt[p] = aux != 0 ? ((f0 + incl) & kCountMask) : (f0 + incl);
            if (swap) { lds->lnk[p] = mk_lnk(l_up(pw), l_hi(pw), l_lo(pw)); }
            if (redo_ends) { lds->rng[p] = (lds->rng[p] & ~kEndsMask) | ends; }
            lds->pend[sp + lane - k0] = ((uint32_t)p << 16) | (uint32_t)out.ch;
        }
        lds_fence();
        XSEC(10)
        if (swaps != 0) {                                             // rare: one pass per exchanged pair
            // my node's own ends: what the lane below computed for its parent (= my node), or the start's
            const uint32_t below_ends = (uint32_t)lane_below((int)ends);
            const uint32_t xF = lane == k0 ? first_leaf((uint32_t)x) : (below_ends & 0x3FFu);
            const uint32_t xL = lane == k0 ? last_leaf((uint32_t)x) : ((below_ends >> 10) & 0x3FFu);
            const uint32_t unc = (uint32_t)lane_above((int)sib);      // my parent's sibling (kNil above the chain)
            uint64_t todo = swaps;
            while (todo != 0) {
                const int k = __builtin_ctzll(todo);
                todo &= todo - 1;
                // after the exchange the lo slot holds the old hi child
                const bool x_hi = __builtin_amdgcn_readlane(ends_hi ? 1 : 0, k) != 0;
                const uint32_t xk = (uint32_t)__builtin_amdgcn_readlane(x, k), sk = (uint32_t)__builtin_amdgcn_readlane((int)sib, k);
                const uint32_t xFk = (uint32_t)__builtin_amdgcn_readlane((int)xF, k), xLk = (uint32_t)__builtin_amdgcn_readlane((int)xL, k);
                const uint32_t sFk = (uint32_t)__builtin_amdgcn_readlane((int)sF, k), sLk = (uint32_t)__builtin_amdgcn_readlane((int)sL, k);
                const uint32_t pk = (uint32_t)__builtin_amdgcn_readlane(p, k);
                const uint32_t uk = k + 1 < levels ? (uint32_t)__builtin_amdgcn_readlane((int)unc, k) : kNil;
                if (x_hi) { swap_fix((int)pk, (uint32_t)(levels - k - 1), sk, xk, sFk, sLk, xFk, xLk, uk, lane); }
                else      { swap_fix((int)pk, (uint32_t)(levels - k - 1), xk, sk, xFk, xLk, sFk, sLk, uk, lane); }
            }
            XSEC(12)
        }
        return sp + span;
    }

    // move_up (huffman.h:110-125) for the pair (p, ch) under g with uncle `uncle`; fch / func / fpar =
    // the counts of ch, of the uncle and of p (refreshed) before the move; dg = depth of g.
    __device__ __forceinline__ void promote(int g, uint32_t dg, int p, int ch, int uncle, int left,
                                            uint32_t fch, uint32_t func, uint32_t fpar, int lane) {
        XSEC_BEGIN
        if (lane == 0) {                                              // :110-119, and p's new sum (:120; g's does not change)
            const uint32_t cw = lds->lnk[ch], gw = lds->lnk[g], pw = lds->lnk[p], uw = lds->lnk[uncle];
            lds->lnk[ch] = (cw & ~0x3FFu) | (uint32_t)g;
            lds->lnk[g] = left ? mk_lnk(l_up(gw), l_lo(gw), (uint32_t)ch) : mk_lnk(l_up(gw), (uint32_t)ch, l_hi(gw));
            lds->lnk[p] = mk_lnk(l_up(pw), l_lo(pw), (uint32_t)uncle);
            lds->lnk[uncle] = (uw & ~0x3FFu) | (uint32_t)p;
            lds->cnt[p] += func - fch;
        }
        lds_fence();
        uint32_t ga = 0, gb = 0, deepest = 0;
        promote_fix(g, dg, p, ch, uncle, left, ga, gb, deepest, lane);
        // :122-124, the three order checks: ch and p under g, x and the uncle under p, p under g again
        // (the same pair as the first: nothing left to do).  All counts are known.
        const uint32_t f_x = fpar - fch, f_p = f_x + func;
        const bool swap_g = left ? (f_p > fch) : (fch > f_p);
        const bool swap_p = f_x > func;
        if (swap_g) {
            if (lane == 0) { const uint32_t w = lds->lnk[g]; lds->lnk[g] = mk_lnk(l_up(w), l_hi(w), l_lo(w)); }
            lds_fence();
            swap_fix_at(g, dg, lane);
        }
        if (swap_p) {
            if (lane == 0) { const uint32_t w = lds->lnk[p]; lds->lnk[p] = mk_lnk(l_up(w), l_hi(w), l_lo(w)); }
            lds_fence();
            swap_fix_at(p, dg + 1u, lane);
        }
        if (aux != 0) {
            if (swap_p) {                                             // p's ends moved: g's follow (the climb redoes everything above g)
                if (lane == 0) {
                    const uint32_t gw = lds->lnk[g];
                    set_ends(g, first_leaf(l_lo(gw)), last_leaf(l_hi(gw)));
                }
                lds_fence();
            }
            mark_range(g, ga, gb, deepest, lane);                     // :125 huffman_update_paths(gix)
        } else {
            if (lane == 0) { relabel(g); }
            uniform_regs();
            lds_fence();
        }
        XSEC(11)
    }

    // huffman_frequency_changed + move_up along the chain `c` of the node whose count changed.  The
    // chain stays valid throughout: a promotion under g changes nothing above g, and the climb that
    // follows it starts at g.  Pending pairs: pend[0, base) are older ones (below the last promotion),
    // pend[base, sp) the ones of the latest climb, already judged in registers.
    __device__ __forceinline__ void changed_all(const Chain& c_in, int lane) {
        Chain c = c_in;
        Climbed r;
        int valid_from = 0;                                           // lanes of `c` from here up are a true root path
        int k0 = 0, base = 0;
        int sp = climb_wave(c, 0, 0, in_insert != 0, r, lane);
        // (a promotion moves a node up, so a symbol's update makes fewer of them than the tree is
        // deep; the bound only keeps a wave from spinning on a corrupted tree: it faults instead)
        for (int guard = 0; sp > 0; guard++) {
            if (guard >= 4 * kWave) { fault = 1; break; }
            int p, ch, g, uncle, left, kg;
            uint32_t fch, func, fpar;
            if (r.hits != 0) {
                // the topmost pair of the latest climb that passes (the reference pops from the top; the
                // ones that fail its tests change nothing): everything is in registers
                const int j = 63 - __builtin_clzll(r.hits);
                p = __builtin_amdgcn_readlane(c.par, j);
                ch = __builtin_amdgcn_readlane(r.ch, j);
                fch = (uint32_t)__builtin_amdgcn_readlane((int)r.fch, j);
                fpar = (uint32_t)__builtin_amdgcn_readlane((int)r.fpar, j);
                g = __builtin_amdgcn_readlane(c.par, j + 1);
                uncle = __builtin_amdgcn_readlane(r.sib, j + 1);
                func = (uint32_t)__builtin_amdgcn_readlane((int)r.fs, j + 1);
                left = __builtin_amdgcn_readlane(r.ends_hi ? 1 : 0, j + 1) != 0 ? 0 : 1;
                kg = j + 2;
                sp = base + (j - k0);
            } else {
                // none of them: they are popped; then the older pairs, from the tree (rare)
                sp = base;
                if (sp == 0) { break; }
                bool hit = false;
                int hp = 0, hc = 0, hg = 0, hu = 0, hl = 0;
                uint32_t h_fch = 0, h_func = 0, h_fpar = 0;
                if (lane < sp) {
                    const uint32_t e = lds->pend[lane];
                    hp = (int)(e >> 16); hc = (int)(e & 0xFFFFu);
                    const uint32_t pw = lds->lnk[hp];
                    if (l_up(pw) != kNil && l_hi(pw) == (uint32_t)hc) {
                        hg = (int)l_up(pw);
                        const uint32_t gw = lds->lnk[hg];
                        hl = l_lo(gw) == (uint32_t)hp ? 1 : 0;
                        hu = (int)(hl ? l_hi(gw) : l_lo(gw));
                        if (hu != (int)kNil) {
                            h_fch = freq(hc); h_func = freq(hu); h_fpar = freq(hp);
                            hit = h_fch > h_func;
                        }
                    }
                }
                const uint64_t hits = __ballot(hit);
                if (hits == 0) { break; }
                const int j = 63 - __builtin_clzll(hits);
                sp = j;
                p = __builtin_amdgcn_readlane(hp, j); ch = __builtin_amdgcn_readlane(hc, j);
                g = __builtin_amdgcn_readlane(hg, j); uncle = __builtin_amdgcn_readlane(hu, j);
                left = __builtin_amdgcn_readlane(hl, j);
                fch = (uint32_t)__builtin_amdgcn_readlane((int)h_fch, j);
                func = (uint32_t)__builtin_amdgcn_readlane((int)h_func, j);
                fpar = (uint32_t)__builtin_amdgcn_readlane((int)h_fpar, j);
                // g's level on the chain gives its depth and where the next climb starts.  A promotion
                // leaves the chain ABOVE its g intact; this older pair may sit off the intact part: take
                // g's root path afresh then.
                const uint64_t at_g = __ballot(c.holds && c.mine == g && lane >= valid_from);
                kg = 0;
                if (at_g != 0) { kg = __builtin_ctzll(at_g); }
                else { c = chain_up(g, lane); }
            }
            valid_from = kg;
            base = sp;
            promote(g, (uint32_t)(c.levels - kg), p, ch, uncle, left, fch, func, fpar, lane);
            k0 = kg;
            sp = climb_wave(c, kg, sp, true, r, lane);                // :126
        }
    }

    // huffman_inc_frequency for an ATTACHED leaf s whose chain is `c`.
    // Returns the ballot of "my node is the hi child" (the stream-order code).  Lane k owns level
    // k of the chain and evaluates exactly the two tests the reference would make at that level
    // with the incremented counts (swap huffman.h:75, promote :108); no flag -> every count on the
    // chain grows by one, which the lanes do in one step; any flag -> the reference sequence.
    __device__ __forceinline__ uint64_t bump_wave(int s, const Chain& c, int lane) {
        XSEC_BEGIN
        const int i_mine = c.holds ? c.mine : kRoot;
        const int i_par = c.active ? c.par : kRoot;
        const uint32_t cw = lds->cnt[i_mine];
        const uint32_t fc = aux != 0 ? c_f(cw) : cw;
        const uint32_t pw = lds->lnk[i_par];
        const bool is_hi = c.active & (l_hi(pw) == (uint32_t)c.mine);
        const uint64_t code_bits = __ballot(is_hi);
        if (complete != 0 || depth >= kFreezeDepth) { complete = 1; return code_bits; }   // huffman.h:228-234
        // my sibling's count; my uncle is my parent's sibling = the lane above
        const uint32_t sib = is_hi ? l_lo(pw) : l_hi(pw);
        const bool has_sib = c.active & (sib != kNil);
        const uint32_t fs = freq(has_sib ? (int)sib : kRoot);
        const uint32_t fu = (uint32_t)lane_above((int)fs);
        const bool has_unc = is_hi & c.has_g & (lane_above(has_sib ? 1 : 0) == 1);
        const uint32_t fc1 = fc + 1;
        const uint32_t big = is_hi ? fs : fc1;                        // swap iff lo count > hi count
        const uint32_t small = is_hi ? fc1 : fs;
        const bool flag = (has_sib & (big > small)) | (has_unc & (fc1 > fu)) |
                          (c.levels >= kMaxFastDepth);
        if (__ballot(flag) == 0) {
            if (c.holds) { lds->cnt[c.mine] = cw + 1; }
            lds_fence();
            XSEC(9)
        } else {
            XSEC(9)
            changed_wave(s, c, lane);
        }
        return code_bits;
    }

    // ---------------- entry points of the slow paths ----------------------------------------------
#ifdef SQZ_STATS
    uint64_t st_cyc[3] = {0, 0, 0};
    uint32_t st_cnt[3] = {0, 0, 0};
#define SQZ_ST_BEGIN const uint64_t st_t0 = __builtin_readcyclecounter();
#define SQZ_ST_END(k) st_cyc[k] += __builtin_readcyclecounter() - st_t0; st_cnt[k]++;
#else
#define SQZ_ST_BEGIN
#define SQZ_ST_END(k)
#endif
    __device__ __forceinline__ bool insert_wave(int i, int lane) {
        SQZ_ST_BEGIN
        const TreeStats keep = stats;
        const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)slow_insert<Tree>(lds, code, pack_regs(), i, lane));
        unpack_regs(r);
        stats = keep;
        load_stats_delta();
        lut_ok = 0;
        SQZ_ST_END(0)
        return (r >> 31) != 0;
    }

    __device__ __forceinline__ void changed_wave(int s, const Chain& c, int lane) {
        SQZ_ST_BEGIN
        const TreeStats keep = stats;
        unpack_regs((uint32_t)__builtin_amdgcn_readfirstlane(
            (int)slow_changed<Tree>(lds, code, pack_regs(), s, c.mine, c.levels, lane)));
        stats = keep;
        load_stats_delta();
        lut_ok = 0;
        SQZ_ST_END(1)
    }

    // the slow calls leave what they counted in pend[61..63] (nothing is pending between calls)
    __device__ __forceinline__ void store_stats_delta(int lane) {
        if (!STATS) { return; }
        if (lane == 0) { lds->pend[61] = stats.updates; lds->pend[62] = stats.swaps; lds->pend[63] = stats.moves; }
        lds_fence();
    }
    __device__ __forceinline__ void load_stats_delta() {
        if (!STATS) { return; }
        stats.updates += (uint32_t)__builtin_amdgcn_readfirstlane((int)lds->pend[61]);
        stats.swaps += (uint32_t)__builtin_amdgcn_readfirstlane((int)lds->pend[62]);
        stats.moves += (uint32_t)__builtin_amdgcn_readfirstlane((int)lds->pend[63]);
    }

    __device__ __forceinline__ void build_lut(int lane) {
        SQZ_ST_BEGIN
        slow_build_lut<Tree>(lds->lnk, lut, lane);
        lds_fence();                                           // entries written by other lanes are read next
        lut_ok = 1;
        SQZ_ST_END(2)
    }

    // ---------------- huffman_insert (huffman.h:149-216), whole wave -------------------------------
    // lane 0 hangs the leaf into the links; positions, codes and partners follow; then the climb and
    // the promotions like any other update, and the closing walk of :213.
    __device__ __forceinline__ bool insert_all(int i, int lane) {
        uint32_t hand = 0;       // start | at << 10 | ok << 20 | split << 21 ; q in the high bits
        if (lane == 0) {
            uint32_t ok = 1, split = 0, q = 0;
            int at = kRoot, leaf = i;
            lds->cnt[leaf] = 1u;                                       // freq 1, depth set below
            uint32_t aw = 0;
            for (int guard = 0; at >= kRoot; guard++) {                // :156-170
                aw = lds->lnk[at];
                if (l_hi(aw) == kNil || l_lo(aw) == kNil) { break; }
                if (guard >= kStack) { fault = 1; break; }             // a corrupted tree must not spin the wave
                at = (int)l_lo(aw);
            }
            if (at >= kRoot) {                                         // :171-173: hangs under the root (the first two leaves)
                const bool to_hi = l_hi(aw) == kNil;
                lds->lnk[at] = to_hi ? mk_lnk(l_up(aw), l_lo(aw), (uint32_t)leaf) : mk_lnk(l_up(aw), (uint32_t)leaf, l_hi(aw));
                lds->lnk[leaf] = mk_lnk((uint32_t)at, kNil, kNil);
                set_freq(at, freq(at) + 1u);
                if (aux != 0) {
                    // the tree is tiny (at IS the root: a split always makes two children): its positions by hand
                    lds->cnt[leaf] = 1u | (1u << kDepthShift);
                    const uint32_t w2 = lds->lnk[at];
                    const uint32_t lo = l_lo(w2), hi = l_hi(w2);
                    uint32_t pos = (uint32_t)POS0;
                    if (lo != kNil) {
                        lds->rng[lo] = pos | (hi << 20);               // a lo child is tested against its sibling
                        if (CODES) { code[code_slot((int)lo)] = 0u; }
                        pos++;
                    }
                    lds->rng[hi] = pos | (kNil << 20);                 // a hi child of the root has no test
                    if (CODES) { code[code_slot((int)hi)] = 1u; }
                    lds->rng[at] = (lo != kNil ? lo : hi) | (hi << 10) | (kNil << 20);
                }
            } else if (next >= kIdEnd) {                               // :180-182
                ok = 0;
                complete = 1;
            } else {                                                   // :184-209: split leaf `at`
                split = 1;
                const int fresh = next++;
                const uint32_t above = l_up(lds->lnk[at]);
                const uint32_t acw = lds->cnt[at];
                q = r_pos(lds->rng[at]);
                lds->lnk[fresh] = mk_lnk(above, (uint32_t)at, (uint32_t)leaf);
                lds->cnt[fresh] = aux != 0 ? c_f(acw) : acw;           // at's count
                const uint32_t bw = lds->lnk[above];
                lds->lnk[above] = l_lo(bw) == (uint32_t)at ? mk_lnk(l_up(bw), (uint32_t)fresh, l_hi(bw))
                                                           : mk_lnk(l_up(bw), l_lo(bw), (uint32_t)fresh);
                lds->lnk[at] = mk_lnk((uint32_t)fresh, kNil, kNil);
                lds->lnk[leaf] = mk_lnk((uint32_t)fresh, kNil, kNil);
                if (aux != 0) {                                        // both children one level below at's old place
                    lds->cnt[at] = acw + (1u << kDepthShift);
                    lds->cnt[leaf] = 1u | ((c_d(acw) + 1u) << kDepthShift);
                    lds->rng[fresh] = (uint32_t)at | ((uint32_t)leaf << 10) | (kNil << 20);
                }
                sum(fresh);
                at = fresh;
            }
            hand = (uint32_t)leaf | ((uint32_t)at << 10) | (ok << 20) | (split << 21) | (q << 22);
        }
        uniform_regs();
        hand = (uint32_t)__builtin_amdgcn_readfirstlane((int)hand);
        lds_fence();
        int start = (int)(hand & 0x3FFu);
        const int at = (int)((hand >> 10) & 0x3FFu);
        const bool ok = ((hand >> 20) & 1u) != 0, split = ((hand >> 21) & 1u) != 0;
        const uint32_t q = hand >> 22;
        if (split && aux != 0) {
            // every leaf behind position q moves one to the right; the new leaf takes q + 1
            const int fresh = at;
            const uint32_t leaf_at = l_lo(lds->lnk[fresh]);
            uint32_t ws[kLeafRows];
#pragma unroll
            for (int r = 0; r < kLeafRows; r++) {
                const int v = BASE + r * kWave + lane;
                ws[r] = lds->rng[v < kRoot ? v : kRoot];
            }
#pragma unroll
            for (int r = 0; r < kLeafRows; r++) {
                const int v = BASE + r * kWave + lane;
                if (BASE + r * kWave >= kRoot) { continue; }
                const uint32_t w = ws[r];
                if (v < kRoot && v != i && r_pos(w) != kNoPos && r_pos(w) > q) { lds->rng[v] = w + 1u; }
            }
            if (lane == 0) {
                lds->rng[i] = (q + 1u) | (kNil << 20);
                if (CODES) {
                    const uint32_t oc = code[code_slot((int)leaf_at)];
                    code[code_slot((int)leaf_at)] = oc << 1;
                    code[code_slot(i)] = (oc << 1) | 1u;
                }
            }
            lds_fence();
            fix_partners((int)l_up(lds->lnk[fresh]), 2, lane);         // fresh, its sibling, and their children
        }
        uint32_t droot = 0;
        if (!split && ok) {                                            // :173 order under the root
            const uint32_t w = lds->lnk[at];
            if (__builtin_amdgcn_readfirstlane(
                    (l_lo(w) != kNil && l_hi(w) != kNil && freq((int)l_lo(w)) > freq((int)l_hi(w))) ? 1 : 0) != 0) {
                if (lane == 0) { lds->lnk[at] = mk_lnk(l_up(w), l_hi(w), l_lo(w)); }
                lds_fence();
                start = start == (int)l_lo(w) ? (int)l_hi(w) : (int)l_lo(w);
                swap_fix_at(at, droot, lane);
            }
        }
        if (depth + 4 >= kMaxFastDepth) {                              // chains longer than the wave: the reference sequence on one lane
            give_up_aux(lane);
            if (lane == 0) { changed(start); relabel(at); }
            uniform_regs();
            lds_fence();
        } else {
            const Chain c = chain_up(start, lane);
            in_insert = 1;
            changed_all(c, lane);                                      // :212
            if (aux != 0) { mark_subtree(at, lane); }                  // :213
            else { if (lane == 0) { relabel(at); } uniform_regs(); lds_fence(); }
            in_insert = 0;
        }
        return ok;
    }
};

template <class T>
__device__ SQZ_INSERT_INLINING uint32_t slow_insert(TreeLds* lds, uint32_t* code, uint32_t regs, int sym, int lane) {
    T t;
    t.lds = lds; t.code = code; t.lut = nullptr; t.lut_ok = 0;
    t.stats.updates = t.stats.swaps = t.stats.moves = 0;
    t.unpack_regs(regs);
    const bool ok = t.insert_all(sym, lane);
    t.store_stats_delta(lane);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(t.pack_regs() | (ok ? 0x80000000u : 0u)));
}

template <class T>
__device__ SQZ_CHANGED_INLINING uint32_t slow_changed(TreeLds* lds, uint32_t* code, uint32_t regs, int sym, int mine, int levels, int lane) {
    T t;
    t.lds = lds; t.code = code; t.lut = nullptr; t.lut_ok = 0;
    t.stats.updates = t.stats.swaps = t.stats.moves = 0;
    t.unpack_regs(regs);
    if (lane == 0) { t.lds->cnt[sym] += 1u; }
    lds_fence();
    if (t.depth + 4 >= kMaxFastDepth) {                 // chains longer than the wave: one lane, explicit stacks
        t.give_up_aux(lane);
        if (lane == 0) { t.changed(sym); }
        t.uniform_regs();
        lds_fence();
    } else {
        t.changed_all(t.make_chain(mine, levels, lane), lane);      // the chain bump_wave already holds
    }
    t.store_stats_delta(lane);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)t.pack_regs());
}

// decoder: table over the next kLutBits stream bits -> (node reached, bits used).
// Entry = node | used << 10; a missing child gives node = kNil.
// One step of the table build: the entry for a prefix one bit longer than `parent`'s.
// A finished entry (leaf, or a missing child) is inherited; an internal node hands
// down the child the new bit selects, with one more bit used.
template <class T>
__device__ __forceinline__ uint32_t lut_descend(const uint32_t* lnk, uint32_t parent, int bit, int level) {
    const uint32_t node = parent & 0x3FFu;
    const bool inside = node != kNil && node >= (uint32_t)T::kRoot;
    const uint32_t w = lnk[inside ? (int)node : (int)T::kRoot];
    const uint32_t child = bit ? l_hi(w) : l_lo(w);
    return inside ? (child | ((uint32_t)level << 10)) : parent;
}

template <class T>
__device__ SQZ_LUT_INLINING void slow_build_lut(const uint32_t* lnk, uint16_t* lut, int lane) {
    // level by level from the root: the table for (L+1)-bit prefixes follows from the one
    // for L-bit prefixes with one link read per entry.  Lane j holds entry j while a level fits the wave.
    static_assert(T::kLutBits == 6 || T::kLutBits == 8, "table widths the decoder uses");
    const int half = (lane >> 1) * 4;                         // byte address of lane j>>1
    const int bit = lane & 1;
    uint32_t e = (uint32_t)T::kRoot;                          // the 0-bit prefix: root, no bits used
#pragma unroll
    for (int level = 1; level <= 6; level++) {
        const uint32_t parent = (uint32_t)__builtin_amdgcn_ds_bpermute(half, (int)e);
        e = lut_descend<T>(lnk, parent, bit, level);
    }
    if (T::kLutBits == 6) {
        lut[lane] = (uint16_t)e;
        return;
    }
    // 128 entries: j and j + 64 descend from entries j>>1 and 32 + (j>>1) of level 6
    const uint32_t p0 = (uint32_t)__builtin_amdgcn_ds_bpermute(half, (int)e);
    const uint32_t p1 = (uint32_t)__builtin_amdgcn_ds_bpermute(half + 128, (int)e);
    const uint32_t r0 = lut_descend<T>(lnk, p0, bit, 7);
    const uint32_t r1 = lut_descend<T>(lnk, p1, bit, 7);
    // 256 entries: j + 64k descends from entry 32k + (j>>1) of level 7 (r0: 0..63, r1: 64..127)
    const uint32_t q0 = (uint32_t)__builtin_amdgcn_ds_bpermute(half, (int)r0);
    const uint32_t q1 = (uint32_t)__builtin_amdgcn_ds_bpermute(half + 128, (int)r0);
    const uint32_t q2 = (uint32_t)__builtin_amdgcn_ds_bpermute(half, (int)r1);
    const uint32_t q3 = (uint32_t)__builtin_amdgcn_ds_bpermute(half + 128, (int)r1);
    lut[lane]       = (uint16_t)lut_descend<T>(lnk, q0, bit, 8);
    lut[lane + 64]  = (uint16_t)lut_descend<T>(lnk, q1, bit, 8);
    lut[lane + 128] = (uint16_t)lut_descend<T>(lnk, q2, bit, 8);
    lut[lane + 192] = (uint16_t)lut_descend<T>(lnk, q3, bit, 8);
}

template <bool CODES, bool STATS> using LitTreeT = Tree<0, kLitLeaves, kLitNodes, 512, 0, 8, CODES, STATS>;
template <bool CODES, bool STATS> using PosTreeT = Tree<kPosBase, kPosLeaves, kPosNodes, 32, kPosPos0, 6, CODES, STATS>;

// ---------------------------------------------------------------------------------------------
// Up to 64 tokens per step: lane = token (its lit-tree leaf `a`, then, for a back reference, its
// distance-tree leaf `b`; absolute node ids, -1 = none).  See the head of this file.
//
// Returns how many leading tokens were applied (0..m).  code_a/depth_a/code_b/depth_b are the
// static tree's and valid for all m lanes (kWantCode: the encoder emits them, the decoder has no
// use for them).  Only callable while both trees keep their positions (aux) and take updates.
template <bool kWantCode, class LIT, class POS>
__device__ __forceinline__ int bump_batch(TreeLds* lds, const uint32_t* code, const LIT& lit, const POS& pos,
                                          int lane, int m, int a, int b,
                                          uint32_t& code_a, int& depth_a, uint32_t& code_b, int& depth_b) {
    // LATENCY.  One wave alone on its SIMD pays ~100 cycles for every LDS round trip it WAITS for, and a
    // step used to wait for two dozen of them one after the other (every node's probe was its own chain of
    // three dependent reads, each behind a branch: 3.5 k cycles).  The reads of a step depend on each other
    // in only three levels -- (1) a node's range word and count, (2) its partner's count and the range words
    // of its first and last leaf, (3) the prefix sums at its two ends -- so all probes go through the levels
    // TOGETHER, branch-free on clamped indices (a lane without a node reads the root's words and is masked
    // out at the end): three round trips with 16-20 reads in flight each, and the histogram's own chain
    // (zero, add, read, scan, write) runs while the level-2 reads are on their way.
    SEC_BEGIN
    const bool take = lane < m;
    const bool has_a = take && a >= 0, has_b = take && b >= 0;
    const int ia = has_a ? a : LIT::kRoot - 1, ib = has_b ? b : POS::kRoot - 1;      // (idle lanes look at a pad leaf)
    constexpr int kLitRows = (kLitNodes - kLitLeaves + kWave - 1) / kWave;      // 5
    constexpr int kRows = kLitRows + 1;                                         // + the distance tree's row
    // ---- level 1: range word and count of the tokens' leaves and of this lane's internal nodes ----------
    int vi[kRows];
    bool on[kRows];
#pragma unroll
    for (int row = 0; row < kLitRows; row++) {
        const int v = LIT::kRoot + row * kWave + lane;
        on[row] = v < lit.next && v != LIT::kRoot;
        vi[row] = on[row] ? v : LIT::kRoot;
    }
    {
        const int v = POS::kRoot + lane;
        on[kLitRows] = v < pos.next && v != POS::kRoot;
        vi[kLitRows] = on[kLitRows] ? v : POS::kRoot;
    }
    const uint32_t ra = lds->rng[ia], rb = lds->rng[ib];
    const uint32_t wa = lds->cnt[ia], wb = lds->cnt[ib];
    uint32_t rw[kRows], cw[kRows];
#pragma unroll
    for (int k = 0; k < kRows; k++) { rw[k] = lds->rng[vi[k]]; cw[k] = lds->cnt[vi[k]]; }
    if (kWantCode) {
        code_a = code[has_a ? ia - LIT::kBase + 0 : 0];
        code_b = code[has_b ? ib - POS::kBase + kPosPos0 : 0];
    }
    const uint32_t qa = r_pos(ra), qb = r_pos(rb);
    depth_a = (int)c_d(wa);
    depth_b = (int)c_d(wb);
    // ---- level 2: partners' counts, the ends' positions (on their way while the histogram is made) -------
    const uint32_t pa_a = r_pa(ra), pa_b = r_pa(rb);
    const uint32_t cpa = lds->cnt[pa_a != kNil ? pa_a : (uint32_t)ia], cpb = lds->cnt[pa_b != kNil ? pa_b : (uint32_t)ib];
    uint32_t cp[kRows], rf[kRows], rl[kRows];
#pragma unroll
    for (int k = 0; k < kRows; k++) {
        const uint32_t pk = r_pa(rw[k]);
        cp[k] = lds->cnt[pk != kNil ? pk : (uint32_t)vi[k]];
        rf[k] = lds->rng[r_first(rw[k])];
        rl[k] = lds->rng[r_last(rw[k])];
    }
    SEC(0)
    // ---- histogram over positions (a byte each), then its prefix sums in place ----------------
    auto histogram = [&](int upto) {
        lds->P64[lane] = 0ull;
        lds_fence();
        if (lane < upto && has_a) { atomicAdd(&lds->P32[qa >> 2], 1u << (8u * (qa & 3u))); }
        if (lane < upto && has_b) { atomicAdd(&lds->P32[qb >> 2], 1u << (8u * (qb & 3u))); }
        lds_fence();
        // lane l owns positions 8l .. 8l+7: byte-wise inclusive sums by one multiply (no byte
        // overflows: a batch holds at most 128 symbols), then the lanes' totals
        const uint64_t h = lds->P64[lane];
        const uint64_t incl = h * 0x0101010101010101ull;
        const uint32_t total = (uint32_t)(incl >> 56);
        const uint32_t base = wave_scan(total) - total;
        lds->P64[lane] = (incl << 8) + (uint64_t)base * 0x0101010101010101ull;     // exclusive: P[i] = symbols at positions < i
        lds_fence();
    };
    histogram(m);
    SEC(1)
    // ---- level 3: how many of the batch's chains pass each node (its position run in the prefix sums) ---
    uint32_t st[kRows], en[kRows], pst[kRows], pen[kRows];
#pragma unroll
    for (int k = 0; k < kRows; k++) { st[k] = r_pos(rf[k]); en[k] = r_pos(rl[k]) + 1u; }
    const uint32_t p_a0 = lds->P8[qa], p_a1 = lds->P8[qa + 1u], p_b0 = lds->P8[qb], p_b1 = lds->P8[qb + 1u];
#pragma unroll
    for (int k = 0; k < kRows; k++) { pst[k] = lds->P8[st[k]]; pen[k] = lds->P8[en[k]]; }
    // ---- every touched node's test: f + n <= count of its partner (a lo child's sibling, a hi child's
    //      uncle); the tokens' own leaves first (duplicates test the same node twice: harmless) -------------
    // per tested node, kept in a register: its position run and how many chains may still pass it
    // (st | en << 9 | min(partner's count - its count, 255) << 18)
    auto pack = [](uint32_t s0, uint32_t e0, uint32_t f, uint32_t fb) {
        const uint32_t allowed = fb > f ? fb - f : 0u;
        return s0 | (e0 << 9) | ((allowed < 255u ? allowed : 255u) << 18);
    };
    uint32_t nl[kLitRows], sel[kLitRows], np = 0, sep = 0;     // per row: chains through my node, its packed test
    uint32_t bad = 0;                                   // bit 0 / 1: my leaves; bit 2 + r: my node of lit row r; bit 7: pos row
    uint32_t sa, sb;
    {
        const uint32_t n = has_a ? p_a1 - p_a0 : 0u, f = c_f(wa), fb = c_f(cpa);
        sa = pack(qa, qa + 1u, f, fb);
        bad |= (n != 0 && pa_a != kNil && f + n > fb) ? 1u : 0u;
    }
    {
        const uint32_t n = has_b ? p_b1 - p_b0 : 0u, f = c_f(wb), fb = c_f(cpb);
        sb = pack(qb, qb + 1u, f, fb);
        bad |= (n != 0 && pa_b != kNil && f + n > fb) ? 2u : 0u;
    }
#pragma unroll
    for (int k = 0; k < kRows; k++) {
        const uint32_t n = on[k] ? pen[k] - pst[k] : 0u, f = c_f(cw[k]), fb = c_f(cp[k]);
        const uint32_t packed = on[k] ? pack(st[k], en[k], f, fb) : 0u;
        const bool fails = n != 0 && r_pa(rw[k]) != kNil && f + n > fb;
        if (k < kLitRows) { nl[k] = n; sel[k] = packed; bad |= fails ? (4u << k) : 0u; }
        else { np = n; sep = packed; bad |= fails ? 128u : 0u; }
    }
    int ok = m;
    SEC(2)
    if (__ballot(bad != 0) != 0) {
        // some node would overtake its partner.  For each such node, the token that may not pass is
        // the (partner's count - its count + 1)-th one whose position lies below the node; the
        // earliest of them over all failing nodes ends the batch.  The failing lane's registers
        // hold all that is needed.
#pragma unroll
        for (int cat = 0; cat < 8; cat++) {
            uint64_t vm = __ballot((bad >> cat) & 1u);
            const uint32_t mine = cat == 0 ? sa : cat == 1 ? sb : cat == 7 ? sep : sel[cat >= 2 && cat < 7 ? cat - 2 : 0];
            while (vm != 0) {
                const int k = __builtin_ctzll(vm);
                vm &= vm - 1;
                const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)mine, k);
                const uint32_t s0 = t & 0x1FFu, e0 = (t >> 9) & 0x1FFu, allowed = t >> 18;
                const bool through = (has_a && qa >= s0 && qa < e0) || (has_b && qb >= s0 && qb < e0);
                const uint64_t tm = __ballot(through);
                const uint64_t first_bad = __ballot(through && lanes_under(tm) == allowed);
                if (first_bad != 0) { const int j = __builtin_ctzll(first_bad); ok = j < ok ? j : ok; }
            }
        }
        if (ok == 0) { return 0; }                              // nothing may be applied: the first token takes the exact path
        // count again for the prefix only (the nodes' position runs are still in registers)
        histogram(ok);
#pragma unroll
        for (int row = 0; row < kLitRows; row++) {
            nl[row] = (uint32_t)lds->P8[(sel[row] >> 9) & 0x1FFu] - (uint32_t)lds->P8[sel[row] & 0x1FFu];
        }
        np = (uint32_t)lds->P8[(sep >> 9) & 0x1FFu] - (uint32_t)lds->P8[sep & 0x1FFu];
    }
    SEC(3)
    // ---- apply the prefix: leaves one add per token (lanes holding the same symbol meet at its
    //      leaf); internal nodes one add per node, by the lane that tested it --------------------------
    if (lane < ok && has_a) { atomicAdd(&lds->cnt[ia], 1u); }
    if (lane < ok && has_b) { atomicAdd(&lds->cnt[ib], 1u); }
#pragma unroll
    for (int row = 0; row < kLitRows; row++) {
        const int v = LIT::kRoot + row * kWave + lane;
        if (nl[row] != 0) { atomicAdd(&lds->cnt[v], nl[row]); }       // (one LDS instruction instead of read + write)
    }
    if (np != 0) { atomicAdd(&lds->cnt[POS::kRoot + lane], np); }
    lds_fence();
    SEC(4)
    return ok;
}

} // namespace sqzk
