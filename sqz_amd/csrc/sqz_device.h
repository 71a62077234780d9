// sqz_amd/csrc/sqz_device.h -- device-side building blocks shared by the
// entropy-stage kernels (huffman_emit.hip, decode.hip), gfx950 only.
//
// Reference semantics restated here (file:line relative to
// /root/reference/attic/map_experiment):
//   huffman.h:13-34   node / tree          -> Tree<> (LDS resident, compact ids)
//   huffman.h:41-62   huffman_update_paths -> Tree::relabel (one lane), relabel_moved / relabel_wave
//   huffman.h:64-86   huffman_swap_siblings-> Tree::order_pair, order_only
//   huffman.h:90-96   huffman_update_freq  -> Tree::sum
//   huffman.h:98-147  move_up / frequency_changed -> Tree::changed (one lane), changed_all (whole wave)
//   huffman.h:149-216 huffman_insert       -> Tree::insert_splice + changed_all
//   huffman.h:218-235 huffman_inc_frequency-> Tree::bump_wave (one symbol), bump_lanes (up to 64 tokens)
//   squeeze.h:29-79,151-172 DEFLATE tables -> len_code()/pos_code() arithmetic
//   bitstream.h:28-63,112-114 bit packer   -> BitQueue (huffman_emit.hip)
//   bitstream.h:65-103 bit reader          -> BitSource
//
// Layout decisions:
//  * one wavefront owns one stream; both trees live in LDS.  Most updates change no link and
//    are applied up to 64 tokens at a time with one lane per token (bump_lanes); a token whose
//    update restructures a tree takes the exact path, where lane k holds level k of the
//    symbol's leaf->root chain and the restructuring itself (climb, promotions, relabel) is
//    spread over the wave.  Only the splice of a new leaf and trees deeper than 56 levels run
//    the reference sequence on a single lane.
//  * node ids are compacted: leaves keep their symbol value, internal nodes
//    are numbered upwards from LEAVES (root == LEAVES).  The reference numbers
//    internals downwards from 2n-2; no emitted bit depends on the numbering.
//  * no code is stored: a symbol's code is the "am I the hi child" bits along its chain, read
//    in STREAM order (first branch = most significant bit), so the bit packer appends codes
//    without reversing them.  Same bits on the wire as the reference's LSB-first `path`.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sqzk {

constexpr int kWave      = 64;
constexpr int kLenMin    = 3;    // squeeze.h:13
constexpr int kLenMax    = 257;  // squeeze.h:15
constexpr int kSymLen0   = 257;  // squeeze.h:10
constexpr int kLitNyt    = 285;  // squeeze.h:23
constexpr int kPosNyt    = 30;   // squeeze.h:24
constexpr int kMaxWindow = 32768;

// errno values used on the device (Linux numbering, same as the host's)
constexpr int kEINVAL = 22;
constexpr int kE2BIG  = 7;

constexpr uint32_t kTokMatch = 0x80000000u;
constexpr uint32_t kRefused = 0xFFFFFFFFu;    // tok_count of a block stage 1 would not touch (offsets beyond the scratch)

// ---------------------------------------------------------------------------
// DEFLATE length / distance codes by arithmetic (squeeze.h:29-79 tables and
// the len_index/pos_index lookups built at :151-172).
struct Code { int code; int xbits; int extra; };

__device__ __forceinline__ Code len_code(int len) {  // 3..257
    Code c;
    const int y = len - 3;
    if (y < 8) { c.code = y; c.xbits = 0; c.extra = 0; return c; }
    const int msb = 31 - __clz(y);                       // >= 3
    c.code  = 4 * msb - 4 + ((y >> (msb - 2)) & 3);
    c.xbits = msb - 2;
    c.extra = y & ((1 << c.xbits) - 1);
    return c;
}

__device__ __forceinline__ Code pos_code(int pos) {  // 1..32767
    Code c;
    if (pos < 5) { c.code = pos - 1; c.xbits = 0; c.extra = 0; return c; }
    const int x = pos - 1;
    const int msb = 31 - __clz(x);                       // >= 2
    c.code  = 2 * msb + ((x >> (msb - 1)) & 1);
    c.xbits = msb - 1;
    c.extra = x & ((1 << c.xbits) - 1);
    return c;
}

__device__ __forceinline__ void len_base_of(int code, int& base, int& xbits) {
    if (code < 8)        { base = code + 3; xbits = 0; }
    else if (code == 28) { base = 258; xbits = 0; }      // never produced
    else { xbits = (code >> 2) - 1; base = ((4 + (code & 3)) << xbits) + 3; }
}

__device__ __forceinline__ void pos_base_of(int code, int& base, int& xbits) {
    if (code < 4) { base = code + 1; xbits = 0; }
    else { xbits = (code >> 1) - 1; base = ((2 + (code & 1)) << xbits) + 1; }
}

// Bit source over global memory (bitstream.h:65-93): the stream is read MSB
// first; the reference fetches it in 8-byte big-endian groups and fails with
// E2BIG when a group is incomplete (:72-80), so only the first limit/8*8 bytes
// are readable.  Driven wave-uniformly: `acc` keeps the next `valid` (>= 32
// after fill()) bits left-aligned.  The stream itself sits in two VGPR rows
// (lane j = dword wbase+j / wbase+64+j, already big-endian), loaded with one
// coalesced vector load per 2048 bits and picked with v_readlane, so no memory
// latency (and no scalar-load wait mixed into the LDS waits) sits between two
// symbols.  Bits past the readable end read as zero and raise E2BIG once they
// are CONSUMED (checked by the caller per symbol).
struct BitSource {
    const uint8_t* in;
    uint64_t readable;   // bits
    uint64_t pos;        // bits consumed
    uint64_t acc;
    int      valid;      // bits in acc
    uint32_t next_dw;    // dword index that follows the bits in acc
    uint32_t wbase;      // dword index held by lane 0 of `cur`
    uint32_t cur, nxt;   // VGPR rows
    int      lane;
    int      error;

    __device__ __forceinline__ uint32_t load_row(uint32_t dw0) const {
        const uint64_t k = (uint64_t)dw0 + (uint32_t)lane;
        if (k * 32 + 32 <= readable) {
            return __builtin_bswap32(reinterpret_cast<const uint32_t*>(in)[k]);
        }
        return 0u;
    }
    __device__ __forceinline__ uint32_t dword_at(uint32_t k) {
        uint32_t d = k - wbase;
        if (d >= 2 * kWave) {                           // outside both rows (also: behind them)
            wbase = k; cur = load_row(k); nxt = load_row(k + kWave); d = 0;
        } else if (d >= kWave) {
            cur = nxt; wbase += kWave; nxt = load_row(wbase + kWave); d -= kWave;
        }
        return (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)d);
    }
    __device__ __forceinline__ void seek(uint64_t bit) {
        pos = bit;
        const uint32_t k = (uint32_t)(bit >> 5);
        const int sh = (int)(bit & 31u);
        const uint64_t hi = dword_at(k), lo = dword_at(k + 1);
        acc = ((hi << 32) | lo) << sh;
        valid = 64 - sh;
        next_dw = k + 2;
    }
    __device__ __forceinline__ void open(const uint8_t* p, uint64_t limit_bytes, uint64_t start_bit, int lane_) {
        in = p;
        lane = lane_;
        readable = (limit_bytes / 8) * 64;
        error = 0;
        wbase = (uint32_t)(start_bit >> 5);
        cur = load_row(wbase);
        nxt = load_row(wbase + kWave);
        seek(start_bit);
    }

    // make at least 32 bits available
    __device__ __forceinline__ void fill() {
        if (valid < 32) {
            acc |= (uint64_t)dword_at(next_dw) << (32 - valid);
            valid += 32;
            next_dw++;
        }
    }

    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)(acc >> (64 - n)); }

    __device__ __forceinline__ void skip(int n) { acc <<= n; valid -= n; pos += (uint64_t)n; }

    // consumed past the readable end?  (bitstream.h:74)
    __device__ __forceinline__ bool overrun() const { return pos > readable; }

    // value LSB first (bitstream.h:95-103), n = 1..16
    __device__ __forceinline__ uint32_t get_lsb(int n) {
        fill();
        const uint32_t v = __brev(peek(n)) >> (32 - n);
        skip(n);
        return v;
    }
};

// ---------------------------------------------------------------------------
// Adaptive Huffman tree in LDS.
//
// Two ways through huffman_inc_frequency (huffman.h:218-235):
//  * slow path = the reference's sequence restated literally (relabel /
//    order_pair / sum / climb / changed / insert below), run by lane 0;
//  * fast path (bump_wave) = the common case in which the update changes no
//    link.  Measured on the benchmark's blocks 99.0 % of the updates are of that
//    kind (2,200 of 222,439 symbols restructure the tree).  Lane k owns level k
//    of the leaf->root chain and evaluates, in parallel, exactly the two tests
//    the reference would make at that level with the incremented counts:
//       swap    (huffman.h:75)   freq'[lo(p)] > freq'[hi(p)]
//       promote (huffman.h:108)  child is hi(p), p is not the root,
//                                freq'[child] > freq[uncle]
//    If no lane raises a flag the reference would only add 1 to every node on
//    the chain (every internal count is the sum of its children), which the
//    lanes then do in one step.  Any flag -> the slow path runs instead, from
//    the untouched state, so the result is the reference's in both cases.
//
// A node's links are one 64-bit word: low dword up | up2 | up3, high dword
// lo | hi (10 bits each, 0x3FF = none) | depth (6 bits).  up2 / up3 (grandparent, great-grandparent)
// exist only to shorten the leaf->root walk of the fast path to one dependent
// LDS read per three levels; the slow path maintains them wherever it moves a
// subtree (relabel), and never reads them.
constexpr uint32_t kNil = 0x3FFu;
constexpr int kStack = 128;          // deepest chain the slow path follows (fault beyond)
constexpr int kMaxFastDepth = 60;

struct Node {                        // unpacked view of a link word
    uint32_t up, up2, up3, lo, hi, bits;
};

__device__ __forceinline__ Node unpack(uint64_t w) {
    Node n;
    const uint32_t a = (uint32_t)w, b = (uint32_t)(w >> 32);
    n.up = a & 0x3FFu; n.up2 = (a >> 10) & 0x3FFu; n.up3 = (a >> 20) & 0x3FFu;
    n.lo = b & 0x3FFu; n.hi = (b >> 10) & 0x3FFu; n.bits = (b >> 20) & 0x3Fu;
    return n;
}

__device__ __forceinline__ uint64_t pack(const Node& n) {
    return (uint64_t)(n.up | (n.up2 << 10) | (n.up3 << 20)) |
           ((uint64_t)(n.lo | (n.hi << 10) | (n.bits << 20)) << 32);
}

constexpr uint64_t kEmptyLinks = 0x3FFFFFFFull | (0xFFFFFull << 32);   // all nil, depth 0

// shared per-wave scratch for both trees
struct TreeScratch {
    uint16_t walk[kStack];
    uint32_t pend[kStack];     // parent << 16 | child
};

// whole-wave shifts by one lane (DPP, no LDS traffic); edge lanes receive kNil
__device__ __forceinline__ int lane_above(int v) {          // lane k <- lane k+1
    return __builtin_amdgcn_update_dpp((int)kNil, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ int lane_below(int v) {          // lane k <- lane k-1
    return __builtin_amdgcn_update_dpp((int)kNil, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t lanes_under(uint64_t mask) {      // bits of mask below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// inclusive prefix sum over the wavefront in six DPP steps (row scans, then the row totals)
__device__ __forceinline__ uint32_t wave_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112 /* row_shr:2 */, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114 /* row_shr:4 */, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118 /* row_shr:8 */, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return v;
}

__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// one lane's share of a root path: its node and that node's parent
struct Chain {
    int mine, par;
    int levels;        // uniform: edges between leaf and root
    bool holds;        // this lane holds a node of the path
    bool active;       // ... and the node has a parent
    bool has_g;        // ... and that parent is not the root
};

// The slow path lives in real (non-inlined) functions so that the per-symbol loop
// of the kernels stays a few hundred instructions: it is taken for ~1 % of the
// symbols, and inlining it at every call site made the kernels ~50 KB of code.
// State crosses the call as plain values: LDS pointers + the packed registers.
template <class T>
__device__ __noinline__ uint32_t slow_insert(uint64_t* link, uint32_t* freq, TreeScratch* scratch,
                                             uint32_t regs, int sym, int lane);
template <class T>
__device__ __noinline__ uint32_t slow_changed(uint64_t* link, uint32_t* freq, TreeScratch* scratch,
                                              uint32_t regs, int sym, int lane);
template <class T>
__device__ __noinline__ void slow_build_lut(const uint64_t* link, uint16_t* lut, int lane);

// REF_LEAVES is the reference's leaf count n (512 / 32, squeeze.h:204-205): it
// only fixes how many leaf splits huffman_insert allows (n - 2, huffman.h:180).
template <int LEAVES, int NODES, int REF_LEAVES, int LUT_BITS>
struct Tree {
    // LDS storage
    uint64_t* link;
    uint32_t* freq;
    TreeScratch* scratch;
    uint16_t* lut;      // decoder only: 2^kLutBits entries, node | bits used << 10
    // wave-uniform registers
    int next;           // next free internal id
    int depth;          // huffman.h:26 high-water mark
    int complete;       // huffman.h:27
    int fault;          // stack / depth guard (never set for realistic streams)
    int lut_ok;         // decoder: the lookup table matches the tree

    static constexpr int kRoot = LEAVES;
    static constexpr int kLutBits = LUT_BITS;      // decoder table: 2^kLutBits entries
    static constexpr int kIdEnd =
        (LEAVES + 1 + REF_LEAVES - 2) < NODES ? (LEAVES + 1 + REF_LEAVES - 2) : NODES;

    // all lanes: clear storage (huffman.h:251-269)
    __device__ __forceinline__ void init_all(int lane) {
        for (int i = lane; i < NODES; i += kWave) {
            link[i] = kEmptyLinks;
            freq[i] = 0;
        }
        next = kRoot + 1; depth = 0; complete = 0; fault = 0; lut_ok = 0;
    }

    __device__ __forceinline__ Node ld(int i) const { return unpack(link[i]); }
    __device__ __forceinline__ void st(int i, const Node& n) { link[i] = pack(n); }
    __device__ __forceinline__ uint32_t up_of(int i) const { return (uint32_t)link[i] & 0x3FFu; }

    // registers <-> one word (next:10 | depth:8 | complete:1 | fault:1 | ok:1 at bit 31)
    __device__ __forceinline__ uint32_t pack_regs() const {
        return (uint32_t)next | ((uint32_t)(depth & 0xFF) << 10) | ((uint32_t)(complete & 1) << 18) |
               ((uint32_t)(fault & 1) << 19);
    }
    __device__ __forceinline__ void unpack_regs(uint32_t r) {
        next = (int)(r & 0x3FFu); depth = (int)((r >> 10) & 0xFFu);
        complete = (int)((r >> 18) & 1u); fault = (int)((r >> 19) & 1u);
    }

    // after a lane-0 section: make the registers uniform again
    __device__ __forceinline__ void sync_regs() {
        next = __builtin_amdgcn_readfirstlane(next);
        depth = __builtin_amdgcn_readfirstlane(depth);
        complete = __builtin_amdgcn_readfirstlane(complete);
        fault = __builtin_amdgcn_readfirstlane(fault);
    }

    // ---------------- slow path: the reference sequence, one lane -----------
    // huffman.h:41-62 (depths; codes are read off the chain when needed).  Also
    // refreshes up2/up3 of everything below `top`.
    __device__ __forceinline__ void relabel(int top) {
        if (top == kRoot) { depth = 0; }
        int sp = 0;
        scratch->walk[sp++] = (uint16_t)top;
        while (sp > 0) {
            const int v = scratch->walk[--sp];
            const Node n = ld(v);
            const int b = (int)n.bits;
            if (b > depth) { depth = b; }
            if (b >= 63) { fault = 1; continue; }             // reference asserts bits < 63
            const uint32_t kids[2] = { n.hi, n.lo };
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint32_t ch = kids[j];
                if (ch == kNil) { continue; }
                Node c = ld((int)ch);
                c.bits = (uint32_t)(b + 1);
                c.up2 = n.up;
                c.up3 = n.up2;
                st((int)ch, c);
                if (ch >= (uint32_t)LEAVES) {
                    if (sp < kStack) { scratch->walk[sp++] = (uint16_t)ch; } else { fault = 1; }
                } else if (b + 1 > depth) { depth = b + 1; }
            }
        }
    }

    // huffman.h:90-96
    __device__ __forceinline__ void sum(int i) {
        const Node n = ld(i);
        const uint32_t a = n.lo != kNil ? freq[n.lo] : 0u;
        const uint32_t b = n.hi != kNil ? freq[n.hi] : 0u;
        freq[i] = a + b;
    }

    // huffman.h:64-86
    __device__ __forceinline__ int order_pair(int i) {
        const uint32_t p = up_of(i);
        if (p == kNil) { return i; }
        Node n = ld((int)p);
        if (n.lo != kNil && n.hi != kNil && freq[n.lo] > freq[n.hi]) {
            const uint32_t l = n.lo, r = n.hi;
            n.lo = r; n.hi = l;
            st((int)p, n);
            relabel((int)p);
            return i == (int)l ? (int)r : (int)l;
        }
        return i;
    }

    // climb of huffman_frequency_changed (huffman.h:132-142): refresh sums and
    // sibling order up to the root, remembering (parent, child) per level
    __device__ __forceinline__ int climb(int i, int sp) {
        for (;;) {
            const uint32_t p = up_of(i);
            if (p == kNil) { sum(i); break; }
            sum((int)p);
            i = order_pair(i);
            if (sp < kStack) { scratch->pend[sp++] = (p << 16) | (uint32_t)i; }
            else { fault = 1; }
            i = (int)p;
        }
        return sp;
    }

    // huffman.h:130-147 with move_up (:98-128) inlined; LIFO order equals the
    // reference's recursion order because both inner calls are tail calls
    __device__ __forceinline__ void changed(int start) {
        int sp = climb(start, 0);
        while (sp > 0) {
            const uint32_t e = scratch->pend[--sp];
            const int p = (int)(e >> 16), c = (int)(e & 0xFFFFu);
            const Node np = ld(p);
            if (np.up == kNil || np.hi != (uint32_t)c) { continue; }    // :143
            const int par = (int)up_of(c);
            const int g = (int)up_of(par);
            Node ng = ld(g);
            const bool par_is_left = (ng.lo == (uint32_t)par);
            const int uncle = (int)(par_is_left ? ng.hi : ng.lo);
            if (!(freq[c] > freq[uncle])) { continue; }              // :108
            Node nc = ld(c);
            nc.up = (uint32_t)g;
            st(c, nc);
            if (par_is_left) { ng.hi = (uint32_t)c; } else { ng.lo = (uint32_t)c; }
            st(g, ng);
            Node npar = ld(par);
            npar.hi = (uint32_t)uncle;
            st(par, npar);
            Node nu = ld(uncle);
            nu.up = (uint32_t)par;
            st(uncle, nu);
            sum(par);
            sum(g);
            (void)order_pair(c);
            (void)order_pair(uncle);
            (void)order_pair(par);
            relabel(g);
            sp = climb(g, sp);                                        // :126
        }
    }

    // huffman.h:149-216
    // huffman.h:149-209, one lane: hang leaf i into the tree.  Leaves every field of the
    // nodes it touches valid (depth, up2, up3 as well), so that the whole-wave climb can start
    // from the result.  Returns  start | at << 10 | ok << 20 : the node huffman_frequency_changed
    // is called on (:212; the OTHER sibling when the pair was reordered, :173) and the node
    // whose subtree is relabelled afterwards (:213).
    __device__ __forceinline__ uint32_t insert_splice(int i) {
        uint32_t ok = 1;
        int at = kRoot;
        freq[i] = 1;
        while (at >= LEAVES) {                                        // :156-170
            Node n = ld(at);
            if (n.hi == kNil || n.lo == kNil) {
                if (n.hi == kNil) { n.hi = (uint32_t)i; } else { n.lo = (uint32_t)i; }
                st(at, n);
                Node ni = ld(i);
                ni.up = (uint32_t)at; ni.up2 = n.up; ni.up3 = n.up2;
                ni.bits = n.bits + 1;
                st(i, ni);
                break;
            }
            at = (int)n.lo;
        }
        if (at >= LEAVES) {                                           // :171-173
            freq[at] += 1;
            i = order_pair(i);
        } else if (next >= kIdEnd) {                                  // :180-182
            ok = 0;
            complete = 1;
        } else {                                                      // :184-209
            const int fresh = next++;
            Node na = ld(at);
            Node nf = na;                     // takes at's place: same up/up2/up3/depth
            nf.lo = (uint32_t)at; nf.hi = (uint32_t)i;
            st(fresh, nf);
            freq[fresh] = freq[at];
            if (na.up != kNil) {
                Node nabove = ld((int)na.up);
                if (nabove.lo == (uint32_t)at) { nabove.lo = (uint32_t)fresh; }
                else                            { nabove.hi = (uint32_t)fresh; }
                st((int)na.up, nabove);
            }
            na.up = (uint32_t)fresh; na.up2 = nf.up; na.up3 = nf.up2;
            na.bits = nf.bits + 1;
            st(at, na);
            Node ni = ld(i);
            ni.up = (uint32_t)fresh; ni.up2 = nf.up; ni.up3 = nf.up2;
            ni.bits = nf.bits + 1;
            st(i, ni);
            sum(fresh);
            at = fresh;
        }
        return (uint32_t)i | ((uint32_t)at << 10) | (ok << 20);
    }

    __device__ __forceinline__ bool insert(int i) {                   // everything on one lane
        const uint32_t h = insert_splice(i);
        changed((int)(h & 0x3FFu));                                   // :212
        relabel((int)((h >> 10) & 0x3FFu));                           // :213
        return (h >> 20) != 0;
    }

    // whole wave: insert through lane 0 (unseen symbols are rare: <= 286 per stream)
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
        const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)slow_insert<Tree>(link, freq, scratch, pack_regs(), i, lane));
        unpack_regs(r);
        lut_ok = 0;
        SQZ_ST_END(0)
        return (r >> 31) != 0;
    }

    __device__ __forceinline__ void changed_wave(int s, int lane) {
        SQZ_ST_BEGIN
        unpack_regs((uint32_t)__builtin_amdgcn_readfirstlane(
            (int)slow_changed<Tree>(link, freq, scratch, pack_regs(), s, lane)));
        lut_ok = 0;
        SQZ_ST_END(1)
    }

    __device__ __forceinline__ void build_lut(int lane) {
        SQZ_ST_BEGIN
        slow_build_lut<Tree>(link, lut, lane);
        lut_ok = 1;
        SQZ_ST_END(2)
    }

    // ---------------- fast path ------------------------------------------------
    // leaf -> root walk, three levels per dependent LDS read; lane k receives
    // level k (0 = the leaf).  Uniform control flow.
    __device__ __forceinline__ Chain chain_up(int s, int lane) const {
        // the leaf's own word gives its depth, so the walk has a known trip count:
        // one dependent read per three levels, no end-of-chain tests inside
        const uint64_t w0 = uni64(link[s]);
        const int levels = (int)((uint32_t)(w0 >> 52) & 0x3Fu);
        const int stop = levels < kMaxFastDepth ? levels : kMaxFastDepth;
        uint32_t w = (uint32_t)w0;
        int mine = (lane == 0) ? s : (int)kNil;
        int k = 0;
        while (k < stop) {
            const uint32_t f = (uint32_t)(lane - k - 1);          // 0..2: up / up2 / up3
            const uint32_t pick = (w >> (f < 3u ? 10u * f : 0u)) & 0x3FFu;
            mine = (f < 3u) ? (int)pick : mine;
            k += 3;
            if (k >= stop) { break; }
            const int a = (int)((w >> 20) & 0x3FFu);
            w = (uint32_t)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const uint32_t*>(link)[2 * a]);
        }
        Chain c;
        c.mine = mine;                       // lanes beyond `levels` hold nil (or junk: not `holds`)
        c.par = lane_above(mine);            // lane k+1 holds the parent
        c.levels = levels;
        c.holds = lane <= levels;
        c.active = lane < levels;
        c.has_g = lane + 1 < levels;
        return c;
    }

    // ---------------- restructuring, whole wave --------------------------------
    // The same sequence as changed() above (huffman.h:130-147, :98-128), with the
    // per-level work of a climb done by one lane per level and the pending
    // (parent, child) pairs tested all at once.  Uniform control flow.

    // sibling order under i's parent (huffman.h:64-86) without the relabel: a swap
    // changes codes, never depths, and codes are read off the chain when needed
    __device__ __forceinline__ void order_only(int i) {
        const uint32_t p = up_of(i);
        if (p == kNil) { return; }
        Node n = ld((int)p);
        if (n.lo != kNil && n.hi != kNil && freq[n.lo] > freq[n.hi]) {
            const uint32_t l = n.lo;
            n.lo = n.hi; n.hi = l;
            st((int)p, n);
        }
    }

    // depth / up2 / up3 of every attached node from its parent's, repeated until
    // nothing moves (one pass per level of the subtree that changed).  `reset`:
    // the reference walked from the root, which restarts the depth high-water mark.
    __device__ __forceinline__ void relabel_wave(bool reset, int lane) {
        if (reset) { depth = 0; }
        uint32_t seen = 0;
        for (int pass = 0; pass < 70; pass++) {
            bool moved = false;
            uint32_t all = 0;
            for (int v = lane; v < next; v += kWave) {
                const uint64_t w = link[v];
                const uint32_t up = (uint32_t)w & 0x3FFu;
                if (up == kNil) { continue; }                       // the root, or not in the tree
                const uint64_t pw = link[up];
                uint32_t d = (((uint32_t)(pw >> 32) >> 20) & 0x3Fu) + 1u;
                if (d > 63u || (d == 63u && v >= LEAVES)) { fault = 1; d = d > 63u ? 63u : d; }
                const uint32_t lo_w = up | (((uint32_t)pw & 0xFFFFFu) << 10);
                const uint32_t hi_w = ((uint32_t)(w >> 32) & 0xFFFFFu) | (d << 20);
                all = d > all ? d : all;
                if (lo_w != (uint32_t)w || hi_w != (uint32_t)(w >> 32)) {
                    link[v] = (uint64_t)lo_w | ((uint64_t)hi_w << 32);
                    seen = d > seen ? d : seen;
                    moved = true;
                }
            }
            if (__ballot(moved) == 0) { if (reset) { seen = all; } break; }
        }
        // wave maximum of `seen` -> the high-water mark
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)seen, o);
            seen = other > seen ? other : seen;
        }
        const int top = __builtin_amdgcn_readfirstlane((int)seen);
        if (top > depth) { depth = top; }
        fault = (__ballot(fault != 0) != 0) ? 1 : 0;
    }

    // After a promotion only the two subtrees that traded places changed depth / up2 / up3.
    // Fix them top-down, one tree level per pass, one lane per node of the level (their
    // children queue up for the next pass).  A level wider than the wave, or a promotion
    // under the root (the reference's walk from the root restarts the depth mark), goes
    // through the fixpoint over all nodes instead.
    __device__ __forceinline__ void relabel_moved(int a, int b, bool reset, int lane) {
        if (reset) { relabel_wave(true, lane); return; }
        uint16_t* const q = scratch->walk;                     // two queues of 64
        if (lane == 0) { q[0] = (uint16_t)a; q[1] = (uint16_t)b; }
        int n = 2, cur = 0;
        uint32_t seen = 0;
        bool wide = false;
        for (int level = 0; n > 0; level++) {
            if (level >= 70) { wide = true; break; }               // never: deeper than any tree
            const bool on = lane < n;
            const int v = on ? (int)q[cur + lane] : kRoot;
            const uint64_t w = link[v];
            const uint32_t up = on ? ((uint32_t)w & 0x3FFu) : (uint32_t)kRoot;
            const uint64_t pw = link[up];
            uint32_t d = (((uint32_t)(pw >> 32) >> 20) & 0x3Fu) + 1u;
            if (on && (d > 63u || (d == 63u && v >= LEAVES))) { fault = 1; }
            d = d > 63u ? 63u : d;
            const uint32_t kids = (uint32_t)(w >> 32) & 0xFFFFFu;
            if (on) {
                link[v] = (uint64_t)(up | (((uint32_t)pw & 0xFFFFFu) << 10)) | ((uint64_t)(kids | (d << 20)) << 32);
                seen = d > seen ? d : seen;
            }
            const uint32_t lo = kids & 0x3FFu, hi = (kids >> 10) & 0x3FFu;
            const bool has_lo = on && lo != kNil, has_hi = on && hi != kNil;
            const uint64_t mlo = __ballot(has_lo), mhi = __ballot(has_hi);
            const int nlo = __builtin_popcountll(mlo), nhi = __builtin_popcountll(mhi);
            if (nlo + nhi > kWave) { wide = true; break; }
            const int nxt = cur ^ kWave;
            if (has_lo) { q[nxt + (int)lanes_under(mlo)] = (uint16_t)lo; }
            if (has_hi) { q[nxt + nlo + (int)lanes_under(mhi)] = (uint16_t)hi; }
            n = nlo + nhi;
            cur = nxt;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)seen, o);
            seen = other > seen ? other : seen;
        }
        const int top = __builtin_amdgcn_readfirstlane((int)seen);
        if (top > depth) { depth = top; }
        fault = (__ballot(fault != 0) != 0) ? 1 : 0;
        if (wide) { relabel_wave(false, lane); }
    }

    // the climb of huffman_frequency_changed from node i (huffman.h:132-142): lane k
    // owns level k of i's root path; new sums by prefix sum, sibling order per level,
    // one pending pair per level (bottom first)
    __device__ __forceinline__ int climb_wave(int i, int sp, int lane) {
        const Chain c = chain_up(i, lane);
        const int levels = c.levels;
        if (levels >= kMaxFastDepth || sp + levels > kWave) { fault = 1; return 0; }
        if (levels == 0) {                                            // i is the root
            if (lane == 0) { sum(i); }
            return sp;
        }
        const uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)freq[i]);
        const int x = c.active ? c.mine : kRoot;
        const int p = c.active ? c.par : kRoot;
        const Node lp = ld(p);
        const bool is_hi = lp.hi == (uint32_t)x;
        const uint32_t sib = is_hi ? lp.lo : lp.hi;
        const bool has_sib = c.active & (sib != kNil);
        const uint32_t fs = has_sib ? freq[sib] : 0u;
        const uint32_t incl = wave_scan(c.active ? fs : 0u);
        const uint32_t fx = f0 + incl - fs;                          // my node's count, refreshed
        const bool swap = has_sib & (is_hi ? (fs > fx) : (fx > fs));  // lo count > hi count
        if (c.active) {
            freq[p] = f0 + incl;
            if (swap) {
                Node n = lp;
                n.lo = lp.hi; n.hi = lp.lo;
                st(p, n);
            }
            scratch->pend[sp + lane] = ((uint32_t)p << 16) | (swap ? sib : (uint32_t)x);
        }
        if ((__ballot(swap) >> (levels - 1)) & 1ull) { relabel_wave(true, lane); }   // swapped under the root
        return sp + levels;
    }

    __device__ __forceinline__ void changed_all(int start, int lane) {
        int sp = climb_wave(start, 0, lane);
        while (sp > 0) {
            // every pending pair at once: the reference pops them from the top and the
            // ones that fail its tests (:143, :108) change nothing
            bool hit = false;
            int p = 0, ch = 0, g = 0, uncle = 0, left = 0;
            if (lane < sp) {
                const uint32_t e = scratch->pend[lane];
                p = (int)(e >> 16); ch = (int)(e & 0xFFFFu);
                const Node np = ld(p);
                if (np.up != kNil && np.hi == (uint32_t)ch) {
                    g = (int)np.up;
                    const Node ng = ld(g);
                    left = ng.lo == (uint32_t)p ? 1 : 0;
                    uncle = (int)(left ? ng.hi : ng.lo);
                    hit = uncle != (int)kNil && freq[ch] > freq[uncle];
                }
            }
            const uint64_t hits = __ballot(hit);
            if (hits == 0) { break; }
            const int j = 63 - __builtin_clzll(hits);
            sp = j;
            p = __builtin_amdgcn_readlane(p, j);
            ch = __builtin_amdgcn_readlane(ch, j);
            g = __builtin_amdgcn_readlane(g, j);
            uncle = __builtin_amdgcn_readlane(uncle, j);
            left = __builtin_amdgcn_readlane(left, j);
            if (lane == 0) {                                          // move_up, :110-125
                Node nc = ld(ch);
                nc.up = (uint32_t)g;
                st(ch, nc);
                Node ng = ld(g);
                if (left) { ng.hi = (uint32_t)ch; } else { ng.lo = (uint32_t)ch; }
                st(g, ng);
                Node npar = ld(p);
                npar.hi = (uint32_t)uncle;
                st(p, npar);
                Node nu = ld(uncle);
                nu.up = (uint32_t)p;
                st(uncle, nu);
                sum(p);
                sum(g);
                order_only(ch);
                order_only(uncle);
                order_only(p);
            }
            relabel_moved(ch, uncle, g == kRoot, lane);
            sp = climb_wave(g, sp, lane);                             // :126
        }
    }

    // huffman_inc_frequency for an ATTACHED leaf s whose chain is `c`.
    // Returns the ballot of "my node is the hi child" (the stream-order code).
    __device__ __forceinline__ uint64_t bump_wave(int s, const Chain& c, int lane) {
        // stage A: my count and my parent's links
        const int i_mine = c.holds ? c.mine : kRoot;
        const int i_par = c.active ? c.par : kRoot;
        const uint32_t fc = freq[i_mine];
        const Node lp = ld(i_par);
        const bool is_hi = c.active & (lp.hi == (uint32_t)c.mine);
        const uint64_t code = __ballot(is_hi);
        if (complete != 0 || depth >= 63) { complete = 1; return code; }   // huffman.h:228-234
        // stage B: my sibling's count; my uncle is my parent's sibling = the lane above
        const uint32_t sib = is_hi ? lp.lo : lp.hi;
        const bool has_sib = c.active & (sib != kNil);
        const uint32_t fs = freq[has_sib ? (int)sib : kRoot];
        const uint32_t fu = (uint32_t)lane_above((int)fs);
        const bool has_unc = is_hi & c.has_g & (lane_above(has_sib ? 1 : 0) == 1);
        const uint32_t fc1 = fc + 1;
        const uint32_t big = is_hi ? fs : fc1;                        // swap iff lo count > hi count
        const uint32_t small = is_hi ? fc1 : fs;
        const bool flag = (has_sib & (big > small)) | (has_unc & (fc1 > fu)) |
                          (c.levels >= kMaxFastDepth);
        if (__ballot(flag) == 0) {
            if (c.holds) { freq[c.mine] = fc + 1; }
        } else {
            changed_wave(s, lane);
        }
        return code;
    }
};

template <class T>
__device__ __noinline__ uint32_t slow_insert(uint64_t* link, uint32_t* freq, TreeScratch* scratch,
                                             uint32_t regs, int sym, int lane) {
    T t;
    t.link = link; t.freq = freq; t.scratch = scratch; t.lut = nullptr; t.lut_ok = 0;
    t.unpack_regs(regs);
    // the splice on lane 0, the climb and the promotions on the whole wave (as for any other
    // restructure), the closing relabel of the small subtree the leaf went into on lane 0
    uint32_t hand = 0;
    if (lane == 0) { hand = t.insert_splice(sym); }
    t.unpack_regs((uint32_t)__builtin_amdgcn_readfirstlane((int)t.pack_regs()));
    hand = (uint32_t)__builtin_amdgcn_readfirstlane((int)hand);
    const int start = (int)(hand & 0x3FFu), at = (int)((hand >> 10) & 0x3FFu);
    if (t.depth + 4 >= kMaxFastDepth) {                 // chains too long for one lane per level
        if (lane == 0) { t.changed(start); }
    } else {
        t.changed_all(start, lane);
    }
    if (lane == 0) { t.relabel(at); }
    const uint32_t r = t.pack_regs() | ((hand >> 20) << 31);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
}

template <class T>
__device__ __noinline__ uint32_t slow_changed(uint64_t* link, uint32_t* freq, TreeScratch* scratch,
                                              uint32_t regs, int sym, int lane) {
    T t;
    t.link = link; t.freq = freq; t.scratch = scratch; t.lut = nullptr; t.lut_ok = 0;
    t.unpack_regs(regs);
    if (t.depth + 4 >= kMaxFastDepth) {                 // chains too long for one lane per level
        if (lane == 0) { t.freq[sym] += 1; t.changed(sym); }
    } else {
        if (lane == 0) { t.freq[sym] += 1; }
        t.changed_all(sym, lane);
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)t.pack_regs());
}

// decoder: table over the next kLutBits stream bits -> (node reached, bits used).
// Entry = node | used << 10; a missing child gives node = kNil.
// One step of the table build: the entry for a prefix one bit longer than `parent`'s.
// A finished entry (leaf, or a missing child) is inherited; an internal node hands
// down the child the new bit selects, with one more bit used.
template <class T>
__device__ __forceinline__ uint32_t lut_descend(const uint64_t* link, uint32_t parent, int bit, int level) {
    const uint32_t node = parent & 0x3FFu;
    const bool inside = node != kNil && node >= (uint32_t)T::kRoot;
    const uint32_t kids = (uint32_t)(link[inside ? (int)node : (int)T::kRoot] >> 32);
    const uint32_t child = (kids >> (bit ? 10 : 0)) & 0x3FFu;
    return inside ? (child | ((uint32_t)level << 10)) : parent;
}

template <class T>
__device__ __noinline__ void slow_build_lut(const uint64_t* link, uint16_t* lut, int lane) {
    // level by level from the root: the table for (L+1)-bit prefixes follows from the one
    // for L-bit prefixes with one link read per entry, 2^(L+1) entries per level instead
    // of a root walk per final entry.  Lane j holds entry j while a level fits the wave.
    static_assert(T::kLutBits == 6 || T::kLutBits == 8, "table widths the decoder uses");
    const int half = (lane >> 1) * 4;                         // byte address of lane j>>1
    const int bit = lane & 1;
    uint32_t e = (uint32_t)T::kRoot;                          // the 0-bit prefix: root, no bits used
#pragma unroll
    for (int level = 1; level <= 6; level++) {
        const uint32_t parent = (uint32_t)__builtin_amdgcn_ds_bpermute(half, (int)e);
        e = lut_descend<T>(link, parent, bit, level);
    }
    if (T::kLutBits == 6) {
        lut[lane] = (uint16_t)e;
        return;
    }
    // 128 entries: j and j + 64 descend from entries j>>1 and 32 + (j>>1) of level 6
    const uint32_t p0 = (uint32_t)__builtin_amdgcn_ds_bpermute(half, (int)e);
    const uint32_t p1 = (uint32_t)__builtin_amdgcn_ds_bpermute(half + 128, (int)e);
    const uint32_t r0 = lut_descend<T>(link, p0, bit, 7);
    const uint32_t r1 = lut_descend<T>(link, p1, bit, 7);
    // 256 entries: j + 64k descends from entry 32k + (j>>1) of level 7 (r0: 0..63, r1: 64..127)
    const uint32_t q0 = (uint32_t)__builtin_amdgcn_ds_bpermute(half, (int)r0);
    const uint32_t q1 = (uint32_t)__builtin_amdgcn_ds_bpermute(half + 128, (int)r0);
    const uint32_t q2 = (uint32_t)__builtin_amdgcn_ds_bpermute(half, (int)r1);
    const uint32_t q3 = (uint32_t)__builtin_amdgcn_ds_bpermute(half + 128, (int)r1);
    lut[lane]       = (uint16_t)lut_descend<T>(link, q0, bit, 8);
    lut[lane + 64]  = (uint16_t)lut_descend<T>(link, q1, bit, 8);
    lut[lane + 128] = (uint16_t)lut_descend<T>(link, q2, bit, 8);
    lut[lane + 192] = (uint16_t)lut_descend<T>(link, q3, bit, 8);
}

constexpr int kLitLeaves = 288;               // symbols 0..285 (+2 pad)
constexpr int kLitNodes  = kLitLeaves + 288;  // root + <=285 splits (+pad)
constexpr int kPosLeaves = 32;
constexpr int kPosNodes  = 64;

using LitTree = Tree<kLitLeaves, kLitNodes, 512, 8>;
using PosTree = Tree<kPosLeaves, kPosNodes, 32, 6>;

// LDS image of one stream's entropy state
struct DecodeLuts {
    uint16_t lit[1 << 8];
    uint16_t pos[1 << 6];
};

struct EntropyLds {
    uint64_t lit_link[kLitNodes];
    uint64_t pos_link[kPosNodes];
    uint32_t lit_freq[kLitNodes];
    uint32_t pos_freq[kPosNodes];
    TreeScratch scratch;
};

// ---------------------------------------------------------------------------
// Up to 64 tokens per step: lane = token (its lit-tree symbol, then, for a back
// reference, its pos-tree symbol); every lane walks its own leaf->root chains.  Both trees
// are one array pair (pos ids + kLitNodes).
//
// Exactness: the reference updates the symbols one after another.  It would
// change no link during the whole step if, for every chain node c (parent p,
// sibling s, uncle u; f0 = counts before the step, n(c) = chains through c):
//     c is lo(p):                 f0(c) + n(c) <= f0(s)      (never overtakes s)
//     c is hi(p):                 f0(s) <= f0(c) + 1         (no inversion waiting)
//     c is hi(p), p not the root: f0(c) + n(c) <= f0(u)      (never overtakes u)
// (induction over the sequential updates: every count only grows, c's count
// never exceeds f0(c)+n(c), s and u never drop below f0; a lo sibling that is
// itself on a chain is covered by its own first test).  Then the result of the
// step is f0 + n on every chain node and every code is the static tree's.  The tests
// may be conservative -- the token a step stops at goes through the one-at-a-time
// path -- but the output is always the reference's.
//
// n(c) is counted in the top byte of the count word with LDS atomic adds whose return
// value tells a lane how many lanes were counted at c before it (its rank r).  A lane is
// `bad` if, at some node of its chains, f0(c) + r + 1 would break one of the three tests.
// Take the tokens in front of the first bad lane: for every node at most `allowed` of them
// pass through it (they all have rank < allowed), so the tests hold for that prefix
// whatever order the hardware serialises same-address atomics in -- a different
// order can only shorten the prefix.  Two sweeps over the chains:
//   1. count + test + read the code off the chain
//   2. prefix lanes: count += 1, n -= 1; the others: n -= 1
// Batching stops once a stream nears 2^24 tokens -- no count may outgrow its 24 bits -- (the
// serial path has no such limit).
constexpr uint32_t kCntShift = 24;
constexpr uint32_t kCntOne = 1u << kCntShift;
constexpr uint32_t kCountMask = kCntOne - 1u;

// One lane's walk up one chain, three levels per round trip: a node's word names
// its parent, grandparent and great-grandparent, so their counters and child links
// are fetched together.  Written without branches so that the walks of a lane's
// two symbols interleave; a finished walk keeps issuing harmless accesses (adds
// of 0, reads of its last node) until the other one is done.
#ifdef SQZ_STATS
__device__ unsigned long long g_st[8];
#endif
struct LaneWalk {
    int base;
    uint32_t root;       // the tree's root (relative to base): an index that is always safe to follow
    uint32_t c;          // node whose counter comes next
    uint64_t w;          // its word
    bool live, bad, prev_hi;
    uint32_t prev_reach;
    int depth;           // of the leaf (0 when there is none)
    uint32_t code;       // hi/lo choices, leaf level first

    __device__ __forceinline__ void open(const uint64_t* link, int leaf) {
        live = leaf >= 0; prev_hi = false; prev_reach = 0; code = 0;
        base = leaf >= kLitNodes ? kLitNodes : 0;
        root = leaf >= kLitNodes ? (uint32_t)kPosLeaves : (uint32_t)kLitLeaves;
        c = live ? (uint32_t)(leaf - base) : root;
        w = link[base + (int)c];
        depth = live ? (int)((uint32_t)(w >> 52) & 0x3Fu) : 0;
        bad = depth >= 29;                                   // never: such a chain is left to the serial path
    }

    // the tests at one level: `node` (counter value before my add: `old`) under a parent
    // whose children are `kids`; `at` = how many levels above the leaf.  Two properties of the
    // tree at rest keep this short (the oracle asserts both after every update,
    // `make -C oracle check-invariants`): a chain node of a seen symbol always has a sibling,
    // and lo <= hi for every pair -- so a hi child needs no test against its own sibling (the
    // reference's swap can only be triggered from the lo side), only against its uncle, which
    // is the next level's sibling.  Levels past the end of a chain look at the root's children
    // instead (always two of them while any symbol is seen), so no index needs a guard.
    template <bool kWantCode>
    __device__ __forceinline__ void level(const uint32_t* freq, uint32_t node, uint32_t old, uint32_t kids,
                                          bool valid, int at) {
        const uint32_t lo = kids & 0x3FFu, hi = (kids >> 10) & 0x3FFu;
        const bool is_hi = (hi == node);
        const uint32_t sib = is_hi ? lo : hi;
        const uint32_t f0s = freq[base + (int)sib] & kCountMask;
        const uint32_t reach = (old & kCountMask) + (old >> kCntShift);  // the node's count BEFORE my add
        bad |= valid & !is_hi & (reach >= f0s);                          // a lo child overtaking its sibling (reach + 1 > f0s)
        bad |= valid & prev_hi & (prev_reach >= f0s);                    // the hi child below overtaking its uncle
        if (kWantCode) { code |= ((valid & is_hi) ? 1u : 0u) << at; }
        prev_hi = is_hi;             // past the end of a chain these are never looked at again:
        prev_reach = reach;          // every later level is invalid too and masks its tests
    }

    // sweep 1: count + test (+ code: the encoder emits it, the decoder has no use for it),
    // levels at, at + 1, at + 2 above the leaf
    template <bool kWantCode>
    __device__ __forceinline__ void count3(const uint64_t* link, uint32_t* freq, int at) {
        const uint32_t p1 = (uint32_t)w & 0x3FFu, p2 = ((uint32_t)w >> 10) & 0x3FFu, p3 = ((uint32_t)w >> 20) & 0x3FFu;
        const bool v1 = live & (p1 != kNil), v2 = v1 & (p2 != kNil), v3 = v2 & (p3 != kNil);
        const uint32_t i1 = v1 ? p1 : root, i2 = v2 ? p2 : root, i3 = v3 ? p3 : root;
        const uint32_t kids1 = (uint32_t)(link[base + (int)i1] >> 32);
        const uint32_t kids2 = (uint32_t)(link[base + (int)i2] >> 32);
        const uint64_t w3 = link[base + (int)i3];
        // the root is not counted: nothing ever compares its count (it is no one's child,
        // sibling or uncle; the climbs recompute it from its children), and it is the one
        // node every lane would hit
        uint32_t old0 = 0, old1 = 0, old2 = 0;
        if (v1) { old0 = atomicAdd(&freq[base + (int)c], kCntOne); }
        if (v2) { old1 = atomicAdd(&freq[base + (int)i1], kCntOne); }
        if (v3) { old2 = atomicAdd(&freq[base + (int)i2], kCntOne); }
        level<kWantCode>(freq, c, old0, kids1, v1, at);
        level<kWantCode>(freq, p1, old1, kids2, v2, at + 1);
        level<kWantCode>(freq, p2, old2, (uint32_t)(w3 >> 32), v3, at + 2);
        live = v3; c = i3; w = w3;
    }

    // sweep 2: three levels
    __device__ __forceinline__ void add3(const uint64_t* link, uint32_t* freq, uint32_t delta) {
        const uint32_t p1 = (uint32_t)w & 0x3FFu, p2 = ((uint32_t)w >> 10) & 0x3FFu, p3 = ((uint32_t)w >> 20) & 0x3FFu;
        const bool v1 = live & (p1 != kNil), v2 = v1 & (p2 != kNil), v3 = v2 & (p3 != kNil);
        const uint32_t i3 = v3 ? p3 : root;
        const uint64_t w3 = link[base + (int)i3];
        if (v1) { atomicAdd(&freq[base + (int)c], delta); }         // not the root (see count3)
        if (v2) { atomicAdd(&freq[base + (int)p1], delta); }
        if (v3) { atomicAdd(&freq[base + (int)p2], delta); }
        live = v3; c = i3; w = w3;
    }
};

// Whole wave: lanes [0, m) offer their symbols (a, then b; unified leaf ids, -1 = none).
// Returns how many leading tokens were applied (0..m); codes/depths are valid for those.
template <bool kWantCode>
__device__ __forceinline__ int bump_lanes(uint64_t* link, uint32_t* freq, int lane, int m, int a, int b,
                                          uint64_t& code_a, int& depth_a, uint64_t& code_b, int& depth_b) {
    const bool take = lane < m;
    const int la = take ? a : -1, lb = take ? b : -1;
    LaneWalk wa, wb;
#ifdef SQZ_STATS
    uint64_t t0_ = __builtin_readcyclecounter();
#define BL_SEC(k) { const uint64_t n_ = __builtin_readcyclecounter(); if (blockIdx.x == 1 && threadIdx.x == 0) { g_st[k] += n_ - t0_; } t0_ = n_; }
#else
#define BL_SEC(k)
#endif
    wa.open(link, la);
    wb.open(link, lb);
    BL_SEC(0)
    for (int at = 0; __ballot(wa.live | wb.live) != 0; at += 3) {
        wa.template count3<kWantCode>(link, freq, at);
        if (__ballot(wb.live) != 0) { wb.template count3<kWantCode>(link, freq, at); }
#ifdef SQZ_STATS
        if (blockIdx.x == 1 && threadIdx.x == 0) { g_st[4] += 1; }
#endif
    }
    BL_SEC(1)
    code_a = wa.code; depth_a = wa.depth;
    code_b = wb.code; depth_b = wb.depth;
    const uint64_t bm = __ballot(wa.bad | wb.bad);
    const int ok = bm != 0 ? __builtin_ctzll(bm) : m;                    // tokens in front of the first bad lane
    const uint32_t delta = lane < ok ? (1u - kCntOne) : (0u - kCntOne);
    wa.open(link, la);
    wb.open(link, lb);
    BL_SEC(2)
    while (__ballot(wa.live | wb.live) != 0) {
        wa.add3(link, freq, delta);
        if (__ballot(wb.live) != 0) { wb.add3(link, freq, delta); }
    }
    BL_SEC(3)
    return ok < m ? ok : m;
}

__device__ __forceinline__ void bind(LitTree& lit, PosTree& pos, EntropyLds* s) {
    lit.link = s->lit_link; lit.freq = s->lit_freq; lit.scratch = &s->scratch; lit.lut = nullptr;
    pos.link = s->pos_link; pos.freq = s->pos_freq; pos.scratch = &s->scratch; pos.lut = nullptr;
}

} // namespace sqzk
