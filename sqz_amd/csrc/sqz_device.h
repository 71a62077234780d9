// sqz_amd/csrc/sqz_device.h -- device-side building blocks shared by the
// entropy-stage kernels (huffman_emit.hip, decode.hip), gfx950 only.
//
// Reference semantics restated here (file:line relative to
// /root/reference/attic/map_experiment):
//   huffman.h:13-34   node / tree          -> Tree<> (LDS resident, compact ids)
//   huffman.h:41-62   huffman_update_paths -> Tree::relabel
//   huffman.h:64-86   huffman_swap_siblings-> Tree::order_pair
//   huffman.h:90-96   huffman_update_freq  -> Tree::sum
//   huffman.h:98-147  move_up / frequency_changed -> Tree::changed
//   huffman.h:149-216 huffman_insert       -> Tree::insert
//   huffman.h:218-235 huffman_inc_frequency-> Tree::bump
//   squeeze.h:29-79,151-172 DEFLATE tables -> len_code()/pos_code() arithmetic
//   bitstream.h:28-63,112-114 bit packer   -> BitSink
//   bitstream.h:65-103 bit reader          -> BitSource
//
// Layout decisions:
//  * one wavefront owns one stream; the two trees live in LDS; the serial
//    update chain runs on lane 0, the other lanes help with bulk data moves.
//  * node ids are compacted: leaves keep their symbol value, internal nodes
//    are numbered upwards from LEAVES (root == LEAVES).  The reference numbers
//    internals downwards from 2n-2; no emitted bit depends on the numbering.
//  * a node stores its code in STREAM order (first branch = most significant
//    bit) instead of the reference's LSB-first `path`; the bit packer then
//    appends codes without reversing them.  Same bits on the wire.
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

// ---------------------------------------------------------------------------
// Bit sink: values go out LSB first, bits fill a 64-bit word from the top,
// words are stored big-endian (bitstream.h:30-43,55-63).
// The sink is driven wave-uniformly (every lane holds the same state, so the
// state lives in SGPRs); only lane 0 touches memory.
struct BitSink {
    uint8_t* out;       // global
    uint64_t capacity;
    uint64_t bytes;
    uint64_t acc;
    int      fill;
    int      error;
    bool     writer;    // lane 0

    __device__ __forceinline__ void word() {
        if (error != 0) { return; }
        if (capacity - bytes >= 8) {
            if (writer) { *reinterpret_cast<uint64_t*>(out + bytes) = __builtin_bswap64(acc); }
            bytes += 8;
        } else {                       // bitstream.h:36-43: byte by byte
            for (int k = 0; k < 8 && error == 0; k++) {
                if (bytes == capacity) { error = kE2BIG; }
                else {
                    if (writer) { out[bytes] = (uint8_t)(acc >> (56 - 8 * k)); }
                    bytes++;
                }
            }
        }
        acc = 0;
        fill = 0;
    }

    // append `nbits` (1..63) whose first-out bit is the MOST significant one
    __device__ __forceinline__ void put_msb(uint64_t v, int nbits) {
        if (error != 0) { return; }
        const int room = 64 - fill;
        if (nbits < room) {
            acc = (acc << nbits) | v;
            fill += nbits;
        } else {
            const int rest = nbits - room;             // 0..62
            acc = (fill == 0 ? 0 : (acc << room)) | (v >> rest);
            word();
            acc = v & ((1ULL << rest) - 1);
            fill = rest;
        }
    }

    // value LSB first (squeeze_write_bits, squeeze.h:231-237), nbits 1..32
    __device__ __forceinline__ void put_lsb(uint32_t v, int nbits) {
        put_msb((uint64_t)(__brev(v) >> (32 - nbits)), nbits);
    }

    __device__ __forceinline__ void flush() {          // bitstream.h:112-114
        if (fill > 0 && error == 0) { acc <<= (64 - fill); word(); }
    }
};

// Bit source over global memory (bitstream.h:65-93): words are big-endian,
// bits leave from the top.  `limit` = available bytes.
struct BitSource {
    const uint8_t* in;
    uint64_t limit;
    uint64_t pos;       // absolute bit position of the next bit
    uint64_t acc;
    int      left;
    int      error;

    __device__ __forceinline__ void seek(uint64_t bitpos) {
        pos = bitpos; left = 0; acc = 0;
        const int skip = (int)(bitpos & 63);
        if (skip != 0) { refill_word(bitpos >> 6); acc <<= skip; left = 64 - skip; }
    }

    __device__ __forceinline__ void refill_word(uint64_t w) {
        const uint64_t byte = w * 8;
        if (byte + 8 <= limit) {
            acc = __builtin_bswap64(*reinterpret_cast<const uint64_t*>(in + byte));
        } else {                      // bitstream.h:72-80: short tail is an error
            error = kE2BIG;
            acc = 0;
        }
        left = 64;
    }

    __device__ __forceinline__ int bit() {
        if (error != 0) { return 0; }
        if (left == 0) { refill_word(pos >> 6); if (error != 0) { return 0; } }
        const int b = (int)(acc >> 63);
        acc <<= 1; left--; pos++;
        return b;
    }

    __device__ __forceinline__ uint32_t get_lsb(int nbits) { // bitstream.h:95-103
        uint32_t v = 0;
        for (int b = 0; b < nbits && error == 0; b++) { v |= (uint32_t)bit() << b; }
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
struct __attribute__((aligned(8))) Links {
    uint16_t up, lo, hi, bits;
};
constexpr uint16_t kNil = 0xFFFFu;

constexpr int kStack = 320;   // >= deepest possible tree (<= 286 leaves)
constexpr int kMaxFastDepth = 60;

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

// one lane's share of a root path: its node, that node's parent and grandparent
struct Chain {
    int mine, par, gpar;
    int levels;        // uniform: edges between leaf and root
    bool holds;        // this lane holds a node of the path
    bool active;       // ... and the node has a parent
    bool has_g;        // ... and that parent is not the root
};

// REF_LEAVES is the reference's leaf count n (512 / 32, squeeze.h:204-205): it
// only fixes how many leaf splits huffman_insert allows (n - 2, huffman.h:180).
template <int LEAVES, int NODES, int REF_LEAVES>
struct Tree {
    // LDS storage
    Links*    link;
    uint32_t* freq;
    TreeScratch* scratch;
    // wave-uniform registers
    int next;           // next free internal id
    int depth;          // huffman.h:26 high-water mark
    int complete;       // huffman.h:27
    int fault;          // stack overflow guard (never set for valid alphabets)

    static constexpr int kRoot = LEAVES;
    static constexpr int kIdEnd =
        (LEAVES + 1 + REF_LEAVES - 2) < NODES ? (LEAVES + 1 + REF_LEAVES - 2) : NODES;

    // all lanes: clear storage (huffman.h:251-269)
    __device__ __forceinline__ void init_all(int lane) {
        for (int i = lane; i < NODES; i += kWave) {
            link[i] = Links{kNil, kNil, kNil, 0};
            freq[i] = 0;
        }
        next = kRoot + 1; depth = 0; complete = 0; fault = 0;
    }

    __device__ __forceinline__ Links ld(int i) const {
        return link[i];    // 8-byte aligned aggregate: one ds_read_b64
    }

    // after a lane-0 section: make the registers uniform again
    __device__ __forceinline__ void sync_regs() {
        next = __builtin_amdgcn_readfirstlane(next);
        depth = __builtin_amdgcn_readfirstlane(depth);
        complete = __builtin_amdgcn_readfirstlane(complete);
        fault = __builtin_amdgcn_readfirstlane(fault);
    }

    // ---------------- slow path: the reference sequence, one lane -----------
    // huffman.h:41-62 (depths only: codes are read off the chain when needed)
    __device__ __forceinline__ void relabel(int top) {
        if (top == kRoot) { depth = 0; }
        int sp = 0;
        scratch->walk[sp++] = (uint16_t)top;
        while (sp > 0) {
            const int v = scratch->walk[--sp];
            const Links n = ld(v);
            const int b = n.bits;
            if (b > depth) { depth = b; }
            if (n.hi != kNil) {
                link[n.hi].bits = (uint16_t)(b + 1);
                if (n.hi >= LEAVES) {
                    if (sp < kStack) { scratch->walk[sp++] = n.hi; } else { fault = 1; }
                } else if (b + 1 > depth) { depth = b + 1; }
            }
            if (n.lo != kNil) {
                link[n.lo].bits = (uint16_t)(b + 1);
                if (n.lo >= LEAVES) {
                    if (sp < kStack) { scratch->walk[sp++] = n.lo; } else { fault = 1; }
                } else if (b + 1 > depth) { depth = b + 1; }
            }
        }
    }

    // huffman.h:90-96
    __device__ __forceinline__ void sum(int i) {
        const Links n = ld(i);
        const uint32_t a = n.lo != kNil ? freq[n.lo] : 0u;
        const uint32_t b = n.hi != kNil ? freq[n.hi] : 0u;
        freq[i] = a + b;
    }

    // huffman.h:64-86
    __device__ __forceinline__ int order_pair(int i) {
        const int p = link[i].up;
        if (p == kNil) { return i; }
        const Links n = ld(p);
        if (n.lo != kNil && n.hi != kNil && freq[n.lo] > freq[n.hi]) {
            link[p].lo = n.hi;
            link[p].hi = n.lo;
            relabel(p);
            return i == n.lo ? n.hi : n.lo;
        }
        return i;
    }

    // climb of huffman_frequency_changed (huffman.h:132-142): refresh sums and
    // sibling order up to the root, remembering (parent, child) per level
    __device__ __forceinline__ int climb(int i, int sp) {
        for (;;) {
            const int p = link[i].up;
            if (p == kNil) { sum(i); break; }
            sum(p);
            i = order_pair(i);
            if (sp < kStack) { scratch->pend[sp++] = ((uint32_t)p << 16) | (uint32_t)i; }
            else { fault = 1; }
            i = p;
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
            const Links np = ld(p);
            if (np.up == kNil || np.hi != c) { continue; }          // :143
            const int par = link[c].up;
            const int g = link[par].up;
            const Links ng = ld(g);
            const bool par_is_left = (ng.lo == par);
            const int uncle = par_is_left ? ng.hi : ng.lo;
            if (!(freq[c] > freq[uncle])) { continue; }              // :108
            link[c].up = (uint16_t)g;
            if (par_is_left) { link[g].hi = (uint16_t)c; } else { link[g].lo = (uint16_t)c; }
            link[par].hi = (uint16_t)uncle;
            link[uncle].up = (uint16_t)par;
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
    __device__ __forceinline__ bool insert(int i) {
        bool ok = true;
        int at = kRoot;
        freq[i] = 1;
        while (at >= LEAVES) {                                        // :156-170
            const Links n = ld(at);
            if (n.hi == kNil) { link[at].hi = (uint16_t)i; link[i].up = (uint16_t)at; break; }
            if (n.lo == kNil) { link[at].lo = (uint16_t)i; link[i].up = (uint16_t)at; break; }
            at = n.lo;
        }
        if (at >= LEAVES) {                                           // :171-173
            freq[at] += 1;
            i = order_pair(i);
        } else if (next >= kIdEnd) {                                  // :180-182
            ok = false;
            complete = 1;
        } else {                                                      // :184-209
            const int fresh = next++;
            const Links na = ld(at);
            link[fresh] = Links{na.up, (uint16_t)at, (uint16_t)i, na.bits};
            freq[fresh] = freq[at];
            if (na.up != kNil) {
                if (link[na.up].lo == at) { link[na.up].lo = (uint16_t)fresh; }
                else                      { link[na.up].hi = (uint16_t)fresh; }
            }
            link[at].up = (uint16_t)fresh;
            link[at].bits = (uint16_t)(na.bits + 1);
            link[i].up = (uint16_t)fresh;
            link[i].bits = (uint16_t)(na.bits + 1);
            sum(fresh);
            at = fresh;
        }
        changed(i);                                                   // :212
        relabel(at);                                                  // :213
        return ok;
    }

    // whole wave: insert through lane 0 (unseen symbols are rare: <= 286 per stream)
    __device__ __forceinline__ bool insert_wave(int i, int lane) {
        int ok = 1;
        if (lane == 0) { ok = insert(i) ? 1 : 0; }
        sync_regs();
        return __builtin_amdgcn_readfirstlane(ok) != 0;
    }

    // ---------------- fast path ------------------------------------------------
    // leaf -> root walk; lane k receives level k (0 = the leaf).  Uniform.
    __device__ __forceinline__ Chain chain_up(int s, int lane) const {
        int mine = kNil;
        int a = s, k = 0;
        for (;;) {
            mine = (lane == k) ? a : mine;
            const int up = __builtin_amdgcn_readfirstlane((int)link[a].up);
            if (up == kNil || k >= kMaxFastDepth) { break; }
            a = up;
            k++;
        }
        Chain c;
        c.mine = mine;
        c.par = lane_above(mine);            // lane k+1 holds the parent
        c.gpar = lane_above(c.par);
        c.levels = k;
        c.holds = lane <= k;
        c.active = lane < k;
        c.has_g = lane + 1 < k;
        return c;
    }

    // huffman_inc_frequency for an ATTACHED leaf s whose chain is `c`
    __device__ __forceinline__ void bump_wave(int s, const Chain& c, int lane) {
        if (complete != 0 || depth >= 63) { complete = 1; return; }   // huffman.h:228-234
        bool flag = false;
        uint32_t fc = 0;
        if (c.levels < kMaxFastDepth) {
            if (c.holds) { fc = freq[c.mine]; }
            if (c.active) {
                const Links lp = ld(c.par);
                const bool is_hi = (lp.hi == c.mine);
                const int sib = is_hi ? lp.lo : lp.hi;
                if (sib != kNil) {
                    const uint32_t fs = freq[sib];
                    flag = is_hi ? (fs > fc + 1) : (fc + 1 > fs);
                }
                if (is_hi && c.has_g) {
                    const Links lg = ld(c.gpar);
                    const int uncle = (lg.lo == c.par) ? lg.hi : lg.lo;
                    flag = flag || (fc + 1 > freq[uncle]);
                }
            }
        } else {
            flag = true;                                              // too deep: slow path
        }
        if (__ballot(flag) == 0) {
            if (c.holds) { freq[c.mine] = fc + 1; }
        } else {
            if (lane == 0) { freq[s] += 1; changed(s); }
            sync_regs();
        }
    }
};

constexpr int kLitLeaves = 288;               // symbols 0..285 (+2 pad)
constexpr int kLitNodes  = kLitLeaves + 288;  // root + <=285 splits (+pad)
constexpr int kPosLeaves = 32;
constexpr int kPosNodes  = 64;

using LitTree = Tree<kLitLeaves, kLitNodes, 512>;
using PosTree = Tree<kPosLeaves, kPosNodes, 32>;

// LDS image of one stream's entropy state
struct EntropyLds {
    Links    lit_link[kLitNodes];
    Links    pos_link[kPosNodes];
    uint32_t lit_freq[kLitNodes];
    uint32_t pos_freq[kPosNodes];
    TreeScratch scratch;
};

__device__ __forceinline__ void bind(LitTree& lit, PosTree& pos, EntropyLds* s) {
    lit.link = s->lit_link; lit.freq = s->lit_freq; lit.scratch = &s->scratch;
    pos.link = s->pos_link; pos.freq = s->pos_freq; pos.scratch = &s->scratch;
}

} // namespace sqzk
