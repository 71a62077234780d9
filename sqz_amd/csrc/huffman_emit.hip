// sqz_amd/csrc/huffman_emit.hip -- encode stage 2 (gfx950).
//
// One wavefront per stream.  Consumes the LZ77 token words of stage 1 and
// produces the reference's bit stream:
//   squeeze.h:278-288 squeeze_encode_literal   (NYT escape + 9 raw bits)
//   squeeze.h:290-298 squeeze_encode_len       (symbol 257+code, extra bits)
//   squeeze.h:300-315 squeeze_encode_pos       (pos tree, NYT + 5 raw bits)
//   squeeze.h:239-246 squeeze_write_huffman    (code from the tree BEFORE the
//                                               frequency update)
//   squeeze.h:248-253 squeeze_flush, bitstream.h:28-63 (MSB-first words)
// The wave runs uniformly.  Up to 64 tokens per step, one per lane: a lane looks its token's
// symbols up (leaf position, code, depth: three LDS reads per symbol, no chain walk) and takes
// part in the batched frequency update of sqz_tree.h (bump_batch); the tokens in front of the
// first one whose update could restructure a tree are applied and their bits -- code, extra bits,
// code, extra bits -- are packed in parallel (prefix sum of the widths, LDS atomic OR into a bit
// image) and leave as whole 8-byte words, coalesced.  The token a step stops at (a restructure,
// an unseen symbol with its NYT escape) takes the one-at-a-time path: lane k = level k of the
// chain, code = one __ballot("am I the hi child?"), the reference's own swap / promote tests,
// restructuring on the whole wave; its bits wait in a register and ride along with the next
// batch's pack.
#define SQZ_SEC_TIMERS            // (stats build: the section timers of sqz_tree.h report through this kernel)
#define SQZ_CHANGED_INLINING __forceinline__
#define SQZ_INSERT_INLINING __forceinline__
#include "sqz_tree.h"
#include "sqz_kernels.h"

namespace sqzk {

constexpr int kPendBits = 58;         // most bits the pending register hands to a pack

struct EmitLds {
    TreeLds  tree;                    // tree.P64 doubles as the bit image of a pack (stream order = MSB first)
    uint32_t code[kCodeSlots];
};


struct BitQueue {
    EmitLds* lds;
    uint8_t* out;        // global
    uint64_t capacity;
    uint64_t bytes;      // bytes produced so far (as the reference counts them)
    uint64_t carry_v;    // the bits of the last partial word, left-aligned (what image[0] starts as)
    int      carry;      // how many (0..63)
    int      error;
    // bits of the one-at-a-time path waiting for the next pack: the low pend_n bits of pend_v,
    // first-out bit on top.  They ride along with the next batch instead of costing a pack of
    // their own (most steps end with one such token).
    uint64_t pend_v;
    int      pend_n;

    // store 8-byte words [0, words) of the image (bitstream.h:33-43)
    __device__ __forceinline__ void store_words(int words, int lane) {
        if (error != 0 || words == 0) { return; }
        const uint64_t room = capacity - bytes;
        for (int k = lane; k < words; k += kWave) {
            const uint64_t w = lds->tree.P64[k];
            const uint64_t at = (uint64_t)k * 8;
            if (at + 8 <= room) {
                *reinterpret_cast<uint64_t*>(out + bytes + at) = __builtin_bswap64(w);
            } else {
                for (int j = 0; j < 8; j++) {              // byte by byte up to the capacity
                    if (at + (uint64_t)j < room) { out[bytes + at + j] = (uint8_t)(w >> (56 - 8 * j)); }
                }
            }
        }
        if ((uint64_t)words * 8 <= room) { bytes += (uint64_t)words * 8; }
        else { bytes = capacity; error = kE2BIG; }
    }

    // n bits of v (first-out bit = most significant of the n) at stream bit o of the image
    __device__ __forceinline__ void deposit(uint64_t v, uint32_t n, uint32_t o) {
        const uint32_t w = o >> 6, s = o & 63u;
        if (s + n <= 64) {
            atomicOr(reinterpret_cast<unsigned long long*>(&lds->tree.P64[w]),
                     (unsigned long long)v << (64 - s - n));
        } else {
            const uint32_t r = s + n - 64;              // bits spilling into the next word
            atomicOr(reinterpret_cast<unsigned long long*>(&lds->tree.P64[w]),
                     (unsigned long long)(v >> r));
            atomicOr(reinterpret_cast<unsigned long long*>(&lds->tree.P64[w + 1]),
                     (unsigned long long)v << (64 - r));
        }
    }

    // the pending bits, then every lane's two fields (n1 then n2 bits, each 0..38, first-out bit =
    // most significant): prefix sum of the widths, LDS atomic OR into the image behind the carried
    // bits, full words leave for HBM
    __device__ __forceinline__ void pack_lanes(uint64_t v1, uint32_t n1, uint64_t v2, uint32_t n2, int lane) {
        for (int k = lane; k < kImageWords; k += kWave) { lds->tree.P64[k] = (k == 0) ? carry_v : 0ull; }
        lds_fence();
        const uint32_t incl = wave_scan(n1 + n2);
        const uint32_t head = (uint32_t)carry + (uint32_t)pend_n;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1) + head;
        if (pend_n != 0 && lane == 0) { deposit(pend_v, (uint32_t)pend_n, (uint32_t)carry); }
        const uint32_t at = head + incl - (n1 + n2);
        if (n1 != 0) { deposit(v1, n1, at); }
        if (n2 != 0) { deposit(v2, n2, at + n1); }
        lds_fence();
        const int words = (int)(total >> 6);
        store_words(words, lane);
        carry_v = uni64(lds->tree.P64[words]);             // partial word becomes the new carry
        lds_fence();
        carry = (int)(total & 63u);
        pend_v = 0; pend_n = 0;
    }

    // what is pending, alone (the register is full, or the stream ends)
    __device__ __forceinline__ void pack(int lane) {
        if (pend_n != 0) { pack_lanes(0ull, 0u, 0ull, 0u, lane); }
    }

    // value's low `width` bits, first-out bit = most significant; width 1..32
    __device__ __forceinline__ void push32(uint32_t value, int width, int lane) {
        if (pend_n + width > kPendBits) { pack(lane); }
        const uint64_t bits = width >= 32 ? (uint64_t)value : (uint64_t)(value & ((1u << width) - 1u));
        pend_v = (pend_v << width) | bits;
        pend_n += width;
    }
    __device__ __forceinline__ void push(uint64_t value, int width, int lane) {
        if (width > 32) {
            push32((uint32_t)(value >> 32), width - 32, lane);
            push32((uint32_t)value, 32, lane);
        } else {
            push32((uint32_t)value, width, lane);
        }
    }
    // value LSB first (squeeze_write_bits, squeeze.h:231-237), width 1..32
    __device__ __forceinline__ void push_lsb(uint32_t value, int width, int lane) {
        push32(__brev(value) >> (32 - width), width, lane);
    }

    __device__ __forceinline__ void flush(int lane) {       // bitstream.h:112-114
        pack(lane);
        if (carry > 0) {
            if (lane == 0) { lds->tree.P64[0] = carry_v; }
            lds_fence();
            store_words(1, lane);
            carry = 0; carry_v = 0;
        }
    }
};

// the code of a chain longer than the wave's ballot covers, by a walk (trees this deep need ~2^60 symbols)
__device__ __noinline__ uint64_t deep_code(const uint32_t* lnk, int leaf, int& width) {
    uint64_t code = 0;
    int n = 0, a = leaf;
    for (;;) {
        const uint32_t up = (uint32_t)__builtin_amdgcn_readfirstlane((int)l_up(lnk[a]));
        if (up == kNil || n >= 63) { break; }
        const uint32_t pw = (uint32_t)__builtin_amdgcn_readfirstlane((int)lnk[up]);
        code |= (uint64_t)(l_hi(pw) == (uint32_t)a ? 1 : 0) << n;
        a = (int)up;
        n++;
    }
    width = n;
    return code;
}

// squeeze.h:278-288 / :300-315: symbol s (a leaf id) through tree t with the NYT escape
template <class T>
__device__ __forceinline__ void emit_coded(BitQueue& q, T& t, int s, int nyt, int raw_bits,
                                           int lane, int& err) {
    Chain c = t.chain_up(s, lane);
    const bool unseen = (c.levels == 0);                    // node[s].bits == 0
    if (unseen) { c = t.chain_up(nyt, lane); }
    const int leaf = unseen ? nyt : s;
    // the code of the tree BEFORE the update (squeeze.h:239-246): the ballot of the chain's hi/lo
    // choices, or, for a chain longer than the wave holds, a walk -- made before the tree moves
    int width = c.levels;                                   // ballot bits beyond `levels` are 0
    uint64_t deep = 0;
    if (width >= kMaxFastDepth) { deep = deep_code(t.lds->lnk, leaf, width); }
    uint64_t code = t.bump_wave(leaf, c, lane);
    if (c.levels >= kMaxFastDepth) { code = deep; }
    q.push(code, width, lane);
    if (unseen) {
        q.push_lsb((uint32_t)(s - T::kBase), raw_bits, lane);
        if (!t.insert_wave(s, lane)) { err = kE2BIG; }
    }
}

// one token word -> its bit fields (squeeze.h:377-394 -> :278-315), one symbol at a time
template <class LitTree, class PosTree>
__device__ __forceinline__ void emit_token(BitQueue& q, LitTree& lit, PosTree& pos, uint32_t t,
                                           int lane, int& err) {
    const bool is_match = (t & kTokMatch) != 0;
    Code lc = {0, 0, 0};
    int s_lit = (int)(t & 0xFFu);
    if (is_match) {
        lc = len_code((int)((t >> 16) & 0x1FFu));                     // squeeze.h:290-298
        s_lit = kSymLen0 + lc.code;
    }
    emit_coded(q, lit, s_lit, kLitNyt, 9, lane, err);
    if (is_match) {
        if (lc.xbits > 0) { q.push_lsb((uint32_t)lc.extra, lc.xbits, lane); }
        const Code pc = pos_code((int)(t & 0x7FFFu));                 // squeeze.h:300-315
        emit_coded(q, pos, kPosBase + pc.code, kPosBase + kPosNyt, 5, lane, err);
        if (pc.xbits > 0) { q.push_lsb((uint32_t)pc.extra, pc.xbits, lane); }
    }
    if (q.error != 0) { err = q.error; }
    if (lit.fault | pos.fault) { err = kE2BIG; }
}

// A token word as stage 1 writes it: a byte, or 0x80000000 | len << 16 | dist with len 3..257
// and dist 1..32767.  The token arrays of sqz_hip_huffman_blocks come from the caller, and a
// length or distance outside the code tables would index past the trees.
__device__ __forceinline__ bool token_ok(uint32_t t) {
    if ((t & kTokMatch) == 0) { return t <= 0xFFu; }
    const uint32_t len = (t >> 16) & 0x7FFFu, dist = t & 0xFFFFu;
    return len >= (uint32_t)kLenMin && len <= (uint32_t)kLenMax && dist >= 1u && dist <= 0x7FFFu;
}

// Tokens of stage 1 -> bit stream.  Up to 64 tokens per step, one per lane (sqz_tree.h:
// bump_batch); a step shrinks to the tokens in front of the first unseen or malformed one, then to
// the longest prefix that changes no link, and the token it stops at (the tree really
// restructures, or it needs the NYT escape) goes through the one-at-a-time path in the same step.
template <bool kStats>                         // kStats: keep the reference's counters (sqz_block_stats)
// four waves per SIMD = 16 streams per CU, as many as the LDS allows; the counters build is a diagnostic
// path and takes the registers it needs instead (bounded to 128 it spills 100 of them)
__global__ __launch_bounds__(kWave, kStats ? 1 : 4)
void huffman_emit_kernel(const uint32_t* __restrict__ tokens,
                         const uint64_t* __restrict__ tok_off,
                         const uint32_t* __restrict__ tok_count,
                         uint8_t* __restrict__ out,
                         const uint64_t* __restrict__ out_off,
                         uint64_t* __restrict__ out_bytes,
                         int32_t* __restrict__ err_out,
                         uint32_t n_blocks,
                         uint64_t prefix_acc, int prefix_fill,
                         sqz_block_stats* __restrict__ stats_out) {
    __shared__ EmitLds lds;
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) { return; }

    using LitTree = LitTreeT<true, kStats>;
    using PosTree = PosTreeT<true, kStats>;
    LitTree lit; PosTree pos;
    lit.lds = &lds.tree; lit.code = lds.code; lit.lut = nullptr;
    pos.lds = &lds.tree; pos.code = lds.code; pos.lut = nullptr;
    lit.init_all(lane);
    pos.init_all(lane);
    for (int k = lane; k < kCodeSlots; k += kWave) { lds.code[k] = 0; }
    __syncthreads();

    const uint32_t* tok = tokens + uni64(tok_off[b]);
    uint32_t count = (uint32_t)__builtin_amdgcn_readfirstlane((int)tok_count[b]);
    const bool too_many = (uint64_t)count > uni64(tok_off[b + 1]) - uni64(tok_off[b]);   // one slot per byte
    if (too_many) { count = 0; }

    BitQueue q;
    q.lds = &lds;
    const uint64_t o0 = uni64(out_off[b]), o1 = uni64(out_off[b + 1]);
    q.out = out + o0;
    q.capacity = o1 - o0;
    q.bytes = 0;
    q.pend_v = 0;
    q.pend_n = 0;
    // header bits that precede the payload (single-stream API): the carry
    q.carry = prefix_fill;
    q.carry_v = prefix_fill > 0 ? (prefix_acc << (64 - prefix_fill)) : 0ull;
    q.error = 0;
    int err = 0;

    if (too_many) { err = kEINVAL; }
    if (!lit.insert_wave(kLitNyt, lane)) { err = kEINVAL; }          // squeeze.h:333-334
    if (!pos.insert_wave(kPosBase + kPosNyt, lane)) { err = kEINVAL; }

    uint32_t cursor = 0;
    uint32_t lit_tokens = 0, match_bytes = 0;         // squeeze.h:327-328 li_bytes / br_bytes
    // The tokens [cursor, cursor + 128) sit in two registers per lane.  After a step the window is
    // shifted by what the step consumed (two ds_bpermute) and its second half reloaded from memory:
    // that load is in flight during the whole next step instead of being waited for.
    uint32_t win0 = (uint32_t)lane < count ? tok[lane] : 0u;
    uint32_t win1 = (uint32_t)lane + (uint32_t)kWave < count ? tok[kWave + lane] : 0u;
#ifdef SQZ_STATS
    uint64_t es[4] = {0, 0, 0, 0}, es_last = __builtin_readcyclecounter(), es_begin = es_last;
    uint32_t es_steps = 0, es_exact = 0;
#define ES(k) { const uint64_t n_ = __builtin_readcyclecounter(); es[k] += n_ - es_last; es_last = n_; }
#else
#define ES(k)
#endif
    // streams that are ahead step back (see entropy_decode_kernel): priority 3 -> 0 by quarters
    const uint32_t quarter = (count >> 2) + 1;
    int prio_now = -1;
    int want = 16, avg4 = 32;                          // (avg4 = 4 x the mean length of the recent steps)
    uint32_t regain_at = 0;                            // next look at a tree that has given up its positions
    while (cursor < count && err == 0) {
        {
            const int qq = (int)(cursor / quarter);             // 0..3
            if (qq != prio_now) {
                prio_now = qq;
                if (qq == 0) { __builtin_amdgcn_s_setprio(3); }
                else if (qq == 1) { __builtin_amdgcn_s_setprio(2); }
                else if (qq == 2) { __builtin_amdgcn_s_setprio(1); }
                else { __builtin_amdgcn_s_setprio(0); }
            }
        }
        // ---- this lane's token ------------------------------------------------------
        const uint32_t step_start = cursor;
        const bool valid = cursor + (uint32_t)lane < count;
        const uint32_t traw = win0;                                    // 0 past the end
        const bool wellformed = token_ok(traw);
        const uint32_t t = wellformed ? traw : 0u;                     // a malformed word is never decoded
        const bool is_match = (t & kTokMatch) != 0;
        Code lc = {0, 0, 0}, pc = {0, 0, 0};
        int a = (int)(t & 0xFFu), bsym = -1;
        uint32_t tlen = 1;
        if (is_match) {
            tlen = (t >> 16) & 0x1FFu;
            lc = len_code((int)tlen);                                  // squeeze.h:290-298
            pc = pos_code((int)(t & 0x7FFFu));                         // squeeze.h:300-315
            a = kSymLen0 + lc.code;
            bsym = kPosBase + pc.code;
        }
        // unseen symbols need the NYT escape + an insert: they end the step
        const bool new_a = valid && l_up(lds.tree.lnk[a]) == kNil;
        const bool new_b = valid && is_match && l_up(lds.tree.lnk[bsym]) == kNil;
        const uint64_t unseen = __ballot(valid && (new_a || new_b || !wellformed));
        const uint64_t vmask = __ballot(valid);
        int m = unseen != 0 ? __builtin_ctzll(unseen) : __builtin_popcountll(vmask);
        // While the trees are still forming a step ends at a restructure every ten tokens or so: offering all
        // 64 then means a histogram over tokens that will not be applied and a dozen nodes failing their test,
        // each of them to be located.  The offer follows the recent step lengths (1.5 x their mean and a
        // little, the decoder's read-ahead policy); the output does not depend on it.
        m = m < want ? m : want;
        if (cursor > kBatchTokens && (lit.aux | pos.aux) != 0) {  // counts are about to outgrow their 24 bits
            lit.give_up_aux(lane);
            pos.give_up_aux(lane);
        }
        if ((lit.aux & pos.aux) == 0 && cursor <= kBatchTokens && cursor >= regain_at) {
            // a tree that was too deep for the position machinery may have settled (sqz_tree.h: regain_aux)
            regain_at = cursor + 512u;
            (void)lit.regain_aux(lane);
            (void)pos.regain_aux(lane);
        }
        const bool frozen = (lit.complete | pos.complete) != 0 || lit.depth >= kFreezeDepth || pos.depth >= kFreezeDepth ||
                            (lit.aux & pos.aux) == 0;
        uint32_t ca = 0, cb = 0;
        int wa = 0, wb = 0;
        const int offered = m;
        ES(0)
        if (m >= 1 && !frozen) { m = bump_batch<true>(&lds.tree, lds.code, lit, pos, lane, m, a, bsym, ca, wa, cb, wb); }
        else { m = 0; }
        ES(1)
        if (m >= 1) {
            // this lane's bits: code [extra] | code extra, first-out bit on top
            uint64_t v1 = ca, v2 = 0;
            uint32_t n1 = (uint32_t)wa, n2 = 0;
            if (is_match) {
                v1 = (v1 << lc.xbits) | (uint64_t)(lc.xbits ? (__brev((uint32_t)lc.extra) >> (32 - lc.xbits)) : 0u);
                n1 += (uint32_t)lc.xbits;
                v2 = ((uint64_t)cb << pc.xbits) | (uint64_t)(pc.xbits ? (__brev((uint32_t)pc.extra) >> (32 - pc.xbits)) : 0u);
                n2 = (uint32_t)(wb + pc.xbits);
            }
            if (lane >= m) { n1 = 0; n2 = 0; }
            q.pack_lanes(v1, n1, v2, n2, lane);                        // what the last one-at-a-time token left goes first
            if (kStats) {                                              // squeeze.h:386,391: source bytes coded as back references / as literals
                const uint64_t mm = __ballot(lane < m && is_match);
                lit_tokens += (uint32_t)(m - __builtin_popcountll(mm));
                match_bytes += wave_sum((lane < m && is_match) ? tlen : 0u);
            }
            cursor += (uint32_t)m;
        }
        ES(2)
        if (m < offered || offered == 0 || frozen) {
            // the token the step stopped at (NYT escape / the tree restructures / frozen tree):
            // one symbol at a time, always exact
            const uint32_t tx = (uint32_t)__builtin_amdgcn_readlane((int)traw, m);
            if (!token_ok(tx)) { err = kEINVAL; break; }
            emit_token(q, lit, pos, tx, lane, err);
            if (kStats) { if ((tx & kTokMatch) != 0) { match_bytes += (tx >> 16) & 0x1FFu; } else { lit_tokens += 1; } }
            cursor += 1;
#ifdef SQZ_STATS
            es_exact++;
#endif
        }
        ES(3)
#ifdef SQZ_EMU_DEBUG_BATCH
        if (lane == 0) {
            static unsigned dbg_steps = 0, dbg_offered = 0, dbg_applied = 0, dbg_exact = 0, dbg_next = 16384;
            dbg_steps++; dbg_offered += (unsigned)offered; dbg_applied += (unsigned)m; dbg_exact += (m < offered || offered == 0 || frozen) ? 1u : 0u;
            if (cursor >= dbg_next || cursor >= count) {
                fprintf(stderr, "cursor %u: steps %u offered %u applied %u exact %u (want %d) depth %d/%d aux %d/%d\n", cursor, dbg_steps, dbg_offered, dbg_applied, dbg_exact, want,
                        lit.depth, pos.depth, lit.aux, pos.aux);
                dbg_steps = dbg_offered = dbg_applied = dbg_exact = 0; dbg_next = cursor + 16384;
            }
        }
#endif
#ifdef SQZ_STATS
        es_steps++;
#endif
        {
            avg4 += (int)(cursor - step_start) - (avg4 >> 2);
            want = ((avg4 * 3) >> 3) + 6;
            want = want < kWave ? want : kWave;
        }
        {   // slide the window by the 1..64 tokens this step consumed
            const uint32_t adv = cursor - step_start;
            const int from = ((lane + (int)adv) & (kWave - 1)) * 4;
            const uint32_t s0 = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)win0);
            const uint32_t s1 = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)win1);
            win0 = (uint32_t)lane + adv < (uint32_t)kWave ? s0 : s1;
            const uint32_t t1 = cursor + (uint32_t)kWave + (uint32_t)lane;
            win1 = t1 < count ? tok[t1] : 0u;
        }
        if (q.error != 0) { err = q.error; }
        if (lit.fault | pos.fault) { err = kE2BIG; }
    }

#ifdef SQZ_STATS
    if (lane == 0 && b == 1) {
        printf("emit block 1: cycles %llu steps %u exact %u: prep %llu bump %llu pack %llu exact %llu\n",
               (unsigned long long)(es_last - es_begin), es_steps, es_exact, (unsigned long long)es[0],
               (unsigned long long)es[1], (unsigned long long)es[2], (unsigned long long)es[3]);
        for (int k = 0; k < 32; k++) { if (g_secn[k]) { printf("sec %d: %llu cycles / %u = %llu\n", k, g_sec[k], g_secn[k], g_sec[k] / g_secn[k]); } }
        printf("emit slow: lit insert %u/%llu changed %u/%llu; pos insert %u/%llu changed %u/%llu\n",
               lit.st_cnt[0], (unsigned long long)lit.st_cyc[0], lit.st_cnt[1], (unsigned long long)lit.st_cyc[1],
               pos.st_cnt[0], (unsigned long long)pos.st_cyc[0], pos.st_cnt[1], (unsigned long long)pos.st_cyc[1]);
    }
#endif
    if (err == 0) { q.flush(lane); err = q.error; }
    else { q.pack(lane); }
    if (lane == 0) {
        out_bytes[b] = q.bytes;
        err_out[b] = err;
        if (kStats && stats_out != nullptr) {                     // (field by field: no 1.3 KB temporary on the stack)
            sqz_block_stats* st = stats_out + b;
            st->lit_updates = lit.stats.updates; st->lit_swaps = lit.stats.swaps; st->lit_moves = lit.stats.moves;
            st->pos_updates = pos.stats.updates; st->pos_swaps = pos.stats.swaps; st->pos_moves = pos.stats.moves;
            st->literal_bytes = lit_tokens; st->backref_bytes = match_bytes;
            st->lit_depth = (uint32_t)lit.depth; st->pos_depth = (uint32_t)pos.depth;
            st->tokens = cursor; st->reserved = 0;
        }
    }
    // leaf counts for huffman_entropy (huffman.h:237-249; the host does the logarithms)
    if (kStats && stats_out != nullptr) {
        for (int k = lane; k < kLitLeaves; k += kWave) { stats_out[b].lit_freq[k] = lit.freq(k); }
        if (lane < kPosLeaves) { stats_out[b].pos_freq[lane] = pos.freq(kPosBase + lane); }
    }
}

// ---------------------------------------------------------------------------------------------
// Debug / test entry: drive ONE tree with a symbol sequence exactly as the kernels above drive
// it (batches of attached symbols through bump_batch, everything else one at a time through
// chain_up / bump_wave / insert_wave) and dump its LDS image, so that the reference's tree dumps
// (tests/golden/trees.npz, huffman.h node arrays) pin the device tree directly.
// dump layout (uint32): [0..7] next, depth mark, complete, aux, fault, updates, swaps, moves;
// then per node id v of the tree (NODES of them): lnk, rng, cnt, code (leaves; 0 otherwise).
using DbgLit = LitTreeT<true, true>;
using DbgPos = PosTreeT<true, true>;
template <class T>
__device__ __forceinline__ void debug_drive(T& t, DbgLit& lit, DbgPos& pos, EmitLds& lds, const int32_t* symbols,
                                            uint32_t count, int batch, int lane, bool is_pos, bool stop_early,
                                            uint32_t max_steps, uint32_t& consumed) {
    uint32_t k = 0;
    for (uint32_t step = 0; k < count && t.fault == 0 && (max_steps == 0 || step < max_steps); step++) {
        if (stop_early && t.aux == 0) { break; }               // (bisecting: do not crawl on after the positions are given up)
        int sym = -1;
        if (k + (uint32_t)lane < count) { sym = symbols[k + lane]; }
        const bool valid = sym >= 0 && sym < T::kLeaves && lane < batch;
        const int leaf = valid ? T::kBase + sym : T::kRoot;
        const uint64_t seen = __ballot(valid && l_up(lds.tree.lnk[leaf]) != kNil);
        int m = (~seen == 0ull) ? kWave : __builtin_ctzll(~seen);   // leading attached symbols (0..64)
        const bool frozen = t.complete != 0 || t.depth >= kFreezeDepth || t.aux == 0;
        uint32_t ca, cb; int wa, wb;
        if (m > 0 && !frozen) {
            m = is_pos ? bump_batch<true>(&lds.tree, lds.code, lit, pos, lane, m, -1, leaf, ca, wa, cb, wb)
                       : bump_batch<true>(&lds.tree, lds.code, lit, pos, lane, m, leaf, -1, ca, wa, cb, wb);
        } else { m = 0; }
        k += (uint32_t)m;
        if (k < count) {                                       // the symbol the batch stopped at, exactly
            const int s = __builtin_amdgcn_readfirstlane(symbols[k]);
            if (s < 0 || s >= T::kLeaves) { break; }
            const int lf = T::kBase + s;
            const Chain c = t.chain_up(lf, lane);
            if (c.levels == 0) { (void)t.insert_wave(lf, lane); }
            else { (void)t.bump_wave(lf, c, lane); }
            k++;
        }
    }
    consumed = k;
}

__global__ __launch_bounds__(kWave, 4)          // four waves per SIMD: 16 streams per CU (the LDS allows as many)
void tree_debug_kernel(const int32_t* __restrict__ symbols, uint32_t count, int which, int batch,
                       uint32_t* __restrict__ dump) {
    __shared__ EmitLds lds;
    const int lane = threadIdx.x;
    DbgLit lit; DbgPos pos;
    lit.lds = &lds.tree; lit.code = lds.code; lit.lut = nullptr;
    pos.lds = &lds.tree; pos.code = lds.code; pos.lut = nullptr;
    lit.init_all(lane);
    pos.init_all(lane);
    for (int k = lane; k < kCodeSlots; k += kWave) { lds.code[k] = 0; }
    __syncthreads();
    const bool stop_early = (batch & 0x100) != 0;
    batch &= 0xFF;
    if (batch < 1) { batch = 1; }
    if (batch > kWave) { batch = kWave; }
    const uint32_t max_steps = (uint32_t)which >> 8;       // development: stop after this many steps (0 = run to the end)
    which &= 1;
    uint32_t consumed = 0;
    if (which == 0) { debug_drive(lit, lit, pos, lds, symbols, count, batch, lane, false, stop_early, max_steps, consumed); }
    else { debug_drive(pos, lit, pos, lds, symbols, count, batch, lane, true, stop_early, max_steps, consumed); }
    lds_fence();
    const int base = which == 0 ? 0 : kPosBase, nodes = which == 0 ? kLitNodes : kPosNodes;
    const int leaves = which == 0 ? kLitLeaves : kPosLeaves, pos0 = which == 0 ? 0 : kPosPos0;
    if (lane == 0) {
        const TreeStats st = which == 0 ? lit.stats : pos.stats;
        dump[0] = (uint32_t)(which == 0 ? lit.next : pos.next);
        dump[1] = (uint32_t)(which == 0 ? lit.depth : pos.depth);
        dump[2] = (uint32_t)(which == 0 ? lit.complete : pos.complete);
        dump[3] = (uint32_t)(which == 0 ? lit.aux : pos.aux);
        dump[4] = (uint32_t)(which == 0 ? lit.fault : pos.fault) | (max_steps != 0 ? consumed << 8 : 0u);
        dump[5] = st.updates; dump[6] = st.swaps; dump[7] = st.moves;
    }
    for (int v = lane; v < nodes; v += kWave) {
        dump[8 + 4 * v + 0] = lds.tree.lnk[base + v];
        dump[8 + 4 * v + 1] = lds.tree.rng[base + v];
        dump[8 + 4 * v + 2] = lds.tree.cnt[base + v];
        dump[8 + 4 * v + 3] = v < leaves ? lds.code[pos0 + v] : 0u;
    }
}

void launch_tree_debug(const int32_t* symbols, uint32_t count, int which, int batch, uint32_t* dump,
                       hipStream_t stream) {
    hipLaunchKernelGGL(tree_debug_kernel, dim3(1), dim3(kWave), 0, stream, symbols, count, which, batch, dump);
}

void launch_huffman_emit(const uint32_t* tokens, const uint64_t* tok_off,
                         const uint32_t* tok_count, uint8_t* out,
                         const uint64_t* out_off, uint64_t* out_bytes,
                         int32_t* err, uint32_t n_blocks,
                         uint64_t prefix_acc, int prefix_fill, sqz_block_stats* stats,
                         hipStream_t stream) {
    if (n_blocks == 0) { return; }
    if (stats != nullptr) {
        hipLaunchKernelGGL(huffman_emit_kernel<true>, dim3(n_blocks), dim3(kWave), 0, stream,
                           tokens, tok_off, tok_count, out, out_off, out_bytes, err, n_blocks,
                           prefix_acc, prefix_fill, stats);
    } else {
        hipLaunchKernelGGL(huffman_emit_kernel<false>, dim3(n_blocks), dim3(kWave), 0, stream,
                           tokens, tok_off, tok_count, out, out_off, out_bytes, err, n_blocks,
                           prefix_acc, prefix_fill, stats);
    }
}

} // namespace sqzk
