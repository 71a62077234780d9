// sqz_amd/csrc/decode.hip -- decoder (gfx950), two kernels.
//
//   entropy_decode_kernel   one wavefront per stream, running uniformly:
//       squeeze.h:502-551 squeeze_decompress   token loop (without the copy)
//       squeeze.h:429-442 squeeze_read_huffman root->leaf walk, then frequency bump
//       squeeze.h:458-474 squeeze_read_length
//       squeeze.h:476-500 squeeze_read_pos
//     With the trees held still, lane l decodes the token that would start at bit l of a
//     staged piece of the stream; the real starts are picked by following the lengths, and
//     up to 64 tokens then update the trees at once (sqz_tree.h: bump_batch).  The token
//     a step stops at (a restructure, an unseen symbol) takes the one-at-a-time path, where
//     the leaf->root chain hands level d of the path to lane d; what was read ahead behind it
//     is kept as far as the update left its codes alone.  Output: the same token words stage 1
//     of the encoder produces (literal / len<<16|dist).
//   lz_expand_kernel        one wavefront per stream: squeeze.h:521-539, 64 tokens per step.
//     The window is the output buffer itself, read back through the L2: literals land in
//     parallel, short back references are copied by their own lanes side by side, the rest by
//     all 64 lanes with the byte-serial overlap rule out[i+k] = out[i-dist + (k mod dist)]
//     (RLE when dist < len).
//
// Hardening (the reference only asserts): a missing
// child, a raw symbol that is out of range or already in the tree, a distance
// reaching before the stream start and a match running past the stream end
// are EINVAL; reading past the compressed bytes is E2BIG (bitstream.h:74).
#define SQZ_CHANGED_INLINING __forceinline__
#define SQZ_INSERT_INLINING __forceinline__
#define SQZ_LUT_INLINING __forceinline__
#include "sqz_tree.h"
#include "sqz_kernels.h"

namespace sqzk {

constexpr int kStageDw = 128;          // staged stream dwords (big-endian values): 4096 bits

struct DecodeLuts {
    uint16_t lit[1 << 8];
    uint16_t pos[1 << 6];
};

struct DecodeLds {
    TreeLds    tree;                    // tree.P32[0 .. 128) doubles as the read-ahead's token slots
                                        // (word | (bits used - 1) << 25): they are in registers
                                        // before the batched update needs the space
    DecodeLuts luts;
    uint32_t   stage[kStageDw];
};

using LitTree = LitTreeT<false, false>;
using PosTree = PosTreeT<false, false>;

// squeeze.h:429-442; returns the leaf or -1 with err set.  The first 8 levels
// of the root->leaf walk come from the lookup table (rebuilt whenever the tree
// restructured, ~1 % of the symbols), deeper leaves continue bit by bit; the
// leaf->root chain for the frequency update is chain_up()'s.
template <class T>
__device__ __forceinline__ int read_symbol(BitSource& r, T& t, int lane, int& err) {
    if (t.lut_ok == 0) { t.build_lut(lane); }
    r.fill();
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.lut[r.peek(T::kLutBits)]);
    int node = (int)(e & 0x3FFu);
    r.skip((int)(e >> 10));
    int d = (int)(e >> 10);
    while (node >= (int)T::kRoot && node != (int)kNil) {           // deeper than 8 levels
        r.fill();
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.lds->lnk[node]);
        node = (int)(r.peek(1) ? l_hi(w) : l_lo(w));
        r.skip(1);
        if (++d > 2 * kStack) { node = (int)kNil; }
    }
    if (r.overrun()) { err = kE2BIG; return -1; }                       // bitstream.h:74
    if (node == (int)kNil) { err = kEINVAL; return -1; }
    const Chain c = t.chain_up(node, lane);
    (void)t.bump_wave(node, c, lane);
    return node;
}

// one symbol through the lookup table WITHOUT touching the tree (the tree is static
// while a step is being read ahead); -1 = ran past the readable bits / missing child
template <class T>
__device__ __forceinline__ int peek_symbol(BitSource& r, const T& t) {
    r.fill();
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.lut[r.peek(T::kLutBits)]);
    int node = (int)(e & 0x3FFu);
    r.skip((int)(e >> 10));
    int d = (int)(e >> 10);
    while (node >= (int)T::kRoot && node != (int)kNil) {           // deeper than 8 levels
        r.fill();
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)t.lds->lnk[node]);
        node = (int)(r.peek(1) ? l_hi(w) : l_lo(w));
        r.skip(1);
        if (++d > 2 * kStack) { node = (int)kNil; }
    }
    if (r.overrun() || node == (int)kNil) { return -1; }
    return node;
}

// Up to 64 tokens per step (sqz_tree.h: bump_batch): the tokens are first read ahead
// with both trees held still -- valid as long as no link changes, which is exactly what
// bump_batch then establishes for a prefix of them; the token the step stops at takes the
// one-at-a-time updates (and, if it is an NYT escape or malformed input, is decoded again
// from its own bit position), so errors and updates are the reference's.
__global__ __launch_bounds__(kWave, 4)          // four waves per SIMD: 16 streams per CU (the LDS allows as many)
void entropy_decode_kernel(const uint8_t* __restrict__ in,
                           const uint64_t* __restrict__ in_off,
                           const uint64_t* __restrict__ out_off,
                           uint32_t* __restrict__ tokens,
                           uint32_t* __restrict__ tok_count,
                           int32_t* __restrict__ err_out,
                           uint64_t* __restrict__ end_bit,   // optional: bit position after the last symbol
                           uint32_t n_blocks,
                           uint64_t start_bit) {
    __shared__ DecodeLds lds;
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) { return; }

    LitTree lit; PosTree pos;
    lit.lds = &lds.tree; lit.code = nullptr; lit.lut = lds.luts.lit;
    pos.lds = &lds.tree; pos.code = nullptr; pos.lut = lds.luts.pos;
    lit.init_all(lane);
    pos.init_all(lane);
    __syncthreads();

    const uint64_t o0 = uni64(out_off[b]), o1 = uni64(out_off[b + 1]);
    const uint64_t bytes = o1 - o0;
    uint32_t* tok = tokens + o0;

    BitSource r;
    const uint64_t i0 = uni64(in_off[b]), i1 = uni64(in_off[b + 1]);
    const uint8_t* const src = in + i0;
    const uint64_t src_bytes = i1 - i0;
    r.open(src, src_bytes, start_bit, lane);
    int err = 0;
    if (!lit.insert_wave(kLitNyt, lane)) { err = kEINVAL; }          // squeeze.h:505-506
    if (!pos.insert_wave(kPosBase + kPosNyt, lane)) { err = kEINVAL; }

    const uint32_t* const lnk = lds.tree.lnk;          // both trees: absolute node ids
    uint32_t* const slot = lds.tree.P32;
    uint64_t i = 0;
    uint32_t ntok = 0;

    // squeeze.h:509-549 for exactly one token, with every check and the tree updates
    auto decode_one = [&]() {
        int s = read_symbol(r, lit, lane, err);
        if (err != 0) { return; }
        if (s == kLitNyt) {                                            // squeeze.h:512-520
            s = (int)r.get_lsb(9);
            if (r.overrun()) { err = kE2BIG; return; }
            if (s == 256 || s >= kLitNyt) { err = kEINVAL; return; }
            const uint32_t up = (uint32_t)__builtin_amdgcn_readfirstlane((int)lit.up_of(s));
            if (up != kNil) { err = kEINVAL; return; }
            if (!lit.insert_wave(s, lane)) { err = kE2BIG; return; }
        }
        uint32_t word;
        if (s <= 0xFF) {
            word = (uint32_t)s;
            i++;
        } else {
            int base, xb;                                              // squeeze.h:458-474
            len_base_of(s - kSymLen0, base, xb);
            int len = base;
            if (xb != 0) {
                len += (int)r.get_lsb(xb);
                if (r.overrun()) { err = kE2BIG; return; }
            }
            if (len < kLenMin || len > kLenMax) { err = kEINVAL; return; }
            int pk = read_symbol(r, pos, lane, err);                   // squeeze.h:476-500
            if (err != 0) { return; }
            pk -= kPosBase;                                             // leaf id -> distance code
            if (pk == kPosNyt) {
                pk = (int)r.get_lsb(5);
                if (r.overrun()) { err = kE2BIG; return; }
                if (pk >= kPosNyt) { err = kEINVAL; return; }
                const uint32_t up = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos.up_of(kPosBase + pk));
                if (up != kNil) { err = kEINVAL; return; }
                if (!pos.insert_wave(kPosBase + pk, lane)) { err = kE2BIG; return; }
            }
            pos_base_of(pk, base, xb);
            int dist = base;
            if (xb != 0) {
                dist += (int)r.get_lsb(xb);
                if (r.overrun()) { err = kE2BIG; return; }
            }
            // squeeze.h:534-541: 0 < pos <= 0x7FFF (code 29 with all 13 extra bits set is 32768)
            if (dist > 0x7FFF || (uint64_t)dist > i || (uint64_t)len > bytes - i) { err = kEINVAL; return; }
            word = kTokMatch | ((uint32_t)len << 16) | (uint32_t)dist;
            i += (uint64_t)len;
        }
        if (lit.fault | pos.fault) { err = kE2BIG; return; }
        if (lane == 0) { tok[ntok] = word; }
        ntok++;
    };

    uint32_t sdw = 0xFFFFFFFFu;                      // stream dword held by stage[0] (none yet)
    int want = 8, avg4 = 16;                         // read-ahead depth follows the recent step lengths (avg4 = 4 x mean)
#ifdef SQZ_STATS
    uint32_t st_steps = 0, st_rounds = 0, st_m = 0, st_done = 0, st_hist[5] = {0, 0, 0, 0, 0};
    uint64_t st_sec[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter(), st_begin = st_last;
    const uint64_t st_wall0 = wall_clock64();
#define ST_SEC(k) { const uint64_t n_ = __builtin_readcyclecounter(); st_sec[k] += n_ - st_last; st_last = n_; }
#else
#define ST_SEC(k)
#endif
    // All streams of the batch start together and the hardware serves the oldest wave of a
    // SIMD first, so without help the first workgroups of a CU race ahead and the last ones
    // finish alone, latency-bound, on a mostly idle CU.  A wave that is ahead steps back:
    // priority 3 in its first quarter down to 0 in its last, which holds the 16 streams of a
    // CU within a quarter of each other and shortens the tail.
    const uint64_t quarter = (bytes >> 2) + 1;
    int prio_now = -1;
    // tokens read ahead in the previous step that its exact update left valid (below): they open this
    // step's batch without being decoded again
    int kept = 0;
    uint32_t kept_bits = 0, kept_slot = 0;
    uint32_t regain_at = 0;                          // next look at a tree that has given up its positions
    while (i < bytes && err == 0) {
        {
            const int q = (int)(i / quarter);                  // 0..3
            if (q != prio_now) {
                prio_now = q;
                if (q == 0) { __builtin_amdgcn_s_setprio(3); }
                else if (q == 1) { __builtin_amdgcn_s_setprio(2); }
                else if (q == 2) { __builtin_amdgcn_s_setprio(1); }
                else { __builtin_amdgcn_s_setprio(0); }
            }
        }
        if (ntok > kBatchTokens && (lit.aux | pos.aux) != 0) {     // counts are about to outgrow their 24 bits
            lit.give_up_aux(lane);
            pos.give_up_aux(lane);
        }
        if ((lit.aux & pos.aux) == 0 && ntok <= kBatchTokens && ntok >= regain_at) {
            // a tree that was too deep for the position machinery may have settled (sqz_tree.h: regain_aux)
            regain_at = ntok + 512u;
            (void)lit.regain_aux(lane);
            (void)pos.regain_aux(lane);
        }
        const bool frozen = (lit.complete | pos.complete) != 0 || lit.depth >= kFreezeDepth || pos.depth >= kFreezeDepth ||
                            (lit.aux & pos.aux) == 0;
        if (lit.lut_ok == 0) { lit.build_lut(lane); }
        if (pos.lut_ok == 0) { pos.build_lut(lane); }
        // ---- read ahead with the trees held still: every lane decodes the token that would
        //      start at ITS bit offset, then the real starts are picked by following the
        //      lengths from the known start (64 offsets per round, up to 64 tokens) ----------
        const uint64_t bit0 = r.pos;
        uint64_t base = bit0 + kept_bits;
        int m = frozen ? 0 : kept;
        if (lane < m) { slot[lane] = kept_slot; }
        kept = 0; kept_bits = 0;
        bool stop = frozen;
        while (m < want && !stop) {
            const uint32_t k0 = (uint32_t)(base >> 5);
            if (k0 < sdw || ((uint32_t)((base + 63) >> 5) + 2 - sdw) >= (uint32_t)kStageDw) {
                sdw = k0;
#pragma unroll
                for (int h = 0; h < kStageDw / kWave; h++) {
                    const uint64_t k = (uint64_t)sdw + (uint32_t)(h * kWave + lane);
                    uint32_t v = 0;
                    if (k * 32 + 32 <= r.readable) { v = __builtin_bswap32(reinterpret_cast<const uint32_t*>(src)[k]); }
                    lds.stage[h * kWave + lane] = v;
                }
                lds_fence();
            }
            // this lane's 64 stream bits from its offset
            const uint64_t o = base + (uint32_t)lane;
            const uint32_t k = (uint32_t)(o >> 5) - sdw;
            const int sh = (int)(o & 31u);
            const uint32_t d0 = lds.stage[k], d1 = lds.stage[k + 1], d2 = lds.stage[k + 2];
            uint64_t w = ((((uint64_t)d0 << 32) | d1) << sh) | (sh ? ((uint64_t)d2 >> (32 - sh)) : 0ull);
            ST_SEC(6)
            // literal / length symbol
            uint32_t e = lds.luts.lit[(uint32_t)(w >> 56)];
            uint32_t node = e & 0x3FFu;
            uint32_t used = e >> 10;
            w <<= used;
            for (int it = 0; it < 56; it++) {                      // deeper than the table
                const bool more = node >= (uint32_t)kLitLeaves && node != kNil;
                if (__ballot(more) == 0) { break; }
                if (more) {
                    const uint32_t kw = lnk[node];
                    node = (w >> 63) ? l_hi(kw) : l_lo(kw);
                    w <<= 1;
                    used++;
                }
            }
            ST_SEC(7)
            bool bad = node >= (uint32_t)kLitLeaves;              // nil / still inside
            bool esc = node == (uint32_t)kLitNyt;
            // squeeze.h:458-500, read only.  Every lane goes through the back-reference
            // fields (some lane nearly always holds a length code, so branches would only
            // cost mask bookkeeping); a literal's lane uses harmless indices and keeps
            // its own word.
            const bool is_len = !bad && !esc && node > 0xFFu;
            int bs, xb;
            len_base_of(is_len ? (int)node - kSymLen0 : 0, bs, xb);
            const uint32_t len = (uint32_t)bs + (__brev((uint32_t)(w >> 32)) & ((1u << xb) - 1u));
            w <<= xb;
            uint32_t more_bits = (uint32_t)xb;
            const uint32_t e2 = lds.luts.pos[(uint32_t)(w >> (64 - PosTree::kLutBits))];
            uint32_t n2 = e2 & 0x3FFu;                               // absolute node id (or nil)
            w <<= (e2 >> 10);
            more_bits += e2 >> 10;
            for (int it = 0; it < 56; it++) {                      // deeper than the table
                const bool more = is_len && n2 >= (uint32_t)PosTree::kRoot && n2 != kNil;
                if (__ballot(more) == 0) { break; }
                const uint32_t kw = lnk[more ? (int)n2 : (int)PosTree::kRoot];
                const uint32_t down = (w >> 63) ? l_hi(kw) : l_lo(kw);
                n2 = more ? down : n2;
                w <<= more ? 1 : 0;
                more_bits += more ? 1u : 0u;
            }
            const bool pos_bad = n2 >= (uint32_t)PosTree::kRoot;  // nil / still inside
            const bool pos_esc = n2 == (uint32_t)(kPosBase + kPosNyt);
            pos_base_of(pos_bad ? 0 : (int)n2 - kPosBase, bs, xb);
            const uint32_t dist = (uint32_t)bs + (__brev((uint32_t)(w >> 32)) & ((1u << xb) - 1u));
            more_bits += (uint32_t)xb;
            uint32_t word = node;
            if (is_len) {
                bad = pos_bad | (len > (uint32_t)kLenMax) | (dist > 0x7FFFu);    // squeeze.h:534: pos <= 0x7FFF
                esc = pos_esc;
                used += more_bits;
                word = kTokMatch | (len << 16) | dist;
            }
            const bool ok = !bad && !esc && used <= 64u && o + used <= r.readable;
#ifdef SQZ_STATS
            st_rounds++;
#endif
            // follow the token lengths from offset 0 of this round: a token that cannot be
            // taken here jumps out of the round (>= 128) and ends the read-ahead
            ST_SEC(0)
            const int hop = ok ? (int)used : 128;
            uint64_t starts = 0;
            uint32_t s = 0;
            do {
                set_bit64(starts, s);
                s += (uint32_t)__builtin_amdgcn_readlane(hop, (int)s);
            } while (s < (uint32_t)kWave);
            if (s >= 128u) {                                      // the last start is the refused one
                stop = true;
                const int last = 63 - __builtin_clzll(starts);
                starts &= ~(1ull << last);
                s = (uint32_t)last;
            }
            base += s;
            // hand this round's picks to the slots, in order (slots past 64 are never used)
            if ((starts >> lane) & 1ull) {
                const int at = m + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(starts >> 32),
                                   __builtin_amdgcn_mbcnt_lo((uint32_t)starts, 0u));
                slot[at] = word | ((used - 1u) << 25);
            }
            m += __builtin_popcountll(starts);
            ST_SEC(1)
        }
        m = m < kWave ? m : kWave;
        ST_SEC(1)
        lds_fence();
        // ---- lane j = token j: positions, validity, symbols ---------------------------------
        uint32_t word_v = 0, used_v = 0, slot_v = 0;
        if (lane < m) { slot_v = slot[lane]; word_v = slot_v & 0x81FFFFFFu; used_v = ((slot_v >> 25) & 63u) + 1u; }
        const bool is_match = (word_v & kTokMatch) != 0;
        const uint32_t tlen_v = lane < m ? (is_match ? ((word_v >> 16) & 0x1FFu) : 1u) : 0u;
        uint32_t scan = (used_v << 16) | tlen_v;               // both sums fit 16 bits
        scan = wave_scan(scan);
        const uint64_t out_before = i + (uint64_t)((scan & 0xFFFFu) - tlen_v);
        const bool invalid = lane < m &&
            (out_before >= bytes ||
             (is_match && ((uint64_t)(word_v & 0x7FFFu) > out_before || (uint64_t)tlen_v > bytes - out_before)));
        const uint64_t inv = __ballot(invalid);
        if (inv != 0) { const int f = __builtin_ctzll(inv); if (f < m) { m = f; stop = true; } }
        int a_v = -1, b_v = -1;
        if (lane < m) {
            if (is_match) {
                a_v = kSymLen0 + len_code((int)tlen_v).code;
                b_v = kPosBase + pos_code((int)(word_v & 0x7FFFu)).code;
            } else {
                a_v = (int)word_v;
            }
        }
        // ---- apply the longest prefix that changes no link ------------------------------------
        uint32_t ca, cb;
        int wa, wb;
        int done = 0;
        ST_SEC(2)
        lds_fence();                                          // the slots are in registers: their space is the batch's now
        if (m > 0) { done = bump_batch<false>(&lds.tree, nullptr, lit, pos, lane, m, a_v, b_v, ca, wa, cb, wb); }
        ST_SEC(3)
        uint64_t resume = bit0;
        if (done > 0) {
            if (lane < done) { tok[ntok + (uint32_t)lane] = word_v; }
            const uint32_t af = (uint32_t)__builtin_amdgcn_readlane((int)scan, done - 1);
            ntok += (uint32_t)done;
            i += (uint64_t)(af & 0xFFFFu);
            resume = bit0 + (uint64_t)(af >> 16);
        }
        // ---- whatever stopped the step: one token, exactly ------------------------------------
#ifdef SQZ_STATS
        st_steps++; st_m += (uint32_t)m; st_done += (uint32_t)done;
#ifdef SQZ_STATS_SERIES
        if (lane == 0 && b == 1) { printf("S %d %d %d %d\n", want, m, done, (int)stop); }
#endif
        st_hist[done == 0 ? 0 : done < 8 ? 1 : done < 24 ? 2 : done < 64 ? 3 : 4]++;
#endif
        ST_SEC(4)
        if (i < bytes && (stop || done < m)) {
            const bool may_keep = done + 1 < m;              // the step ended on a token the batch refused: the
            if (may_keep && lane == 0) {                     // ones behind it were decoded with the tree as it was
                lds.tree.pend[LitTree::kChgSlot] = 0x7FFFFFFFu; lds.tree.pend[LitTree::kChgSlot + 1] = 0u;
                lds.tree.pend[PosTree::kChgSlot] = 0x7FFFFFFFu; lds.tree.pend[PosTree::kChgSlot + 1] = 0u;
            }
            lds_fence();
            if (done < m) {
                // the token the batch refused sits in lane `done`, read ahead and checked like the others:
                // only its updates are left to do, one symbol at a time (squeeze.h:509-549 without the reads)
                const int sa = __builtin_amdgcn_readlane(a_v, done), sb = __builtin_amdgcn_readlane(b_v, done);
                const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)word_v, done);
                const uint32_t u1 = (uint32_t)__builtin_amdgcn_readlane((int)used_v, done);
                const uint32_t l1 = (uint32_t)__builtin_amdgcn_readlane((int)tlen_v, done);
                { const Chain c = lit.chain_up(sa, lane); (void)lit.bump_wave(sa, c, lane); }
                if (sb >= 0) { const Chain c = pos.chain_up(sb, lane); (void)pos.bump_wave(sb, c, lane); }
                if (lit.fault | pos.fault) { err = kE2BIG; }
                else {
                    if (lane == 0) { tok[ntok] = w1; }
                    ntok++;
                    i += (uint64_t)l1;
                    r.seek(resume + (uint64_t)u1);
                }
            } else {
                r.seek(resume);
                decode_one();
            }
            lds_fence();
            const bool still = (lit.complete | pos.complete) == 0 && lit.depth < kFreezeDepth && pos.depth < kFreezeDepth &&
                               (lit.aux & pos.aux) != 0;
            if (may_keep && still && err == 0 && i < bytes) {
                // A token behind the exact one is still right if neither of its leaves got a new code: the
                // same bits then lead to the same leaf in the new tree.  Keep the run of such tokens.
                const uint32_t la = lds.tree.pend[LitTree::kChgSlot], ha = lds.tree.pend[LitTree::kChgSlot + 1];
                const uint32_t lb = lds.tree.pend[PosTree::kChgSlot], hb = lds.tree.pend[PosTree::kChgSlot + 1];
                const bool cand = lane > done && lane < m;
                bool moved = false;
                if (cand) {
                    const uint32_t qa = r_pos(lds.tree.rng[a_v]);
                    moved = qa >= la && qa < ha;
                    if (is_match) { const uint32_t qb = r_pos(lds.tree.rng[b_v]); moved |= qb >= lb && qb < hb; }
                }
                const uint64_t behind = ~0ull << (done + 1);
                const uint64_t ends = (__ballot(!cand || moved) & behind);      // lanes >= m are not candidates: never empty
                const int first_out = ends != 0 ? __builtin_ctzll(ends) : kWave;
                kept = first_out - (done + 1);
                if (kept > 0) {
                    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)scan, done) >> 16;
                    const uint32_t s1 = (uint32_t)__builtin_amdgcn_readlane((int)scan, done + kept) >> 16;
                    kept_bits = s1 - s0;
                    kept_slot = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + done + 1) & (kWave - 1)) << 2, (int)slot_v);
                }
            }
        } else {
            r.seek(resume);
        }
        ST_SEC(5)
        {   // a step that ran short is usually followed by more short ones (the tree is still moving)
            avg4 += done - (avg4 >> 2);
            want = ((avg4 * 3) >> 3) + 6;                     // 1.5 x the recent mean, and a little (offline
                                                              // what-if over a recorded step series: tools/readahead_policy.py)
            want = want < kWave ? want : kWave;
        }
    }
#ifdef SQZ_STATS
    if (lane == 0 && (b % 512 == 7 || b == n_blocks - 1)) {
        printf("wall block %u: start %llu end %llu (100 MHz ticks), cycles %llu\n", b, (unsigned long long)st_wall0,
               (unsigned long long)wall_clock64(), (unsigned long long)(__builtin_readcyclecounter() - st_begin));
    }
    if (lane == 0 && b == 1) {
        printf("block %u: tokens %u steps %u rounds %u sum_m %u sum_done %u hist[0,<8,<24,<64,64] %u %u %u %u %u\n", b, ntok,
               st_steps, st_rounds, st_m, st_done, st_hist[0], st_hist[1], st_hist[2], st_hist[3], st_hist[4]);
        printf("cycles total %llu: round %llu hop+slot %llu post %llu bump %llu store %llu exact %llu\n", (unsigned long long)(st_last - st_begin),
               (unsigned long long)st_sec[0], (unsigned long long)st_sec[1], (unsigned long long)st_sec[2], (unsigned long long)st_sec[3], (unsigned long long)st_sec[4], (unsigned long long)st_sec[5]);
        printf("round parts: stage+window %llu lit lut+deep %llu match %llu\n", (unsigned long long)st_sec[6], (unsigned long long)st_sec[7], (unsigned long long)st_sec[0]);
        printf("lit slow: insert %u/%llu changed %u/%llu lut %u/%llu; pos: insert %u/%llu changed %u/%llu lut %u/%llu\n",
               lit.st_cnt[0], (unsigned long long)lit.st_cyc[0], lit.st_cnt[1], (unsigned long long)lit.st_cyc[1], lit.st_cnt[2], (unsigned long long)lit.st_cyc[2],
               pos.st_cnt[0], (unsigned long long)pos.st_cyc[0], pos.st_cnt[1], (unsigned long long)pos.st_cyc[1], pos.st_cnt[2], (unsigned long long)pos.st_cyc[2]);
    }
#endif
    if (lane == 0) {
        tok_count[b] = ntok;
        err_out[b] = err;
        if (end_bit != nullptr) { end_bit[b] = r.pos; }
    }
}


// ---------------------------------------------------------------------------------------------
// The same decoder with W wavefronts per stream (W = 2, 4), for batches that leave most of the chip idle
// (at 512 blocks per GPU -- BASELINE.json configs[3] on 8 GPUs -- half of the 1024 SIMDs have no wave at
// all and one stream is a 60 ms dependent chain).  The read-ahead is what the extra waves are for: with the
// trees held still, wave w decodes the candidate tokens at bit offsets base + 64 w + lane, all W waves at
// once; the real token starts are then picked wave after wave (a wave's walk starts where the previous
// one's left its 64 offsets) -- 64 W offsets per round instead of 64.  Wave 0 alone owns the trees and runs
// the update side exactly as the one-wave kernel does (bump_batch, the exact path, the stores); the waves
// meet at workgroup barriers and share their wave-uniform state through a few LDS words, so every wave takes
// every barrier: there is no spinning and no wave waits for a flag another wave might never set.
// Tokens a round decodes beyond the 64 a step applies stay in the slots and open the next step's batch.
// Wave-uniform state the waves hand to each other: written by ONE wave before a barrier, read by all behind
// it.  Two copies used alternately (steps by their parity, rounds by theirs): a wave that is slow to read
// what the last barrier published never meets the next writer, who is already filling the other copy --
// and by the time a copy is written again every wave has passed at least one more barrier.
struct MwStep {
    uint32_t run;                       // 0: the stream is finished (or failed): every wave leaves
    uint32_t m;                         // tokens in the slots
    uint32_t want;                      // read ahead until this many
    uint32_t stop;                      // no read-ahead in this step (frozen / very deep trees)
    uint32_t bit_lo, bit_hi;            // stream bit at which the read-ahead goes on
    uint32_t solo;                      // this step's read-ahead is wave 0's alone
    uint32_t sdw;                       // stream dword the staged window starts at
};
// Per round and wave: for every one of its 64 offsets, where a walk that STARTS there leaves the wave's share
// and how many tokens it picks on the way (found for all 64 starts at once by pointer doubling).  With these
// tables in LDS every wave follows the round's walk through all W shares by itself -- W dependent reads --
// and then marks its own starts: one barrier per round instead of one per wave.
//   low byte:  0..63   the walk goes on at that offset of the next wave's share
//              64 + l  the walk ends AT offset l of this share: the token there cannot be taken
//   high byte: tokens picked in this share
struct MwRound { uint16_t exit[8][kWave]; };

constexpr int kMwSoloWant = 24;         // steps that want no more tokens than this are read ahead by wave 0 alone

template <int W>
struct DecodeMwLds {
    TreeLds    tree;
    DecodeLuts luts;
    uint32_t   stage[kStageDw];
    uint32_t   slot[kWave + kWave * W]; // word | (bits used - 1) << 25, in stream order
    MwStep     step[2];
    MwRound    round[2];
};

template <int W>
__global__ __launch_bounds__(kWave * W, 4)      // 128 VGPRs: 16 waves per CU = 16 / W streams
void entropy_decode_mw_kernel(const uint8_t* __restrict__ in,
                              const uint64_t* __restrict__ in_off,
                              const uint64_t* __restrict__ out_off,
                              uint32_t* __restrict__ tokens,
                              uint32_t* __restrict__ tok_count,
                              int32_t* __restrict__ err_out,
                              uint64_t* __restrict__ end_bit,
                              uint32_t n_blocks,
                              uint64_t start_bit) {
    __shared__ DecodeMwLds<W> lds;
    const int lane = (int)(threadIdx.x & (kWave - 1));
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) { return; }                 // (the whole workgroup)

    LitTree lit; PosTree pos;
    lit.lds = &lds.tree; lit.code = nullptr; lit.lut = lds.luts.lit;
    pos.lds = &lds.tree; pos.code = nullptr; pos.lut = lds.luts.pos;
    lit.init_all(lane);                            // (registers in every wave; the LDS words by wave 0's lanes below)
    pos.init_all(lane);
    lds_barrier();

    const uint64_t o0 = uni64(out_off[b]), o1 = uni64(out_off[b + 1]);
    const uint64_t bytes = o1 - o0;
    uint32_t* tok = tokens + o0;
    const uint64_t i0 = uni64(in_off[b]), i1 = uni64(in_off[b + 1]);
    const uint8_t* const src = in + i0;
    const uint64_t src_bytes = i1 - i0;
    const uint64_t readable = (src_bytes / 8) * 64;
    BitSource r;
    r.open(src, src_bytes, start_bit, lane);       // (wave 0's is the one that is used)
    int err = 0;
    if (wave == 0) {
        if (!lit.insert_wave(kLitNyt, lane)) { err = kEINVAL; }          // squeeze.h:505-506
        if (!pos.insert_wave(kPosBase + kPosNyt, lane)) { err = kEINVAL; }
    }
    const uint32_t* const lnk = lds.tree.lnk;
    uint32_t* const slot = lds.slot;
    uint64_t i = 0;
    uint32_t ntok = 0;

    auto decode_one = [&]() {                      // squeeze.h:509-549 for exactly one token (wave 0)
        int s = read_symbol(r, lit, lane, err);
        if (err != 0) { return; }
        if (s == kLitNyt) {
            s = (int)r.get_lsb(9);
            if (r.overrun()) { err = kE2BIG; return; }
            if (s == 256 || s >= kLitNyt) { err = kEINVAL; return; }
            const uint32_t up = (uint32_t)__builtin_amdgcn_readfirstlane((int)lit.up_of(s));
            if (up != kNil) { err = kEINVAL; return; }
            if (!lit.insert_wave(s, lane)) { err = kE2BIG; return; }
        }
        uint32_t word;
        if (s <= 0xFF) {
            word = (uint32_t)s;
            i++;
        } else {
            int base, xb;
            len_base_of(s - kSymLen0, base, xb);
            int len = base;
            if (xb != 0) {
                len += (int)r.get_lsb(xb);
                if (r.overrun()) { err = kE2BIG; return; }
            }
            if (len < kLenMin || len > kLenMax) { err = kEINVAL; return; }
            int pk = read_symbol(r, pos, lane, err);
            if (err != 0) { return; }
            pk -= kPosBase;
            if (pk == kPosNyt) {
                pk = (int)r.get_lsb(5);
                if (r.overrun()) { err = kE2BIG; return; }
                if (pk >= kPosNyt) { err = kEINVAL; return; }
                const uint32_t up = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos.up_of(kPosBase + pk));
                if (up != kNil) { err = kEINVAL; return; }
                if (!pos.insert_wave(kPosBase + pk, lane)) { err = kE2BIG; return; }
            }
            pos_base_of(pk, base, xb);
            int dist = base;
            if (xb != 0) {
                dist += (int)r.get_lsb(xb);
                if (r.overrun()) { err = kE2BIG; return; }
            }
            if (dist > 0x7FFF || (uint64_t)dist > i || (uint64_t)len > bytes - i) { err = kEINVAL; return; }
            word = kTokMatch | ((uint32_t)len << 16) | (uint32_t)dist;
            i += (uint64_t)len;
        }
        if (lit.fault | pos.fault) { err = kE2BIG; return; }
        if (lane == 0) { tok[ntok] = word; }
        ntok++;
    };

    uint32_t sdw = 0xFFFFFFFFu;                      // stream dword held by stage[0] (every wave keeps the same value)
    int want = 8, avg4 = 16;
    const uint64_t quarter = (bytes >> 2) + 1;
    int prio_now = -1;
    int kept = 0;                                    // tokens in slot[0, kept) that the last step left valid (wave 0)
    uint32_t kept_bits = 0;
    uint32_t step_no = 0, round_no = 0;              // shared steps and rounds (every wave counts them alike)
    uint32_t regain_at = 0;                          // (wave 0) next look at a tree that has given up its positions
#ifdef SQZ_STATS
    uint64_t mw_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, mw_last = __builtin_readcyclecounter();
    uint32_t mw_solo_rounds = 0, mw_steps = 0, mw_done = 0;
#define MW_SEC(k) { const uint64_t n_ = __builtin_readcyclecounter(); mw_t[k] += n_ - mw_last; mw_last = n_; }
#else
#define MW_SEC(k)
#endif
    // Wave 0 meets the other waves only in the steps whose read-ahead they take part in (and at the very end):
    // a SOLO step -- wave 0 reads ahead by itself, see below -- has no barrier in it at all, and while wave 0
    // makes one solo step after the other the other waves stand at the barrier at the top of their loop.
    for (;;) {
        MwStep& sc = lds.step[step_no & 1u];                  // (step_no counts the steps the waves share)
        bool shared = true;
        uint64_t base = 0;
        int m = 0, want_now = 0;
        bool stop = false, solo = false;
        // ---- wave 0: where the stream stands; the lookup tables -----------------------------------------
        if (wave == 0) {
            const bool run = i < bytes && err == 0;
            bool frozen = true;
            if (run) {
                const int q = (int)(i / quarter);
                if (q != prio_now) {
                    prio_now = q;
                    if (q == 0) { __builtin_amdgcn_s_setprio(3); }
                    else if (q == 1) { __builtin_amdgcn_s_setprio(2); }
                    else if (q == 2) { __builtin_amdgcn_s_setprio(1); }
                    else { __builtin_amdgcn_s_setprio(0); }
                }
                if (ntok > kBatchTokens && (lit.aux | pos.aux) != 0) { lit.give_up_aux(lane); pos.give_up_aux(lane); }
                if ((lit.aux & pos.aux) == 0 && ntok <= kBatchTokens && ntok >= regain_at) {
                    regain_at = ntok + 512u;
                    (void)lit.regain_aux(lane);
                    (void)pos.regain_aux(lane);
                }
                frozen = (lit.complete | pos.complete) != 0 || lit.depth >= kFreezeDepth || pos.depth >= kFreezeDepth ||
                         (lit.aux & pos.aux) == 0;
                if (lit.lut_ok == 0) { lit.build_lut(lane); }
                if (pos.lut_ok == 0) { pos.build_lut(lane); }
            }
            if (frozen) { kept = 0; kept_bits = 0; }
            const uint64_t at = r.pos + kept_bits;
            solo = run && (frozen || want <= kMwSoloWant);
            shared = !solo;
            if (shared) {   // lane k writes word k of the step's shared words (run, m, want, stop, bit_lo, bit_hi, -, sdw)
                const uint32_t v = lane == 0 ? (run ? 1u : 0u) : lane == 1 ? (uint32_t)kept : lane == 2 ? (uint32_t)want
                                 : lane == 3 ? (frozen ? 1u : 0u) : lane == 4 ? (uint32_t)at : lane == 5 ? (uint32_t)(at >> 32)
                                 : lane == 6 ? 0u : sdw;
                if (lane < 8) { reinterpret_cast<uint32_t*>(&sc)[lane] = v; }
            } else {
                base = at; m = kept; want_now = want; stop = frozen;
            }
        }
        MW_SEC(0)
        const uint64_t bit0 = r.pos;                                     // (wave 0's: where this step's first token starts)
        if (shared) {
            lds_barrier();
            MW_SEC(1)
            step_no++;
            // the step's shared words: ONE read (lane k takes word k), then out of the register
            const uint32_t scw = reinterpret_cast<const uint32_t*>(&sc)[lane & 7];
            static_assert(sizeof(MwStep) == 32, "eight words");
            if (__builtin_amdgcn_readlane((int)scw, 0) == 0) { break; }      // run
            base = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)scw, 5) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane((int)scw, 4);
            m = __builtin_amdgcn_readlane((int)scw, 1);
            want_now = __builtin_amdgcn_readlane((int)scw, 2);
            stop = __builtin_amdgcn_readlane((int)scw, 3) != 0;
            sdw = (uint32_t)__builtin_amdgcn_readlane((int)scw, 7);
        }
        // the candidate token at stream bit o (squeeze.h:458-500, read only): its word, the bits it takes, and
        // whether it can be taken here (no escape, nothing malformed, inside the readable bits)
        auto candidate = [&](uint64_t o, uint32_t& word_out, uint32_t& used_out, bool& ok) {
            // this lane's 64 stream bits from its offset
            const uint32_t k = (uint32_t)(o >> 5) - sdw;
            const int sh = (int)(o & 31u);
            const uint32_t d0 = lds.stage[k], d1 = lds.stage[k + 1], d2 = lds.stage[k + 2];
            uint64_t w = ((((uint64_t)d0 << 32) | d1) << sh) | (sh ? ((uint64_t)d2 >> (32 - sh)) : 0ull);
            // literal / length symbol
            uint32_t e = lds.luts.lit[(uint32_t)(w >> 56)];
            uint32_t node = e & 0x3FFu;
            uint32_t used = e >> 10;
            w <<= used;
            for (int it = 0; it < 56; it++) {                      // deeper than the table
                const bool more = node >= (uint32_t)kLitLeaves && node != kNil;
                if (__ballot(more) == 0) { break; }
                if (more) {
                    const uint32_t kw = lnk[node];
                    node = (w >> 63) ? l_hi(kw) : l_lo(kw);
                    w <<= 1;
                    used++;
                }
            }
            bool bad = node >= (uint32_t)kLitLeaves;              // nil / still inside
            bool esc = node == (uint32_t)kLitNyt;
            const bool is_len = !bad && !esc && node > 0xFFu;
            int bs, xb;
            len_base_of(is_len ? (int)node - kSymLen0 : 0, bs, xb);
            const uint32_t len = (uint32_t)bs + (__brev((uint32_t)(w >> 32)) & ((1u << xb) - 1u));
            w <<= xb;
            uint32_t more_bits = (uint32_t)xb;
            const uint32_t e2 = lds.luts.pos[(uint32_t)(w >> (64 - PosTree::kLutBits))];
            uint32_t n2 = e2 & 0x3FFu;
            w <<= (e2 >> 10);
            more_bits += e2 >> 10;
            for (int it = 0; it < 56; it++) {
                const bool more = is_len && n2 >= (uint32_t)PosTree::kRoot && n2 != kNil;
                if (__ballot(more) == 0) { break; }
                const uint32_t kw = lnk[more ? (int)n2 : (int)PosTree::kRoot];
                const uint32_t down = (w >> 63) ? l_hi(kw) : l_lo(kw);
                n2 = more ? down : n2;
                w <<= more ? 1 : 0;
                more_bits += more ? 1u : 0u;
            }
            const bool pos_bad = n2 >= (uint32_t)PosTree::kRoot;
            const bool pos_esc = n2 == (uint32_t)(kPosBase + kPosNyt);
            pos_base_of(pos_bad ? 0 : (int)n2 - kPosBase, bs, xb);
            const uint32_t dist = (uint32_t)bs + (__brev((uint32_t)(w >> 32)) & ((1u << xb) - 1u));
            more_bits += (uint32_t)xb;
            uint32_t word = node;
            if (is_len) {
                bad = pos_bad | (len > (uint32_t)kLenMax) | (dist > 0x7FFFu);
                esc = pos_esc;
                used += more_bits;
                word = kTokMatch | (len << 16) | dist;
            }
            ok = !bad && !esc && used <= 64u && o + used <= readable;
            word_out = word;
            used_out = used;
        };
        // ---- read ahead -----------------------------------------------------------------------------------
        if (solo) {
            // Short steps (the trees are still forming, or the data keeps them moving: a restructure every few
            // tokens): a round of 64 W offsets would decode far more than the step can use and pay a barrier
            // for it.  Wave 0 reads ahead by itself, 64 offsets per round, as the one-wave kernel does, and
            // meets no barrier in such a step.
            if (wave == 0) {
                while (m < want_now && !stop) {
                    const uint32_t k0 = (uint32_t)(base >> 5);
                    if (k0 < sdw || ((uint32_t)((base + 63) >> 5) + 2 - sdw) >= (uint32_t)kStageDw) {
                        sdw = k0;
                        for (int h = lane; h < kStageDw; h += kWave) {
                            const uint64_t k = (uint64_t)sdw + (uint32_t)h;
                            uint32_t v = 0;
                            if (k * 32 + 32 <= readable) { v = __builtin_bswap32(reinterpret_cast<const uint32_t*>(src)[k]); }
                            lds.stage[h] = v;
                        }
                        lds_fence();
                    }
                    uint32_t word, used; bool ok;
                    candidate(base + (uint32_t)lane, word, used, ok);
#ifdef SQZ_STATS
                    mw_solo_rounds++;
#endif
                    const int hop = ok ? (int)used : 128;
                    uint64_t starts = 0;
                    uint32_t s1 = 0;
                    do {
                        set_bit64(starts, s1);
                        s1 += (uint32_t)__builtin_amdgcn_readlane(hop, (int)s1);
                    } while (s1 < (uint32_t)kWave);
                    if (s1 >= 128u) {                              // the last start is the refused one
                        stop = true;
                        const int last = 63 - __builtin_clzll(starts);
                        starts &= ~(1ull << last);
                        s1 = (uint32_t)last;
                    }
                    base += s1;
                    if ((starts >> lane) & 1ull) { slot[m + (int)lanes_under(starts)] = word | ((used - 1u) << 25); }
                    m += __builtin_popcountll(starts);
                }
            }
        } else {
        // all waves: 64 W bit offsets per round
        while (m < want_now && !stop) {
            const uint32_t k0 = (uint32_t)(base >> 5);
            if (k0 < sdw || ((uint32_t)((base + (uint64_t)(kWave * W - 1) + 63) >> 5) + 2 - sdw) >= (uint32_t)kStageDw) {
                sdw = k0;                                                // (the last round's readers are past their barriers)
                for (int h = wave * kWave + lane; h < kStageDw; h += kWave * W) {
                    const uint64_t k = (uint64_t)sdw + (uint32_t)h;
                    uint32_t v = 0;
                    if (k * 32 + 32 <= readable) { v = __builtin_bswap32(reinterpret_cast<const uint32_t*>(src)[k]); }
                    lds.stage[h] = v;
                }
                lds_barrier();
            }
            uint32_t word, used; bool ok;
            candidate(base + (uint32_t)(wave * kWave + lane), word, used, ok);
            const int hop = ok ? (int)used : 1024;               // a token that cannot be taken jumps far out of the round
            MW_SEC(2)
            // ---- the real token starts, wave after wave: wave v follows the lengths through its own 64
            //      offsets from where wave v-1 left them ---------------------------------------------------
            MwRound& rc = lds.round[round_no & 1u];
            round_no++;
            {   // pointer doubling over this wave's 64 offsets: to = where the walk stands after cnt tokens
                uint32_t to = ok ? (uint32_t)lane + used : 128u + (uint32_t)lane;     // < 64: inside; 64..127: out; >= 128: refused here
                uint32_t cnt = ok ? 1u : 0u;
#pragma unroll
                for (int step = 0; step < 6; step++) {
                    const int from = (int)(to & 63u) << 2;
                    const uint32_t to2 = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)to);
                    const uint32_t cnt2 = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)cnt);
                    const bool inside = to < (uint32_t)kWave;
                    cnt += inside ? cnt2 : 0u;
                    to = inside ? to2 : to;
                }
                rc.exit[wave][lane] = (uint16_t)((to - (uint32_t)kWave) | (cnt << 8));   // 0..63 next share, 64 + l refused at l
            }
            lds_barrier();
            // the round's walk through all shares (every wave does this alike)
            uint32_t entry = 0, my_entry = 0;
            int my_base = m;
            bool mine = false;
            uint32_t advance = 0;
#pragma unroll
            for (int v = 0; v < W; v++) {
                if (!stop) {
                    const uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)rc.exit[v][entry]);
                    if (v == wave) { mine = true; my_entry = entry; my_base = m; }
                    m += (int)(a >> 8);
                    const uint32_t fin = a & 0xFFu;
                    if (fin >= (uint32_t)kWave) { stop = true; advance = (uint32_t)(kWave * v) + (fin - (uint32_t)kWave); }
                    else { entry = fin; advance = (uint32_t)(kWave * (v + 1)) + fin; }
                }
            }
            if (mine) {                                            // my starts, into the slots (a refused start ends the walk)
                uint64_t starts = 0;
                uint32_t s1 = my_entry;
                while (s1 < (uint32_t)kWave) {
                    const uint32_t h = (uint32_t)__builtin_amdgcn_readlane(hop, (int)s1);
                    if (h >= 1024u) { break; }
                    set_bit64(starts, s1);
                    s1 += h;
                }
                if ((starts >> lane) & 1ull) { slot[my_base + (int)lanes_under(starts)] = word | ((used - 1u) << 25); }
            }
            base += advance;                                        // (stopped: AT the token that cannot be taken)
            MW_SEC(3)
        }
        }
        MW_SEC(5)
        if (shared) { lds_barrier(); }                            // every wave's slots are in
        if (wave != 0) { continue; }                              // (back to the barrier at the top)
        // =================================================================================================
        // wave 0: the update side, as in entropy_decode_kernel
        const int m_total = m;
        m = m < kWave ? m : kWave;
        lds_fence();
        uint32_t word_v = 0, used_v = 0, slot_v = 0;
        if (lane < m) { slot_v = slot[lane]; word_v = slot_v & 0x81FFFFFFu; used_v = ((slot_v >> 25) & 63u) + 1u; }
        const bool is_match = (word_v & kTokMatch) != 0;
        const uint32_t tlen_v = lane < m ? (is_match ? ((word_v >> 16) & 0x1FFu) : 1u) : 0u;
        uint32_t scan = (used_v << 16) | tlen_v;
        scan = wave_scan(scan);
        const uint64_t out_before = i + (uint64_t)((scan & 0xFFFFu) - tlen_v);
        const bool invalid = lane < m &&
            (out_before >= bytes ||
             (is_match && ((uint64_t)(word_v & 0x7FFFu) > out_before || (uint64_t)tlen_v > bytes - out_before)));
        const uint64_t inv = __ballot(invalid);
        if (inv != 0) { const int f = __builtin_ctzll(inv); if (f < m) { m = f; stop = true; } }
        int a_v = -1, b_v = -1;
        if (lane < m) {
            if (is_match) {
                a_v = kSymLen0 + len_code((int)tlen_v).code;
                b_v = kPosBase + pos_code((int)(word_v & 0x7FFFu)).code;
            } else {
                a_v = (int)word_v;
            }
        }
        uint32_t ca, cb;
        int wa, wb;
        int done = 0;
        MW_SEC(6)
#ifdef SQZ_STATS
        mw_steps++;
#endif
        if (m > 0) { done = bump_batch<false>(&lds.tree, nullptr, lit, pos, lane, m, a_v, b_v, ca, wa, cb, wb); }
        MW_SEC(7)
#ifdef SQZ_STATS
        mw_done += (uint32_t)done;
#endif
        uint64_t resume = bit0;
        if (done > 0) {
            if (lane < done) { tok[ntok + (uint32_t)lane] = word_v; }
            const uint32_t af = (uint32_t)__builtin_amdgcn_readlane((int)scan, done - 1);
            ntok += (uint32_t)done;
            i += (uint64_t)(af & 0xFFFFu);
            resume = bit0 + (uint64_t)(af >> 16);
        }
        kept = 0; kept_bits = 0;
        int keep_from = done;                                   // first slot that may open the next step's batch
        if (i < bytes && (stop || done < m)) {
            const bool may_keep = done + 1 < m;
            if (may_keep && lane == 0) {
                lds.tree.pend[LitTree::kChgSlot] = 0x7FFFFFFFu; lds.tree.pend[LitTree::kChgSlot + 1] = 0u;
                lds.tree.pend[PosTree::kChgSlot] = 0x7FFFFFFFu; lds.tree.pend[PosTree::kChgSlot + 1] = 0u;
            }
            lds_fence();
            if (done < m) {
                const int sa = __builtin_amdgcn_readlane(a_v, done), sb = __builtin_amdgcn_readlane(b_v, done);
                const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)word_v, done);
                const uint32_t u1 = (uint32_t)__builtin_amdgcn_readlane((int)used_v, done);
                const uint32_t l1 = (uint32_t)__builtin_amdgcn_readlane((int)tlen_v, done);
                { const Chain c = lit.chain_up(sa, lane); (void)lit.bump_wave(sa, c, lane); }
                if (sb >= 0) { const Chain c = pos.chain_up(sb, lane); (void)pos.bump_wave(sb, c, lane); }
                if (lit.fault | pos.fault) { err = kE2BIG; }
                else {
                    if (lane == 0) { tok[ntok] = w1; }
                    ntok++;
                    i += (uint64_t)l1;
                    r.seek(resume + (uint64_t)u1);
                }
            } else {
                r.seek(resume);
                decode_one();
            }
            lds_fence();
            const bool still = (lit.complete | pos.complete) == 0 && lit.depth < kFreezeDepth && pos.depth < kFreezeDepth &&
                               (lit.aux & pos.aux) != 0;
            if (may_keep && still && err == 0 && i < bytes) {
                const uint32_t la = lds.tree.pend[LitTree::kChgSlot], ha = lds.tree.pend[LitTree::kChgSlot + 1];
                const uint32_t lb = lds.tree.pend[PosTree::kChgSlot], hb = lds.tree.pend[PosTree::kChgSlot + 1];
                const bool cand = lane > done && lane < m;
                bool moved = false;
                if (cand) {
                    const uint32_t qa = r_pos(lds.tree.rng[a_v]);
                    moved = qa >= la && qa < ha;
                    if (is_match) { const uint32_t qb = r_pos(lds.tree.rng[b_v]); moved |= qb >= lb && qb < hb; }
                }
                const uint64_t behind = ~0ull << (done + 1);
                const uint64_t ends = (__ballot(!cand || moved) & behind);
                const int first_out = ends != 0 ? __builtin_ctzll(ends) : kWave;
                kept = first_out - (done + 1);
                keep_from = done + 1;
                if (kept > 0) {
                    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)scan, done) >> 16;
                    const uint32_t s1 = (uint32_t)__builtin_amdgcn_readlane((int)scan, done + kept) >> 16;
                    kept_bits = s1 - s0;
                }
            }
        } else {
            r.seek(resume);
            // the whole batch went through and nothing stopped the read-ahead: what it decoded beyond the 64
            // tokens of a step is as valid as they were
            if (done == m && !stop && m_total > m && i < bytes) {
                kept = m_total - m < kWave ? m_total - m : kWave;       // (one lane moves one token)
                keep_from = m;
                const uint32_t sv = lane < kept ? slot[m + lane] : 0u;
                kept_bits = wave_sum(lane < kept ? ((sv >> 25) & 63u) + 1u : 0u);
            }
        }
        if (kept > 0 && keep_from > 0) {                           // move them to the front of the slots
            const uint32_t sv = lane < kept ? slot[keep_from + lane] : 0u;
            lds_fence();
            if (lane < kept) { slot[lane] = sv; }
            lds_fence();
        }
        {
            avg4 += done - (avg4 >> 2);
            want = ((avg4 * 3) >> 3) + 6;
            want = want < kWave ? want : kWave;
        }
        MW_SEC(4)
    }
#ifdef SQZ_STATS
    if (lane == 0 && b == 1) {
        printf("mw all steps %u solo rounds %u tokens %u done-by-batches %u\n", mw_steps, mw_solo_rounds, ntok, mw_done);
        printf("mw W=%d wave %d: shared steps %u rounds %u: top %llu top-barrier %llu decode %llu chain %llu solo-readahead %llu post %llu bump %llu rest %llu\n", W, wave, step_no, round_no,
               (unsigned long long)mw_t[0], (unsigned long long)mw_t[1], (unsigned long long)mw_t[2], (unsigned long long)mw_t[3], (unsigned long long)mw_t[5],
               (unsigned long long)mw_t[6], (unsigned long long)mw_t[7], (unsigned long long)mw_t[4]);
    }
#endif
    if (wave == 0 && lane == 0) {
        tok_count[b] = ntok;
        err_out[b] = err;
        if (end_bit != nullptr) { end_bit[b] = r.pos; }
    }
}

// ---------------------------------------------------------------------------
// LZ77 expansion (squeeze.h:521-539) with the window where it already is: the output buffer,
// served by the L2.  A window staged in LDS needs 40 KB per stream -- one wave per SIMD, four
// rounds of workgroups for the batch, every latency exposed (22.4 ms); this needs no LDS, so
// all streams are resident at once and four waves per SIMD hide each other's round trips
// (6.8 ms).
//
// Bytes written by one lane are read by others, through memory: stores are write-through to
// the L2, a workgroup-scope fence waits for them, and the source bytes of a back reference
// are loaded with agent-scope (L2) loads, so nothing depends on what the L1 holds.
// 64 tokens per step: literals all at once; back references whose source lies entirely in
// front of the step (nearly all of them) by their own lanes, side by side; the rest -- the
// source overlaps this step's output, or the copy is long -- one after the other with the
// whole wave, in token order.
constexpr int kExpOwnLane = 24;               // longest copy a single lane does by itself

__device__ __forceinline__ uint8_t load_l2(const uint8_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kWave, 4)          // four waves per SIMD: 16 streams per CU (the LDS allows as many)
void lz_expand_kernel(const uint32_t* __restrict__ tokens,
                         const uint32_t* __restrict__ tok_count,
                         uint8_t* __restrict__ out,
                         const uint64_t* __restrict__ out_off,
                         uint32_t n_blocks) {
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) { return; }
    uint8_t* dst = out + out_off[b];
    const uint32_t* tok = tokens + out_off[b];
    const uint32_t count = tok_count[b];      // tokens decoded before an error are still expanded

    uint64_t i = 0;                            // next output byte
    uint32_t next_word = (uint32_t)lane < count ? tok[lane] : 0u;
    for (uint32_t t0 = 0; t0 < count; t0 += kWave) {
        const uint32_t left = count - t0;
        const bool mine_in = (uint32_t)lane < left;
        const uint32_t word = next_word;
        const uint32_t t1 = t0 + kWave + (uint32_t)lane;   // the next step's tokens are on their way
        next_word = t1 < count ? tok[t1] : 0u;
        const bool is_match = mine_in && (word & kTokMatch) != 0;
        const uint32_t mylen = mine_in ? (is_match ? ((word >> 16) & 0x1FFu) : 1u) : 0u;
        const uint32_t incl = wave_scan(mylen);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
        const uint64_t o = i + (incl - mylen);             // where my token's bytes go
        const uint32_t dist = word & 0x7FFFu;
        // everything earlier steps wrote is in the L2 from here on
        __threadfence_block();
        if (mine_in && !is_match) { dst[o] = (uint8_t)word; }
        // source entirely in front of this step, and short: my own lane copies it
        const bool own = is_match && (o - dist) + mylen <= i && mylen <= (uint32_t)kExpOwnLane;
        {   // four bytes per trip: the loads of a trip are independent of each other
            const uint8_t* sp = dst + (own ? o - dist : 0);
            uint8_t* dp = dst + (own ? o : 0);
            const uint32_t n_own = own ? mylen : 0u;
            for (uint32_t k0 = 0; __ballot(k0 < n_own) != 0; k0 += 4) {
                uint8_t v[4];
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) { v[j] = k0 + j < n_own ? load_l2(sp + k0 + j) : (uint8_t)0; }
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) { if (k0 + j < n_own) { dp[k0 + j] = v[j]; } }
            }
        }
        uint64_t mm = __ballot(is_match && !own);
        while (mm != 0) {
            __threadfence_block();                         // literals, own-lane copies, the copy before
            const int ml = __builtin_ctzll(mm);
            mm &= mm - 1;
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)word, ml);
            const uint64_t at = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(o >> 32), ml) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)o, ml);
            const int len = (int)((w >> 16) & 0x1FFu);
            const int d = (int)(w & 0x7FFFu);
            const uint8_t* sp = dst + (at - (uint64_t)d);
            // out[at+k] = out[at-d+(k mod d)]: only bytes in front of `at` are read
            if (d >= len) {
                for (int k = lane; k < len; k += kWave) { dst[at + (uint64_t)k] = load_l2(sp + k); }
            } else {
                for (int k = lane; k < len; k += kWave) { dst[at + (uint64_t)k] = load_l2(sp + (k % d)); }
            }
        }
        i += total;
    }
}

void launch_entropy_decode(const uint8_t* in, const uint64_t* in_off, const uint64_t* out_off,
                           uint32_t* tokens, uint32_t* tok_count, int32_t* err, uint64_t* end_bit,
                           uint32_t n_blocks, uint64_t start_bit, int waves, hipStream_t stream) {
    if (n_blocks == 0) { return; }
    // waves per stream: 1 when the batch fills the chip by itself (16 streams per CU), 2 / 4 when it does not
    if (waves >= 8) {
        hipLaunchKernelGGL(entropy_decode_mw_kernel<8>, dim3(n_blocks), dim3(kWave * 8), 0, stream,
                           in, in_off, out_off, tokens, tok_count, err, end_bit, n_blocks, start_bit);
    } else if (waves >= 4) {
        hipLaunchKernelGGL(entropy_decode_mw_kernel<4>, dim3(n_blocks), dim3(kWave * 4), 0, stream,
                           in, in_off, out_off, tokens, tok_count, err, end_bit, n_blocks, start_bit);
    } else if (waves >= 2) {
        hipLaunchKernelGGL(entropy_decode_mw_kernel<2>, dim3(n_blocks), dim3(kWave * 2), 0, stream,
                           in, in_off, out_off, tokens, tok_count, err, end_bit, n_blocks, start_bit);
    } else {
        hipLaunchKernelGGL(entropy_decode_kernel, dim3(n_blocks), dim3(kWave), 0, stream,
                           in, in_off, out_off, tokens, tok_count, err, end_bit, n_blocks, start_bit);
    }
}

void launch_lz_expand(const uint32_t* tokens, const uint32_t* tok_count, uint8_t* out,
                      const uint64_t* out_off, uint32_t n_blocks, hipStream_t stream) {
    if (n_blocks == 0) { return; }
    hipLaunchKernelGGL(lz_expand_kernel, dim3(n_blocks), dim3(kWave), 0, stream,
                       tokens, tok_count, out, out_off, n_blocks);
}

} // namespace sqzk
