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
// The wave runs uniformly: tokens are staged 128 at a time into LDS by all
// lanes; per symbol lane k takes level k of the leaf->root chain, so the
// Huffman code is one __ballot ("am I the hi child?") and the frequency update
// is the parallel fast path of sqz_device.h (slow path on lane 0 when the tree
// restructures).  Codes and raw bit fields are queued as (value, width) pairs;
// every ~60 fields the wave packs them in parallel (prefix sum of the widths,
// LDS atomic OR into a bit image) and stores whole 8-byte words, coalesced.
#include "sqz_device.h"
#include "sqz_kernels.h"

namespace sqzk {

constexpr int kTokStrip = 128;
constexpr int kQueue = 64;           // fields per pack
constexpr int kQueueRoom = 8;        // a token adds at most 6 fields (+2 for a 63-bit code)

struct EmitLds {
    EntropyLds entropy;
    uint64_t   field[kQueue];        // width << 32 | value   (width 1..32)
    uint64_t   image[kQueue / 2 + 2];// packed bits of one batch, stream order = MSB first
};

struct BitQueue {
    EmitLds* lds;
    uint8_t* out;        // global
    uint64_t capacity;
    uint64_t bytes;      // bytes produced so far (as the reference counts them)
    int      count;      // queued fields
    int      carry;      // bits already sitting in image[0] (0..63)
    int      error;

    // value's low `width` bits, first-out bit = most significant; width 1..32
    __device__ __forceinline__ void push32(uint32_t value, int width, int lane) {
        if (lane == 0) { lds->field[count] = ((uint64_t)(uint32_t)width << 32) | (uint64_t)value; }
        count++;
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

    // store 8-byte words [0, words) of the image (bitstream.h:33-43)
    __device__ __forceinline__ void store_words(int words, int lane) {
        if (error != 0 || words == 0) { return; }
        const uint64_t room = capacity - bytes;
        if (lane < words) {
            const uint64_t w = lds->image[lane];
            const uint64_t at = (uint64_t)lane * 8;
            if (at + 8 <= room) {
                *reinterpret_cast<uint64_t*>(out + bytes + at) = __builtin_bswap64(w);
            } else {
                for (int k = 0; k < 8; k++) {              // byte by byte up to the capacity
                    if (at + (uint64_t)k < room) { out[bytes + at + k] = (uint8_t)(w >> (56 - 8 * k)); }
                }
            }
        }
        if ((uint64_t)words * 8 <= room) { bytes += (uint64_t)words * 8; }
        else { bytes = capacity; error = kE2BIG; }
    }

    // pack the queued fields behind the carried bits, store the full words
    __device__ __forceinline__ void pack(int lane) {
        if (count == 0) { return; }
        uint32_t v = 0, n = 0;
        if (lane < count) {
            const uint64_t f = lds->field[lane];
            v = (uint32_t)f;
            n = (uint32_t)(f >> 32);
        }
        uint32_t incl = n;                                  // inclusive scan of the widths
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const uint32_t up = __shfl_up(incl, d);
            if (lane >= d) { incl += up; }
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1) + (uint32_t)carry;
        if (n != 0) {
            const uint32_t o = (uint32_t)carry + incl - n;  // first stream bit of this field
            const uint32_t w = o >> 6, s = o & 63u;
            if (s + n <= 64) {
                atomicOr(reinterpret_cast<unsigned long long*>(&lds->image[w]),
                         (unsigned long long)v << (64 - s - n));
            } else {
                const uint32_t r = s + n - 64;              // bits spilling into the next word
                atomicOr(reinterpret_cast<unsigned long long*>(&lds->image[w]),
                         (unsigned long long)(v >> r));
                atomicOr(reinterpret_cast<unsigned long long*>(&lds->image[w + 1]),
                         (unsigned long long)v << (64 - r));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const int words = (int)(total >> 6);
        store_words(words, lane);
        const uint64_t rest = lds->image[words];           // partial word becomes the new carry
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane < kQueue / 2 + 2) { lds->image[lane] = (lane == 0) ? rest : 0ull; }
        carry = (int)(total & 63u);
        count = 0;
    }

    __device__ __forceinline__ void flush(int lane) {       // bitstream.h:112-114
        pack(lane);
        if (carry > 0) { store_words(1, lane); carry = 0; }
    }
};

// code of a leaf deeper than the wave is wide (never seen in practice): serial walk
__device__ __noinline__ uint64_t deep_code(const uint64_t* link, int leaf, int& width) {
    uint64_t code = 0;
    int n = 0, a = leaf;
    for (;;) {
        const uint32_t up = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)link[a] & 0x3FFu));
        if (up == kNil || n >= 63) { break; }
        const Node pn = unpack(uni64(link[up]));
        code |= (uint64_t)(pn.hi == (uint32_t)a ? 1 : 0) << n;
        a = (int)up;
        n++;
    }
    width = n;
    return code;
}

// squeeze.h:278-288 / :300-315: symbol s through tree t with the NYT escape
template <class T>
__device__ __forceinline__ void emit_coded(BitQueue& q, T& t, int s, int nyt, int raw_bits,
                                           int lane, int& err) {
    Chain c = t.chain_up(s, lane);
    const bool unseen = (c.levels == 0);                    // node[s].bits == 0
    if (unseen) { c = t.chain_up(nyt, lane); }
    const int leaf = unseen ? nyt : s;
    uint64_t code = t.bump_wave(leaf, c, lane);             // code of the tree BEFORE the update
    int width = c.levels;                                   // ballot bits beyond `levels` are 0
    if (width >= kMaxFastDepth) { code = deep_code(t.link, leaf, width); }
    q.push(code, width, lane);
    if (unseen) {
        q.push_lsb((uint32_t)s, raw_bits, lane);
        if (!t.insert_wave(s, lane)) { err = kE2BIG; }
    }
}

// one token word -> its bit fields (squeeze.h:377-394 -> :278-315), one symbol at a time
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
        emit_coded(q, pos, pc.code, kPosNyt, 5, lane, err);
        if (pc.xbits > 0) { q.push_lsb((uint32_t)pc.extra, pc.xbits, lane); }
    }
    if (q.error != 0) { err = q.error; }
    if (lit.fault | pos.fault) { err = kE2BIG; }
}

// The token source: a 128-entry window of token words in two VGPRs (lane j holds
// entries j and 64+j), refilled with coalesced loads; a word is fetched with
// v_readlane -- no LDS round trip in front of every symbol.
//   kFromMatch = false: entry k = token word k of stage 1 (any finder)
//   kFromMatch = true : entry p = what starts at byte p according to the indexed
//                       finder's match table (kTokMatch|len<<16|dist, or the byte);
//                       the greedy step (squeeze.h:377-394) is the cursor advance
// Offsets are 32-bit: a stream is at most 2^31 bytes (include/sqz/sqz.h).
template <bool kFromMatch>
struct TokenWindow {
    const uint32_t* words;    // token words | match table
    const uint8_t*  bytes_in; // input bytes (kFromMatch)
    uint32_t total;           // tokens | bytes
    uint32_t wbase;           // entry held by lane 0 of `cur`
    uint32_t cur, nxt;

    __device__ __forceinline__ uint32_t load(uint32_t at, int lane) const {
        const uint32_t k = at + (uint32_t)lane;
        if (k >= total) { return 0u; }
        if (!kFromMatch) { return words[k]; }
        const uint32_t m = (k + 2 < total) ? words[k] : 0u;          // last 2 bytes: literals
        return m != 0 ? (kTokMatch | m) : (uint32_t)bytes_in[k];
    }
    __device__ __forceinline__ void open(uint32_t at, int lane) {
        wbase = at;
        cur = load(at, lane);
        nxt = load(at + kWave, lane);
    }
    // once per step: afterwards entries [pos, pos+64) are in the window
    __device__ __forceinline__ void cover(uint32_t pos, int lane) {
        const uint32_t d = pos - wbase;
        if (d >= 2 * kWave) { open(pos, lane); }
        else if (d >= kWave) { cur = nxt; wbase += kWave; nxt = load(wbase + kWave, lane); }
    }
    __device__ __forceinline__ bool has(uint32_t pos) const { return pos - wbase < 2 * kWave; }
    __device__ __forceinline__ uint32_t get(uint32_t pos) const {      // has(pos)
        const uint32_t d = pos - wbase;
        const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)(d & 63u));
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)(d & 63u));
        return d < kWave ? a : b;
    }
};

// what one token needs on the way out: its symbols' ids and the extra-bit fields
struct TokenPlan {
    uint32_t word;
    uint32_t lx, px;          // extra bits (value | width << 16) for length / distance
};

template <bool kFromMatch>
__global__ __launch_bounds__(kWave)
void huffman_emit_kernel(const uint32_t* __restrict__ tokens,    // token words | match table
                         const uint64_t* __restrict__ tok_off,   // = in_off
                         const uint32_t* __restrict__ tok_count, // unused when kFromMatch
                         uint32_t* __restrict__ tok_count_out,   // kFromMatch: tokens per stream (or null)
                         const uint8_t* __restrict__ in,         // input bytes (kFromMatch)
                         uint8_t* __restrict__ out,
                         const uint64_t* __restrict__ out_off,
                         uint64_t* __restrict__ out_bytes,
                         int32_t* __restrict__ err_out,
                         uint32_t n_blocks,
                         uint64_t prefix_acc, int prefix_fill) {
    __shared__ EmitLds lds;
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) { return; }

    LitTree lit; PosTree pos;
    bind(lit, pos, &lds.entropy);
    lit.init_all(lane);
    pos.init_all(lane);
    if (lane < kQueue / 2 + 2) {
        // header bits that precede the payload (single-stream API): the carry
        lds.image[lane] = (lane == 0 && prefix_fill > 0) ? (prefix_acc << (64 - prefix_fill)) : 0ull;
    }
    __syncthreads();

    const uint64_t t0 = uni64(tok_off[b]), t1 = uni64(tok_off[b + 1]);

    BitQueue q;
    q.lds = &lds;
    const uint64_t o0 = uni64(out_off[b]), o1 = uni64(out_off[b + 1]);
    q.out = out + o0;
    q.capacity = o1 - o0;
    q.bytes = 0;
    q.count = 0;
    q.carry = prefix_fill;
    q.error = 0;
    int err = 0;

    if (!lit.insert_wave(kLitNyt, lane)) { err = kEINVAL; }          // squeeze.h:333-334
    if (!pos.insert_wave(kPosNyt, lane)) { err = kEINVAL; }

    TokenWindow<kFromMatch> win;
    win.words = tokens + t0;
    win.bytes_in = kFromMatch ? in + t0 : nullptr;
    win.total = kFromMatch ? (uint32_t)(t1 - t0)
                           : (uint32_t)__builtin_amdgcn_readfirstlane((int)tok_count[b]);
    win.open(0, lane);

    uint64_t* const link = lds.entropy.lit_link;      // both trees: pos ids + kLitNodes
    uint32_t* const freq = lds.entropy.lit_freq;
    uint32_t cursor = 0, ntok = 0;
    while (cursor < win.total && err == 0) {
        win.cover(cursor, lane);
        // ---- plan up to 4 symbols (whole tokens) ------------------------------------
        TokenPlan tp[kBatch];
        int s0 = kUnifiedDummy, s1 = kUnifiedDummy, s2 = kUnifiedDummy, s3 = kUnifiedDummy;
        int nt = 0, ns = 0;
        bool open_step = true;
        uint32_t c2 = cursor;
        auto put = [&](int leaf) {          // uniform selects keep the ids in SGPRs
            s0 = ns == 0 ? leaf : s0;
            s1 = ns == 1 ? leaf : s1;
            s2 = ns == 2 ? leaf : s2;
            s3 = ns == 3 ? leaf : s3;
            ns++;
        };
#pragma unroll
        for (int j = 0; j < kBatch; j++) {
            tp[j].word = 0; tp[j].lx = 0; tp[j].px = 0;
            if (open_step && c2 < win.total && ns < kBatch && win.has(c2)) {
                const uint32_t t = win.get(c2);
                if ((t & kTokMatch) == 0) {
                    tp[j].word = t;
                    nt = j + 1;
                    put((int)(t & 0xFFu));
                    c2 += 1;
                } else if (ns + 2 <= kBatch) {
                    const Code lc = len_code((int)((t >> 16) & 0x1FFu));          // squeeze.h:290-298
                    const Code pc = pos_code((int)(t & 0x7FFFu));                 // squeeze.h:300-315
                    tp[j].word = t;
                    tp[j].lx = (uint32_t)lc.extra | ((uint32_t)lc.xbits << 16);
                    tp[j].px = (uint32_t)pc.extra | ((uint32_t)pc.xbits << 16);
                    nt = j + 1;
                    put(kSymLen0 + lc.code);
                    put(kLitNodes + pc.code);
                    c2 += kFromMatch ? ((t >> 16) & 0x1FFu) : 1u;
                } else {
                    open_step = false;       // no room for this match: close the step
                }
            } else {
                open_step = false;
            }
        }
        // ---- all of them at once, if no link can change (sqz_device.h) ---------------
        BatchOut bo;
        bool done = false;
        const bool frozen = (lit.complete | pos.complete) != 0 || lit.depth >= 63 || pos.depth >= 63;
        if (!frozen && ns > 1) {
            done = bump_batch(link, freq, s0, s1, s2, s3, ns, lane, bo);
        }
        if (done) {
            uint64_t codes = bo.code_bits;
            uint32_t depths = bo.depths;
#pragma unroll
            for (int j = 0; j < kBatch; j++) {
                if (j < nt) {
                    q.push32((uint32_t)codes & 0xFFFFu, (int)(depths & 0xFFu), lane);
                    codes >>= 16; depths >>= 8;
                    if (tp[j].word & kTokMatch) {
                        if (tp[j].lx >> 16) { q.push_lsb(tp[j].lx & 0xFFFFu, (int)(tp[j].lx >> 16), lane); }
                        q.push32((uint32_t)codes & 0xFFFFu, (int)(depths & 0xFFu), lane);
                        codes >>= 16; depths >>= 8;
                        if (tp[j].px >> 16) { q.push_lsb(tp[j].px & 0xFFFFu, (int)(tp[j].px >> 16), lane); }
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < kBatch; j++) {
                if (j < nt && err == 0) { emit_token(q, lit, pos, tp[j].word, lane, err); }
            }
        }
        cursor = c2;
        ntok += (uint32_t)nt;
        if (q.count > kQueue - 2 * kQueueRoom) { q.pack(lane); }
        if (q.error != 0) { err = q.error; }
    }
    if (kFromMatch && lane == 0 && tok_count_out != nullptr) { tok_count_out[b] = ntok; }

    if (err == 0) { q.flush(lane); err = q.error; }
    else { q.pack(lane); }
    if (lane == 0) {
        out_bytes[b] = q.bytes;
        err_out[b] = err;
    }
}

void launch_huffman_emit(const uint32_t* tokens, const uint64_t* tok_off,
                         const uint32_t* tok_count, uint8_t* out,
                         const uint64_t* out_off, uint64_t* out_bytes,
                         int32_t* err, uint32_t n_blocks,
                         uint64_t prefix_acc, int prefix_fill, hipStream_t stream) {
    if (n_blocks == 0) { return; }
    hipLaunchKernelGGL(huffman_emit_kernel<false>, dim3(n_blocks), dim3(kWave), 0, stream,
                       tokens, tok_off, tok_count, (uint32_t*)nullptr, (const uint8_t*)nullptr, out,
                       out_off, out_bytes, err, n_blocks, prefix_acc, prefix_fill);
}

void launch_huffman_emit_from_match(const uint8_t* in, const uint64_t* in_off,
                                    const uint32_t* match, uint32_t* tok_count_out, uint8_t* out,
                                    const uint64_t* out_off, uint64_t* out_bytes,
                                    int32_t* err, uint32_t n_blocks,
                                    uint64_t prefix_acc, int prefix_fill, hipStream_t stream) {
    if (n_blocks == 0) { return; }
    hipLaunchKernelGGL(huffman_emit_kernel<true>, dim3(n_blocks), dim3(kWave), 0, stream,
                       match, in_off, (const uint32_t*)nullptr, tok_count_out, in, out, out_off,
                       out_bytes, err, n_blocks, prefix_acc, prefix_fill);
}

} // namespace sqzk
