// sqz_amd/csrc/huffman_emit.hip -- encode stage 2 (gfx950).
//
// One wavefront per stream.  Consumes the LZ77 token words of stage 1 and
// produces the reference's bit stream:
//   squeeze.h:278-288 squeeze_encode_literal   (NYT escape + 9 raw bits)
//   squeeze.h:290-298 squeeze_encode_len       (symbol 257+code, extra bits)
//   squeeze.h:300-315 squeeze_encode_pos       (pos tree, NYT + 5 raw bits)
//   squeeze.h:239-246 squeeze_write_huffman    (code from the tree BEFORE the
//                                               frequency update)
//   squeeze.h:248-253 squeeze_flush
// The wave runs uniformly: tokens are staged 256 at a time into LDS by all
// lanes; per symbol lane k takes level k of the leaf->root chain, so the
// Huffman code is one __ballot ("am I the hi child?") and the frequency update
// is the parallel fast path of sqz_device.h (slow path on lane 0 when the tree
// restructures).  The bit sink state is wave-uniform; lane 0 stores the words.
#include "sqz_device.h"
#include "sqz_kernels.h"

namespace sqzk {

constexpr int kTokStrip = 256;

struct EmitLds {
    EntropyLds entropy;
    uint32_t   strip[kTokStrip];
};

// huffman code of attached leaf s, read off its chain: lane k contributes the
// branch taken at level k; ballot bit k is then bit k of the stream-order code
template <class T>
__device__ __forceinline__ void emit_attached(BitSink& w, T& t, int s, const Chain& c, int lane) {
    if (c.levels < kMaxFastDepth) {
        bool is_hi = false;
        if (c.active) { is_hi = (t.link[c.par].hi == c.mine); }
        const uint64_t code = __ballot(is_hi);
        w.put_msb(code, c.levels);
    } else {                                   // deeper than the wave is wide: serial
        uint64_t code = 0;
        int n = 0, a = s;
        for (;;) {
            const int up = __builtin_amdgcn_readfirstlane((int)t.link[a].up);
            if (up == kNil || n >= 63) { break; }
            const int hi = __builtin_amdgcn_readfirstlane((int)t.link[up].hi);
            code |= (uint64_t)(hi == a ? 1 : 0) << n;
            a = up;
            n++;
        }
        w.put_msb(code, n);
    }
    t.bump_wave(s, c, lane);                   // squeeze.h:245: after the code is out
}

// squeeze.h:278-288 / :300-315: symbol through tree t with NYT escape
template <class T>
__device__ __forceinline__ void emit_coded(BitSink& w, T& t, int s, int nyt, int raw_bits,
                                           int lane, int& err) {
    const Chain c = t.chain_up(s, lane);
    if (c.levels == 0) {                       // unseen (node[s].bits == 0)
        const Chain cn = t.chain_up(nyt, lane);
        emit_attached(w, t, nyt, cn, lane);
        w.put_lsb((uint32_t)s, raw_bits);
        if (!t.insert_wave(s, lane)) { err = kE2BIG; }
    } else {
        emit_attached(w, t, s, c, lane);
    }
}

__global__ __launch_bounds__(kWave)
void huffman_emit_kernel(const uint32_t* __restrict__ tokens,
                         const uint64_t* __restrict__ tok_off,
                         const uint32_t* __restrict__ tok_count,
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
    __syncthreads();

    const uint32_t* tok = tokens + tok_off[b];
    const uint32_t count = tok_count[b];

    BitSink w;
    w.out = out + out_off[b];
    w.capacity = out_off[b + 1] - out_off[b];
    w.bytes = 0;
    w.acc = prefix_acc;
    w.fill = prefix_fill;
    w.error = 0;
    w.writer = (lane == 0);
    int err = 0;

    if (!lit.insert_wave(kLitNyt, lane)) { err = kEINVAL; }          // squeeze.h:333-334
    if (!pos.insert_wave(kPosNyt, lane)) { err = kEINVAL; }

    for (uint32_t base = 0; base < count && err == 0; base += kTokStrip) {
        const uint32_t left = count - base;
        const uint32_t take = left < (uint32_t)kTokStrip ? left : (uint32_t)kTokStrip;
        __syncthreads();
        for (uint32_t k = lane; k < take; k += kWave) { lds.strip[k] = tok[base + k]; }
        __syncthreads();
        for (uint32_t k = 0; k < take && err == 0; k++) {
            const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds.strip[k]);
            if ((t & kTokMatch) == 0) {
                emit_coded(w, lit, (int)(t & 0xFFu), kLitNyt, 9, lane, err);
            } else {
                const int len = (int)((t >> 16) & 0x1FFu);
                const int dist = (int)(t & 0x7FFFu);
                const Code lc = len_code(len);                        // squeeze.h:290-298
                emit_coded(w, lit, kSymLen0 + lc.code, kLitNyt, 9, lane, err);
                if (lc.xbits > 0) { w.put_lsb((uint32_t)lc.extra, lc.xbits); }
                const Code pc = pos_code(dist);                       // squeeze.h:300-315
                emit_coded(w, pos, pc.code, kPosNyt, 5, lane, err);
                if (pc.xbits > 0) { w.put_lsb((uint32_t)pc.extra, pc.xbits); }
            }
            if (w.error != 0) { err = w.error; }
            if (lit.fault | pos.fault) { err = kE2BIG; }
        }
    }

    if (err == 0) { w.flush(); err = w.error; }
    if (lane == 0) {
        out_bytes[b] = w.bytes;
        err_out[b] = err;
    }
}

void launch_huffman_emit(const uint32_t* tokens, const uint64_t* tok_off,
                         const uint32_t* tok_count, uint8_t* out,
                         const uint64_t* out_off, uint64_t* out_bytes,
                         int32_t* err, uint32_t n_blocks,
                         uint64_t prefix_acc, int prefix_fill, hipStream_t stream) {
    if (n_blocks == 0) { return; }
    hipLaunchKernelGGL(huffman_emit_kernel, dim3(n_blocks), dim3(kWave), 0, stream,
                       tokens, tok_off, tok_count, out, out_off, out_bytes, err,
                       n_blocks, prefix_acc, prefix_fill);
}

} // namespace sqzk
