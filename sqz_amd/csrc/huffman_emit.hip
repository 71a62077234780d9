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
// Tokens are fetched 256 at a time by all 64 lanes (16-byte coalesced loads)
// into an LDS staging strip; the serial adaptive-Huffman chain runs on lane 0.
#include "sqz_device.h"
#include "sqz_kernels.h"

namespace sqzk {

constexpr int kTokStrip = 256;

struct EmitLds {
    EntropyLds entropy;
    uint32_t   strip[kTokStrip];
};

__device__ __forceinline__ void emit_symbol(BitSink& w, LitTree& t, int s) {
    const Links n = t.ld(s);
    w.put_msb(t.code[s], n.bits);
    t.bump(s);
}

__device__ __forceinline__ void emit_symbol(BitSink& w, PosTree& t, int s) {
    const Links n = t.ld(s);
    w.put_msb(t.code[s], n.bits);
    t.bump(s);
}

__device__ __forceinline__ void emit_lit(BitSink& w, LitTree& lit, int s, int& err) {
    if (lit.link[s].bits == 0) {                         // squeeze.h:280
        emit_symbol(w, lit, kLitNyt);
        w.put_lsb((uint32_t)s, 9);
        if (!lit.insert(s)) { err = kE2BIG; }
    } else {
        emit_symbol(w, lit, s);
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
    int err = 0;

    if (lane == 0) {                                      // squeeze.h:333-334
        if (!lit.insert(kLitNyt)) { err = kEINVAL; }
        if (!pos.insert(kPosNyt)) { err = kEINVAL; }
    }

    for (uint32_t base = 0; base < count; base += kTokStrip) {
        // all lanes: stage the next strip of tokens
        const uint32_t left = count - base;
        const uint32_t take = left < (uint32_t)kTokStrip ? left : (uint32_t)kTokStrip;
        for (uint32_t k = lane; k < take; k += kWave) { lds.strip[k] = tok[base + k]; }
        __syncthreads();
        if (lane == 0) {
            for (uint32_t k = 0; k < take && err == 0; k++) {
                const uint32_t t = lds.strip[k];
                if ((t & kTokMatch) == 0) {
                    emit_lit(w, lit, (int)(t & 0xFFu), err);
                } else {
                    const int len = (int)((t >> 16) & 0x1FFu);
                    const int dist = (int)(t & 0x7FFFu);
                    const Code lc = len_code(len);        // squeeze.h:290-298
                    emit_lit(w, lit, kSymLen0 + lc.code, err);
                    if (lc.xbits > 0) { w.put_lsb((uint32_t)lc.extra, lc.xbits); }
                    const Code pc = pos_code(dist);       // squeeze.h:300-315
                    if (pos.link[pc.code].bits == 0) {
                        emit_symbol(w, pos, kPosNyt);
                        w.put_lsb((uint32_t)pc.code, 5);
                        if (!pos.insert(pc.code)) { err = kE2BIG; }
                    } else {
                        emit_symbol(w, pos, pc.code);
                    }
                    if (pc.xbits > 0) { w.put_lsb((uint32_t)pc.extra, pc.xbits); }
                }
                if (w.error != 0) { err = w.error; }
                if (lit.fault | pos.fault) { err = kE2BIG; }
            }
        }
        __syncthreads();
    }

    if (lane == 0) {
        if (err == 0) { w.flush(); err = w.error; }
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
