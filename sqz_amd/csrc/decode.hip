// sqz_amd/csrc/decode.hip -- decoder (gfx950).
//
// One wavefront per stream:
//   squeeze.h:502-551 squeeze_decompress   token loop
//   squeeze.h:429-442 squeeze_read_huffman root->leaf walk, then frequency bump
//   squeeze.h:458-474 squeeze_read_length
//   squeeze.h:476-500 squeeze_read_pos
//   squeeze.h:537-539 byte-serial overlapped copy (RLE when dist < len)
// Lane 0 walks the adaptive trees (LDS resident) and reads bits; a back
// reference is broadcast to the wave and copied by all 64 lanes with the
// overlap rule  out[i+k] = out[i-dist + (k mod dist)].
//
// Hardening (the reference only asserts): a missing
// child, a raw symbol that is out of range or already in the tree, a distance
// reaching before the stream start and a match running past the stream end
// are EINVAL; reading past the compressed bytes is E2BIG (bitstream.h:74).
#include "sqz_device.h"
#include "sqz_kernels.h"

namespace sqzk {

struct DecodeLds {
    EntropyLds entropy;
};

template <class T>
__device__ __forceinline__ int read_symbol(BitSource& r, T& t, int& err) {
    int i = T::kRoot;
    for (;;) {
        const int bit = r.bit();
        if (r.error != 0) { err = r.error; return -1; }
        const Links n = t.ld(i);
        i = bit ? n.hi : n.lo;
        if (i == kNil) { err = kEINVAL; return -1; }
        if (i < (int)T::kRoot) { break; }                 // leaf ids < LEAVES
    }
    t.bump(i);
    return i;
}

__global__ __launch_bounds__(kWave)
void decode_kernel(const uint8_t* __restrict__ in,
                   const uint64_t* __restrict__ in_off,
                   uint8_t* out,
                   const uint64_t* __restrict__ out_off,
                   int32_t* __restrict__ err_out,
                   uint64_t* __restrict__ end_bit,   // optional: bit position after the last symbol
                   uint32_t n_blocks,
                   uint64_t start_bit) {
    __shared__ DecodeLds lds;
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) { return; }

    LitTree lit; PosTree pos;
    bind(lit, pos, &lds.entropy);
    lit.init_all(lane);
    pos.init_all(lane);
    __syncthreads();

    uint8_t* dst = out + out_off[b];
    const uint64_t bytes = out_off[b + 1] - out_off[b];

    BitSource r;
    r.in = in + in_off[b];
    r.limit = in_off[b + 1] - in_off[b];
    r.error = 0;
    int err = 0;

    if (lane == 0) {
        r.seek(start_bit);
        if (r.error != 0) { err = r.error; }
        if (!lit.insert(kLitNyt)) { err = kEINVAL; }      // squeeze.h:505-506
        if (!pos.insert(kPosNyt)) { err = kEINVAL; }
    }

    uint64_t i = 0;
    for (;;) {
        // ---- lane 0: decode literals until a back reference or the end ----
        int len = 0, dist = 0;
        if (lane == 0) {
            while (i < bytes && err == 0) {
                int s = read_symbol(r, lit, err);
                if (err != 0) { break; }
                if (s == kLitNyt) {                        // squeeze.h:512-520
                    s = (int)r.get_lsb(9);
                    if (r.error != 0) { err = r.error; break; }
                    if (s == 256 || s >= kLitNyt || lit.link[s].up != kNil) {
                        err = kEINVAL; break;
                    }
                    if (!lit.insert(s)) { err = kE2BIG; break; }
                }
                if (s <= 0xFF) { dst[i++] = (uint8_t)s; continue; }
                int base, xb;                              // squeeze.h:458-474
                len_base_of(s - kSymLen0, base, xb);
                len = base;
                if (xb != 0) {
                    len += (int)r.get_lsb(xb);
                    if (r.error != 0) { err = r.error; break; }
                }
                if (len < kLenMin || len > kLenMax) { err = kEINVAL; break; }
                int pk = read_symbol(r, pos, err);         // squeeze.h:476-500
                if (err != 0) { break; }
                if (pk == kPosNyt) {
                    pk = (int)r.get_lsb(5);
                    if (r.error != 0) { err = r.error; break; }
                    if (pk >= kPosNyt || pos.link[pk].up != kNil) { err = kEINVAL; break; }
                    if (!pos.insert(pk)) { err = kE2BIG; break; }
                }
                pos_base_of(pk, base, xb);
                dist = base;
                if (xb != 0) {
                    dist += (int)r.get_lsb(xb);
                    if (r.error != 0) { err = r.error; break; }
                }
                if ((uint64_t)dist > i || (uint64_t)len > bytes - i) {
                    err = kEINVAL; break;
                }
                if (lit.fault | pos.fault) { err = kE2BIG; }
                break;                                     // hand the copy to the wave
            }
            if (err != 0) { len = 0; }
        }
        // ---- whole wave: overlapped copy ----
        len = __shfl(len, 0);
        if (len == 0) { break; }                           // end of stream or error
        dist = __shfl(dist, 0);
        i = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(i >> 32), 0) << 32) |
            (uint64_t)(uint32_t)__shfl((int)(uint32_t)i, 0);
        // the source bytes were written by this wave: drain the stores, then
        // read them back from L2 (sc1), never from a possibly stale L1 line
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        const uint8_t* src = dst + i - (uint64_t)dist;
        for (int k = lane; k < len; k += kWave) {
            const int m = (dist >= len) ? k : (k % dist);
            const uint8_t v = __hip_atomic_load(src + m, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
            dst[i + (uint64_t)k] = v;
        }
        i += (uint64_t)len;
    }

    if (lane == 0) {
        err_out[b] = err;
        if (end_bit != nullptr) { end_bit[b] = r.pos; }
    }
}

void launch_decode(const uint8_t* in, const uint64_t* in_off, uint8_t* out,
                   const uint64_t* out_off, int32_t* err, uint64_t* end_bit,
                   uint32_t n_blocks, uint64_t start_bit, hipStream_t stream) {
    if (n_blocks == 0) { return; }
    hipLaunchKernelGGL(decode_kernel, dim3(n_blocks), dim3(kWave), 0, stream,
                       in, in_off, out, out_off, err, end_bit, n_blocks, start_bit);
}

} // namespace sqzk
