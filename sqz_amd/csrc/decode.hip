// sqz_amd/csrc/decode.hip -- decoder (gfx950).
//
// One wavefront per stream, running uniformly:
//   squeeze.h:502-551 squeeze_decompress   token loop
//   squeeze.h:429-442 squeeze_read_huffman root->leaf walk, then frequency bump
//   squeeze.h:458-474 squeeze_read_length
//   squeeze.h:476-500 squeeze_read_pos
//   squeeze.h:537-539 byte-serial overlapped copy (RLE when dist < len)
// The root->leaf walk hands level d of the path to lane d, so the frequency
// update right after it is the parallel fast path of sqz_device.h (the walk has
// already produced the chain the update needs).  Literals are byte stores by
// lane 0; a back reference is copied by all 64 lanes with the overlap rule
// out[i+k] = out[i-dist + (k mod dist)].
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

// squeeze.h:429-442; returns the leaf or -1 with err set
template <class T>
__device__ __forceinline__ int read_symbol(BitSource& r, T& t, int lane, int& err) {
    int node = T::kRoot, d = 0;
    int mine = (lane == 0) ? node : (int)kNil;
    for (;;) {
        const int bit = r.bit();
        if (r.error != 0) { err = r.error; return -1; }
        const int child = __builtin_amdgcn_readfirstlane(
            (int)(bit ? t.link[node].hi : t.link[node].lo));
        if (child == kNil) { err = kEINVAL; return -1; }
        d++;
        mine = (lane == d) ? child : mine;
        node = child;
        if (child < (int)T::kRoot) { break; }            // leaf ids < LEAVES
        if (d >= kMaxFastDepth) { break; }               // finish serially below
    }
    if (node >= (int)T::kRoot) {                         // deeper than the wave is wide
        for (;;) {
            const int bit = r.bit();
            if (r.error != 0) { err = r.error; return -1; }
            const int child = __builtin_amdgcn_readfirstlane(
                (int)(bit ? t.link[node].hi : t.link[node].lo));
            if (child == kNil) { err = kEINVAL; return -1; }
            d++;
            node = child;
            if (child < (int)T::kRoot) { break; }
            if (d >= 2 * kStack) { err = kEINVAL; return -1; }
        }
    }
    // lane d' holds depth d' (0 = root, d = leaf)
    Chain c;
    c.mine = mine;
    c.par = lane_below(mine);                            // lane d-1 holds the parent
    c.gpar = lane_below(c.par);
    c.levels = d;
    c.holds = lane <= d;
    c.active = lane >= 1 && lane <= d;
    c.has_g = lane >= 2 && lane <= d;
    t.bump_wave(node, c, lane);
    return node;
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
    r.seek(start_bit);
    if (r.error != 0) { err = r.error; }
    if (!lit.insert_wave(kLitNyt, lane)) { err = kEINVAL; }          // squeeze.h:505-506
    if (!pos.insert_wave(kPosNyt, lane)) { err = kEINVAL; }

    uint64_t i = 0;
    while (i < bytes && err == 0) {
        int s = read_symbol(r, lit, lane, err);
        if (err != 0) { break; }
        if (s == kLitNyt) {                                            // squeeze.h:512-520
            s = (int)r.get_lsb(9);
            if (r.error != 0) { err = r.error; break; }
            if (s == 256 || s >= kLitNyt) { err = kEINVAL; break; }
            const int up = __builtin_amdgcn_readfirstlane((int)lit.link[s].up);
            if (up != kNil) { err = kEINVAL; break; }
            if (!lit.insert_wave(s, lane)) { err = kE2BIG; break; }
        }
        if (s <= 0xFF) {
            if (lane == 0) { dst[i] = (uint8_t)s; }
            i++;
            continue;
        }
        int base, xb;                                                  // squeeze.h:458-474
        len_base_of(s - kSymLen0, base, xb);
        int len = base;
        if (xb != 0) {
            len += (int)r.get_lsb(xb);
            if (r.error != 0) { err = r.error; break; }
        }
        if (len < kLenMin || len > kLenMax) { err = kEINVAL; break; }
        int pk = read_symbol(r, pos, lane, err);                       // squeeze.h:476-500
        if (err != 0) { break; }
        if (pk == kPosNyt) {
            pk = (int)r.get_lsb(5);
            if (r.error != 0) { err = r.error; break; }
            if (pk >= kPosNyt) { err = kEINVAL; break; }
            const int up = __builtin_amdgcn_readfirstlane((int)pos.link[pk].up);
            if (up != kNil) { err = kEINVAL; break; }
            if (!pos.insert_wave(pk, lane)) { err = kE2BIG; break; }
        }
        pos_base_of(pk, base, xb);
        int dist = base;
        if (xb != 0) {
            dist += (int)r.get_lsb(xb);
            if (r.error != 0) { err = r.error; break; }
        }
        if ((uint64_t)dist > i || (uint64_t)len > bytes - i) { err = kEINVAL; break; }
        if (lit.fault | pos.fault) { err = kE2BIG; break; }
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
