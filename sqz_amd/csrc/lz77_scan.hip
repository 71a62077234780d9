// sqz_amd/csrc/lz77_scan.hip -- encode stage 1 (gfx950): the O(window)
// LZ77 longest-match scan + greedy parse of the reference,
//   attic/map_experiment/squeeze.h:338-358  (scan)   :377-394 (greedy step).
//
// Exact semantics restated: at token start i the
// candidates are the distances d = 1 .. min(i, window-1), nearest first; a
// candidate's length k counts equal bytes up to min(bytes-i, 257); it is taken
// iff k >= 3 && k > best  => longest match, nearest among equals; len >= 3
// emits a match and advances by len, otherwise a literal.
//
// Mapping to CDNA4: one workgroup of W wavefronts per stream (W = 4 by default,
// so the 4 streams that fit a CU's LDS put 4 waves on every SIMD; W = 1 is the
// single-wavefront form).  A sliding region of the input (window + look-ahead)
// sits in LDS, refilled with coalesced 16-byte global loads, so every input
// byte leaves HBM once.  Per token, in every wave:
//   * lane j keeps bytes i+4j..i+4j+3 of the string at i in a VGPR
//     (257 look-ahead bytes = 64 lanes x 4 + 1);
//   * the candidate sweep walks the window nearest-first in steps of 256
//     positions, wave w taking steps w, w+W, ...: every lane reads two aligned
//     LDS dwords and tests its 4 byte positions against the 3-byte prefix
//     (v_perm_b32 + v_cmp), the 4 result masks land in SGPRs (__ballot) --
//     no hit => next step;
//   * hits are first thinned in parallel (the byte at offset `best` must match,
//     a necessary condition for k > best), survivors are extended one at a
//     time by the whole wave: 64 dword compares + __ballot give k in one go;
//   * a wave stops as soon as its best == min(bytes-i, 257) (the reference's
//     early exit at 257, squeeze.h:353; stopping at the cap is equivalent
//     because only a strictly longer match replaces the best) and raises a
//     flag the other waves poll.
// The waves' results meet in LDS once per token (one barrier, double-buffered
// slots): max over (len, -dist) = longest, nearest among equals.
// Tokens are collected one per lane of wave 0 and stored 64 at a time.
#include "sqz_device.h"
#include "sqz_kernels.h"

namespace sqzk {

constexpr int kFront  = 272;                 // slack below the oldest byte
constexpr int kBack   = 288;                 // slack above the newest byte
constexpr int kRegion = 39888;               // resident input bytes (x16)
constexpr int kLdsBytes = kFront + kRegion + kBack;   // 40448 -> 4 streams/CU
constexpr int kAhead  = 264;                 // look-ahead that must be resident
constexpr int kMaxScanWaves = 8;

static_assert(kRegion % 16 == 0 && kFront % 16 == 0, "16-byte staging");
static_assert(kRegion >= kMaxWindow + kAhead + 16, "window + look-ahead must fit");

struct ScanLds {
    __attribute__((aligned(16))) uint8_t buf[kLdsBytes];
    uint32_t result[2][kMaxScanWaves];       // per token parity, per wave: len<<16 | (0xFFFF-dist)
    uint32_t stop[2];                        // a wave reached the cap
};

// pin a wave-uniform value into SGPRs (the compiler cannot always prove it)
__device__ __forceinline__ uint32_t uni(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint64_t uni(uint64_t v) {
    return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v);
}

__device__ __forceinline__ uint32_t lds_dword_at(const uint8_t* buf, int p) {
    // unaligned 4 bytes at byte index p from two aligned dwords
    const uint32_t* w = reinterpret_cast<const uint32_t*>(buf + (p & ~3));
    return __builtin_amdgcn_alignbyte(w[1], w[0], (uint32_t)(p & 3));
}

template <int W>
__global__ __launch_bounds__(W * kWave)
void lz77_scan_kernel(const uint8_t* __restrict__ in,
                      const uint64_t* __restrict__ in_off,
                      uint32_t n_blocks, uint32_t window,
                      uint32_t* __restrict__ tokens,
                      uint32_t* __restrict__ tok_count, uint64_t slots) {
    __shared__ ScanLds lds;
    uint8_t* const buf = lds.buf;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) { return; }
    if (in_off[b + 1] > slots) {                  // the caller's arrays do not reach this far: refuse the block
        if (tid == 0) { tok_count[b] = kRefused; }
        return;
    }

    const uint8_t* src = in + in_off[b];
    const uint64_t bytes = in_off[b + 1] - in_off[b];
    uint32_t* tok = tokens + in_off[b];
    const bool src_aligned = ((reinterpret_cast<uintptr_t>(src) & 15u) == 0);

    if (tid < 2) { lds.stop[tid] = 0; }

    uint64_t base = 0;       // stream offset of buf[kFront]
    uint32_t valid = 0;      // resident bytes
    uint64_t i = 0;          // token start
    uint32_t ntok = 0;
    uint32_t tok_reg = 0;    // wave 0: lane (ntok & 63) holds the pending token
    uint32_t par = 0;        // token parity (result slot)

    while (i < bytes) {
        i = uni(i); base = uni(base); valid = uni(valid); ntok = uni(ntok); par = uni(par);
        const uint32_t reach = (uint32_t)(i < (uint64_t)(window - 1) ? i : (uint64_t)(window - 1));

        // ---- keep [i-reach, i+kAhead) resident (all waves) --------------------
        if (i + kAhead > base + valid && base + valid < bytes) {
            const uint64_t new_base = (i - reach) & ~(uint64_t)15;
            const uint32_t shift = (uint32_t)(new_base - base);
            if (W > 1) { __syncthreads(); }                 // everyone is done with the old image
            if (shift > 0 && shift < valid) {
                const uint32_t keep = valid - shift;
                for (uint32_t off0 = 0; off0 < keep; off0 += W * kWave * 16) {
                    const uint32_t off = off0 + (uint32_t)tid * 16;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (off < keep) { v = *reinterpret_cast<const uint4*>(buf + kFront + shift + off); }
                    if (W > 1) { __syncthreads(); }         // reads of this chunk before its writes
                    else { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier(); }
                    if (off < keep) { *reinterpret_cast<uint4*>(buf + kFront + off) = v; }
                }
                valid = keep;
            } else if (shift >= valid) {
                valid = 0;
            }
            base = new_base;
            // the stream offset base+valid is a multiple of 16 here
            const uint64_t from = base + valid;
            const uint64_t room = (uint64_t)kRegion - valid;
            const uint64_t want = (bytes - from) < room ? (bytes - from) : room;
            const uint32_t full = (uint32_t)(want & ~(uint64_t)15);
            if (src_aligned) {
                for (uint32_t off = (uint32_t)tid * 16; off < full; off += W * kWave * 16) {
                    const uint4 v = *reinterpret_cast<const uint4*>(src + from + off);
                    *reinterpret_cast<uint4*>(buf + kFront + valid + off) = v;
                }
            } else {
                for (uint32_t off = (uint32_t)tid; off < full; off += W * kWave) {
                    buf[kFront + valid + off] = src[from + off];
                }
            }
            for (uint32_t off = full + (uint32_t)tid; off < (uint32_t)want; off += W * kWave) {
                buf[kFront + valid + off] = src[from + off];
            }
            valid += (uint32_t)want;
            __syncthreads();
        }

        const int li = kFront + (int)(i - base);             // LDS index of byte i
        const uint64_t n_left = bytes - i;
        const uint32_t cap = n_left < (uint64_t)kLenMax ? (uint32_t)n_left : (uint32_t)kLenMax;

        uint32_t best_len = 0, best_dist = 0;
        bool searched = false;

        if (cap >= (uint32_t)kLenMin && reach >= 1) {
            searched = true;
            // string at i: lane j holds bytes 4j..4j+3
            const uint32_t tgt = lds_dword_at(buf, li + 4 * lane);
            const uint32_t T = (uint32_t)__builtin_amdgcn_readfirstlane((int)tgt) & 0x00FFFFFFu;

            const int near_p = li - 1;                        // distance 1
            const int far_p  = li - (int)reach;               // distance reach
            const int a0 = near_p & ~3;
            const int steps = (a0 + 3 - far_p) / 256 + 1;
            bool done = false;

            // one step: this lane's 4 positions a..a+3 against the 3-byte prefix
            auto probe = [&](int a, bool (&c)[4]) {
                const uint32_t* w = reinterpret_cast<const uint32_t*>(buf + a);
                const uint32_t lo = w[0], hi = w[1];
                c[0] = __builtin_amdgcn_perm(hi, lo, 0x0c020100u) == T;
                c[1] = __builtin_amdgcn_perm(hi, lo, 0x0c030201u) == T;
                c[2] = __builtin_amdgcn_perm(hi, lo, 0x0c040302u) == T;
                c[3] = __builtin_amdgcn_perm(hi, lo, 0x0c050403u) == T;
            };

            // some lane of step t saw the prefix: validate, thin, extend
            auto settle = [&](int t, int a, const bool (&c)[4]) {
                // drop positions outside [far_p, near_p]; thin by the byte at
                // offset best_len (needed for a strictly longer match)
                uint32_t want_byte = 0;
                if (best_len >= (uint32_t)kLenMin) {     // best_len < cap <= 257
                    want_byte = uni((uint32_t)buf[li + (int)best_len]);
                }
                uint64_t m[4];
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const int p = a + s;
                    bool ok = c[s] && p >= far_p && p <= near_p;
                    if (best_len >= (uint32_t)kLenMin) {
                        ok = ok && (buf[ok ? p + (int)best_len : li] == want_byte);
                    }
                    m[s] = __ballot(ok);
                }
                uint64_t any = m[0] | m[1] | m[2] | m[3];
                while (any != 0 && !done) {
                    const int hl = __builtin_ctzll(any);     // nearest group first
                    any &= any - 1;
                    const int ha = a0 - 4 * (64 * t + hl);
                    for (int s = 3; s >= 0 && !done; s--) {  // nearest position first
                        if (((m[s] >> hl) & 1ull) == 0) { continue; }
                        const int p = ha + s;
                        // whole-wave extension: lane j compares bytes 4j..4j+3
                        const uint32_t x = lds_dword_at(buf, p + 4 * lane) ^ tgt;
                        const uint64_t ne = __ballot(x != 0);
                        uint32_t k;
                        if (ne == 0) {
                            k = 256 + uni((uint32_t)(buf[p + 256] == buf[li + 256]));
                        } else {
                            const int fl = __builtin_ctzll(ne);
                            const uint32_t xf = (uint32_t)__builtin_amdgcn_readlane((int)x, fl);
                            k = 4u * (uint32_t)fl + ((uint32_t)__builtin_ctz(xf) >> 3);
                        }
                        if (k > cap) { k = cap; }
                        if (k > best_len) {                   // k >= 3 by the prefix test
                            best_len = k;
                            best_dist = (uint32_t)(li - p);
                            if (best_len == cap) {
                                done = true;
                                if (W > 1 && lane == 0) { lds.stop[par] = 1; }
                            }
                        }
                    }
                }
            };

            // sweep: 4 steps per trip, the LDS reads of a trip issue together
            int t = wave;
            const int my_a = a0 - 4 * lane;
            while (t + 3 * W < steps && !done) {
                bool c[4][4];
                const int ab = my_a - 256 * t;
                uint32_t stop_now = 0;
                if (W > 1) { stop_now = lds.stop[par]; }      // another wave hit the cap?
#pragma unroll
                for (int j = 0; j < 4; j++) { probe(ab - 256 * W * j, c[j]); }
                bool hit = false;
#pragma unroll
                for (int j = 0; j < 4; j++) { hit = hit | c[j][0] | c[j][1] | c[j][2] | c[j][3]; }
                if (__ballot(hit) != 0) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (!done && __ballot(c[j][0] | c[j][1] | c[j][2] | c[j][3]) != 0) {
                            settle(t + W * j, ab - 256 * W * j, c[j]);
                        }
                    }
                    best_len = uni(best_len);
                    best_dist = uni(best_dist);
                    done = uni((uint32_t)done) != 0;
                }
                t += 4 * W;
                if (W > 1 && uni(stop_now) != 0) { done = true; }
            }
            for (; t < steps && !done; t += W) {
                bool c[4];
                const int a = my_a - 256 * t;
                probe(a, c);
                if (__ballot(c[0] | c[1] | c[2] | c[3]) != 0) {
                    settle(t, a, c);
                    best_len = uni(best_len);
                    best_dist = uni(best_dist);
                    done = uni((uint32_t)done) != 0;
                }
            }
        }

        // ---- the waves' results meet (longest, then nearest) -------------------
        if (W > 1 && searched) {
            if (lane == 0) {
                lds.result[par][wave] = best_len != 0 ? ((best_len << 16) | (0xFFFFu - best_dist)) : 0u;
            }
            if (tid == 0) { lds.stop[par ^ 1] = 0; }          // next token's flag
            __syncthreads();
            uint32_t key = 0;
#pragma unroll
            for (int w2 = 0; w2 < W; w2++) {
                const uint32_t r = lds.result[par][w2];
                key = r > key ? r : key;
            }
            key = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
            best_len = key >> 16;
            best_dist = 0xFFFFu - (key & 0xFFFFu);
            par ^= 1;
        }

        // ---- greedy step: squeeze.h:377-394 -----------------------------------
        uint32_t word;
        if (best_len >= (uint32_t)kLenMin) {
            word = kTokMatch | (best_len << 16) | best_dist;
            i += best_len;
        } else {
            word = (uint32_t)__builtin_amdgcn_readfirstlane((int)buf[li]);
            i += 1;
        }
        if (wave == 0) {
            if ((uint32_t)lane == (ntok & 63u)) { tok_reg = word; }
            if (((ntok + 1) & 63u) == 0) { tok[ntok - 63 + lane] = tok_reg; }
        }
        ntok++;
    }

    if (wave == 0) {
        if ((uint32_t)lane < (ntok & 63u)) { tok[(ntok & ~63u) + lane] = tok_reg; }
        if (lane == 0) { tok_count[b] = ntok; }
    }
}

void launch_lz77_scan(const uint8_t* in, const uint64_t* in_off, uint32_t n_blocks,
                      uint32_t window, uint32_t* tokens, uint32_t* tok_count,
                      int waves_per_stream, uint64_t slots, hipStream_t stream) {
    if (n_blocks == 0) { return; }
    switch (waves_per_stream) {
    case 1:
        hipLaunchKernelGGL(lz77_scan_kernel<1>, dim3(n_blocks), dim3(kWave), 0, stream,
                           in, in_off, n_blocks, window, tokens, tok_count, slots);
        break;
    case 2:
        hipLaunchKernelGGL(lz77_scan_kernel<2>, dim3(n_blocks), dim3(2 * kWave), 0, stream,
                           in, in_off, n_blocks, window, tokens, tok_count, slots);
        break;
    case 8:
        hipLaunchKernelGGL(lz77_scan_kernel<8>, dim3(n_blocks), dim3(8 * kWave), 0, stream,
                           in, in_off, n_blocks, window, tokens, tok_count, slots);
        break;
    default:
        hipLaunchKernelGGL(lz77_scan_kernel<4>, dim3(n_blocks), dim3(4 * kWave), 0, stream,
                           in, in_off, n_blocks, window, tokens, tok_count, slots);
        break;
    }
}

} // namespace sqzk
