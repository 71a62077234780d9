// sqz_amd/csrc/sqz_device.h -- device-side building blocks shared by the
// entropy-stage kernels (huffman_emit.hip, decode.hip), gfx950 only.
//
// Reference semantics restated here (file:line relative to
// /root/reference/attic/map_experiment):
//   squeeze.h:29-79,151-172 DEFLATE tables -> len_code()/pos_code() arithmetic
//   bitstream.h:28-63,112-114 bit packer   -> BitQueue (huffman_emit.hip)
//   bitstream.h:65-103 bit reader          -> BitSource
// The adaptive Huffman trees (huffman.h) are in sqz_tree.h.
//
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
// whole-wave helpers (DPP, no LDS traffic)

// shifts by one lane; edge lanes receive 0x3FF (the trees' "no node")
__device__ __forceinline__ int lane_above(int v) {          // lane k <- lane k+1
    return __builtin_amdgcn_update_dpp(0x3FF, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ int lane_below(int v) {          // lane k <- lane k-1
    return __builtin_amdgcn_update_dpp(0x3FF, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
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

// mask |= 1 << bit on a wave-uniform mask: one scalar op (tests/emu compiles these kernels for the
// CPU wave emulator, which has no scalar unit)
__device__ __forceinline__ void set_bit64(uint64_t& mask, uint32_t bit) {
#ifdef SQZ_WAVE_EMU
    mask |= 1ull << bit;
#else
    asm("s_bitset1_b64 %0, %1" : "+s"(mask) : "s"(bit));
#endif
}

// Workgroup barrier for waves that talk to each other through LDS ONLY.  __syncthreads() also waits for the
// wave's outstanding global stores (s_waitcnt vmcnt(0): a workgroup-scope release covers global memory too) --
// the token stores of the decoder's update wave, one to two microseconds each time; here only the LDS
// traffic has to be in (s_waitcnt lgkmcnt(0)) before the wave stands at the barrier.
__device__ __forceinline__ void lds_barrier() {
#ifdef SQZ_WAVE_EMU
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

__device__ __forceinline__ uint64_t uni64(uint64_t v) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

} // namespace sqzk
