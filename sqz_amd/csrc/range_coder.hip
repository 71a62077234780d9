// sqz_amd/csrc/range_coder.hip -- the reference's HEAD ("R-era") codec on gfx950 (SURVEY.md section
// 8f-1): an adaptive order-0 range coder over byte models.  File:line = /root/reference/src/sqz.c.
//
//   struct prob_model + Fenwick tree  :398-472   -> RcModel: 256 counts in REGISTERS, 4 per lane, and
//                                                   beside them the sum of the lanes in front -- kept
//                                                   up to date by one add per symbol (the lanes behind
//                                                   the coded one), so a cumulative count is one
//                                                   v_readlane and a lookup by cumulative count one
//                                                   __ballot: no tree of partial sums, no prefix sum,
//                                                   no LDS.  RcFlag: the two-symbol literal flag, two
//                                                   wave-uniform counts
//   rc_emit / rc_encode               :474-521   -> RcEncoder::put / encode
//   rc_consume / rc_decode            :499-548   -> RcDecoder::consume / decode
//   sqz_compress as HEAD runs it      :590-743   -> rc_encode_kernel: the finders are compiled out
//                                                   (SURVEY.md section 0), every byte is a literal:
//                                                   flag 1 + the byte, then flag 0 + size 0xFF, flush
//   sqz_decompress                    :793-839   -> rc_decode_kernel, as written (back references included)
//
// One wavefront per independent stream, running uniformly: the coder's state (low, range, code) is a
// serial chain of 64-bit divisions per symbol, so the parallelism is across streams and, inside
// one, across the 256 counts of a model.  Throughput is bounded by that chain; what the batch buys is
// thousands of streams at once.  The divisions are rc_div(): a double-precision estimate made low on
// purpose, the exact remainder, a second estimate of what is left and at most two steps up -- about 60
// instructions where the compiler's 64-bit division takes 200 (it was most of the 500 per byte).
#include "sqz_device.h"
#include "sqz_kernels.h"

namespace sqzk {

constexpr int kRcEINVAL = 22, kRcERANGE = 34, kRcEILSEQ = 84, kRcENOBUFS = 105;   // Linux errno values
constexpr int kRcMinLen = 2, kRcMaxLen = 254;                                    // src/sqz.c:29-30

struct RcLds {
    uint32_t dist[32][4];                                    // pm_dist[32], two symbols each (inc/sqz/sqz.h:77)
};

// floor(a / d), exact, for wave-uniform operands (d == 0: all ones -- the callers test for that case
// themselves, as the reference's EILSEQ / EINVAL checks do).  The quotient is estimated in double precision
// with a reciprocal made LOW on purpose -- (1 - 2^-40) / d, good to 2^-41 -- so that the estimate cannot
// exceed it: q1 = floor(a rdl), the remainder r = a - q1 d is then exact, non-negative and below
// a 2^-39 + 2d < 2^26 d; the same estimate of r / d leaves less than 2d + 1: two conditional steps.  The
// reciprocal belongs to the MODEL (its total changes by one per symbol): it is made when the total moves,
// off the chain low/range -> next symbol that bounds a stream.
__device__ __forceinline__ double rc_recip_low(uint64_t d) {
    const double f = (double)d;
#ifdef SQZ_WAVE_EMU
    double x = 1.0 / f;
#else
    double x = __builtin_amdgcn_rcp(f);                      // v_rcp_f64, then two Newton steps (four fma):
    x = __builtin_fma(__builtin_fma(-f, x, 1.0), x, x);      // whatever the first estimate's accuracy, far
    x = __builtin_fma(__builtin_fma(-f, x, 1.0), x, x);      // inside the 2^-41 the margin allows
#endif
    return x * (1.0 - 0x1p-40);
}

__device__ __forceinline__ uint64_t rc_div_lanes(uint64_t a, uint64_t d, double rdl) {
    if (d == 0) { return ~0ull; }
    uint64_t q = (uint64_t)((double)a * rdl);
    uint64_t r = a - q * d;
    const uint32_t q2 = (uint32_t)((double)r * rdl);         // < 2^26
    q += q2;
    r -= (uint64_t)q2 * d;
    if (r >= d) { q++; r -= d; }
    if (r >= d) { q++; }
    return q;
}
__device__ __forceinline__ uint64_t rc_div_lanes(uint64_t a, uint32_t d, double rdl) {   // the usual case: a model's total
    if (d == 0) { return ~0ull; }
    uint64_t q = (uint64_t)((double)a * rdl);
    uint64_t r = a - ((uint64_t)(uint32_t)q * d + (((uint64_t)((uint32_t)(q >> 32) * d)) << 32));   // mod 2^64
    const uint32_t q2 = (uint32_t)((double)r * rdl);
    q += q2;
    r -= (uint64_t)q2 * d;
    if (r >= d) { q++; r -= d; }
    if (r >= d) { q++; }
    return q;
}
// (the result is wave-uniform, and said to be: what follows -- low, range, the byte loop -- runs on the scalar unit)
__device__ __forceinline__ uint64_t rc_div(uint64_t a, uint32_t d, double rdl) { return uni64(rc_div_lanes(a, d, rdl)); }
__device__ __forceinline__ uint64_t rc_div(uint64_t a, uint64_t d) { return uni64(rc_div_lanes(a, d, rc_recip_low(d))); }

struct RcModel {        // struct prob_model (inc/sqz/sqz.h:40-43) with up to 256 symbols
    uint32_t c0, c1, c2, c3;   // this lane's counts: symbols 4 lane .. 4 lane + 3
    uint32_t excl;             // the counts of the lanes in front of this one
    uint32_t total;            // pm_total_freq (:451)
    double rdl;                // rc_recip_low(total)

    __device__ __forceinline__ void init(uint32_t n, int lane) {                       // pm_init :453-458
        const uint32_t s0 = 4u * (uint32_t)lane;
        c0 = s0 < n ? 1u : 0u; c1 = s0 + 1u < n ? 1u : 0u; c2 = s0 + 2u < n ? 1u : 0u; c3 = s0 + 3u < n ? 1u : 0u;
        excl = s0 < n ? s0 : n;
        total = n;
        rdl = rc_recip_low(n);
    }
    // start (counts below sym) and size of sym: pm_sum_of / freq (:447, :510)
    __device__ __forceinline__ void span_of(uint32_t sym, int lane, uint32_t& start, uint32_t& size) const {
        (void)lane;
        const uint32_t k = sym & 3u;
        const uint32_t below = excl + (k > 0 ? c0 : 0u) + (k > 1 ? c1 : 0u) + (k > 2 ? c2 : 0u);
        const uint32_t mine = k == 0 ? c0 : k == 1 ? c1 : k == 2 ? c2 : c3;
        start = (uint32_t)__builtin_amdgcn_readlane((int)below, (int)(sym >> 2));
        size = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)(sym >> 2));
    }
    // pm_index_of (:451, ft_index_of :432-445) of sum = floor(x / unit): the symbol whose run holds it, with its
    // span.  The quotient itself is never formed: start <= floor(x / unit) < start + size is the same as
    // start unit <= x < (start + size) unit (no product exceeds total unit <= range), and every lane tests
    // its own run at once -- a 64-bit division with a divisor that is only known on the chain (two per
    // symbol was most of the decoder) becomes five multiplications per lane.  A sum at or past the total
    // (a damaged stream) is symbol 0 there (ft_index_of's -1, plus 1), not an error.
    __device__ __forceinline__ int find(uint64_t x, uint64_t unit, int lane, uint32_t& start, uint32_t& size) const {
        (void)lane;
        const uint32_t e1 = excl + c0, e2 = e1 + c1, e3 = e2 + c2, e4 = e3 + c3;
        const uint64_t p0 = (uint64_t)excl * unit, p4 = (uint64_t)e4 * unit;
        const bool here = x >= p0 && x < p4;                       // (the last run ends at total unit: sum < total)
        const uint64_t m = __ballot(here);
        if (m == 0) {
            start = 0;
            size = (uint32_t)__builtin_amdgcn_readlane((int)c0, 0);
            return x >= (uint64_t)total * unit ? 0 : -1;
        }
        const int L = __builtin_ctzll(m);
        const uint32_t k = x < (uint64_t)e1 * unit ? 0u : x < (uint64_t)e2 * unit ? 1u : x < (uint64_t)e3 * unit ? 2u : 3u;
        const uint32_t below = k == 0 ? excl : k == 1 ? e1 : k == 2 ? e2 : e3;
        const uint32_t mine = k == 0 ? c0 : k == 1 ? c1 : k == 2 ? c2 : c3;
        start = (uint32_t)__builtin_amdgcn_readlane((int)below, L);
        size = (uint32_t)__builtin_amdgcn_readlane((int)mine, L);
        return 4 * L + (int)(uint32_t)__builtin_amdgcn_readlane((int)k, L);
    }
    __device__ __forceinline__ void update(uint32_t sym, int lane) {                    // pm_update :466-472
        // (the reference stops at a total of 2^56; a stream here is shorter than 2^31 symbols)
        const uint32_t at = sym >> 2, k = sym & 3u;
        const uint32_t one = (uint32_t)lane == at ? 1u : 0u;
        c0 += k == 0 ? one : 0u; c1 += k == 1 ? one : 0u; c2 += k == 2 ? one : 0u; c3 += k == 3 ? one : 0u;
        excl += (uint32_t)lane > at ? 1u : 0u;
        total += 1u;
        rdl = rc_recip_low(total);
    }
};

struct RcFlag {         // the literal flag's model: two symbols (pm_literal, inc/sqz/sqz.h:73), wave-uniform
    uint32_t f0, f1, total;
    double rdl;
    __device__ __forceinline__ void init(uint32_t n, int lane) {
        (void)lane;
        f0 = n > 0 ? 1u : 0u; f1 = n > 1 ? 1u : 0u; total = n;
        rdl = rc_recip_low(n);
    }
    __device__ __forceinline__ void span_of(uint32_t sym, int lane, uint32_t& start, uint32_t& size) const {
        (void)lane;
        start = sym != 0 ? f0 : 0u;
        size = sym != 0 ? f1 : f0;
    }
    __device__ __forceinline__ int find(uint64_t x, uint64_t unit, int lane, uint32_t& start, uint32_t& size) const {
        (void)lane;
        const bool one = x >= (uint64_t)f0 * unit && x < (uint64_t)total * unit;   // (past the total: symbol 0, as above)
        start = one ? f0 : 0u;
        size = one ? f1 : f0;
        return one ? 1 : 0;
    }
    __device__ __forceinline__ void update(uint32_t sym, int lane) {
        (void)lane;
        f0 += sym == 0 ? 1u : 0u; f1 += sym != 0 ? 1u : 0u;
        total += 1u;
        rdl = rc_recip_low(total);
    }
};

struct RcEncoder {
    uint64_t low, range;
    uint8_t* out; uint64_t cap, written;
    uint32_t pending; int n_pending;          // lane k holds pending byte k
    int error;

    __device__ __forceinline__ void put(int lane) {                                     // rc_emit :474-479
        const uint32_t b = (uint32_t)(low >> 56);
        if (lane == n_pending) { pending = b; }
        n_pending++;
        if (n_pending == kWave) { flush(lane); }
        low <<= 8;
        range <<= 8;
    }
    __device__ __forceinline__ void flush(int lane) {
        if (lane < n_pending) {
            if (written + (uint64_t)lane < cap) { out[written + (uint64_t)lane] = (uint8_t)pending; }
        }
        if (written + (uint64_t)n_pending > cap) { error = kRcENOBUFS; }
        written += (uint64_t)n_pending;
        n_pending = 0;
    }
    __device__ __forceinline__ bool same_top() const { return (low >> 56) == ((low + range) >> 56); }   // :481-483

    template <class Model>
    __device__ __forceinline__ void encode(Model& m, uint32_t sym, int lane) {           // rc_encode :506-521
        const uint64_t total = m.total;
        uint32_t start, size;
        m.span_of(sym, lane, start, size);
        range = rc_div(range, m.total, m.rdl);
        low += (uint64_t)start * range;
        range *= (uint64_t)size;
        m.update(sym, lane);
        while (same_top()) { put(lane); }
        if (range < total + 1) {
            put(lane);
            put(lane);
            range = ~0ull - low;
        }
    }
};

__global__ __launch_bounds__(kWave)
void rc_encode_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                      uint8_t* __restrict__ out, const uint64_t* __restrict__ out_off,
                      uint64_t* __restrict__ out_bytes, int32_t* __restrict__ err_out, uint32_t n_blocks) {
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) { return; }
    RcFlag lit;
    RcModel size, byte;
    lit.init(2, lane); size.init(256, lane); byte.init(256, lane);                       // sqz_init :550-565
    const uint8_t* src = in + uni64(in_off[b]);
    const uint64_t bytes = uni64(in_off[b + 1]) - uni64(in_off[b]);
    RcEncoder rc;
    rc.low = 0; rc.range = ~0ull;                                                        // rc_init :485-490
    rc.out = out + uni64(out_off[b]); rc.cap = uni64(out_off[b + 1]) - uni64(out_off[b]);
    rc.written = 0; rc.pending = 0; rc.n_pending = 0; rc.error = 0;
    for (uint64_t i0 = 0; i0 < bytes; i0 += kWave) {                                     // :614-739, as HEAD runs it
        const uint32_t mine = i0 + (uint64_t)lane < bytes ? src[i0 + lane] : 0u;
        const int n = bytes - i0 < (uint64_t)kWave ? (int)(bytes - i0) : kWave;
        for (int k = 0; k < n; k++) {
            rc.encode(lit, 1u, lane);                                                    // :722
            rc.encode(byte, (uint32_t)__builtin_amdgcn_readlane((int)mine, k), lane);    // :723
        }
    }
    rc.encode(lit, 0u, lane);                                                            // :741-742: end of stream
    rc.encode(size, 0xFFu, lane);
    for (int k = 0; k < 8; k++) { rc.range = ~0ull; rc.put(lane); }                      // rc_flush :492-497
    rc.flush(lane);
    if (lane == 0) { out_bytes[b] = rc.written; err_out[b] = rc.error; }
}

struct RcDecoder {
    uint64_t low, range, code;
    const uint8_t* in; uint64_t avail, consumed;
    uint32_t row; uint64_t row_base;          // lane k holds in[row_base + k]
    int error;
    int dry_error;                            // what a read past the end sets (0: it reads as 0 and that is all)

    // rc.read.  Past the end of the stream a byte reads as 0 (test.c:113-122); the reference's own callback
    // ALSO sets rc.error there when its source failed (test.c:112-121: rc->error = io->error) and the
    // decoder's loop ends at its next check of it (src/sqz.c:800-802) -- dry_error is that error, handed
    // in by a caller whose source ended with one (sqz_rc_decompress), so the byte count is the reference's.
    __device__ __forceinline__ uint32_t get(int lane) {
        if (dry_error != 0 && consumed >= avail && error == 0) { error = dry_error; }
        if (consumed < row_base || consumed >= row_base + (uint64_t)kWave) {
            row_base = consumed & ~(uint64_t)(kWave - 1);
            row = row_base + (uint64_t)lane < avail ? in[row_base + lane] : 0u;
        }
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)row, (int)(consumed - row_base));
        consumed++;
        return b;
    }
    __device__ __forceinline__ void consume(int lane) {                                  // rc_consume :499-504
        code = (code << 8) + get(lane);
        low <<= 8;
        range <<= 8;
    }
    __device__ __forceinline__ bool same_top() const { return (low >> 56) == ((low + range) >> 56); }

    template <class Model>
    __device__ __forceinline__ uint32_t decode(Model& m, int lane) {                      // rc_decode :528-548
        const uint64_t total = m.total;
        if (total < 1) { error = kRcEINVAL; return 0; }
        if (range < total) {
            consume(lane);
            consume(lane);
            range = ~0ull - low;
        }
        const uint64_t unit = rc_div(range, m.total, m.rdl);           // (0 when range < total: EILSEQ below)
        uint32_t start, size;
        const int sym = m.find(code - low, unit, lane, start, size);   // (code - low) / unit, without the division
        if (sym < 0 || size == 0 || range < total) { error = kRcEILSEQ; return 0; }
        range = unit;
        low += (uint64_t)start * range;
        range *= (uint64_t)size;
        m.update((uint32_t)sym, lane);
        while (same_top()) { consume(lane); }
        return (uint32_t)sym;
    }
};

__global__ __launch_bounds__(kWave)
void rc_decode_kernel(const uint8_t* __restrict__ in, const uint64_t* __restrict__ in_off,
                      uint8_t* __restrict__ out, const uint64_t* __restrict__ out_off,
                      uint64_t* __restrict__ out_bytes, uint64_t* __restrict__ consumed_out,
                      int32_t* __restrict__ err_out, uint32_t n_blocks, int dry_error) {
    __shared__ RcLds lds;
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) { return; }
    RcFlag lit;
    RcModel size, byte, bits;
    lit.init(2, lane); size.init(256, lane); byte.init(256, lane); bits.init(32, lane);
    if (lane < 32) { lds.dist[lane][0] = 1; lds.dist[lane][1] = 1; lds.dist[lane][2] = 0; lds.dist[lane][3] = 0; }
    __syncthreads();
    uint8_t* d = out + uni64(out_off[b]);
    const uint64_t cap = uni64(out_off[b + 1]) - uni64(out_off[b]);
    RcDecoder rc;
    rc.low = 0; rc.range = ~0ull; rc.code = 0; rc.error = 0; rc.dry_error = dry_error;
    rc.in = in + uni64(in_off[b]); rc.avail = uni64(in_off[b + 1]) - uni64(in_off[b]);
    rc.consumed = 0; rc.row_base = ~0ull; rc.row = 0;
    for (int k = 0; k < 8; k++) { rc.code = (rc.code << 8) + rc.get(lane); }             // :794-797
    uint64_t i = 0;
    uint32_t pend = 0; int n_pend = 0;                       // literals wait in lane k and leave 64 at a time
    auto flush = [&]() {
        if (lane < n_pend) { d[i - (uint64_t)n_pend + (uint64_t)lane] = (uint8_t)pend; }
        n_pend = 0;
    };
    uint32_t dist_total[32];                                 // (only touched by streams with back references)
#pragma unroll
    for (int k = 0; k < 32; k++) { dist_total[k] = 2; }
    while (rc.error == 0) {                                                              // :800-837
        const uint32_t is_lit = rc.decode(lit, lane);
        if (rc.error != 0) { break; }
        if (is_lit != 0) {
            if (i < cap) {
                const uint32_t v = rc.decode(byte, lane);
                if (lane == n_pend) { pend = v; }
                n_pend++; i++;
                if (n_pend == kWave) { flush(); }
            } else { rc.error = kRcENOBUFS; }
        } else {
            const uint32_t sz = rc.decode(size, lane);
            if (sz == 0xFFu) { break; }                                                  // end of stream :808
            if (sz < (uint32_t)kRcMinLen || sz > (uint32_t)kRcMaxLen) { rc.error = kRcERANGE; }
            else {
                const uint32_t nb = rc.decode(bits, lane);
                if (rc.error != 0) { break; }
                uint32_t dd = 0;
                for (int k = 0; k + 1 < (int)nb && rc.error == 0; k++) {                 // :817-819, one two-symbol model per bit
                    // (the two-symbol arithmetic done directly on the pair's counts in LDS)
                    const uint64_t total = dist_total[k];
                    if (rc.range < total) { rc.consume(lane); rc.consume(lane); rc.range = ~0ull - rc.low; }
                    const uint64_t unit = rc_div(rc.range, total);
                    const uint64_t x = rc.code - rc.low;
                    const uint32_t f0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds.dist[k][0]);
                    const uint32_t f1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds.dist[k][1]);
                    const uint32_t bit = (x < (uint64_t)f0 * unit || x >= total * unit) ? 0u : 1u;   // (pm_index_of: past the total is symbol 0)
                    const uint32_t start = bit ? f0 : 0u, sz2 = bit ? f1 : f0;
                    if (rc.range < total) { rc.error = kRcEILSEQ; break; }
                    rc.range = unit;
                    rc.low += (uint64_t)start * rc.range;
                    rc.range *= (uint64_t)sz2;
                    if (lane == 0) { lds.dist[k][bit] += 1u; }
                    dist_total[k] += 1u;
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                    while (rc.same_top()) { rc.consume(lane); }
                    dd |= bit << k;
                }
                if (nb > 0) { dd |= nb < 32u ? (1u << nb) : 0u; }                         // :821
                if (rc.error == 0) {
                    const uint64_t n = i + sz;
                    if (i < (uint64_t)dd) { rc.error = kRcERANGE; }
                    else if (n <= cap) {                                                 // :826-830 byte-serial, overlap allowed
                        flush();
                        __threadfence_block();
                        for (uint64_t k = i; k < n; k++) {                               // (the wave stays uniform: every lane fences)
                            if (lane == 0) { d[k] = __hip_atomic_load(d + k - dd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                            __threadfence_block();
                        }
                        i = n;
                    } else { rc.error = kRcENOBUFS; }
                }
            }
        }
    }
    flush();
    if (lane == 0) {
        out_bytes[b] = i;
        err_out[b] = rc.error;
        if (consumed_out != nullptr) { consumed_out[b] = rc.consumed; }
    }
}

void launch_rc_encode(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                      uint64_t* out_bytes, int32_t* err, uint32_t n_blocks, hipStream_t stream) {
    if (n_blocks == 0) { return; }
    hipLaunchKernelGGL(rc_encode_kernel, dim3(n_blocks), dim3(kWave), 0, stream, in, in_off, out, out_off, out_bytes, err, n_blocks);
}

void launch_rc_decode(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                      uint64_t* out_bytes, uint64_t* consumed, int32_t* err, uint32_t n_blocks, int dry_error,
                      hipStream_t stream) {
    if (n_blocks == 0) { return; }
    hipLaunchKernelGGL(rc_decode_kernel, dim3(n_blocks), dim3(kWave), 0, stream, in, in_off, out, out_off, out_bytes, consumed, err,
                       n_blocks, dry_error);
}

} // namespace sqzk
