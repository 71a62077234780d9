// sqz_amd/csrc/lz77_index.hip -- encode stage 1, indexed form (gfx950).
//
// Same result as the brute-force scan of lz77_scan.hip (and therefore as the
// reference, attic/map_experiment/squeeze.h:338-358,377-394), found without
// visiting every distance: a candidate can only matter if it shares the first
// 3 bytes with the string at i (squeeze_deflate_len_min = 3, squeeze.h:13), so
// it is enough to visit, nearest first, the earlier positions with the same
// 3-byte prefix.  This is SURVEY.md section 8f-3 ("faster exact match
// finders ... every in-window candidate sharing the 3-byte prefix visited
// nearest-first with strict >"), validated like bst.c:254-308 validates its
// finder: token-for-token equality with the brute-force scan (tests).
//
// Three kernels:
//   index_sort_kernel   one 1024-thread workgroup per stream: stable LSD radix
//                       sort (3 passes x 8 bits) of the positions 0..n-3 by their
//                       3-byte prefix; the three histograms come from one sweep over
//                       the bytes, every pass orders 4096-element tiles in LDS so that
//                       a digit's elements leave as contiguous runs.  Equal prefixes
//                       end up adjacent, positions ascending inside a run.
//   index_match_kernel  one thread per position (all positions, not only token
//                       starts -- there is no serial dependence here): walk the
//                       run backwards = nearest first, stop at distance
//                       min(i, window-1), keep the first strictly longer match,
//                       stop at len == min(bytes-i, 257).  -> match[i].  A wave whose
//                       64 ranks lie inside one run walks the candidates once for
//                       all its lanes; a stream's workgroups share one XCD.
//   index_parse_kernel  one wavefront per stream: the greedy step
//                       (squeeze.h:377-394) over match[] -> the token words of
//                       stage 1 (same format as lz77_scan.hip), found by per-chunk
//                       walks that merge with the real path instead of one serial walk.
#include "sqz_device.h"
#include "sqz_kernels.h"

namespace sqzk {

constexpr int kSharedGiveUp = 8;         // index_match: lanes gone their own way before a wave stops sharing its walk
constexpr int kXcds = 8;                 // MI355X: 8 XCDs, workgroups dispatched round-robin over them
constexpr int kSortThreads = 1024;
constexpr int kSortWaves = kSortThreads / kWave;

constexpr int kSortTile = 4 * kSortThreads;      // elements per workgroup tile (4 rows of 64 per wave)

// The three digit histograms do not depend on the order of the elements (a position's key
// bytes are fixed), so one sweep over the stream's BYTES gives all three; each pass is then a
// single sweep over the elements.  The scatter does not write elements where they fall: every
// 4096-element tile is first ordered by digit in LDS (stable: wave order, then row and lane order
// inside a wave), then copied out, so a digit's elements of one tile leave as one contiguous run.
// Writing each element straight to its bucket kept ~256 half-filled lines open per wave -- 16 MB
// of open lines per XCD against 4 MB of L2 -- and they left as partial-line writes: 53 GB written
// for 12 GB of elements.
//
// Which key bits a pass sorts by is free: all that is needed is that equal 24-bit keys end up
// adjacent with positions ascending (a stable LSD sort over ANY split of the 24 bits).  Blocks of
// up to 256 KB (positions < 2^18) use 10 + 7 + 7 bits: pass 0 reads the positions in order, so the
// whole key is at hand, and the 14 bits it does not sort by travel in the element's top bits -- the
// later passes never gather a byte at random (those gathers were half of the kernel's 28.8 GB of
// fetches: 64 streams of 256 KB per XCD against 4 MB of L2).  Longer blocks use 8 + 8 + 8: pass 1
// gathers bytes p and p+1 and carries byte 0 (blocks up to 16 MB), pass 2 of longer ones gathers.
constexpr int kSortBins = 1024;                  // most digits of a pass
struct SortLds {
    uint32_t elem[kSortTile];                    // the tile in digit order
    uint16_t dig[kSortTile];                     // digit per tile slot
    uint16_t cnt[kSortWaves][kSortBins];         // per tile: elements of (wave, digit) so far -> offset inside the digit's run
    uint32_t gbase[3][kSortBins];                // per pass: where the next tile's run of digit d goes
    uint32_t tstart[kSortBins];                  // tile slot where digit d's run starts
    uint32_t total[kSortBins];
};                                               // 76 KB: two workgroups per CU

struct __attribute__((packed)) U32u { uint32_t v; };

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
    return reinterpret_cast<const U32u*>(p)->v;
}

// lanes of this wave with the same digit (of `bits` bits) and valid, as a 64-bit mask
__device__ __forceinline__ uint64_t peers_of(uint32_t digit, bool valid, int bits) {
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 10; bit++) {
        if (bit < bits) {
            const bool set = (digit >> bit) & 1u;
            const uint64_t m = __ballot(set);
            peers &= set ? m : ~m;
        }
    }
    return peers;
}

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {   // popcount of mask below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__global__ __launch_bounds__(kSortThreads)
void index_sort_kernel(const uint8_t* __restrict__ in,
                       const uint64_t* __restrict__ in_off,
                       uint32_t n_blocks,
                       uint32_t* __restrict__ buf_a,      // result lands here
                       uint32_t* __restrict__ buf_b,
                       uint32_t* __restrict__ tmp,        // 4 bytes per position of scratch
                       uint64_t slots) {
    __shared__ SortLds lds;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks || in_off[b + 1] > slots) { return; }   // beyond the caller's arrays: index_parse refuses the block
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const uint8_t* src = in + in_off[b];
    const uint64_t bytes = in_off[b + 1] - in_off[b];
    if (bytes < 3) { return; }
    const uint32_t count = (uint32_t)(bytes - 2);             // positions with a 3-byte prefix
    uint32_t* const pa = buf_a + in_off[b];
    uint32_t* const pb = buf_b + in_off[b];
    (void)tmp;                                                // (was the per-row byte cache of the counting sweeps)
    // the key of position k as one number: byte k first (most significant), byte k+2 last
    auto key_of = [&](uint32_t k) {
        return (uint64_t)k + 4 <= bytes ? (__builtin_bswap32(load_u32_unaligned(src + k)) >> 8)
                                        : (((uint32_t)src[k] << 16) | ((uint32_t)src[k + 1] << 8) | (uint32_t)src[k + 2]);
    };
    const bool small = bytes <= (1u << 18);                   // the unsorted key bits fit above the position
    const bool carry = bytes <= (1u << 24);                   // (8 + 8 + 8) byte 0 fits above the position
    // pass p sorts by key bits [shift[p], shift[p] + width[p])
    const int w0 = small ? 10 : 8, w1 = small ? 7 : 8, w2 = small ? 7 : 8;
    const int s1 = w0, s2 = w0 + w1;
    const uint32_t m0 = (1u << w0) - 1u, m1 = (1u << w1) - 1u;
    // ---- all three histograms from the bytes ------------------------------------------------
    for (int d = tid; d < 3 * kSortBins; d += kSortThreads) { (&lds.gbase[0][0])[d] = 0; }
    __syncthreads();
    for (uint32_t k = (uint32_t)tid; k < count; k += (uint32_t)kSortThreads) {
        const uint32_t key = key_of(k);
        atomicAdd(&lds.gbase[0][key & m0], 1u);
        atomicAdd(&lds.gbase[1][(key >> s1) & m1], 1u);
        atomicAdd(&lds.gbase[2][key >> s2], 1u);
    }
    __syncthreads();
    if (wave < 3) {                                        // exclusive scans: counts -> bases
        uint32_t* const g = lds.gbase[wave];
        const int per = (1 << (wave == 0 ? w0 : wave == 1 ? w1 : w2)) / kWave;     // 16, 4 or 2 digits per lane
        uint32_t sum = 0;
        for (int j = 0; j < per; j++) { sum += g[per * lane + j]; }
        uint32_t excl = wave_scan(sum) - sum;
        for (int j = 0; j < per; j++) { const uint32_t v = g[per * lane + j]; g[per * lane + j] = excl; excl += v; }
    }
    __syncthreads();

    for (int pass = 0; pass < 3; pass++) {
        // pass 0: identity -> A ; pass 1: A -> B ; pass 2: B -> A
        const uint32_t* from = pass == 1 ? pa : pb;
        uint32_t* to = pass == 1 ? pb : pa;
        uint32_t* const gbase = lds.gbase[pass];
        const int bits = pass == 0 ? w0 : pass == 1 ? w1 : w2;
        const int bins = 1 << bits;

        // element k of the input order belongs to tile k / 4096, wave (k / 256) % 16, row (k / 64) % 4
        auto load_rows = [&](uint32_t tile, uint32_t (&elem)[4], uint32_t (&digit)[4], bool (&valid)[4]) {
            uint32_t pos[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t k = tile + (uint32_t)(wave * 256 + j * kWave + lane);
                valid[j] = k < count;
                pos[j] = valid[j] ? (pass == 0 ? k : from[k]) : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                digit[j] = 0; elem[j] = pos[j];
                if (!valid[j]) { continue; }
                if (small) {
                    if (pass == 0) {
                        const uint32_t key = key_of(pos[j]);
                        digit[j] = key & m0;
                        elem[j] = pos[j] | ((key >> 10) << 18);          // 14 key bits above an 18-bit position
                    } else if (pass == 1) {
                        digit[j] = (pos[j] >> 18) & m1;
                    } else {
                        digit[j] = pos[j] >> 25;
                        elem[j] = pos[j] & 0x3FFFFu;
                    }
                } else if (pass == 0) {
                    digit[j] = src[pos[j] + 2];
                } else if (pass == 1) {                              // the one random gather: bytes p and p+1
                    const uint32_t w = (uint32_t)src[pos[j]] | ((uint32_t)src[pos[j] + 1] << 8);
                    digit[j] = w >> 8;
                    if (carry) { elem[j] = pos[j] | ((w & 0xFFu) << 24); }
                } else {
                    digit[j] = carry ? (pos[j] >> 24) : (uint32_t)src[pos[j]];
                    if (carry) { elem[j] = pos[j] & 0x00FFFFFFu; }
                }
            }
        };

        // ---- scatter, tile by tile ------------------------------------------------------
        for (uint32_t tile = 0; tile < count; tile += (uint32_t)kSortTile) {
            const uint32_t n_tile = count - tile < (uint32_t)kSortTile ? count - tile : (uint32_t)kSortTile;
            for (int d = lane; d < bins; d += kWave) { lds.cnt[wave][d] = 0; }
            uint32_t elem[4], digit[4], wrank[4];
            bool valid[4];
            load_rows(tile, elem, digit, valid);
#pragma unroll
            for (int j = 0; j < 4; j++) {                      // rows in order: stability
                const uint64_t peers = peers_of(digit[j], valid[j], bits);
                const uint32_t rank = lanes_below(peers);
                wrank[j] = valid[j] ? (uint32_t)lds.cnt[wave][digit[j]] + rank : 0u;
                __builtin_amdgcn_wave_barrier();               // all reads before the leaders' writes
                if (valid[j] && rank == 0) {
                    lds.cnt[wave][digit[j]] = (uint16_t)(lds.cnt[wave][digit[j]] + (uint32_t)__builtin_popcountll(peers));
                }
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
            uint32_t tcount = 0;
            if (tid < bins) {                                  // offsets of the waves inside a digit's run
                for (int w = 0; w < kSortWaves; w++) {
                    const uint32_t c = lds.cnt[w][tid];
                    lds.cnt[w][tid] = (uint16_t)tcount;
                    tcount += c;
                }
                lds.total[tid] = tcount;
            }
            __syncthreads();
            if (wave == 0) {                                   // where each digit's run starts in the tile
                const int per = bins / kWave;
                uint32_t sum = 0;
                for (int j = 0; j < per; j++) { sum += lds.total[per * lane + j]; }
                uint32_t excl = wave_scan(sum) - sum;
                for (int j = 0; j < per; j++) { lds.tstart[per * lane + j] = excl; excl += lds.total[per * lane + j]; }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (valid[j]) {
                    const uint32_t slot = lds.tstart[digit[j]] + (uint32_t)lds.cnt[wave][digit[j]] + wrank[j];
                    if (slot < (uint32_t)kSortTile) {          // always: slots are a permutation of [0, n_tile)
                        lds.elem[slot] = elem[j];
                        lds.dig[slot] = (uint16_t)digit[j];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; j++) {                      // runs leave contiguously
                const uint32_t slot = (uint32_t)(j * kSortThreads + tid);
                if (slot < n_tile) {
                    const uint32_t d = lds.dig[slot];
                    const uint32_t dest = gbase[d] + (slot - lds.tstart[d]);
                    if (dest < count) { to[dest] = lds.elem[slot]; }   // always: a permutation of [0, count)
                }
            }
            __syncthreads();
            if (tid < bins) { gbase[tid] += tcount; }
        }
        // the next pass reads what other waves of this workgroup wrote
        __threadfence_block();
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void index_match_kernel(const uint8_t* __restrict__ in,
                        const uint64_t* __restrict__ in_off,
                        uint32_t n_blocks, uint32_t window,
                        const uint32_t* __restrict__ sorted,
                        uint32_t* __restrict__ match, uint32_t groups, uint64_t slots) {
    // A stream's workgroups all run on ONE XCD, one stream after the other: workgroup k is
    // dispatched to XCD k % 8, so stream b = 8 * (k / 8 / groups) + k % 8.  Its bytes and
    // sorted positions (~1.3 MB for 256 KB) then stay in that XCD's 4 MB L2 while they are
    // gathered at random.  Measured (PMC, 1 GiB batch): FETCH 185 GB with the stream on the
    // fast grid axis, 16.3 GB with consecutive workgroups sharing a stream across all eight
    // L2s, 3.5 GB with this mapping (61 -> 38 ms).  The 4-byte scatter into match[] is NOT
    // helped by it: WRITE_SIZE stayed at 41-47 GB for 4 GB of match words.  Measured ceiling
    // of fixing that: the same kernel storing to match[rank] (contiguous) takes 30 ms.
    const uint32_t xcd = blockIdx.x % (uint32_t)kXcds;
    const uint32_t local = blockIdx.x / (uint32_t)kXcds;
    const uint32_t b = (local / groups) * (uint32_t)kXcds + xcd;
    const uint32_t group = local % groups;
    if (b >= n_blocks || in_off[b + 1] > slots) { return; }
    const uint8_t* src = in + in_off[b];
    const uint64_t bytes = in_off[b + 1] - in_off[b];
    if (bytes < 3) { return; }
    const uint32_t n = (uint32_t)bytes;
    const uint32_t count = n - 2;
    const uint32_t* S = sorted + in_off[b];
    uint32_t* M = match + in_off[b];

    const int lane = (int)(threadIdx.x & (kWave - 1));
    // A wave owns a CONTIGUOUS run of pages (a page = 64 consecutive ranks), so that the page it has just
    // finished is still in its registers when the next one looks back across the page's first rank.
    const uint32_t pages = (count + (uint32_t)kWave - 1u) / (uint32_t)kWave;
    const uint32_t waves = groups * (blockDim.x / (uint32_t)kWave);
    const uint32_t per_wave = (pages + waves - 1u) / waves;
    const uint32_t wave_id = group * (blockDim.x / (uint32_t)kWave) + (threadIdx.x - (uint32_t)lane) / (uint32_t)kWave;
    const uint32_t page_lo = wave_id * per_wave;
    const uint32_t page_hi = page_lo + per_wave < pages ? page_lo + per_wave : pages;
    // a position's first 16 bytes (zeros beyond the stream's end: never compared, lengths stop at cap <= n - i)
    auto load16 = [&](uint32_t at, uint32_t& w0, uint32_t& w1, uint32_t& w2, uint32_t& w3) {
        if (at + 16 <= n) {
            w0 = load_u32_unaligned(src + at); w1 = load_u32_unaligned(src + at + 4);
            w2 = load_u32_unaligned(src + at + 8); w3 = load_u32_unaligned(src + at + 12);
        } else {
            uint32_t w[4] = {0, 0, 0, 0};
            for (uint32_t k = 0; k < 16 && at + k < n; k++) { w[k >> 2] |= (uint32_t)src[at + k] << (8 * (k & 3)); }
            w0 = w[0]; w1 = w[1]; w2 = w[2]; w3 = w[3];
        }
    };
    // the page in front of the current one: positions and their first 8 bytes (lane = rank inside the page)
    uint32_t prev_i = 0, prev0 = 0, prev1 = 0, prev2 = 0, prev3 = 0;
    if (page_lo > 0 && page_lo < page_hi) {                  // (every rank of an earlier page exists)
        prev_i = S[(page_lo - 1u) * (uint32_t)kWave + (uint32_t)lane];
        load16(prev_i, prev0, prev1, prev2, prev3);
    }
    for (uint32_t pg = page_lo; pg < page_hi; pg++) {
        const uint32_t r0 = pg * (uint32_t)kWave;
        const bool have_prev = pg > 0;
        const uint32_t r = r0 + (uint32_t)lane;              // a wave owns 64 consecutive ranks
        const bool valid = r < count;
        const uint32_t i = valid ? S[r] : 0u;
        const uint32_t cap = (n - i) < (uint32_t)kLenMax ? (n - i) : (uint32_t)kLenMax;
        const uint32_t reach = i < window - 1 ? i : window - 1;
        uint32_t own0 = 0, own1 = 0, own2 = 0, own3 = 0;
        if (valid) { load16(i, own0, own1, own2, own3); }
        const uint32_t key = own0 & 0x00FFFFFFu;             // i <= n - 3: the key's bytes are the stream's
        uint32_t best = 0, dist = 0;
        // one lane on its own: the candidates of ranks q_from-1, q_from-2, ... (nearest first).
        // Where the walk ends -- the first rank of the run, or the first candidate within reach, whichever is
        // later -- is found FIRST (ranks are in position order inside a run, so "same key and within reach" is
        // monotone: doubling steps, then bisection), and the lane's own byte at the length to beat is kept in
        // a register: what is left per candidate is its position (neighbouring lanes read neighbouring ranks)
        // and ONE scattered byte, where it used to be three scattered loads.  These walks are gather-bound
        // (executables: a thousand candidates per position).
        auto walk = [&](uint32_t q_from) {
            if (q_from == 0 || best >= cap) { return; }
            auto inside = [&](uint32_t r) {                    // r < q_from <= own rank: S[r] < i when the key is the same
                const uint32_t p = S[r];
                if (p + 4 > n) { return false; }               // (another key's position may end the stream)
                return (load_u32_unaligned(src + p) & 0x00FFFFFFu) == key && i - p <= reach;
            };
            uint32_t q_lo;                                     // candidates are ranks [q_lo, q_from)
            {
                uint32_t d = 4, bad = 0, good = q_from;       // ranks < bad... : `bad - 1` is outside or bad == 0
                bool open = true;
                while (open) {
                    if (d >= q_from) { bad = 0; open = false; if (inside(0)) { good = 0; } else { bad = 1; } }
                    else if (inside(q_from - d)) { good = q_from - d; d <<= 1; }
                    else { bad = q_from - d + 1; open = false; }
                }
                // first inside rank is in [bad, good]
                uint32_t lo = bad, hi = good;
                while (lo < hi) {
                    const uint32_t mid = lo + ((hi - lo) >> 1);
                    if (inside(mid)) { hi = mid; } else { lo = mid + 1; }
                }
                q_lo = lo;
            }
            uint8_t own_next = best >= (uint32_t)kLenMin ? src[i + best] : (uint8_t)0;   // best < cap: i + best < n
            for (uint32_t q = q_from; q > q_lo && best < cap; ) {
                q--;
                const uint32_t p = S[q];
                if (best >= (uint32_t)kLenMin && src[p + best] != own_next) { continue; }
                uint32_t k = 3;
                while (k < cap) {
                    if (i + k + 4 <= n) {
                        const uint32_t x = load_u32_unaligned(src + p + k) ^ load_u32_unaligned(src + i + k);
                        if (x != 0) { k += (uint32_t)__builtin_ctz(x) >> 3; break; }
                        k += 4;
                    } else {
                        if (src[p + k] != src[i + k]) { break; }
                        k++;
                    }
                }
                if (k > cap) { k = cap; }
                if (k > best) {                                // strictly longer: nearest among equals
                    best = k; dist = i - p;
                    if (best < cap) { own_next = src[i + best]; }
                }
            }
        };
        const uint32_t key0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
        if (__ballot(valid && key == key0 && i + 16 <= n) == ~0ull) {
            // ---- the whole wave sits inside one run of equal keys (a frequent 3-byte string) ---
            // Its lanes walk the same candidates, one rank apart.  Walk them ONCE: every lane compares the
            // candidate with its own first 16 bytes, held in registers.  Same order (nearest first), same strict >.
            bool done = false, deferred = false;
            uint32_t resume = 0;
            // page = sorted positions of ranks [page_base, +64) AND their first 16 bytes, one rank per lane: a
            // candidate's position and bytes come out of these registers (v_readlane) -- one gather per 64
            // candidates instead of a dependent load per candidate (each turn of this loop used to wait for it)
            uint32_t page_base = r0, page = i, pb0 = own0, pb1 = own1, pb2 = own2, pb3 = own3;
            for (int64_t c = (int64_t)r0 + kWave - 2; c >= 0; c--) {
                if (c < (int64_t)page_base) {
                    page_base -= (uint32_t)kWave;            // r0 is a multiple of 64: so is every page
                    page = S[page_base + (uint32_t)lane];
                    load16(page, pb0, pb1, pb2, pb3);        // (zeros beyond the stream's end)
                }
                const int at = (int)(c - (int64_t)page_base);
                const uint32_t pc = (uint32_t)__builtin_amdgcn_readlane((int)page, at);
                const bool below = (int64_t)r > c;           // the candidate comes before my position
                const uint32_t d = i - pc;
                if (below && d > reach) { done = true; }     // everything further is farther
                if (__ballot(!done) == 0) { break; }
                // its key first: a rank in front of the run can be any position
                const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)pb0, at);
                if ((c0 & 0x00FFFFFFu) != key0) { break; }   // left the run: so have all earlier ranks
                const uint32_t c1 = (uint32_t)__builtin_amdgcn_readlane((int)pb1, at);
                const uint32_t c2 = (uint32_t)__builtin_amdgcn_readlane((int)pb2, at);
                const uint32_t c3 = (uint32_t)__builtin_amdgcn_readlane((int)pb3, at);
                const uint32_t x0 = own0 ^ c0, x1 = own1 ^ c1, x2 = own2 ^ c2, x3 = own3 ^ c3;
                uint32_t len = x0 != 0 ? ((uint32_t)__builtin_ctz(x0) >> 3)
                             : x1 != 0 ? 4u + ((uint32_t)__builtin_ctz(x1) >> 3)
                             : x2 != 0 ? 8u + ((uint32_t)__builtin_ctz(x2) >> 3)
                             : x3 != 0 ? 12u + ((uint32_t)__builtin_ctz(x3) >> 3) : 16u;
                const bool want = below && !done;
                if (want && len == 16u && cap > 16u) {       // longer than the registers hold:
                    deferred = true;                         // this lane goes on by itself from here
                    resume = (uint32_t)c + 1u;               // (a long compare inside this loop would
                    done = true;                             // hold up the other 63 lanes)
                } else {
                    if (len > cap) { len = cap; }
                    if (want && len > best) {                // strictly longer: nearest among equals
                        best = len; dist = d;
                        if (best >= cap) { done = true; }
                    }
                }
                // a run of long matches (padding, repeated records): nearly every lane ends up on
                // its own anyway, so stop sharing early instead of trickling them out one per turn
                if (__builtin_popcountll(__ballot(deferred)) >= kSharedGiveUp) {
                    if (!done) {
                        deferred = true;
                        resume = below ? (uint32_t)c : r;    // ranks [c, r) are behind me / nothing is
                        done = true;
                    }
                    break;
                }
            }
            if (deferred) { walk(resume); }
            M[i] = best >= (uint32_t)kLenMin ? ((best << 16) | dist) : (key & 0xFFu);   // no match: the literal itself
            prev_i = i; prev0 = own0; prev1 = own1; prev2 = own2; prev3 = own3;
            continue;
        }
        // ---- several runs in one wave (the usual case: 2.7 candidates per position on Zipf bytes) ------
        // The candidates of rank r are the ranks r-1, r-2, ... of its run, nearest first -- and those are
        // the positions the NEIGHBOURING LANES hold, in this page or in the one before it (kept from the
        // previous turn of the loop).  A candidate's position and bytes come out of lane l-k's registers
        // through the LDS crossbar (ds_bpermute): no gather and no dependent load per candidate (a lane that
        // walks by itself pays three dependent global loads per candidate, and one such lane holds up its
        // wave: with per-lane walks the kernel ran at 13 us per page).  Same order, same strict >.  What the
        // registers cannot settle leaves the loop and is finished by the lane itself with walk(): a match
        // longer than the 16 bytes held, and a run that reaches back more than 64 ranks.
        bool done = !valid;
        bool later = false;                                      // this lane finishes by itself, from rank `resume` down
        uint32_t resume = 0;
        for (int k = 1; k <= kWave; k++) {
            const int from = lane - k;
            const bool in_cur = from >= 0;
            if (!done && !in_cur && !have_prev) { done = true; }  // nothing lies in front of rank 0
            if (__ballot(!done) == 0) { break; }
            const int addr = (from & (kWave - 1)) << 2;
            uint32_t pc = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)i);
            uint32_t c0 = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)own0);
            uint32_t c1 = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)own1);
            const bool back = __ballot(!done && !in_cur) != 0;   // someone looks into the page before
            if (back) {
                const uint32_t qc = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)prev_i);
                const uint32_t q0 = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)prev0);
                const uint32_t q1 = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)prev1);
                pc = in_cur ? pc : qc; c0 = in_cur ? c0 : q0; c1 = in_cur ? c1 : q1;
            }
            const uint32_t x0 = own0 ^ c0, x1 = own1 ^ c1;
            if (!done && (x0 & 0x00FFFFFFu) != 0) { done = true; }           // left the run: so have all earlier ranks
            const uint32_t d = i - pc;
            if (!done && d > reach) { done = true; }                        // everything further is farther
            uint32_t len = x0 != 0 ? ((uint32_t)__builtin_ctz(x0) >> 3) : x1 != 0 ? 4u + ((uint32_t)__builtin_ctz(x1) >> 3) : 8u;
            if (__ballot(!done && len == 8u && cap > 8u) != 0) {             // (rare on Zipf bytes: the other two words)
                uint32_t c2 = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)own2);
                uint32_t c3 = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)own3);
                if (back) {
                    const uint32_t q2 = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)prev2);
                    const uint32_t q3 = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)prev3);
                    c2 = in_cur ? c2 : q2; c3 = in_cur ? c3 : q3;
                }
                const uint32_t x2 = own2 ^ c2, x3 = own3 ^ c3;
                if (len == 8u) {
                    len = x2 != 0 ? 8u + ((uint32_t)__builtin_ctz(x2) >> 3) : x3 != 0 ? 12u + ((uint32_t)__builtin_ctz(x3) >> 3) : 16u;
                }
            }
            if (!done) {
                if (len == 16u && cap > 16u) {                   // longer than the registers hold: this lane goes on
                    later = true;                                // by itself, starting with this very candidate
                    resume = r - (uint32_t)k + 1u;
                    done = true;
                } else {
                    if (len > cap) { len = cap; }
                    if (len > best) {                            // strictly longer: nearest among equals
                        best = len; dist = d;
                        if (best >= cap) { done = true; }
                    }
                    if (!done && k == kWave) {                   // 64 ranks back and still inside the run
                        later = true;
                        resume = r - (uint32_t)kWave;
                        done = true;
                    }
                }
            }
        }
        if (later) { walk(resume); }
        if (valid) { M[i] = best >= (uint32_t)kLenMin ? ((best << 16) | dist) : (key & 0xFFu); }   // no match: the literal itself
        prev_i = i; prev0 = own0; prev1 = own1; prev2 = own2; prev3 = own3;
    }
}

// ---------------------------------------------------------------------------
// Greedy parse (squeeze.h:377-394) over the match table, without the serial walk.
// Token starts are the positions reachable from 0 through next(i) = i + (match ? len : 1).
// A tile of 64 chunks is staged in LDS; lane l first walks chunk l from its first position
// on its own (a guess: the real way in is not known yet).  Two walks that ever land on the
// same position are the same from there on, so the real path through a chunk -- entered
// where the previous chunk's path left -- only has to be followed until it steps on a
// position the guess also visited; from there the guess is right.  That fix-up runs chunk
// by chunk, a few hops each, instead of one hop per token.
constexpr int kChunk = 32;                          // positions per lane (<= 32: one mask bit each)
constexpr int kTile = kChunk * kWave;               // positions per pass
constexpr int kChunkRow = kChunk + 1;               // padded: same-offset reads of all lanes spread over the banks

// 8,448 B per stream: 16 streams per CU, the whole 4096-block batch in one round (with the tile's
// bytes staged as well it was 10,496 B, 15 per CU and two rounds: 9.9 ms instead of 5).  The
// literal travels in the match word instead: len << 16 | dist for a match (len >= 3), the byte
// itself (len field 0) where index_match_kernel found none.
struct ParseLds {
    uint32_t m[kChunkRow * kWave];                  // match word per position
};

__device__ __forceinline__ uint32_t parse_slot(uint32_t k) { return k + (k / (uint32_t)kChunk); }

// how far a token reaches: its match length, 1 for a literal.  Match words come from
// index_match_kernel (length >= 3); the floor of 1 is there so that the walks below end
// whatever the table holds -- a word with a zero length field must not stall a wave.
__device__ __forceinline__ uint32_t parse_step(uint32_t w) {
    const uint32_t len = w >> 16;
    return len != 0 ? len : 1u;
}

__global__ __launch_bounds__(kWave)
void index_parse_kernel(const uint8_t* __restrict__ in,
                        const uint64_t* __restrict__ in_off,
                        uint32_t n_blocks,
                        const uint32_t* __restrict__ match,
                        uint32_t* __restrict__ tokens,
                        uint32_t* __restrict__ tok_count, uint64_t slots) {
    __shared__ ParseLds lds;
    const uint32_t b = blockIdx.x;
    if (b >= n_blocks) { return; }
    const int lane = threadIdx.x;
    if (in_off[b + 1] > slots) {                  // the caller's arrays do not reach this far: refuse the block
        if (lane == 0) { tok_count[b] = kRefused; }
        return;
    }
    const uint8_t* src = in + in_off[b];
    const uint64_t bytes = in_off[b + 1] - in_off[b];
    const uint32_t* M = match + in_off[b];
    uint32_t* tok = tokens + in_off[b];

    uint32_t ntok = 0;
    uint32_t entry = 0;                              // where the real path enters the tile (tile-relative)
    for (uint64_t tile = 0; tile < bytes; tile += (uint64_t)kTile) {
        const uint64_t left = bytes - tile;
        const uint32_t have = left < (uint64_t)kTile ? (uint32_t)left : (uint32_t)kTile;
        if (entry >= have) { entry -= (uint32_t)kTile; continue; }         // a token spans the whole tile
        __syncthreads();
        for (uint32_t k = lane; k < have; k += kWave) {
            // positions bytes-2, bytes-1 have no 3-byte prefix: literal
            lds.m[parse_slot(k)] = (tile + k + 2 < bytes) ? M[tile + k] : (uint32_t)src[tile + k];
        }
        __syncthreads();

        // ---- every lane: its chunk from the chunk's first position ---------------------
        const uint32_t lo = (uint32_t)lane * (uint32_t)kChunk;
        const uint32_t room = have > lo ? (have - lo < (uint32_t)kChunk ? have - lo : (uint32_t)kChunk) : 0u;
        const uint32_t* row = &lds.m[lane * kChunkRow];
        uint32_t guess = 0;                          // one bit per position of the chunk (kChunk <= 32)
        uint32_t p = 0;
        while (p < room) {
            guess |= 1u << p;
            const uint32_t w = row[p];
            p += parse_step(w);
        }
        const uint32_t guess_out = lo + p;           // where the guess leaves the chunk (>= lo + room)

        // ---- the real path, chunk by chunk (uniform) -----------------------------------
        uint32_t mine = 0;                           // this lane's chunk: real token starts
        uint32_t e = entry;
        for (int l = 0; l < kWave; l++) {
            const uint32_t clo = (uint32_t)l * (uint32_t)kChunk;
            if (clo >= have) { break; }
            const uint32_t chi = clo + (uint32_t)kChunk < have ? clo + (uint32_t)kChunk : have;
            if (e >= chi) { continue; }              // the path jumps over this chunk
            const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)guess, l);
            const uint32_t gout = (uint32_t)__builtin_amdgcn_readlane((int)guess_out, l);
            uint32_t real = 0;
            uint32_t q = e - clo;
            const uint32_t croom = chi - clo;
            for (;;) {
                if ((g >> q) & 1u) {                 // the guess was here too: the rest is the guess's
                    real |= g & ~((1u << q) - 1u);
                    e = gout;
                    break;
                }
                real |= 1u << q;
                const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds.m[l * kChunkRow + (int)q]);
                q += parse_step(w);
                if (q >= croom) { e = clo + q; break; }
            }
            if (lane == l) { mine = real; }
        }
        entry = e - (uint32_t)kTile;                 // e >= have here; carried into the next tile

        // ---- token words, in order ----------------------------------------------------
        const uint32_t cnt = (uint32_t)__builtin_popcount(mine);
        const uint32_t incl = wave_scan(cnt);
        uint32_t at = ntok + incl - cnt;
        uint32_t bits = mine;
        while (bits != 0) {
            const uint32_t k = (uint32_t)__builtin_ctz(bits);
            bits &= bits - 1u;
            const uint32_t w = row[k];
            tok[at++] = (w >> 16) != 0 ? (kTokMatch | w) : (w & 0xFFu);
        }
        ntok += (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
    }
    if (lane == 0) { tok_count[b] = ntok; }
}

void launch_index_sort(const uint8_t* in, const uint64_t* in_off, uint32_t n_blocks,
                       uint32_t* buf_a, uint32_t* buf_b, uint32_t* tmp, uint64_t slots,
                       hipStream_t stream) {
    if (n_blocks == 0) { return; }
    hipLaunchKernelGGL(index_sort_kernel, dim3(n_blocks), dim3(kSortThreads), 0, stream,
                       in, in_off, n_blocks, buf_a, buf_b, tmp, slots);
}

void launch_index_match(const uint8_t* in, const uint64_t* in_off, uint32_t n_blocks,
                        uint32_t window, const uint32_t* sorted, uint32_t* match,
                        uint32_t match_groups, uint64_t slots, hipStream_t stream) {
    if (n_blocks == 0) { return; }
    if (match_groups < 1) { match_groups = 1; }
    const uint32_t rounded = (n_blocks + (uint32_t)kXcds - 1) / (uint32_t)kXcds * (uint32_t)kXcds;
    while ((uint64_t)match_groups * rounded > 0x7FFFFFFFull) { match_groups = (match_groups + 1) / 2; }
    hipLaunchKernelGGL(index_match_kernel, dim3(rounded * match_groups), dim3(256), 0, stream,
                       in, in_off, n_blocks, window, sorted, match, match_groups, slots);
}

void launch_index_parse(const uint8_t* in, const uint64_t* in_off, uint32_t n_blocks,
                        const uint32_t* match, uint32_t* tokens, uint32_t* tok_count,
                        uint64_t slots, hipStream_t stream) {
    if (n_blocks == 0) { return; }
    hipLaunchKernelGGL(index_parse_kernel, dim3(n_blocks), dim3(kWave), 0, stream,
                       in, in_off, n_blocks, match, tokens, tok_count, slots);
}

} // namespace sqzk
