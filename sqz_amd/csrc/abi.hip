// sqz_amd/csrc/abi.hip -- the C ABI of libsqz_amd.so (include/sqz/sqz.h).
//
// Host side of the drop-in boundary: argument checks
// with the reference's errno conventions, H2D / D2H staging for the host
// flavour, kernel launches for the device flavour.  There is no CPU code path
// for the codec itself: without a gfx950 device every call reports ENODEV.
#include "../../include/sqz/sqz.h"
#include "../../include/sqz/sqz_rc.h"
#include "sqz_kernels.h"

#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

namespace {

constexpr uint64_t kMaxStream = 1ull << 31;     // per-stream limit (32-bit freq/tokens)

// ---------------------------------------------------------------- device ctx
struct DevBuf {
    void*  p = nullptr;
    size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) { return 0; }
        if (p != nullptr) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = n + n / 4 + 4096;
        if (hipMalloc(&p, want) != hipSuccess) {
            p = nullptr;
            if (hipMalloc(&p, n) != hipSuccess) { p = nullptr; return ENOMEM; }
            want = n;
        }
        cap = want;
        return 0;
    }
};

struct Ctx {
    std::mutex mu;                  // guards the probe only: no call holds it while the device works
    int  state = 0;                 // 0 = unprobed, 1 = ok, -1 = no device
    char name[256] = {0};
    int  cus = 0;
    uint64_t lds = 0;
};

Ctx& ctx() { static Ctx c; return c; }

// Staging of the host-buffer flavour: one LANE per call in flight -- grow-only device buffers and a
// HIP stream of its own (SURVEY.md section 8b "Threading": one stream per call or caller-provided).
// A call leases a lane for its H2D -> kernels -> D2H and hands it back; concurrent callers get
// different lanes, so two threads neither serialise on a lock nor share the NULL stream.
struct Lane {
    hipStream_t own = nullptr;      // created with the lane (non-blocking: independent of the NULL stream)
    DevBuf in, out, in_off, out_off, tokens, tok_count, out_bytes, err, end_bit, work_a, work_m,
           dense, dense_off;
    std::vector<uint8_t> host_dense;   // landing area of the host flavour's one device-to-host copy
};

struct LanePool {
    std::mutex mu;
    std::vector<Lane*> idle;
    std::atomic<int> made{0};
};
LanePool& lanes() { static LanePool p; return p; }

struct LaneLease {
    Lane* lane = nullptr;
    LaneLease() {
        LanePool& p = lanes();
        {
            std::lock_guard<std::mutex> g(p.mu);
            if (!p.idle.empty()) { lane = p.idle.back(); p.idle.pop_back(); }
        }
        if (lane == nullptr) {
            lane = new Lane();
            if (hipStreamCreateWithFlags(&lane->own, hipStreamNonBlocking) != hipSuccess) { lane->own = nullptr; }
            p.made.fetch_add(1);
        }
    }
    ~LaneLease() {
        LanePool& p = lanes();
        std::lock_guard<std::mutex> g(p.mu);
        p.idle.push_back(lane);
    }
    // the caller's stream (struct sqz.stream) when given, else the lane's own
    hipStream_t stream(void* callers) const { return callers != nullptr ? (hipStream_t)callers : lane->own; }
};

int probe_locked(Ctx& c) {
    if (c.state != 0) { return c.state > 0 ? 0 : ENODEV; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { c.state = -1; return ENODEV; }
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { c.state = -1; return ENODEV; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "libsqz_amd: device '%s' is not gfx950; kernels are built for "
                        "MI355X only\n", prop.gcnArchName);
        c.state = -1;
        return ENODEV;
    }
    snprintf(c.name, sizeof(c.name), "%s (%s)", prop.name, prop.gcnArchName);
    c.cus = prop.multiProcessorCount;
    c.lds = (uint64_t)prop.maxSharedMemoryPerMultiProcessor;
    c.state = 1;
    return 0;
}

int device_ready(void) {
    Ctx& c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    return probe_locked(c);
}

int hip_errno(hipError_t e) {
    if (e == hipSuccess) { return 0; }
    if (e == hipErrorOutOfMemory) { return ENOMEM; }
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) { return ENODEV; }
    fprintf(stderr, "libsqz_amd: HIP error %d (%s)\n", (int)e, hipGetErrorString(e));
    return EIO;
}

#define HIP_TRY(expr) do { const int _e = hip_errno(expr); if (_e != 0) { return _e; } } while (0)

// ---------------------------------------------------------------- timing
struct TimedSpan { hipEvent_t a, b; int kind; };
struct Timing {
    std::mutex mu;
    bool enabled = false;
    std::vector<TimedSpan> spans;
    sqz_hip_timing acc = {};
    // fold finished spans into acc and release their events (caller holds mu)
    void drain_locked() {
        for (TimedSpan& s : spans) {
            float ms = 0.0f;
            if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
                if (s.kind >= 0 && s.kind < SQZ_HIP_KERNELS) { acc.ms[s.kind] += ms; acc.launches[s.kind]++; }
            }
            (void)hipEventDestroy(s.a);
            (void)hipEventDestroy(s.b);
        }
        spans.clear();
    }
};
Timing& timing() { static Timing t; return t; }
constexpr size_t kMaxPendingSpans = 4096;       // a caller that never reads the timing must not grow it forever

struct SpanGuard {
    hipStream_t s; int kind; hipEvent_t a = nullptr, b = nullptr; bool on = false;
    SpanGuard(hipStream_t stream, int k) : s(stream), kind(k) {
        Timing& t = timing();
        std::lock_guard<std::mutex> g(t.mu);
        on = t.enabled;
        if (on) {
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
            (void)hipEventRecord(a, s);
        }
    }
    ~SpanGuard() {
        if (!on) { return; }
        (void)hipEventRecord(b, s);
        Timing& t = timing();
        std::lock_guard<std::mutex> g(t.mu);
        t.spans.push_back({a, b, kind});
        if (t.spans.size() >= kMaxPendingSpans) { t.drain_locked(); }
    }
};

// ---------------------------------------------------------------- host bit I/O
// bitstream.h:28-63 / :65-103, both modes, used only for the stream headers.
bool writer_is_memory(const bitstream* bs) { return bs->data != NULL && bs->capacity > 0; }     // bitstream.h:34
bool writer_is_callback(const bitstream* bs) { return bs->data == NULL && bs->capacity == 0 && bs->output != NULL; }
bool reader_is_callback(const bitstream* bs) { return bs->data == NULL && bs->bytes == 0 && bs->input != NULL; }

void host_put_bit(bitstream* bs, int bit) {
    if (bs->error != 0) { return; }
    bs->b64 = (bs->b64 << 1) | (uint64_t)(bit & 1);
    if (++bs->bits == 64) {
        if (writer_is_memory(bs)) {
            for (int k = 0; k < 8 && bs->error == 0; k++) {
                if (bs->bytes == bs->capacity) { bs->error = E2BIG; }
                else { bs->data[bs->bytes++] = (uint8_t)(bs->b64 >> (56 - 8 * k)); }
            }
        } else if (writer_is_callback(bs)) {                    // bitstream.h:44-48
            bs->error = bs->output(bs);
            if (bs->error == 0) { bs->bytes += 8; }
        } else {
            bs->error = EINVAL;
        }
        bs->bits = 0;
        bs->b64 = 0;
    }
}

void host_put_bits(bitstream* bs, uint64_t v, int n) {
    for (int b = 0; b < n && bs->error == 0; b++) { host_put_bit(bs, (int)((v >> b) & 1)); }
}

uint64_t reader_limit(const bitstream* bs) { return bs->bytes != 0 ? bs->bytes : bs->capacity; }

int host_get_bit(bitstream* bs) {
    if (bs->error != 0) { return 0; }
    if (bs->bits == 0) {
        bs->b64 = 0;
        if (bs->data != NULL) {
            const uint64_t limit = reader_limit(bs);
            for (int k = 0; k < 8 && bs->error == 0; k++) {
                if (bs->read == limit) { bs->error = E2BIG; }
                else { bs->b64 |= (uint64_t)bs->data[bs->read++] << (56 - 8 * k); }
            }
        } else if (reader_is_callback(bs)) {                    // bitstream.h:81-85
            bs->error = bs->input(bs);
            if (bs->error == 0) { bs->read += 8; }
        } else {
            bs->error = EINVAL;
        }
        bs->bits = 64;
    }
    const int bit = (int)(bs->b64 >> 63);
    bs->b64 <<= 1;
    bs->bits--;
    return bit;
}

uint64_t host_get_bits(bitstream* bs, int n) {
    uint64_t v = 0;
    for (int b = 0; b < n && bs->error == 0; b++) { v |= (uint64_t)host_get_bit(bs) << b; }
    return v;
}

uint64_t load_be64(const uint8_t* p) {
    uint64_t w = 0;
    for (int k = 0; k < 8; k++) { w = (w << 8) | p[k]; }
    return w;
}

void store_be64(uint8_t* p, uint64_t w) {
    for (int k = 0; k < 8; k++) { p[k] = (uint8_t)(w >> (56 - 8 * k)); }
}

bool window_ok(uint32_t w) { return w >= 2 && w <= 32768; }

// wavefronts that share one stream's LDS window in the scan kernel (tuning knob:
// SQZ_SCAN_WAVES=1|2|4|8, default 4 = four waves on every SIMD at 4 streams per CU)
int scan_waves() {
    static const int w = [] {
        const char* e = getenv("SQZ_SCAN_WAVES");
        const int v = e != NULL ? atoi(e) : 4;
        return (v == 1 || v == 2 || v == 4 || v == 8) ? v : 4;
    }();
    return w;
}

int check_offsets(const uint64_t* off, uint32_t n, bool need8) {
    for (uint32_t b = 0; b < n; b++) {
        if (off[b + 1] < off[b] || off[b + 1] - off[b] > kMaxStream) { return EINVAL; }
        if (need8 && (off[b] & 7u) != 0) { return EINVAL; }
    }
    return 0;
}

uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// match finder of encode stage 1: 1 = indexed (default), 0 = brute-force scan
// (SQZ_FINDER=scan|index).  Both produce the reference's tokens.
std::atomic<int> g_finder{-1};      // -1 = not chosen yet
int finder_default() {
    int f = g_finder.load(std::memory_order_relaxed);
    if (f < 0) {
        const char* e = getenv("SQZ_FINDER");
        const int chosen = (e != NULL && strcmp(e, "scan") == 0) ? 0 : 1;
        // a concurrent sqz_hip_set_finder wins over the environment's default
        f = g_finder.compare_exchange_strong(f, chosen) ? chosen : g_finder.load();
    }
    return f;
}

// Wavefronts per stream in the entropy decoder.  A batch of more than 4 streams per CU runs one wave per
// stream (16 streams per CU fit: the batch fills the chip by itself); smaller batches leave SIMDs idle and the
// decoder spends them on the read-ahead: 4 waves per stream up to 4 streams per CU, 8 up to 2 (measured,
// MI355X, entropy_decode_kernel: 512 blocks 62.4 -> 51.4 -> 49.4 ms with 1 / 4 / 8 waves, 1024 blocks
// 63.6 -> 54.0 ms with 4; 2 waves per stream gain 2-4 % at best and lose at 2048 blocks: not taken).
// SQZ_DECODE_WAVES=1|2|4|8 overrides.
int decode_waves_for(uint32_t n_blocks) {
    static const int forced = [] { const char* e = getenv("SQZ_DECODE_WAVES"); return e != NULL ? atoi(e) : 0; }();
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8) { return forced; }
    int cus = ctx().cus;
    if (cus <= 0) { cus = 256; }
    if ((uint64_t)n_blocks <= (uint64_t)cus * 2) { return 8; }
    return (uint64_t)n_blocks <= (uint64_t)cus * 4 ? 4 : 1;
}

uint32_t match_groups_for(uint64_t avg_block_bytes) {
    static const int div = [] { const char* e = getenv("SQZ_MATCH_DIV"); const int v = e != NULL ? atoi(e) : 1024; return v >= 256 ? v : 1024; }();
    uint64_t g = (avg_block_bytes + div - 1) / div;
    if (g < 1) { g = 1; }
    if (g > 65535) { g = 65535; }
    return (uint32_t)g;
}

// stage 1 on device buffers -> token words; work = 2 arrays of one uint32 slot per input byte
void run_stage1(int finder, const uint8_t* d_in, const uint64_t* d_in_off, uint32_t n,
                uint32_t window, uint32_t* tokens, uint32_t* counts,
                uint32_t* work_a, uint32_t* work_m, uint64_t avg_block, uint64_t slots,
                hipStream_t st) {
    if (finder == 0 || work_a == nullptr || work_m == nullptr) {
        SpanGuard g(st, SQZ_HIP_K_LZ77_SCAN);
        sqzk::launch_lz77_scan(d_in, d_in_off, n, window, tokens, counts, scan_waves(), slots, st);
    } else {
        { SpanGuard g(st, SQZ_HIP_K_INDEX_SORT);
          // two arrays serve the three steps: the sort leaves the positions in work_a (work_m is its second
          // buffer), the match table goes to work_m (the sort is done with it), and the token words may take
          // work_a's place (tokens == work_a in the encode path: the sorted positions are dead by then)
          sqzk::launch_index_sort(d_in, d_in_off, n, work_a, work_m /* ping-pong */, nullptr, slots, st); }
        { SpanGuard g(st, SQZ_HIP_K_INDEX_MATCH);
          sqzk::launch_index_match(d_in, d_in_off, n, window, work_a, work_m,
                                   match_groups_for(avg_block), slots, st); }
        { SpanGuard g(st, SQZ_HIP_K_INDEX_PARSE);
          sqzk::launch_index_parse(d_in, d_in_off, n, work_m, tokens, counts, slots, st); }
    }
}

// the whole encode on device buffers: stage 1 (either finder) -> token words -> emit
void run_encode(int finder, const uint8_t* d_in, const uint64_t* d_in_off, uint32_t n,
                uint32_t window, uint32_t* tokens, uint32_t* counts, uint32_t* work_a,
                uint32_t* work_m, uint64_t avg_block, uint8_t* d_out, const uint64_t* d_out_off,
                uint64_t* d_out_bytes, int32_t* d_err, uint64_t prefix_acc, int prefix_fill,
                uint64_t slots, sqz_block_stats* d_stats, hipStream_t st) {
    run_stage1(finder, d_in, d_in_off, n, window, tokens, counts, work_a, work_m, avg_block, slots, st);
    SpanGuard g(st, SQZ_HIP_K_HUFFMAN_EMIT);
    sqzk::launch_huffman_emit(tokens, d_in_off, counts, d_out, d_out_off, d_out_bytes, d_err, n,
                              prefix_acc, prefix_fill, d_stats, st);
}

// host-buffer encode of n blocks; prefix = pending header bits of block 0
// (single-stream API only).  `c` is the caller's leased lane, `st` the stream of this call.
int encode_host(Lane& c, hipStream_t st, const uint8_t* in, const uint64_t* in_off, uint32_t n,
                       uint32_t window, uint8_t* out, const uint64_t* out_off,
                       uint64_t* out_bytes, int32_t* err,
                       uint64_t prefix_acc, int prefix_fill, uint64_t* tokens_total) {
    const uint64_t in_base = in_off[0], out_base = out_off[0];
    const uint64_t total_in = in_off[n] - in_base, total_out = out_off[n] - out_base;
    std::vector<uint64_t> io(n + 1), oo(n + 1);
    for (uint32_t b = 0; b <= n; b++) { io[b] = in_off[b] - in_base; oo[b] = out_off[b] - out_base; }

    int e;
    if ((e = c.in.reserve(total_in + 16)) || (e = c.out.reserve(total_out + 16)) ||
        (e = c.in_off.reserve((n + 1) * 8)) || (e = c.out_off.reserve((n + 1) * 8)) ||
        (e = c.tok_count.reserve((size_t)n * 4)) ||
        (e = c.work_a.reserve((total_in + 64) * 4)) || (e = c.work_m.reserve((total_in + 64) * 4)) ||
        (e = c.out_bytes.reserve((size_t)n * 8)) || (e = c.err.reserve((size_t)n * 4))) {
        return e;
    }
    if (total_in > 0) { HIP_TRY(hipMemcpyAsync(c.in.p, in + in_base, total_in, hipMemcpyHostToDevice, st)); }
    HIP_TRY(hipMemcpyAsync(c.in_off.p, io.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c.out_off.p, oo.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    uint64_t widest = 0;
    for (uint32_t b = 0; b < n; b++) { widest = io[b + 1] - io[b] > widest ? io[b + 1] - io[b] : widest; }
    run_encode(finder_default(), (const uint8_t*)c.in.p, (const uint64_t*)c.in_off.p, n, window,
               (uint32_t*)c.work_a.p /* token words take the sorted positions' place */, (uint32_t*)c.tok_count.p,
               (uint32_t*)c.work_a.p, (uint32_t*)c.work_m.p, widest, (uint8_t*)c.out.p, (const uint64_t*)c.out_off.p,
               (uint64_t*)c.out_bytes.p, (int32_t*)c.err.p, prefix_acc, prefix_fill, total_in + 64, nullptr, st);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_bytes, c.out_bytes.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(err, c.err.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    // the streams leave as ONE transfer: packed back to back on the device first (the slabs are
    // sized for the worst case, 2.3 x what a Zipf block produces), then spread over the caller's
    // slabs on the host.  Only out[out_off[b] .. + out_bytes[b]) is written.
    std::vector<uint64_t> dense_off(n + 1);
    dense_off[0] = 0;
    for (uint32_t b = 0; b < n; b++) { dense_off[b + 1] = dense_off[b] + ((out_bytes[b] + 7) & ~7ull); }
    const uint64_t dense_total = dense_off[n];
    if (dense_total > 0) {
        if (n == 1) {
            HIP_TRY(hipMemcpyAsync(out + out_off[0], c.out.p, out_bytes[0], hipMemcpyDeviceToHost, st));
        } else {
            if ((e = c.dense.reserve(dense_total + 16)) || (e = c.dense_off.reserve((n + 1) * 8))) { return e; }
            HIP_TRY(hipMemcpyAsync(c.dense_off.p, dense_off.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
            sqzk::launch_compact_blocks((const uint8_t*)c.out.p, (const uint64_t*)c.out_off.p,
                                        (const uint64_t*)c.out_bytes.p, n, (uint8_t*)c.dense.p,
                                        (const uint64_t*)c.dense_off.p, dense_total / n, st);
            HIP_TRY(hipGetLastError());
            if (c.host_dense.size() < dense_total) { c.host_dense.resize(dense_total); }
            HIP_TRY(hipMemcpyAsync(c.host_dense.data(), c.dense.p, dense_total, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            for (uint32_t b = 0; b < n; b++) {
                memcpy(out + out_off[b], c.host_dense.data() + dense_off[b], out_bytes[b]);
            }
        }
    }
    if (tokens_total != nullptr) {
        std::vector<uint32_t> tc(n);
        HIP_TRY(hipMemcpyAsync(tc.data(), c.tok_count.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        uint64_t sum = 0;
        for (uint32_t b = 0; b < n; b++) { sum += tc[b]; }
        *tokens_total = sum;
    }
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int decode_host(Lane& c, hipStream_t st, const uint8_t* in, const uint64_t* in_off, uint32_t n,
                uint8_t* out, const uint64_t* out_off, int32_t* err,
                uint64_t start_bit, uint64_t* end_bit) {
    const uint64_t in_base = in_off[0], out_base = out_off[0];
    const uint64_t total_in = in_off[n] - in_base, total_out = out_off[n] - out_base;
    std::vector<uint64_t> io(n + 1), oo(n + 1);
    for (uint32_t b = 0; b <= n; b++) { io[b] = in_off[b] - in_base; oo[b] = out_off[b] - out_base; }
    int e;
    if ((e = c.in.reserve(total_in + 16)) || (e = c.out.reserve(total_out + 16)) ||
        (e = c.in_off.reserve((n + 1) * 8)) || (e = c.out_off.reserve((n + 1) * 8)) ||
        (e = c.err.reserve((size_t)n * 4)) || (e = c.end_bit.reserve((size_t)n * 8)) ||
        (e = c.tokens.reserve((total_out + 64) * 4)) || (e = c.tok_count.reserve((size_t)n * 4))) {
        return e;
    }
    if (total_in > 0) { HIP_TRY(hipMemcpyAsync(c.in.p, in + in_base, total_in, hipMemcpyHostToDevice, st)); }
    HIP_TRY(hipMemcpyAsync(c.in_off.p, io.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c.out_off.p, oo.data(), (n + 1) * 8, hipMemcpyHostToDevice, st));
    {
        SpanGuard g(st, SQZ_HIP_K_ENTROPY_DECODE);
        sqzk::launch_entropy_decode((const uint8_t*)c.in.p, (const uint64_t*)c.in_off.p,
                                    (const uint64_t*)c.out_off.p, (uint32_t*)c.tokens.p,
                                    (uint32_t*)c.tok_count.p, (int32_t*)c.err.p,
                                    (uint64_t*)c.end_bit.p, n, start_bit, decode_waves_for(n), st);
    }
    {
        SpanGuard g(st, SQZ_HIP_K_LZ_EXPAND);
        sqzk::launch_lz_expand((const uint32_t*)c.tokens.p, (const uint32_t*)c.tok_count.p,
                               (uint8_t*)c.out.p, (const uint64_t*)c.out_off.p, n, st);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(err, c.err.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (total_out > 0) {
        HIP_TRY(hipMemcpyAsync(out + out_base, c.out.p, total_out, hipMemcpyDeviceToHost, st));
    }
    if (end_bit != nullptr) {
        HIP_TRY(hipMemcpyAsync(end_bit, c.end_bit.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

} // namespace

// =========================================================================
extern "C" {

const char* sqz_version(void) { return "sqz_amd 0.1 (gfx950; H0 semantics, H1 names)"; }

int sqz_hip_device_info(char* name, size_t name_cap, int* compute_units,
                        uint64_t* lds_bytes_per_cu) {
    Ctx& c = ctx();
    std::lock_guard<std::mutex> g(c.mu);
    const int e = probe_locked(c);
    if (e != 0) { return e; }
    if (name != NULL && name_cap > 0) { snprintf(name, name_cap, "%s", c.name); }
    if (compute_units != NULL) { *compute_units = c.cus; }
    if (lds_bytes_per_cu != NULL) { *lds_bytes_per_cu = c.lds; }
    return 0;
}

uint64_t sqz_bound(uint64_t bytes) { return (2 * bytes + 1024 + 7) & ~(uint64_t)7; }

int sqz_file_words(const uint8_t* in, uint64_t bytes, uint8_t* out) {
    if ((bytes & 7u) != 0 || (bytes != 0 && (in == NULL || out == NULL))) { return EINVAL; }
    for (uint64_t k = 0; k < bytes; k += 8) {
        uint64_t w = 0;                                   // the word's value: stream bytes are MSB first
        for (int j = 0; j < 8; j++) { w = (w << 8) | in[k + j]; }
        memcpy(out + k, &w, 8);                           // what fwrite(&b64, 8, 1, f) puts in the file
    }
    return 0;
}

void sqz_init(struct sqz* s) {
    if (s == NULL) { return; }
    memset(s, 0, sizeof(*s));
    s->device = -1;
}

void sqz_write_header(struct bitstream* bs, uint64_t bytes) { host_put_bits(bs, bytes, 64); }

void sqz_read_header(struct bitstream* bs, uint64_t* bytes) {
    const uint64_t b = host_get_bits(bs, 64);
    if (bs->error == 0 && bytes != NULL) { *bytes = b; }
}

void sqz_write_header_h0(struct bitstream* bs, uint64_t bytes, uint8_t win_bits) {
    if (win_bits < sqz_min_win_bits || win_bits > sqz_max_win_bits) {   // squeeze.h:257-258
        bs->error = EINVAL;
    } else {
        host_put_bits(bs, bytes, 64);
        host_put_bits(bs, win_bits, 8);
    }
}

void sqz_read_header_h0(struct bitstream* bs, uint64_t* bytes, uint8_t* win_bits) {
    const uint64_t b = host_get_bits(bs, 64);
    const uint64_t w = host_get_bits(bs, 8);
    if (bs->error == 0) {
        if (w < sqz_min_win_bits || w > sqz_max_win_bits) {             // squeeze.h:449-450
            bs->error = EINVAL;
        } else {
            if (bytes != NULL) { *bytes = b; }
            if (win_bits != NULL) { *win_bits = (uint8_t)w; }
        }
    }
}

void sqz_compress(struct sqz* s, struct bitstream* bs,
                  const uint8_t* data, size_t bytes, uint32_t window) {
    if (s == NULL || bs == NULL) { return; }
    s->bs = bs;                                          // squeeze.h:332
    if (s->error != 0) { return; }                       // sticky: squeeze.h:337
    if (bs->error != 0) { s->error = bs->error; return; } // squeeze.h:226-227
    const bool by_callback = writer_is_callback(bs);
    if (!window_ok(window) || (!by_callback && !writer_is_memory(bs)) ||
        (bytes > 0 && data == NULL) || bytes > kMaxStream || (!by_callback && bs->bytes > bs->capacity) ||
        bs->bits < 0 || bs->bits > 63) {
        s->error = EINVAL;
        return;
    }
    // callback mode: the stream lands in a buffer of ours and is handed over word by word below
    std::vector<uint8_t> own;
    uint8_t* dst = by_callback ? nullptr : bs->data + bs->bytes;
    uint64_t room = by_callback ? 0 : bs->capacity - bs->bytes;
    if (by_callback) {
        room = sqz_bound(bytes) + 16;
        own.resize(room);
        dst = own.data();
    }
    const uint64_t in_off[2] = {0, (uint64_t)bytes};
    const uint64_t out_off[2] = {0, room};
    uint64_t produced = 0;
    int32_t err = 0;
    {
        int e = device_ready();
        if (e != 0) { s->error = e; return; }
        // the device writes into a staging slab; it lands behind the header bytes
        LaneLease lease;
        e = encode_host(*lease.lane, lease.stream(s->stream), data, in_off, 1, window, dst, out_off, &produced,
                        &err, bs->b64, bs->bits, &s->tokens);
        if (e != 0) { s->error = e; return; }
    }
    bs->b64 = 0;
    bs->bits = 0;
    if (by_callback) {                                   // bitstream.h:44-48, one call per word
        for (uint64_t k = 0; k + 8 <= produced && bs->error == 0; k += 8) {
            bs->b64 = load_be64(dst + k);
            bs->error = bs->output(bs);
            if (bs->error == 0) { bs->bytes += 8; }
            bs->b64 = 0;
        }
        if (err == 0) { err = bs->error; }               // squeeze.h:226-236 mirrors the stream's error
    } else {
        bs->bytes += produced;
        if (err == E2BIG) { bs->error = E2BIG; }
    }
    s->error = err;
}

// leave the reader where the reference's would be after the symbol that ended at `end_bit`
// of buf[0 .. limit): bitstream.h:65-93 (the last fetched word, shifted by what was consumed)
static void park_reader(bitstream* bs, const uint8_t* buf, uint64_t limit, uint64_t end_bit,
                        uint64_t* words_out) {
    const uint64_t words = (end_bit + 63) / 64;
    bs->bits = (int32_t)(words * 64 - end_bit);
    bs->b64 = 0;
    if (bs->bits > 0 && words * 8 <= limit) { bs->b64 = load_be64(buf + (words - 1) * 8) << (64 - bs->bits); }
    *words_out = words;
}

static void decompress_by_callback(struct sqz* s, struct bitstream* bs, uint8_t* data, size_t bytes) {
    // What the header reader left: `bits` unread bits at the top of b64.  They become word 0 of
    // our buffer (the bits in front of them are gone; the decoder starts behind them).
    std::vector<uint8_t> buf;
    uint64_t start_bit = 0;
    if (bs->bits > 0) {
        buf.resize(8);
        store_be64(buf.data(), bs->b64 >> (64 - bs->bits));
        start_bit = 64 - (uint64_t)bs->bits;
    }
    const uint64_t head_words = buf.size() / 8;
    const uint64_t most_words = head_words + (sqz_bound(bytes) + 16) / 8 + 2;   // no stream of `bytes` bytes is longer
    // The reference pulls one word when its bit reader runs dry (bitstream.h:81-85); a device decode wants
    // its input up front and a word cannot be handed back, so the pull runs AHEAD of the decoder: 4 words,
    // then twice as many whenever the decoder ran dry.  Words pulled <= max(4, 2 x the words the stream
    // holds) whatever its compression ratio; every retry decodes again from the stream's start.
    uint64_t want_words = head_words + 4;
    int src_error = 0;
    bool dry = false;
    for (;;) {
        if (want_words > most_words) { want_words = most_words; }
        while (buf.size() / 8 < want_words && !dry) {        // bitstream.h:81-85
            bs->b64 = 0;
            const int e = bs->input(bs);
            if (e != 0) { src_error = e; dry = true; break; }
            const size_t at = buf.size();
            buf.resize(at + 8);
            store_be64(buf.data() + at, bs->b64);
        }
        const uint64_t limit = buf.size();
        const uint64_t in_off[2] = {0, limit};
        const uint64_t out_off[2] = {0, (uint64_t)bytes};
        int32_t err = 0;
        uint64_t end_bit = 0;
        {
            int e = device_ready();
            if (e == 0) {
                LaneLease lease;
                e = decode_host(*lease.lane, lease.stream(s->stream), buf.data(), in_off, 1, data, out_off, &err,
                                start_bit, &end_bit);
            }
            if (e != 0) { s->error = e; return; }
        }
        if (err == E2BIG && !dry && limit / 8 < most_words) {   // ran out of words: pull more and decode again
            want_words = head_words + 2 * (limit / 8 - head_words);
            continue;
        }
        uint64_t words = 0;
        park_reader(bs, buf.data(), limit, end_bit, &words);
        if (words > limit / 8) { words = limit / 8; }
        bs->read += 8 * (words - head_words);                // what the reference's reader would have fetched
        if (err == E2BIG) {                                  // the source ended before the stream did
            bs->error = src_error != 0 ? src_error : E2BIG;
            err = bs->error;
        }
        s->error = err;
        return;
    }
}

void sqz_decompress(struct sqz* s, struct bitstream* bs, uint8_t* data, size_t bytes) {
    if (s == NULL || bs == NULL) { return; }
    s->bs = bs;                                          // squeeze.h:504
    if (s->error != 0) { return; }
    if (bs->error != 0) { s->error = bs->error; return; }
    if ((bytes > 0 && data == NULL) || bytes > kMaxStream || bs->bits < 0 || bs->bits > 63) {
        s->error = EINVAL;
        return;
    }
    if (reader_is_callback(bs)) {
        if (bytes != 0) { decompress_by_callback(s, bs, data, bytes); }
        return;
    }
    const uint64_t limit = reader_limit(bs);
    if (bs->data == NULL || bs->read > limit || (uint64_t)bs->bits > bs->read * 8) {
        s->error = EINVAL;
        return;
    }
    if (bytes == 0) { return; }
    const uint64_t start_bit = bs->read * 8 - (uint64_t)bs->bits;
    const uint64_t in_off[2] = {0, limit};
    const uint64_t out_off[2] = {0, (uint64_t)bytes};
    int32_t err = 0;
    uint64_t end_bit = 0;
    {
        int e = device_ready();
        if (e == 0) {
            LaneLease lease;
            e = decode_host(*lease.lane, lease.stream(s->stream), bs->data, in_off, 1, data, out_off, &err,
                            start_bit, &end_bit);
        }
        if (e != 0) { s->error = e; return; }
    }
    uint64_t words = 0;
    park_reader(bs, bs->data, limit, end_bit, &words);
    bs->read = words * 8 <= limit ? words * 8 : limit;
    if (err == E2BIG) { bs->error = E2BIG; }
    s->error = err;
}

// ------------------------------------------------------------------ vtable
static squeeze_type* vt_alloc(uint8_t map_bits) {
    if (map_bits != 0) { return NULL; }
    squeeze_type* s = (squeeze_type*)calloc(1, sizeof(squeeze_type));
    if (s != NULL) { sqz_init(s); }
    return s;
}
static int vt_init_with(squeeze_type* s, void* memory, size_t size, uint8_t map_bits) {
    // squeeze.h:191-199: size == squeeze_sizeof(map_bits).  The codec state lives on the device, so the
    // caller's block only has to hold the struct; a block sized for the reference's trees is accepted too.
    if (map_bits != 0 || memory == NULL || s != memory || size < squeeze_sizeof(0)) { return EINVAL; }
    sqz_init(s);
    return 0;
}
static void vt_compress(squeeze_type* s, bitstream* bs, const uint8_t* data, size_t bytes,
                        uint16_t window) {
    sqz_compress(s, bs, data, bytes, window);
}
static void vt_free(squeeze_type* s) { free(s); }

squeeze_interface squeeze = {
    vt_alloc, vt_init_with, sqz_write_header_h0, vt_compress,
    sqz_read_header_h0, sqz_decompress, vt_free
};

// ------------------------------------------------------------------ batch, host
int sqz_encode_blocks(const uint8_t* in, const uint64_t* in_off, uint32_t n, uint32_t window,
                      uint8_t* out, const uint64_t* out_off, uint64_t* out_bytes, int32_t* err) {
    if (n == 0) { return 0; }
    if (in_off == NULL || out_off == NULL || out == NULL || out_bytes == NULL || err == NULL ||
        !window_ok(window)) { return EINVAL; }
    if (check_offsets(in_off, n, false) != 0 || check_offsets(out_off, n, true) != 0) { return EINVAL; }
    if (in == NULL && in_off[n] != in_off[0]) { return EINVAL; }
    const int e = device_ready();
    if (e != 0) { return e; }
    LaneLease lease;
    return encode_host(*lease.lane, lease.stream(nullptr), in, in_off, n, window, out, out_off, out_bytes, err,
                       0, 0, nullptr);
}

int sqz_decode_blocks(const uint8_t* in, const uint64_t* in_off, uint32_t n,
                      uint8_t* out, const uint64_t* out_off, int32_t* err) {
    if (n == 0) { return 0; }
    if (in == NULL || in_off == NULL || out_off == NULL || err == NULL) { return EINVAL; }
    if (check_offsets(in_off, n, true) != 0 || check_offsets(out_off, n, false) != 0) { return EINVAL; }
    if (out == NULL && out_off[n] != out_off[0]) { return EINVAL; }
    const int e = device_ready();
    if (e != 0) { return e; }
    LaneLease lease;
    return decode_host(*lease.lane, lease.stream(nullptr), in, in_off, n, out, out_off, err, 0, nullptr);
}

// ------------------------------------------------------------------ batch, device
uint64_t sqz_hip_encode_scratch_bytes(uint32_t n, uint64_t total_in_bytes) {
    // token counts + two uint32 slots per input byte: sorted positions, later the token words / the sort's
    // second buffer, later the match table
    return align_up((uint64_t)n * 4, 256) + 2 * (total_in_bytes + 64) * 4;
}

int sqz_hip_lz77_blocks(const void* d_in, const uint64_t* d_in_off, uint32_t n, uint32_t window,
                        uint32_t* d_tokens, uint32_t* d_token_count, void* stream) {
    if (n == 0) { return 0; }
    if (d_in == NULL || d_in_off == NULL || d_tokens == NULL || d_token_count == NULL ||
        !window_ok(window)) { return EINVAL; }
    const int e = device_ready();
    if (e != 0) { return e; }
    run_stage1(0, (const uint8_t*)d_in, d_in_off, n, window, d_tokens, d_token_count,
               nullptr, nullptr, 0, ~0ull /* d_tokens: one slot per input byte, by contract */,
               (hipStream_t)stream);
    return hip_errno(hipGetLastError());
}

int sqz_hip_lz77_blocks_ex(const void* d_in, const uint64_t* d_in_off, uint32_t n, uint32_t window,
                           uint32_t* d_tokens, uint32_t* d_token_count, int finder,
                           void* d_work, uint64_t work_bytes, void* stream) {
    if (n == 0) { return 0; }
    if (d_in == NULL || d_in_off == NULL || d_tokens == NULL || d_token_count == NULL ||
        !window_ok(window) || (finder != 0 && finder != 1)) { return EINVAL; }
    if (finder == 1 && (d_work == NULL || work_bytes < 8 * 64)) { return EINVAL; }
    const int e = device_ready();
    if (e != 0) { return e; }
    // work = two arrays of work_bytes/8 uint32 slots (sorted positions, match table)
    const uint64_t slots = work_bytes / 8;
    uint32_t* wa = (uint32_t*)d_work;
    uint32_t* wm = wa != nullptr ? wa + slots : nullptr;
    run_stage1(finder, (const uint8_t*)d_in, d_in_off, n, window, d_tokens, d_token_count,
               finder == 1 ? wa : nullptr, finder == 1 ? wm : nullptr,
               slots / (n > 0 ? n : 1), finder == 1 ? slots : ~0ull, (hipStream_t)stream);
    return hip_errno(hipGetLastError());
}

int sqz_hip_huffman_blocks(const uint32_t* d_tokens, const uint64_t* d_in_off,
                           const uint32_t* d_token_count, uint32_t n, void* d_out,
                           const uint64_t* d_out_off, uint64_t* d_out_bytes, int32_t* d_err,
                           void* stream) {
    if (n == 0) { return 0; }
    if (d_tokens == NULL || d_in_off == NULL || d_token_count == NULL || d_out == NULL ||
        d_out_off == NULL || d_out_bytes == NULL || d_err == NULL) { return EINVAL; }
    const int e = device_ready();
    if (e != 0) { return e; }
    SpanGuard g((hipStream_t)stream, SQZ_HIP_K_HUFFMAN_EMIT);
    sqzk::launch_huffman_emit(d_tokens, d_in_off, d_token_count, (uint8_t*)d_out, d_out_off,
                             d_out_bytes, d_err, n, 0, 0, nullptr, (hipStream_t)stream);
    return hip_errno(hipGetLastError());
}

int sqz_hip_encode_blocks(const void* d_in, const uint64_t* d_in_off, uint32_t n, uint32_t window,
                          void* d_out, const uint64_t* d_out_off, uint64_t* d_out_bytes,
                          int32_t* d_err, void* d_scratch, uint64_t scratch_bytes, void* stream) {
    return sqz_hip_encode_blocks_stats(d_in, d_in_off, n, window, d_out, d_out_off, d_out_bytes, d_err,
                                       d_scratch, scratch_bytes, NULL, stream);
}

double sqz_stats_entropy(const uint32_t* freq, uint32_t n) {          // huffman.h:237-249
    if (freq == NULL) { return 0.0; }
    double total = 0.0, e = 0.0;
    for (uint32_t i = 0; i < n; i++) { total += (double)freq[i]; }
    for (uint32_t i = 0; i < n; i++) {
        if (freq[i] > 0) {
            const double p = (double)freq[i] / total;
            e += p * log2(p);
        }
    }
    return -e;
}

int sqz_hip_encode_blocks_stats(const void* d_in, const uint64_t* d_in_off, uint32_t n, uint32_t window,
                                void* d_out, const uint64_t* d_out_off, uint64_t* d_out_bytes,
                                int32_t* d_err, void* d_scratch, uint64_t scratch_bytes,
                                sqz_block_stats* d_stats, void* stream) {
    if (n == 0) { return 0; }
    if (d_scratch == NULL || scratch_bytes < sqz_hip_encode_scratch_bytes(n, 0)) { return EINVAL; }
    if (d_in == NULL || d_in_off == NULL || d_out == NULL || d_out_off == NULL ||
        d_out_bytes == NULL || d_err == NULL || !window_ok(window)) { return EINVAL; }
    int e = device_ready();
    if (e != 0) { return e; }
    const uint64_t head = align_up((uint64_t)n * 4, 256);
    // The offsets live on the device, so whether in_off[n] fits the scratch cannot be checked here
    // without a round trip: the kernels get `slots` and refuse (EINVAL, that block only) any
    // block whose in_off[b+1] lies beyond it.  Offsets are absolute slot indices: in_off[0]
    // need not be 0, but the arrays are addressed by them.
    const uint64_t slots = (scratch_bytes - head) / 8;     // >= total_in_bytes + 64 by contract
    uint32_t* counts = (uint32_t*)d_scratch;
    uint32_t* tokens = (uint32_t*)((uint8_t*)d_scratch + head);
    uint32_t* work_a = tokens;                             // sorted positions first, the token words afterwards
    uint32_t* work_m = work_a + slots;
    run_encode(finder_default(), (const uint8_t*)d_in, d_in_off, n, window, tokens, counts,
               work_a, work_m, slots / n, (uint8_t*)d_out, d_out_off, d_out_bytes, d_err, 0, 0,
               slots, d_stats, (hipStream_t)stream);
    return hip_errno(hipGetLastError());
}

uint64_t sqz_hip_decode_scratch_bytes(uint32_t n, uint64_t total_out_bytes) {
    return align_up((uint64_t)n * 4, 256) + (total_out_bytes + 64) * 4;
}

int sqz_hip_decode_blocks(const void* d_in, const uint64_t* d_in_off, uint32_t n, void* d_out,
                          const uint64_t* d_out_off, int32_t* d_err,
                          void* d_scratch, uint64_t scratch_bytes, void* stream) {
    if (n == 0) { return 0; }
    if (d_in == NULL || d_in_off == NULL || d_out == NULL || d_out_off == NULL || d_err == NULL ||
        d_scratch == NULL || scratch_bytes < sqz_hip_decode_scratch_bytes(n, 0)) {
        return EINVAL;
    }
    const int e = device_ready();
    if (e != 0) { return e; }
    uint32_t* counts = (uint32_t*)d_scratch;
    uint32_t* tokens = (uint32_t*)((uint8_t*)d_scratch + align_up((uint64_t)n * 4, 256));
    { SpanGuard g((hipStream_t)stream, SQZ_HIP_K_ENTROPY_DECODE);
      sqzk::launch_entropy_decode((const uint8_t*)d_in, d_in_off, d_out_off, tokens, counts, d_err,
                                  nullptr, n, 0, decode_waves_for(n), (hipStream_t)stream); }
    { SpanGuard g((hipStream_t)stream, SQZ_HIP_K_LZ_EXPAND);
      sqzk::launch_lz_expand(tokens, counts, (uint8_t*)d_out, d_out_off, n, (hipStream_t)stream); }
    return hip_errno(hipGetLastError());
}

int sqz_hip_debug_tree(const int32_t* d_symbols, uint32_t count, int which, int batch,
                       uint32_t* d_dump, void* stream) {
    if (d_symbols == NULL || d_dump == NULL || ((which & 0xFF) != 0 && (which & 0xFF) != 1)) { return EINVAL; }
    const int e = device_ready();
    if (e != 0) { return e; }
    sqzk::launch_tree_debug(d_symbols, count, which, batch, d_dump, (hipStream_t)stream);
    return hip_errno(hipGetLastError());
}

int sqz_hip_pack_blocks(const void* d_slabs, const uint64_t* d_slab_off, const uint64_t* d_bytes,
                        uint32_t n, void* d_dense, const uint64_t* d_dense_off,
                        uint64_t avg_bytes, void* stream) {
    if (n == 0) { return 0; }
    if (d_slabs == NULL || d_slab_off == NULL || d_bytes == NULL || d_dense == NULL ||
        d_dense_off == NULL) { return EINVAL; }
    const int e = device_ready();
    if (e != 0) { return e; }
    sqzk::launch_compact_blocks((const uint8_t*)d_slabs, d_slab_off, d_bytes, n, (uint8_t*)d_dense,
                                d_dense_off, avg_bytes, (hipStream_t)stream);
    return hip_errno(hipGetLastError());
}

void sqz_hip_set_finder(int finder) { g_finder.store(finder == 0 ? 0 : 1); }
int  sqz_hip_get_finder(void) { return finder_default(); }

// ------------------------------------------------------------------ R-era API (include/sqz/sqz_rc.h)
uint64_t sqz_rc_bound(uint64_t bytes) { return 2 * bytes + 64; }

void sqz_rc_init(struct sqz_rc* s, struct sqz_rc_map_entry entry[], size_t n) {       // src/sqz.c:550-565
    (void)entry; (void)n;                     // HEAD disables its map inside sqz_compress (src/sqz.c:591)
    if (s == NULL) { return; }
    s->rc.low = 0;                            // rc_init :485-490 (write / read / that are the caller's)
    s->rc.range = ~0ull;
    s->rc.code = 0;
    s->rc.error = 0;
    memset(s->reserved, 0, sizeof(s->reserved));
}

int sqz_hip_rc_encode_blocks(const void* d_in, const uint64_t* d_in_off, uint32_t n, void* d_out,
                             const uint64_t* d_out_off, uint64_t* d_out_bytes, int32_t* d_err, void* stream) {
    if (n == 0) { return 0; }
    if (d_in == NULL || d_in_off == NULL || d_out == NULL || d_out_off == NULL || d_out_bytes == NULL ||
        d_err == NULL) { return EINVAL; }
    const int e = device_ready();
    if (e != 0) { return e; }
    SpanGuard g((hipStream_t)stream, SQZ_HIP_K_RC_ENCODE);
    sqzk::launch_rc_encode((const uint8_t*)d_in, d_in_off, (uint8_t*)d_out, d_out_off, d_out_bytes, d_err, n,
                           (hipStream_t)stream);
    return hip_errno(hipGetLastError());
}

int sqz_hip_rc_decode_blocks(const void* d_in, const uint64_t* d_in_off, uint32_t n, void* d_out,
                             const uint64_t* d_out_off, uint64_t* d_out_bytes, uint64_t* d_consumed,
                             int32_t* d_err, void* stream) {
    if (n == 0) { return 0; }
    if (d_in == NULL || d_in_off == NULL || d_out == NULL || d_out_off == NULL || d_out_bytes == NULL ||
        d_err == NULL) { return EINVAL; }
    const int e = device_ready();
    if (e != 0) { return e; }
    SpanGuard g((hipStream_t)stream, SQZ_HIP_K_RC_DECODE);
    sqzk::launch_rc_decode((const uint8_t*)d_in, d_in_off, (uint8_t*)d_out, d_out_off, d_out_bytes, d_consumed,
                           d_err, n, 0, (hipStream_t)stream);
    return hip_errno(hipGetLastError());
}

// one stream through the device, host buffers; returns a HIP-side errno (0 = the kernel ran)
static int rc_run_host(bool decode, const uint8_t* in, uint64_t in_bytes, uint8_t* out, uint64_t out_cap,
                       uint64_t* produced, uint64_t* consumed, int32_t* err, int dry_error) {
    int e = device_ready();
    if (e != 0) { return e; }
    LaneLease lease;
    Lane& c = *lease.lane;
    hipStream_t st = lease.stream(nullptr);
    if ((e = c.in.reserve(in_bytes + 16)) || (e = c.out.reserve(out_cap + 16)) || (e = c.in_off.reserve(16)) ||
        (e = c.out_off.reserve(16)) || (e = c.out_bytes.reserve(16)) || (e = c.err.reserve(8))) { return e; }
    const uint64_t io[2] = {0, in_bytes}, oo[2] = {0, out_cap};
    if (in_bytes > 0) { HIP_TRY(hipMemcpyAsync(c.in.p, in, in_bytes, hipMemcpyHostToDevice, st)); }
    HIP_TRY(hipMemcpyAsync(c.in_off.p, io, 16, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(c.out_off.p, oo, 16, hipMemcpyHostToDevice, st));
    uint64_t* d_sizes = (uint64_t*)c.out_bytes.p;               // [0] produced, [1] consumed
    if (decode) {
        sqzk::launch_rc_decode((const uint8_t*)c.in.p, (const uint64_t*)c.in_off.p, (uint8_t*)c.out.p,
                               (const uint64_t*)c.out_off.p, d_sizes, d_sizes + 1, (int32_t*)c.err.p, 1, dry_error, st);
    } else {
        sqzk::launch_rc_encode((const uint8_t*)c.in.p, (const uint64_t*)c.in_off.p, (uint8_t*)c.out.p,
                               (const uint64_t*)c.out_off.p, d_sizes, (int32_t*)c.err.p, 1, st);
    }
    HIP_TRY(hipGetLastError());
    uint64_t sizes[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(sizes, d_sizes, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(err, c.err.p, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const uint64_t got = sizes[0] < out_cap ? sizes[0] : out_cap;
    if (got > 0) {
        HIP_TRY(hipMemcpyAsync(out, c.out.p, got, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    *produced = sizes[0];
    if (consumed != NULL) { *consumed = decode ? sizes[1] : 0; }
    return 0;
}

void sqz_rc_compress(struct sqz_rc* s, const void* d, size_t b, uint32_t window) {   // src/sqz.c:590-791
    (void)window;                               // unused at HEAD but for an assert (:656)
    if (s == NULL || s->rc.error != 0) { return; }
    if ((b > 0 && d == NULL) || s->rc.write == NULL || b > kMaxStream) { s->rc.error = EINVAL; return; }
    std::vector<uint8_t> buf(sqz_rc_bound(b));
    uint64_t produced = 0;
    int32_t err = 0;
    const int e = rc_run_host(false, (const uint8_t*)d, b, buf.data(), buf.size(), &produced, NULL, &err, 0);
    if (e != 0) { s->rc.error = e; return; }
    if (err != 0) { s->rc.error = err; return; }
    for (uint64_t k = 0; k < produced && s->rc.error == 0; k++) { s->rc.write(&s->rc, buf[k]); }   // rc_emit :474-476
}

uint64_t sqz_rc_decompress(struct sqz_rc* s, void* data, size_t bytes) {               // src/sqz.c:793-839
    if (s == NULL || s->rc.error != 0) { return 0; }
    if ((bytes > 0 && data == NULL) || s->rc.read == NULL || bytes > kMaxStream) { s->rc.error = EINVAL; return 0; }
    std::vector<uint8_t> in;
    const uint64_t most = sqz_rc_bound(bytes) + 64;             // no stream of `bytes` literals is longer
    // The reference reads a byte when its decoder needs one (rc_consume :499-500); a device decode wants its
    // input up front, so the pull runs AHEAD of the decoder: 64 bytes, then twice as much each time the decoder
    // ran dry, decoding again from the start.  OVER-READ BOUND: at most max(64, 2 x the bytes the decoder
    // consumes) are pulled through rc.read (include/sqz/sqz_rc.h).
    uint64_t want = 64;
    bool dry = false;
    int src_error = 0;
    for (;;) {
        if (want > most) { want = most; }
        while (in.size() < want && !dry) {                      // rc_consume :499-500, ahead of the decoder
            const uint8_t v = s->rc.read(&s->rc);
            if (s->rc.error != 0) { src_error = s->rc.error; s->rc.error = 0; dry = true; break; }
            in.push_back(v);
        }
        uint64_t produced = 0, consumed = 0;
        int32_t err = 0;
        // a source that ended with an error: the decoder's first read past it sets that error, as the
        // reference's read callback does (test.c:112-121), and the loop ends where the reference's would
        const int e = rc_run_host(true, in.data(), in.size(), (uint8_t*)data, bytes, &produced, &consumed, &err,
                                  dry ? src_error : 0);
        if (e != 0) { s->rc.error = e; return 0; }
        if (consumed > in.size() && !dry && in.size() < most) { // ran past what was pulled: pull more, decode again
            want = 2 * in.size();
            continue;
        }
        s->rc.error = err;
        return produced < bytes ? produced : bytes;
    }
}

// ------------------------------------------------------------------ timing
void sqz_hip_set_timing(int enabled) {
    Timing& t = timing();
    std::lock_guard<std::mutex> g(t.mu);
    t.enabled = enabled != 0;
}

int sqz_hip_get_timing(sqz_hip_timing* out, int reset) {
    Timing& t = timing();
    std::lock_guard<std::mutex> g(t.mu);
    t.drain_locked();
    if (out != NULL) { *out = t.acc; }
    if (reset) { t.acc = sqz_hip_timing{}; }
    return 0;
}

} // extern "C"
