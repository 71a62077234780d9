/* tests/emu/emu_rc.cpp -- TEST INFRASTRUCTURE ONLY: the range-coder kernels of sqz_amd/csrc/range_coder.hip,
 * compiled for the CPU wave emulator (tests/emu/hip/hip_runtime.h). */
#include "../../sqz_amd/csrc/range_coder.hip"

extern "C" {
int emu_rc_encode(const uint8_t* in, const uint64_t* in_off, uint32_t n, uint8_t* out, const uint64_t* out_off,
                  uint64_t* out_bytes, int32_t* err) {
    sqzk::launch_rc_encode(in, in_off, out, out_off, out_bytes, err, n, nullptr);
    return 0;
}
int emu_rc_decode(const uint8_t* in, const uint64_t* in_off, uint32_t n, uint8_t* out, const uint64_t* out_off,
                  uint64_t* out_bytes, uint64_t* consumed, int32_t* err) {
    sqzk::launch_rc_decode(in, in_off, out, out_off, out_bytes, consumed, err, n, 0, nullptr);
    return 0;
}
/* rc_div against the host's 64-bit division: the number of pairs that differ (first one in *bad) */
uint64_t emu_rc_div_mismatches(const uint64_t* a, const uint64_t* d, uint64_t n, uint64_t* bad) {
    uint64_t wrong = 0;
    for (uint64_t k = 0; k < n; k++) {
        if (d[k] == 0) { continue; }
        bool ok = sqzk::rc_div_lanes(a[k], d[k], sqzk::rc_recip_low(d[k])) == a[k] / d[k];
        if (d[k] <= 0xFFFFFFFFull) { ok = ok && sqzk::rc_div_lanes(a[k], (uint32_t)d[k], sqzk::rc_recip_low(d[k])) == a[k] / d[k]; }   // a model's total
        if (!ok) { if (wrong == 0 && bad != nullptr) { *bad = k; } wrong++; }
    }
    return wrong;
}
/* a source that fails at its end: the first read past it sets `dry_error` (the reference's read callback, test.c:112-121) */
int emu_rc_decode_dry(const uint8_t* in, const uint64_t* in_off, uint32_t n, uint8_t* out, const uint64_t* out_off,
                      uint64_t* out_bytes, uint64_t* consumed, int32_t* err, int dry_error) {
    sqzk::launch_rc_decode(in, in_off, out, out_off, out_bytes, consumed, err, n, dry_error, nullptr);
    return 0;
}
}
