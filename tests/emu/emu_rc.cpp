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
/* a source that fails at its end: the first read past it sets `dry_error` (the reference's read callback, test.c:112-121) */
int emu_rc_decode_dry(const uint8_t* in, const uint64_t* in_off, uint32_t n, uint8_t* out, const uint64_t* out_off,
                      uint64_t* out_bytes, uint64_t* consumed, int32_t* err, int dry_error) {
    sqzk::launch_rc_decode(in, in_off, out, out_off, out_bytes, consumed, err, n, dry_error, nullptr);
    return 0;
}
}
