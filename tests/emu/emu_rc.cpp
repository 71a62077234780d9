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
    sqzk::launch_rc_decode(in, in_off, out, out_off, out_bytes, consumed, err, n, nullptr);
    return 0;
}
}
