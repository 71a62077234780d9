/* tests/emu/emu_decode.cpp -- TEST INFRASTRUCTURE ONLY: the decoder kernels of sqz_amd/csrc/decode.hip,
 * compiled for the CPU wave emulator (tests/emu/hip/hip_runtime.h). */
#include "../../sqz_amd/csrc/decode.hip"

extern "C" int emu_decode_waves(const uint8_t* in, const uint64_t* in_off, uint32_t n, uint8_t* out, const uint64_t* out_off,
                                uint32_t* tokens, uint32_t* tok_count, int32_t* err, int waves) {
    sqzk::launch_entropy_decode(in, in_off, out_off, tokens, tok_count, err, nullptr, n, 0, waves, nullptr);
    sqzk::launch_lz_expand(tokens, tok_count, out, out_off, n, nullptr);
    return 0;
}

extern "C" int emu_decode(const uint8_t* in, const uint64_t* in_off, uint32_t n, uint8_t* out, const uint64_t* out_off,
                          uint32_t* tokens, uint32_t* tok_count, int32_t* err) {
    sqzk::launch_entropy_decode(in, in_off, out_off, tokens, tok_count, err, nullptr, n, 0, 1, nullptr);
    sqzk::launch_lz_expand(tokens, tok_count, out, out_off, n, nullptr);
    return 0;
}
