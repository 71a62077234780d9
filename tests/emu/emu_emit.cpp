/* tests/emu/emu_emit.cpp -- TEST INFRASTRUCTURE ONLY: encode stage 2 and the tree debug entry of
 * sqz_amd/csrc/huffman_emit.hip, compiled for the CPU wave emulator (tests/emu/hip/hip_runtime.h). */
#include "../../sqz_amd/csrc/huffman_emit.hip"

extern "C" {
int emu_tree_debug(const int32_t* symbols, uint32_t count, int which, int batch, uint32_t* dump) {
    sqzk::launch_tree_debug(symbols, count, which, batch, dump, nullptr);
    return 0;
}
int emu_huffman_emit(const uint32_t* tokens, const uint64_t* tok_off, const uint32_t* tok_count, uint32_t n,
                     uint8_t* out, const uint64_t* out_off, uint64_t* out_bytes, int32_t* err,
                     sqz_block_stats* stats) {
    sqzk::launch_huffman_emit(tokens, tok_off, tok_count, out, out_off, out_bytes, err, n, 0, 0, stats, nullptr);
    return 0;
}
}
