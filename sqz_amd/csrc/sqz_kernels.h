// sqz_amd/csrc/sqz_kernels.h -- host-visible launchers of the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sqz/sqz.h"      // sqz_block_stats

namespace sqzk {

// stage 1: LZ77 brute-force longest-match scan + greedy parse
// (attic/map_experiment/squeeze.h:338-358, :377-394).  Block b reads
// in[in_off[b] .. in_off[b+1]) and writes its token words at
// tokens + in_off[b]; the count goes to tok_count[b].
// waves_per_stream: 1, 2, 4 (default) or 8 wavefronts share one stream's window.
// slots: how many uint32 slots the per-byte arrays (tokens, buf_a, buf_b, match) hold; a block
// whose in_off[b+1] lies beyond them is refused (tok_count[b] = kRefused -> EINVAL in stage 2)
// instead of being written out of bounds.
void launch_lz77_scan(const uint8_t* in, const uint64_t* in_off, uint32_t n_blocks,
                      uint32_t window, uint32_t* tokens, uint32_t* tok_count,
                      int waves_per_stream, uint64_t slots, hipStream_t stream);

// stage 1, indexed form: same tokens as launch_lz77_scan without visiting every
// distance (lz77_index.hip): sort -> match -> parse.  buf_a / buf_b / match: one
// uint32 slot per input byte each, addressed like `tokens`.
void launch_index_sort(const uint8_t* in, const uint64_t* in_off, uint32_t n_blocks,
                       uint32_t* buf_a, uint32_t* buf_b, uint32_t* tmp, uint64_t slots,
                       hipStream_t stream);
void launch_index_match(const uint8_t* in, const uint64_t* in_off, uint32_t n_blocks,
                        uint32_t window, const uint32_t* sorted, uint32_t* match,
                        uint32_t match_groups, uint64_t slots, hipStream_t stream);
void launch_index_parse(const uint8_t* in, const uint64_t* in_off, uint32_t n_blocks,
                        const uint32_t* match, uint32_t* tokens, uint32_t* tok_count,
                        uint64_t slots, hipStream_t stream);

// stage 2: adaptive-Huffman emit (squeeze.h:278-315, huffman.h, bitstream.h)
void launch_huffman_emit(const uint32_t* tokens, const uint64_t* tok_off,
                         const uint32_t* tok_count, uint8_t* out,
                         const uint64_t* out_off, uint64_t* out_bytes,
                         int32_t* err, uint32_t n_blocks,
                         uint64_t prefix_acc, int prefix_fill, sqz_block_stats* stats,
                         hipStream_t stream);

// test entry: one tree driven by a symbol sequence, its LDS image dumped (huffman_emit.hip)
void launch_tree_debug(const int32_t* symbols, uint32_t count, int which, int batch, uint32_t* dump,
                       hipStream_t stream);

// decode (squeeze.h:502-551): entropy stage -> token words -> LZ77 expansion.
// tokens: one uint32 slot per OUTPUT byte, addressed by out_off; tok_count[n].
void launch_entropy_decode(const uint8_t* in, const uint64_t* in_off, const uint64_t* out_off,
                           uint32_t* tokens, uint32_t* tok_count, int32_t* err, uint64_t* end_bit,
                           uint32_t n_blocks, uint64_t start_bit, int waves, hipStream_t stream);
void launch_lz_expand(const uint32_t* tokens, const uint32_t* tok_count, uint8_t* out,
                      const uint64_t* out_off, uint32_t n_blocks, hipStream_t stream);

// the reference's HEAD range coder (range_coder.hip): literal-only encode as HEAD
// runs it, decode as written.  Block b: in[in_off[b] .. in_off[b+1]) -> out + out_off[b], at most
// out_off[b+1] - out_off[b] bytes; sizes to out_bytes[b], errno (EINVAL/EILSEQ/ERANGE/ENOBUFS) to err[b].
void launch_rc_encode(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                      uint64_t* out_bytes, int32_t* err, uint32_t n_blocks, hipStream_t stream);
void launch_rc_decode(const uint8_t* in, const uint64_t* in_off, uint8_t* out, const uint64_t* out_off,
                      uint64_t* out_bytes, uint64_t* consumed, int32_t* err, uint32_t n_blocks, int dry_error,
                      hipStream_t stream);

// slabs -> dense image of a batch's streams (blocks.hip): block b moves from src + src_off[b]
// to dst + dst_off[b].  bytes[b] and all offsets are multiples of 8.
void launch_compact_blocks(const uint8_t* src, const uint64_t* src_off, const uint64_t* bytes,
                           uint32_t n_blocks, uint8_t* dst, const uint64_t* dst_off,
                           uint64_t avg_bytes, hipStream_t stream);

} // namespace sqzk
