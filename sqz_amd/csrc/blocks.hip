// sqz_amd/csrc/blocks.hip -- moving a batch's streams between the fixed-stride slabs the
// encoder writes and a dense image (gfx950).
//
// The encoder leaves stream b at out + out_off[b] with out_bytes[b] bytes of a slab sized for
// the worst case (sqz_bound).  Whoever ships the batch elsewhere -- the host flavour's one
// device-to-host copy, the multi-GPU gather of SURVEY.md section 8e -- wants the streams back
// to back: block b at dst + dst_off[b], dst_off = exclusive prefix sum of out_bytes.  Every
// stream is a multiple of 8 bytes (bitstream.h:112-114) and every offset is too, so the copy
// runs in 16-byte rows where the alignment allows and 8-byte words otherwise.
#include "sqz_kernels.h"

namespace sqzk {

__global__ __launch_bounds__(256)
void compact_blocks_kernel(const uint8_t* __restrict__ src, const uint64_t* __restrict__ src_off,
                           const uint64_t* __restrict__ bytes, uint32_t n_blocks,
                           uint8_t* __restrict__ dst, const uint64_t* __restrict__ dst_off,
                           uint32_t groups) {
    const uint32_t b = blockIdx.x / groups, g = blockIdx.x % groups;
    if (b >= n_blocks) { return; }
    const uint64_t nbytes = bytes[b];          // a multiple of 8 unless the stream was cut short (E2BIG)
    const uint8_t* from = src + src_off[b];
    uint8_t* to = dst + dst_off[b];
    const uint64_t words = nbytes / 8;
    if ((((uintptr_t)from | (uintptr_t)to) & 15u) == 0) {
        const uint64_t rows = words / 2;
        for (uint64_t k = (uint64_t)g * blockDim.x + threadIdx.x; k < rows; k += (uint64_t)groups * blockDim.x) {
            reinterpret_cast<uint4*>(to)[k] = reinterpret_cast<const uint4*>(from)[k];
        }
        if ((words & 1u) != 0 && g == 0 && threadIdx.x == 0) {
            reinterpret_cast<uint64_t*>(to)[words - 1] = reinterpret_cast<const uint64_t*>(from)[words - 1];
        }
    } else {
        for (uint64_t k = (uint64_t)g * blockDim.x + threadIdx.x; k < words; k += (uint64_t)groups * blockDim.x) {
            reinterpret_cast<uint64_t*>(to)[k] = reinterpret_cast<const uint64_t*>(from)[k];
        }
    }
    if (g == 0 && threadIdx.x == 0) {
        for (uint64_t k = words * 8; k < nbytes; k++) { to[k] = from[k]; }
    }
}

void launch_compact_blocks(const uint8_t* src, const uint64_t* src_off, const uint64_t* bytes,
                           uint32_t n_blocks, uint8_t* dst, const uint64_t* dst_off,
                           uint64_t avg_bytes, hipStream_t stream) {
    if (n_blocks == 0) { return; }
    // enough workgroups to fill the chip when the batch is small, one per 64 KB otherwise
    uint64_t groups = (avg_bytes + 65535) / 65536;
    if (groups < 1) { groups = 1; }
    while (groups * n_blocks > 0x7FFFFFFFull) { groups = (groups + 1) / 2; }
    hipLaunchKernelGGL(compact_blocks_kernel, dim3((unsigned)(groups * n_blocks)), dim3(256), 0, stream,
                       src, src_off, bytes, n_blocks, dst, dst_off, (uint32_t)groups);
}

} // namespace sqzk
