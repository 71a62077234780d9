/* include/sqz/sqz_workload.h -- synthetic benchmark workload of libsqz_amd.so.
 *
 * Not part of the codec path: fills device memory with the Zipf(s=1) byte
 * blocks that BASELINE.json configs[2] names (4096 x 256 KB), generated as
 * pinned in SURVEY.md section 8d, so bench.py starts with HBM-resident input.
 */
#ifndef SQZ_AMD_WORKLOAD_H
#define SQZ_AMD_WORKLOAD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#ifndef SQZ_API
#if defined(__GNUC__)
#define SQZ_API __attribute__((visibility("default")))
#else
#define SQZ_API
#endif
#endif

/* d_out: device pointer, n_blocks * block_bytes bytes; block b holds the
 * generator's block (first_block + b).  block_bytes must be a multiple of 4. */
SQZ_API int sqz_hip_zipf_blocks(void* d_out, uint64_t first_block, uint64_t n_blocks,
                        uint64_t block_bytes, void* stream);

/* the 256-entry CDF the generator uses (host memory) */
SQZ_API const uint32_t* sqz_zipf_cdf_table(void);

#ifdef __cplusplus
}
#endif
#endif
