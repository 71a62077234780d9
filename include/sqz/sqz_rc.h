/* include/sqz/sqz_rc.h -- the reference's HEAD ("R-era") interface on libsqz_amd.so (MI355X / gfx950).
 *
 * SURVEY.md section 8f-1.  At HEAD the reference is an adaptive order-0 range coder behind
 *     void     sqz_init(struct sqz*, struct map_entry entry[], size_t n);            inc/sqz/sqz.h:87
 *     void     sqz_compress(struct sqz*, const void* d, size_t b, uint32_t window);  inc/sqz/sqz.h:88
 *     uint64_t sqz_decompress(struct sqz*, void* data, size_t bytes);                inc/sqz/sqz.h:89
 * with byte I/O through rc.write / rc.read callbacks (inc/sqz/sqz.h:45-53) and the caller's context in
 * `that` (the `rc`-is-first-field cast, inc/sqz/sqz.h:71,81; used at test.c:49-50,114-115).  Those
 * names collide with the Huffman-era API of <sqz/sqz.h> (same library, different signatures), so the
 * entry points here carry an `rc` in their names; define SQZ_RC_REFERENCE_NAMES before including this
 * header (and do not include <sqz/sqz.h> in the same file) to get the reference's own spellings, so
 * that HEAD's callers (shl.c:23-68, test.c:57-180) compile against it as they are.
 *
 * What is computed is what HEAD computes (/root/reference/src/sqz.c): its match finders are compiled
 * out (:630-631, :660, :591), so sqz_compress codes every byte as a literal through the range coder
 * (:722-723) and closes the stream with flag 0 + size 0xFF + an 8-byte flush (:741-743); `window` and
 * the map entries are accepted and unused, as at HEAD.  sqz_decompress is :793-839 as written.
 * Bit-exact against the reference compiled here (tests/golden/golden_rc.json).
 *
 * The coder itself runs on the device (sqz_amd/csrc/range_coder.hip).  The byte callbacks cannot
 * cross to it, so the shim serves them: compress runs to a host buffer and hands every byte to
 * rc.write in order (stopping at the first error the callback raises in rc.error); decompress pulls
 * bytes with rc.read ahead of the decoder -- 64 bytes, then twice as much each time the decoder ran
 * past what was pulled (it starts over: the total work stays within twice the last attempt).
 * OVER-READ BOUND: at most max(64, 2 x the bytes the reference would have asked for) are read through
 * rc.read; the reference asks for exactly the stream (rc_consume, src/sqz.c:499-500), so a caller whose
 * source continues behind the stream must re-position it itself.  A source that FAILS at its end (the
 * callback sets rc.error, as the reference's own does, test.c:112-121) ends the decode where the
 * reference's loop ends (:800-802): same return value, same rc.error.  rc.low / rc.range / rc.code are
 * not meaningful after a call.
 * Errors are the reference's errno values in rc.error: EINVAL, EILSEQ (src/sqz.c:523-541), ERANGE,
 * ENOBUFS (:807-833), plus ENODEV (no gfx950 device: there is no CPU fallback) and ENOMEM.
 */
#ifndef SQZ_AMD_SQZ_RC_H
#define SQZ_AMD_SQZ_RC_H

#include <stddef.h>
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

struct sqz_rc_range_coder {                 /* struct range_coder, inc/sqz/sqz.h:45-53: same fields, same order */
    uint64_t low;
    uint64_t range;
    uint64_t code;
    void    (*write)(struct sqz_rc_range_coder*, uint8_t);
    uint8_t (*read)(struct sqz_rc_range_coder*);
    int32_t  error;                         /* sticky error (e.g. errno_t from read/write) */
    int32_t  padding;
};

struct sqz_rc_map_entry { const uint8_t* data; uint64_t hash; int32_t bytes; };   /* inc/sqz/sqz.h:55-59 (unused at HEAD) */

struct sqz_rc {                             /* struct sqz, inc/sqz/sqz.h:69-79: rc first, then `that` */
    struct sqz_rc_range_coder rc;
    void*    that;                          /* convenience for caller i/o override */
    uint64_t reserved[8];                   /* (the reference keeps its models here; they live on the device) */
};

SQZ_API void     sqz_rc_init(struct sqz_rc* s, struct sqz_rc_map_entry entry[], size_t n);
SQZ_API void     sqz_rc_compress(struct sqz_rc* s, const void* d, size_t b, uint32_t window);
SQZ_API uint64_t sqz_rc_decompress(struct sqz_rc* s, void* data, size_t bytes);

/* most bytes sqz_rc_compress hands to rc.write for an input of `bytes` bytes */
SQZ_API uint64_t sqz_rc_bound(uint64_t bytes);

/* Batch, device resident (every pointer a DEVICE pointer, `stream` a hipStream_t): n independent
 * streams, block b = in[in_off[b] .. in_off[b+1]) -> out + out_off[b], at most out_off[b+1] - out_off[b]
 * bytes (ENOBUFS beyond); produced sizes in out_bytes[b], the reference's errno in err[b].  Decode stops at
 * the end-of-stream symbol; consumed[b] (optional) = stream bytes the decoder asked for.  Asynchronous. */
SQZ_API int sqz_hip_rc_encode_blocks(const void* d_in, const uint64_t* d_in_off, uint32_t n,
                                     void* d_out, const uint64_t* d_out_off, uint64_t* d_out_bytes,
                                     int32_t* d_err, void* stream);
SQZ_API int sqz_hip_rc_decode_blocks(const void* d_in, const uint64_t* d_in_off, uint32_t n,
                                     void* d_out, const uint64_t* d_out_off, uint64_t* d_out_bytes,
                                     uint64_t* d_consumed, int32_t* d_err, void* stream);

#ifdef SQZ_RC_REFERENCE_NAMES               /* the spellings of /root/reference/inc/sqz/sqz.h */
#define range_coder    sqz_rc_range_coder
#define map_entry      sqz_rc_map_entry
#define sqz            sqz_rc
#define sqz_init       sqz_rc_init
#define sqz_compress   sqz_rc_compress
#define sqz_decompress sqz_rc_decompress
#endif

#ifdef __cplusplus
}
#endif
#endif /* SQZ_AMD_SQZ_RC_H */
