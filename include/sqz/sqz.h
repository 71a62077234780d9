/* include/sqz/sqz.h -- C ABI of libsqz_amd.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE path of leok7v/sqz: the LZ77 longest-match scan +
 * adaptive-Huffman emit (encode) and its inverse (decode).  C99, plain pointers
 * and sizes; no HIP or torch types appear in any signature (streams and device
 * pointers travel as void*).
 *
 * Three layers, all backed by the same HIP kernels (there is NO CPU fallback:
 * every entry point reports ENODEV when no gfx950 device can be opened):
 *
 *  1. single-stream API with the reference's documented names
 *     (sqz_init / sqz_write_header / sqz_compress / sqz_read_header /
 *      sqz_decompress, `struct sqz`, `sqz_type`, `struct bitstream`):
 *       - names + call shape: /root/reference/shl/README.md:31-62,
 *         /root/reference/README.md:66-131 ("H1" in SURVEY.md section 0)
 *       - behaviour (bit-exact): attic/map_experiment/squeeze.h ("H0"):
 *         compress :319-409, decompress :502-551, header :255-265/:444-456
 *  2. the H0 vtable spelling `squeeze` (squeeze.h:109-131) for callers of the
 *     attic harness (attic/map_experiment/test.c:54-61,114-134)
 *  3. batch API over independent blocks -- the data-parallel hot path:
 *     host-buffer flavour and device-resident flavour (what bench.py times).
 *
 * Errors are the reference's sticky errno integers (squeeze.h:82,224-237;
 * bitstream.h:15,38,74): 0, EINVAL, E2BIG, ENOMEM; plus ENODEV (no GPU).
 */
#ifndef SQZ_AMD_SQZ_H
#define SQZ_AMD_SQZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define SQZ_API __attribute__((visibility("default")))
#else
#define SQZ_API
#endif

enum {
    sqz_min_win_bits = 10,  /* squeeze.h:19  squeeze_min_win_bits */
    sqz_max_win_bits = 15,  /* squeeze.h:20  squeeze_max_win_bits */
    sqz_min_len      = 3,   /* squeeze.h:13  squeeze_deflate_len_min */
    sqz_max_len      = 257, /* squeeze.h:15  squeeze_deflate_len_max */
    sqz_header_bits  = 72   /* squeeze.h:255-265: 64 (bytes) + 8 (win_bits) */
};

/* The public constants of the H0 header under the reference's own names (squeeze.h:9-25), same
 * values: what a caller of the `squeeze` vtable compiles against (attic/map_experiment/test.c
 * passes win_bits in [squeeze_min_win_bits, squeeze_max_win_bits] and sizes init_with's block
 * with squeeze_sizeof).                                                                      */
enum {
    squeeze_deflate_sym_min = 257,  /* squeeze.h:10 first length symbol of the literal/length alphabet */
    squeeze_deflate_sym_max = 284,  /* squeeze.h:11 last one                                           */
    squeeze_deflate_pos_max = 29,   /* squeeze.h:12 last distance code                                 */
    squeeze_deflate_len_min = 3,    /* squeeze.h:13 */
    squeeze_deflate_len_max = 257   /* squeeze.h:15 */
};
enum {
    squeeze_min_win_bits = 10,      /* squeeze.h:19 */
    squeeze_max_win_bits = 15,      /* squeeze.h:20 */
    squeeze_min_map_bits = 16,      /* squeeze.h:21 (the map experiment: map_bits != 0 is refused here) */
    squeeze_max_map_bits = 28,      /* squeeze.h:22 */
    squeeze_lit_nyt = squeeze_deflate_sym_max + 1,   /* squeeze.h:23 = 285 */
    squeeze_pos_nyt = squeeze_deflate_pos_max + 1    /* squeeze.h:24 = 30  */
};

/* ------------------------------------------------------------------ */
/* Bit stream: `bitstream` of attic/map_experiment/bitstream.h:7-18, same fields in the same
 * order, so the reference's designated initialisers compile unchanged
 * (attic/map_experiment/test.c:53,110):
 *
 *   memory mode   { .data = buf, .capacity = sizeof buf }            writer -> .bytes produced
 *                 { .data = buf, .bytes = n }                        reader (H0, test.c:110)
 *                 { .data = buf, .capacity = n }                     reader (H1, shl/README.md:47-48)
 *   callback mode { .stream = f, .output = write_file }              writer (bitstream.h:44-48)
 *                 { .stream = f, .input  = read_file  }              reader (bitstream.h:81-85)
 *
 * Callback mode is served by the shim, since a device cannot call the host per word
 * (SURVEY.md section 8b "Ownership"):
 *   writer  the stream is produced on the device into a host buffer, then every 64-bit word
 *           is handed over exactly as bitstream.h:45-47 does: bs->b64 = word, bs->error =
 *           bs->output(bs), bs->bytes += 8 on success, stop at the first error.  Same calls,
 *           same order, same values as the reference makes.
 *   reader  words are pulled with bs->input(bs) (bs->b64 = the word, bitstream.h:83) into a host
 *           buffer and decoded from there.  The reference pulls a word when its bit reader
 *           runs dry; the device decodes a whole buffer at once, so the shim pulls ahead:
 *           4 words first, then twice as many each time the decoder runs dry (E2BIG), decoding
 *           again from the start.  OVER-READ BOUND: the words pulled are at most
 *           max(4, 2 x the words the stream holds), whatever its compression ratio -- the
 *           reference pulls exactly the words the stream holds (bitstream.h:81-85), so a
 *           caller whose source continues behind the stream (another record in the same file)
 *           must re-position the source from bs->read, which IS the reference's figure; a
 *           pulled word cannot be handed back.  A callback error only surfaces if the decoder
 *           needed words beyond it.  bs->read / bs->bits / bs->b64 are left as the reference's
 *           reader would leave them (read = words the DECODER consumed x 8).               */
typedef struct bitstream {
    void*    stream;   /* callback context (bitstream.h:8); exclusive with (data, capacity) */
    uint8_t* data;
    uint64_t capacity; /* data[capacity] */
    uint64_t bytes;    /* bytes written (writer) / available (reader) */
    uint64_t read;     /* bytes consumed by the reader */
    uint64_t b64;      /* bit shifting buffer (bitstream.h:13) */
    int32_t  bits;     /* bit count inside b64 (bitstream.h:14) */
    int32_t  error;    /* sticky errno (bitstream.h:15; errno_t is int) */
    int (*output)(struct bitstream* bs); /* write b64 as 8 bytes (bitstream.h:16) */
    int (*input)(struct bitstream* bs);  /* read b64 as 8 bytes  (bitstream.h:17) */
} bitstream;

/* codec state (squeeze_type, squeeze.h:81-92).  Caller-owned, single-use per stream like the
 * reference's (squeeze.h:333-334 inserts the NYT leaves at the start of every call).  The
 * trees themselves live on the device for the duration of a call; the struct carries what the
 * reference's callers read afterwards: the sticky error (first member, as in the reference)
 * and the bit stream of the last call (`s->bs->bytes`, attic test.c:84).                  */
struct sqz {
    int32_t  error;       /* sticky errno: README.md:126-131, squeeze.h:82 */
    int32_t  device;      /* HIP device ordinal used by the last call, -1 = default */
    uint64_t tokens;      /* LZ77 tokens of the last sqz_compress */
    struct bitstream* bs; /* squeeze.h:89: set by compress / decompress */
    void*    stream;      /* optional hipStream_t the call's copies and kernels are enqueued on; NULL (what
                           * sqz_init / alloc / init_with leave) = a stream of the library's own, one per
                           * call in flight.  Calls from different threads do not serialise on each other:
                           * every call stages through buffers of its own. */
    uint64_t reserved[3];
};
typedef struct sqz sqz_type; /* README.md:128 */

/* squeeze.h:94-107 squeeze_sizeof(map_bits): the size of the block `init_with` takes (squeeze.h:191-199).
 * The reference lays its trees out behind the struct; here they live in the device's LDS for the duration of
 * a call, so the block is the struct.  init_with accepts any size >= this (a caller that kept the reference's
 * larger figure still passes); map_bits must be 0.                                                       */
#define squeeze_sizeof(map_bits) (sizeof(struct sqz))

/* shl/README.md:37-38  `static struct sqz s; sqz_init(&s);` */
SQZ_API void sqz_init(struct sqz* s);

/* shl/README.md:34 spelling (2 arguments).  The H1 source is absent from the
 * reference snapshot, so its framing cannot be pinned: this writes the 64-bit
 * length only, LSB first (the first field of squeeze.h:255-265).            */
SQZ_API void sqz_write_header(struct bitstream* bs, uint64_t bytes);
SQZ_API void sqz_read_header(struct bitstream* bs, uint64_t* bytes);

/* squeeze.h:255-265 / :444-456 framing, pinned by tests/golden: 64 bits of
 * length + 8 bits of win_bits (10..15, else bs->error = EINVAL), LSB first,
 * no alignment before the payload.                                          */
SQZ_API void sqz_write_header_h0(struct bitstream* bs, uint64_t bytes, uint8_t win_bits);
SQZ_API void sqz_read_header_h0(struct bitstream* bs, uint64_t* bytes, uint8_t* win_bits);

/* squeeze.h:319-409.  `window` = 1u << win_bits (2..32768 accepted, like the
 * reference's uint16_t argument; max distance is window-1).  Continues the bit
 * stream wherever sqz_write_header* left it and zero-pads to a 64-bit boundary
 * (bitstream.h:112-114).  Result in s->error (mirrored to bs->error), output
 * size in bs->bytes.  E2BIG when bs->capacity is too small (bitstream.h:38).  In callback
 * mode (bs->data == NULL, bs->output set) every word goes through bs->output.            */
SQZ_API void sqz_compress(struct sqz* s, struct bitstream* bs,
                          const uint8_t* data, size_t bytes, uint32_t window);

/* squeeze.h:502-551.  Decodes exactly `bytes` bytes (taken from the header). */
SQZ_API void sqz_decompress(struct sqz* s, struct bitstream* bs,
                            uint8_t* data, size_t bytes);

/* ------------------------------------------------------------------ */
/* H0 vtable spelling: squeeze.h:109-131.  `map_bits` must be 0 (the map
 * experiment is disabled in the reference's default configuration,
 * attic/map_experiment/test.c:30-31) -- anything else is EINVAL / NULL.     */
typedef struct sqz squeeze_type;
typedef struct {
    squeeze_type* (*alloc)(uint8_t map_bits);
    int  (*init_with)(squeeze_type* s, void* memory, size_t size, uint8_t map_bits);
    void (*write_header)(bitstream* bs, uint64_t bytes, uint8_t win_bits);
    void (*compress)(squeeze_type* s, bitstream* bs,
                     const uint8_t* data, size_t bytes, uint16_t window);
    void (*read_header)(bitstream* bs, uint64_t* bytes, uint8_t* win_bits);
    void (*decompress)(squeeze_type* s, bitstream* bs, uint8_t* data, size_t bytes);
    void (*free)(squeeze_type* s);
} squeeze_interface;
SQZ_API extern squeeze_interface squeeze;

/* ------------------------------------------------------------------ */
/* Batch API: n independent blocks, each a self-contained stream with fresh
 * trees (squeeze.h:333-336), payload only (no header), every output a multiple
 * of 8 bytes.  Block b reads  in[in_off[b] .. in_off[b+1])  and writes at most
 * out_off[b+1]-out_off[b] bytes at out + out_off[b]; the size goes to
 * out_bytes[b], the errno to err[b].  Offsets arrays have n+1 entries.
 * out_off[b] must be a multiple of 8.  Only out[out_off[b] .. + out_bytes[b]) is written.     */

/* worst-case compressed size of one block of `bytes` bytes (multiple of 8) */
SQZ_API uint64_t sqz_bound(uint64_t bytes);

/* File-mode bit streams (attic/map_experiment/test.c:39-42,98-101; bitstream.h `.stream`,
 * `.output`, `.input`): the reference hands every 64-bit word of the stream to
 * fwrite(&b64, 8, 1, f) / fread(&b64, 8, 1, f), i.e. in HOST byte order, while a memory-mode
 * stream holds the words most-significant byte first.  This converts one image into the
 * other (the operation is its own inverse; on a big-endian host it is a copy): write
 * out[] with fwrite to get the file the reference's harness writes, or pass a file's bytes
 * through it to get the stream sqz_decompress / sqz_decode_blocks read.  `bytes` must be a
 * multiple of 8 (every stream is, bitstream.h:112-114); in == out is allowed.
 * Returns 0 or EINVAL.  Host code: no device is touched.                              */
SQZ_API int sqz_file_words(const uint8_t* in, uint64_t bytes, uint8_t* out);

/* host buffers in, host buffers out (H2D + kernels + D2H inside) */
SQZ_API int sqz_encode_blocks(const uint8_t* in, const uint64_t* in_off, uint32_t n,
                              uint32_t window,
                              uint8_t* out, const uint64_t* out_off,
                              uint64_t* out_bytes, int32_t* err);
SQZ_API int sqz_decode_blocks(const uint8_t* in, const uint64_t* in_off, uint32_t n,
                              uint8_t* out, const uint64_t* out_off,
                              int32_t* err);

/* Device-resident flavour: every pointer is a DEVICE pointer on the current
 * HIP device (hipMalloc / torch.cuda tensor storage), `stream` is a
 * hipStream_t (NULL = default stream).  Asynchronous: returns after enqueue.
 * `scratch` must hold sqz_hip_encode_scratch_bytes(n, in_off[n]): the per-byte work arrays
 * are addressed by the ABSOLUTE offsets in d_in_off (in_off[0] need not be 0).  The offsets
 * live on the device, so a scratch that is too small cannot be told at the call: the kernels
 * refuse every block whose in_off[b+1] lies beyond it -- err[b] = EINVAL, out_bytes[b] = 0,
 * nothing written out of bounds -- and the other blocks are unaffected.                    */
SQZ_API uint64_t sqz_hip_encode_scratch_bytes(uint32_t n, uint64_t total_in_bytes);
SQZ_API int sqz_hip_encode_blocks(const void* d_in, const uint64_t* d_in_off, uint32_t n,
                                  uint32_t window,
                                  void* d_out, const uint64_t* d_out_off,
                                  uint64_t* d_out_bytes, int32_t* d_err,
                                  void* d_scratch, uint64_t scratch_bytes,
                                  void* stream);
SQZ_API uint64_t sqz_hip_decode_scratch_bytes(uint32_t n, uint64_t total_out_bytes);
SQZ_API int sqz_hip_decode_blocks(const void* d_in, const uint64_t* d_in_off, uint32_t n,
                                  void* d_out, const uint64_t* d_out_off,
                                  int32_t* d_err, void* d_scratch, uint64_t scratch_bytes,
                                  void* stream);

/* The two encode stages on their own (parity tests pin each independently,
 * SURVEY.md section 8c; also what a caller with its own entropy stage binds):
 *  stage 1  squeeze.h:338-358 + greedy step :377-394 -> token words
 *           literal 0x000000bb ; match 0x80000000 | len<<16 | dist
 *           block b's tokens start at d_tokens + in_off[b] (one slot per byte)
 *  stage 2  squeeze.h:278-315 + huffman.h + bitstream.h -> payload bytes    */
SQZ_API int sqz_hip_lz77_blocks(const void* d_in, const uint64_t* d_in_off, uint32_t n,
                                uint32_t window, uint32_t* d_tokens,
                                uint32_t* d_token_count, void* stream);
/* stage 1 with an explicit match finder: 0 = brute-force scan (squeeze.h:340-358 as
 * written), 1 = indexed (same tokens; visits only the earlier positions that share
 * the 3-byte prefix, nearest first -- SURVEY.md section 8f-3).  finder 1 needs
 * d_work = 8 bytes per input byte (+512).                                        */
SQZ_API int sqz_hip_lz77_blocks_ex(const void* d_in, const uint64_t* d_in_off, uint32_t n,
                                   uint32_t window, uint32_t* d_tokens,
                                   uint32_t* d_token_count, int finder,
                                   void* d_work, uint64_t work_bytes, void* stream);
/* stage 2 alone, on the caller's token words.  Every word is checked (a byte, or len 3..257 and
 * dist 1..32767, no stray bits) and a block may not hold more tokens than it has slots: a
 * violation fails that block with err[b] = EINVAL, other blocks are unaffected.  Whether the
 * tokens describe a consistent text (distances within what precedes them, lengths adding up) is
 * the caller's business: an inconsistent sequence encodes to a stream the decoder rejects. */
SQZ_API int sqz_hip_huffman_blocks(const uint32_t* d_tokens, const uint64_t* d_in_off,
                                   const uint32_t* d_token_count, uint32_t n,
                                   void* d_out, const uint64_t* d_out_off,
                                   uint64_t* d_out_bytes, int32_t* d_err,
                                   void* stream);

/* Counters the reference keeps next to the hot path, per block, opt-in:
 *   huffman.h:29-33   stats.updates (huffman_update_paths calls, one per node visited, :42),
 *                     stats.swaps (:76), stats.moves (:111), for each tree
 *   squeeze.h:327-328,386,391  li_bytes / br_bytes: source bytes coded as literals / as back
 *                     references (what SQUEEZE_MAP_STATS prints as percentages, :397-403)
 *   huffman.h:26      the depth marks at the end;  huffman.h:237-249 huffman_entropy is
 *                     sqz_stats_entropy() over the leaf counts returned here
 * Pass a device array of n entries to sqz_hip_encode_blocks_stats (NULL = no counters).
 * The update counter follows the reference while a tree is shallower than 26 levels (any
 * stream of less than 2^25 symbols); deeper trees keep swaps / moves only.                  */
typedef struct sqz_block_stats {
    uint32_t lit_updates, lit_swaps, lit_moves;
    uint32_t pos_updates, pos_swaps, pos_moves;
    uint32_t literal_bytes, backref_bytes;
    uint32_t lit_depth, pos_depth;
    uint32_t tokens, reserved;
    uint32_t lit_freq[288];      /* count of every leaf of the literal/length tree (symbol = index) */
    uint32_t pos_freq[32];       /* ... of the distance tree */
} sqz_block_stats;
SQZ_API int sqz_hip_encode_blocks_stats(const void* d_in, const uint64_t* d_in_off, uint32_t n,
                                        uint32_t window,
                                        void* d_out, const uint64_t* d_out_off,
                                        uint64_t* d_out_bytes, int32_t* d_err,
                                        void* d_scratch, uint64_t scratch_bytes,
                                        sqz_block_stats* d_stats, void* stream);
/* huffman.h:237-249: Shannon entropy (bits per symbol) of `n` leaf counts.  Host arithmetic. */
SQZ_API double sqz_stats_entropy(const uint32_t* freq, uint32_t n);

/* Test entry: drive ONE adaptive Huffman tree on the device with a symbol sequence through
 * huffman_inc_frequency semantics (huffman.h:218-235; unseen symbol -> huffman_insert :149) --
 * exactly the code paths the encoder uses: runs of attached symbols in batches of `batch`
 * (1..64) through the batched update, everything else one at a time -- and dump its node arrays.
 * which: 0 = the literal/length tree (n = 512 in the reference, 288 leaf ids here), 1 = the
 * distance tree (n = 32).  d_dump: 8 + 4 * nodes uint32 (nodes = 576 / 64):
 *   [0..7]  next id, depth mark (huffman.h:26), complete (:27), intervals kept, fault,
 *           stats.updates, stats.swaps, stats.moves (huffman.h:29-33)
 *   then per node id v:  up | lo << 10 | hi << 20 (0x3FF = none; ids are absolute: + 576 for the
 *           distance tree),  st | en << 9 | partner << 18,  count | depth << 24,  code (leaves)
 * Leaves keep their symbol value as id, the root is id `leaves`, internal nodes count up from it
 * (the reference counts down from 2n-2; no emitted bit depends on the numbering).           */
SQZ_API int sqz_hip_debug_tree(const int32_t* d_symbols, uint32_t count, int which, int batch,
                               uint32_t* d_dump, void* stream);

/* Streams of a batch, back to back: stream b moves from d_slabs + d_slab_off[b] (where
 * sqz_hip_encode_blocks left it) to d_dense + d_dense_off[b]; d_dense_off = exclusive prefix
 * sum of d_bytes, every entry a multiple of 8 (n + 1 entries, computed by the caller on the device).  The dense image with
 * d_dense_off as offsets is what sqz_hip_decode_blocks reads, and what the multi-GPU gather
 * of SURVEY.md section 8e ships (110 MB per 512 blocks instead of 269 MB of slabs).
 * avg_bytes sizes the launch (total / n is fine).  All device pointers, asynchronous.       */
SQZ_API int sqz_hip_pack_blocks(const void* d_slabs, const uint64_t* d_slab_off,
                                const uint64_t* d_bytes, uint32_t n,
                                void* d_dense, const uint64_t* d_dense_off,
                                uint64_t avg_bytes, void* stream);

/* which finder sqz_compress / sqz_*encode_blocks use: 1 = indexed (default),
 * 0 = brute-force scan; also settable with SQZ_FINDER=scan|index.           */
SQZ_API void sqz_hip_set_finder(int finder);
SQZ_API int  sqz_hip_get_finder(void);

/* Live timing of the last kernels enqueued through this library on the
 * calling thread's context, measured with HIP events ON THE LAUNCH STREAM.
 * Enabled with sqz_hip_set_timing(1); values in milliseconds.               */
enum {
    SQZ_HIP_K_LZ77_SCAN = 0,      /* lz77_scan_kernel (brute-force finder)            */
    SQZ_HIP_K_HUFFMAN_EMIT = 1,   /* huffman_emit_kernel                              */
    SQZ_HIP_K_ENTROPY_DECODE = 2, /* entropy_decode_kernel                            */
    SQZ_HIP_K_INDEX_SORT = 3,     /* index_sort_kernel   } indexed finder             */
    SQZ_HIP_K_INDEX_MATCH = 4,    /* index_match_kernel  }                            */
    SQZ_HIP_K_INDEX_PARSE = 5,    /* index_parse_kernel  }                            */
    SQZ_HIP_K_LZ_EXPAND = 6,      /* lz_expand_kernel                                 */
    SQZ_HIP_K_RC_ENCODE = 7,      /* rc_encode_kernel  } R-era range coder            */
    SQZ_HIP_K_RC_DECODE = 8,      /* rc_decode_kernel  } (include/sqz/sqz_rc.h)       */
    SQZ_HIP_KERNELS = 12
};
typedef struct sqz_hip_timing {
    float    ms[SQZ_HIP_KERNELS];        /* summed launch durations per kernel        */
    uint32_t launches[SQZ_HIP_KERNELS];
} sqz_hip_timing;
SQZ_API void sqz_hip_set_timing(int enabled);
SQZ_API int  sqz_hip_get_timing(sqz_hip_timing* out, int reset);

/* device / build information; returns 0 when a gfx950 device is usable */
SQZ_API int sqz_hip_device_info(char* name, size_t name_cap, int* compute_units,
                                uint64_t* lds_bytes_per_cu);
SQZ_API const char* sqz_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SQZ_AMD_SQZ_H */
