/* oracle/ref_wrap.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin wrapper translation unit that compiles the reference's own H0 codec
 * (attic/map_experiment/{bitstream,huffman,map,squeeze}.h) *where it lies*
 * under /root/reference (via -I, see oracle/Makefile) into oracle/_ref/.
 * No reference source is copied into this repository; this file only calls
 * the reference's public vtable `squeeze` (squeeze.h:109-131, 557-565) with
 * in-memory bitstreams (bitstream.h:7-18, memory mode :34-43 / :70-80).
 *
 * The two things the reference takes from the MSVC CRT / its rt.h
 * (`errno_t`, `null`) are supplied as -D flags by the Makefile.
 *
 * Exposed C ABI (used by tests/ and oracle/gen_golden.py only):
 *   sqz_ref_compress / sqz_ref_decompress / sqz_ref_available /
 *   sqz_ref_tree_run / sqz_ref_compress_file / sqz_ref_decompress_file / sqz_ref_compress_stats
 */
#include <stdbool.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <errno.h>
#include <assert.h>

#include "bitstream.h"
#include "squeeze.h"

int sqz_ref_available(void) { return 1; }

/* returns bytes written (>=0) or -errno */
int64_t sqz_ref_compress(const uint8_t* data, uint64_t bytes, int win_bits,
                         int with_header, uint8_t* out, uint64_t capacity) {
    bitstream bs = { .data = out, .capacity = capacity };
    if (with_header) {
        squeeze.write_header(&bs, bytes, (uint8_t)win_bits);
        if (bs.error != 0) { return -(int64_t)bs.error; }
    }
    squeeze_type* s = squeeze.alloc(0);
    if (s == NULL) { return -(int64_t)ENOMEM; }
    squeeze.compress(s, &bs, data, bytes, (uint16_t)(1u << win_bits));
    int64_t r = s->error != 0 ? -(int64_t)s->error : (int64_t)bs.bytes;
    squeeze.free(s);
    return r;
}

/* the reference's own counters after a compress (huffman.h:29-33 per tree; entropy
 * huffman.h:237-249; the tree depth marks): what SQUEEZE_MAP_STATS would print from
 * (squeeze.h:397-403).  out[0..5] = lit updates/swaps/moves, pos updates/swaps/moves;
 * ent[0..1] = entropy lit/pos; depth[0..1].  returns bytes written or -errno */
int64_t sqz_ref_compress_stats(const uint8_t* data, uint64_t bytes, int win_bits,
                               uint8_t* out, uint64_t capacity, uint64_t counters[6],
                               double ent[2], int32_t depth[2]) {
    bitstream bs = { .data = out, .capacity = capacity };
    squeeze_type* s = squeeze.alloc(0);
    if (s == NULL) { return -(int64_t)ENOMEM; }
    squeeze.compress(s, &bs, data, bytes, (uint16_t)(1u << win_bits));
    counters[0] = s->lit.stats.updates; counters[1] = s->lit.stats.swaps; counters[2] = s->lit.stats.moves;
    counters[3] = s->pos.stats.updates; counters[4] = s->pos.stats.swaps; counters[5] = s->pos.stats.moves;
    ent[0] = huffman_entropy(&s->lit); ent[1] = huffman_entropy(&s->pos);
    depth[0] = s->lit.depth; depth[1] = s->pos.depth;
    int64_t r = s->error != 0 ? -(int64_t)s->error : (int64_t)bs.bytes;
    squeeze.free(s);
    return r;
}

/* `with_header`: parse the 72-bit header first (then *bytes_io is an output),
 * otherwise decode exactly *bytes_io bytes. returns 0 or errno */
int sqz_ref_decompress(const uint8_t* in, uint64_t in_bytes, int with_header,
                       uint8_t* data, uint64_t capacity, uint64_t* bytes_io,
                       int* win_bits_out) {
    bitstream bs = { .data = (uint8_t*)in, .bytes = in_bytes };
    uint64_t bytes = *bytes_io;
    uint8_t win_bits = 0;
    if (with_header) {
        squeeze.read_header(&bs, &bytes, &win_bits);
        if (bs.error != 0) { return bs.error; }
        if (win_bits_out) { *win_bits_out = win_bits; }
    }
    if (bytes > capacity) { return E2BIG; }
    squeeze_type* s = squeeze.alloc(0);
    if (s == NULL) { return ENOMEM; }
    squeeze.decompress(s, &bs, data, bytes);
    int r = s->error;
    squeeze.free(s);
    *bytes_io = bytes;
    return r;
}

/* The reference's FILE mode (attic test.c:39-42, 98-101): the bit stream hands every
 * full 64-bit word to a callback that fwrite()s / fread()s `b64` in HOST byte order
 * (bitstream.h: `.stream`, `.output`, `.input`).  These two drive exactly that path of
 * the reference, so the file image -- which differs from the memory-mode bytes on a
 * little-endian host -- is pinned by the reference itself, not by a reading of it. */
static int ref_write_file(bitstream* bs) {
    return fwrite(&bs->b64, 8, 1, (FILE*)bs->stream) == 1 ? 0 : (errno != 0 ? errno : EIO);
}

static int ref_read_file(bitstream* bs) {
    return fread(&bs->b64, 8, 1, (FILE*)bs->stream) == 1 ? 0 : (errno != 0 ? errno : EIO);
}

/* header + payload into `path`; returns bytes written (>=0) or -errno */
int64_t sqz_ref_compress_file(const uint8_t* data, uint64_t bytes, int win_bits, const char* path) {
    FILE* f = fopen(path, "wb");
    if (f == NULL) { return -(int64_t)errno; }
    bitstream bs = { .stream = f, .output = ref_write_file };
    squeeze.write_header(&bs, bytes, (uint8_t)win_bits);
    int64_t r = bs.error != 0 ? -(int64_t)bs.error : 0;
    if (r == 0) {
        squeeze_type* s = squeeze.alloc(0);
        if (s == NULL) { r = -(int64_t)ENOMEM; }
        else {
            squeeze.compress(s, &bs, data, bytes, (uint16_t)(1u << win_bits));
            r = s->error != 0 ? -(int64_t)s->error : (int64_t)bs.bytes;
            squeeze.free(s);
        }
    }
    if (fclose(f) != 0 && r >= 0) { r = -(int64_t)errno; }
    return r;
}

/* reads `path` through the reference's `.input` callback; returns 0 or errno */
int sqz_ref_decompress_file(const char* path, uint8_t* data, uint64_t capacity,
                            uint64_t* bytes_out, int* win_bits_out) {
    FILE* f = fopen(path, "rb");
    if (f == NULL) { return errno; }
    bitstream bs = { .stream = f, .input = ref_read_file };
    uint64_t bytes = 0;
    uint8_t win_bits = 0;
    squeeze.read_header(&bs, &bytes, &win_bits);
    int r = bs.error;
    if (r == 0 && bytes > capacity) { r = E2BIG; }
    if (r == 0) {
        squeeze_type* s = squeeze.alloc(0);
        if (s == NULL) { r = ENOMEM; }
        else {
            squeeze.decompress(s, &bs, data, bytes);
            r = s->error;
            squeeze.free(s);
        }
    }
    fclose(f);
    if (bytes_out) { *bytes_out = bytes; }
    if (win_bits_out) { *win_bits_out = win_bits; }
    return r;
}

/* Drives the reference's huffman.h directly (huffman_init :251,
 * huffman_inc_frequency :218 -> huffman_insert :149 for unseen symbols) and
 * dumps every node, so the restatement's tree is pinned independently of
 * the LZ77 stage.  info = {n, next, depth, complete}. */
int sqz_ref_tree_run(int32_t n, const int32_t* symbols, uint64_t count,
                     uint64_t* freq, uint64_t* path, int32_t* bits,
                     int32_t* pix, int32_t* lix, int32_t* rix,
                     int32_t info[4]) {
    const int32_t m = 2 * n - 1;
    huffman_node* nodes = (huffman_node*)calloc((size_t)m, sizeof(huffman_node));
    if (nodes == NULL) { return ENOMEM; }
    huffman_tree t;
    huffman_init(&t, nodes, (size_t)m);
    for (uint64_t k = 0; k < count; k++) { huffman_inc_frequency(&t, symbols[k]); }
    for (int32_t i = 0; i < m; i++) {
        freq[i] = nodes[i].freq; path[i] = nodes[i].path; bits[i] = nodes[i].bits;
        pix[i] = nodes[i].pix; lix[i] = nodes[i].lix; rix[i] = nodes[i].rix;
    }
    info[0] = t.n; info[1] = t.next; info[2] = t.depth; info[3] = t.complete;
    free(nodes);
    return 0;
}

#define squeeze_implementation
#include "squeeze.h"
