/* oracle/sqz_rc_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the reference's HEAD ("R-era") codec, SURVEY.md section 8f-1: an adaptive
 * order-0 range coder.  File:line citations are relative to /root/reference/src/sqz.c unless
 * they name inc/sqz/sqz.h.
 *
 * What HEAD does: the LZ77 finders are compiled out (src/sqz.c:630-631,
 * :660, :591), best_size stays 0 (:657-658), so sqz_compress codes EVERY byte as
 *     rc_encode(pm_literal, 1); rc_encode(pm_byte, d[i])                       (:722-723)
 * and ends with rc_encode(pm_literal, 0); rc_encode(pm_size, 0xFF); rc_flush   (:741-743).
 * The decoder (:793-839) is restated as written, including its back-reference branch (which no
 * stream of HEAD's own encoder reaches, and whose distance format the reference's two sides
 * disagree on).
 *
 * The reference keeps cumulative frequencies in Fenwick trees (:398-445); a cumulative sum is a
 * cumulative sum, so this restatement keeps plain frequency arrays and adds them up -- the GPU
 * kernels do the same with a prefix sum over the lanes.  Same numbers, same bytes.
 *
 * Parity status: PINNED against the reference compiled where it lies (oracle/_ref/libsqz_ref_rc.so,
 * `make -C oracle ref-rc`: clang -std=c23 -include errno.h) and the
 * fingerprints of SURVEY.md section 8f-1, both through tests/golden/golden_rc.json.
 */
#include "sqz_oracle.h"

#include <errno.h>
#include <stdlib.h>
#include <string.h>

enum { RC_MIN_LEN = 2, RC_MAX_LEN = 254 };       /* src/sqz.c:29-30 */

typedef struct { uint64_t freq[256]; uint64_t total; } rc_model;   /* struct prob_model, inc/sqz/sqz.h:40-43 */

static void model_init(rc_model* m, uint32_t n) {                 /* pm_init :453-458 */
    for (uint32_t i = 0; i < 256; i++) { m->freq[i] = i < n ? 1 : 0; }
    m->total = n;
}

static uint64_t model_below(const rc_model* m, uint32_t sym) {    /* pm_sum_of :447-449 */
    uint64_t s = 0;
    for (uint32_t i = 0; i < sym; i++) { s += m->freq[i]; }
    return s;
}

static void model_update(rc_model* m, uint8_t sym) {              /* pm_update :466-472 */
    if (m->total < (1ull << 56)) { m->freq[sym]++; m->total++; }
}

typedef struct {
    uint64_t low, range, code;                                    /* struct range_coder, inc/sqz/sqz.h:45-53 */
    uint8_t* out; uint64_t cap, written;
    const uint8_t* in; uint64_t avail, consumed;
    int error;
    int dry_error;             /* what a read past the end sets (0: nothing), see rc_get */
} rc_state;

static void rc_put(rc_state* rc) {                                /* rc_emit :474-479 */
    if (rc->written < rc->cap) { rc->out[rc->written] = (uint8_t)(rc->low >> 56); } else { rc->error = ENOBUFS; }
    rc->written++;
    rc->low <<= 8;
    rc->range <<= 8;
}

static int rc_same_top(const rc_state* rc) {                      /* rc_leftmost_byte_is_same :481-483 */
    return (rc->low >> 56) == ((rc->low + rc->range) >> 56);
}

static void rc_encode(rc_state* rc, rc_model* m, uint8_t sym) {   /* rc_encode :506-521 */
    const uint64_t total = m->total, start = model_below(m, sym), size = m->freq[sym];
    rc->range /= total;
    rc->low += start * rc->range;
    rc->range *= size;
    model_update(m, sym);
    while (rc_same_top(rc)) { rc_put(rc); }
    if (rc->range < total + 1) {
        rc_put(rc);
        rc_put(rc);
        rc->range = UINT64_MAX - rc->low;
    }
}

/* sqz_compress as HEAD runs it (:590-596, :717-743): literal-only + end of stream + flush.
 * Returns 0 or ENOBUFS (the reference's write callback has no capacity; the harness buffers do);
 * *out_bytes = bytes the stream takes (also when it did not fit). */
int sqzo_rc_encode(const uint8_t* data, uint64_t bytes, uint8_t* out, uint64_t capacity, uint64_t* out_bytes) {
    rc_model lit, size, byte;
    model_init(&lit, 2); model_init(&size, 256); model_init(&byte, 256);       /* sqz_init :550-565 */
    rc_state rc = { 0, UINT64_MAX, 0, out, capacity, 0, NULL, 0, 0, 0, 0 };    /* rc_init :485-490 */
    for (uint64_t i = 0; i < bytes; i++) {
        rc_encode(&rc, &lit, 1);
        rc_encode(&rc, &byte, data[i]);
    }
    rc_encode(&rc, &lit, 0);                                                   /* :741-742 */
    rc_encode(&rc, &size, 0xFF);
    for (int k = 0; k < 8; k++) { rc.range = UINT64_MAX; rc_put(&rc); }        /* rc_flush :492-497 */
    *out_bytes = rc.written;
    return rc.error;
}

/* The harness's read callback returns 0 past the end AND carries its source's error into rc.error
 * (test.c:112-121: `if (rc->error == 0) { b = io_get(io); rc->error = io->error; }`); dry_error is that
 * source error (0 = a source that just ends, as an in-memory one does). */
static uint8_t rc_get(rc_state* rc) {
    uint8_t b = 0;
    if (rc->dry_error != 0 && rc->consumed >= rc->avail && rc->error == 0) { rc->error = rc->dry_error; }
    if (rc->consumed < rc->avail) { b = rc->in[rc->consumed]; }
    rc->consumed++;
    return b;
}

static void rc_consume(rc_state* rc) {                            /* rc_consume :499-504 */
    rc->code = (rc->code << 8) + rc_get(rc);
    rc->low <<= 8;
    rc->range <<= 8;
}

static uint8_t rc_decode(rc_state* rc, rc_model* m) {             /* rc_decode :528-548 */
    const uint64_t total = m->total;
    if (total < 1) { rc->error = EINVAL; return 0; }
    if (rc->range < total) {
        rc_consume(rc);
        rc_consume(rc);
        rc->range = UINT64_MAX - rc->low;
    }
    const uint64_t sum = (rc->code - rc->low) / (rc->range / total);
    /* pm_index_of (:451, ft_index_of :432-445): the symbol whose run of the cumulative counts holds `sum`.
     * A sum at or past the total (a damaged stream) makes ft_index_of return -1 and pm_index_of add 1 to
     * it: symbol 0, not an error. */
    int32_t sym = 0;
    if (sum < total) {
        sym = -1;
        uint64_t acc = 0;
        for (int32_t i = 0; i < 256; i++) {
            if (sum < acc + m->freq[i]) { sym = i; break; }
            acc += m->freq[i];
        }
    }
    if (sym < 0 || m->freq[sym] == 0) { rc->error = EILSEQ; return 0; }
    const uint64_t start = model_below(m, (uint32_t)sym), size = m->freq[sym];
    if (size == 0 || rc->range < total) { rc->error = EILSEQ; return 0; }
    rc->range /= total;
    rc->low += start * rc->range;
    rc->range *= size;
    model_update(m, (uint8_t)sym);
    while (rc_same_top(rc)) { rc_consume(rc); }
    return (uint8_t)sym;
}

/* sqz_decompress (:793-839).  Returns the reference's rc.error (0, EINVAL, EILSEQ, ERANGE, ENOBUFS);
 * *out_bytes = bytes produced, *consumed = stream bytes the decoder asked for. */
int sqzo_rc_decode(const uint8_t* in, uint64_t in_bytes, uint8_t* data, uint64_t capacity,
                   uint64_t* out_bytes, uint64_t* consumed) {
    return sqzo_rc_decode_dry(in, in_bytes, data, capacity, out_bytes, consumed, 0);
}

/* the same with a source that FAILS at its end: the first read past it sets rc.error = dry_error */
int sqzo_rc_decode_dry(const uint8_t* in, uint64_t in_bytes, uint8_t* data, uint64_t capacity,
                       uint64_t* out_bytes, uint64_t* consumed, int dry_error) {
    rc_model lit, size, byte, bits, dist[32];
    model_init(&lit, 2); model_init(&size, 256); model_init(&byte, 256); model_init(&bits, 32);
    for (int b = 0; b < 32; b++) { model_init(&dist[b], 2); }
    rc_state rc = { 0, UINT64_MAX, 0, NULL, 0, 0, in, in_bytes, 0, 0, dry_error };
    for (int k = 0; k < 8; k++) { rc.code = (rc.code << 8) + rc_get(&rc); }   /* :794-797 */
    uint64_t i = 0;
    while (rc.error == 0) {
        const uint8_t is_lit = rc_decode(&rc, &lit);
        if (rc.error != 0) { break; }
        if (is_lit) {
            if (i < capacity) { data[i++] = rc_decode(&rc, &byte); } else { rc.error = ENOBUFS; }
        } else {
            const uint8_t sz = rc_decode(&rc, &size);
            if (sz == 0xFF) { break; }                                       /* end of stream :808 */
            if (sz < RC_MIN_LEN || sz > RC_MAX_LEN) { rc.error = ERANGE; }
            else {
                const uint8_t nb = rc_decode(&rc, &bits);
                if (rc.error != 0) { break; }
                uint32_t d = 0;
                for (int b = 0; b < nb - 1 && rc.error == 0; b++) { d |= (uint32_t)rc_decode(&rc, &dist[b]) << b; }
                if (nb > 0) { d |= nb < 32 ? (1u << nb) : 0u; }               /* :821 (1u << 32 is the reference's UB) */
                if (rc.error == 0) {
                    const uint64_t n = i + sz;
                    if (i < d) { rc.error = ERANGE; }
                    else if (n <= capacity) { while (i < n) { data[i] = data[i - d]; i++; } }
                    else { rc.error = ENOBUFS; }
                }
            }
        }
    }
    *out_bytes = i;
    if (consumed != NULL) { *consumed = rc.consumed; }
    return rc.error;
}
