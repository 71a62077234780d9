/* oracle/sqz_oracle.c -- TEST INFRASTRUCTURE ONLY (see sqz_oracle.h).
 *
 * Restatement of the reference H0 codec written from its behaviour; the
 * reference files are cited as  file:line  relative to
 * /root/reference/attic/map_experiment.  Structure differs on purpose
 * (struct-of-arrays tree, explicit work stacks instead of recursion, word-wise
 * bit packer) -- the GPU kernels in sqz_amd/csrc mirror THIS formulation, so a
 * mismatch between the two localises a bug quickly.
 */
#include "sqz_oracle.h"

#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>

/* huffman.h:228: the tree stops taking updates once its depth mark reaches 63.  No real stream gets
 * there (a chain of depth d needs ~2^d symbols), so the tests that want to SEE the freeze build a
 * variant of this file and of the kernels with a lower threshold (tests/test_emu.py); the pinned
 * oracle always uses 63. */
#ifndef SQZO_FREEZE_DEPTH
#define SQZO_FREEZE_DEPTH 63
#endif

/* ------------------------------------------------------------------ */
/* alphabet constants: squeeze.h:9-25                                  */
enum {
    LEN_MIN      = 3,     /* squeeze_deflate_len_min */
    LEN_MAX      = 257,   /* squeeze_deflate_len_max */
    SYM_LEN0     = 257,   /* squeeze_deflate_sym_min */
    LIT_NYT      = 285,   /* squeeze_lit_nyt */
    POS_NYT      = 30,    /* squeeze_pos_nyt */
    LIT_LEAVES   = 512,   /* squeeze.h:204 */
    POS_LEAVES   = 32,    /* squeeze.h:205 */
    MAX_LEAVES   = 512,
    MAX_NODES    = 2 * MAX_LEAVES - 1
};

/* DEFLATE tables: squeeze.h:29-79 (values are RFC 1951's) */
static const uint16_t k_len_base[29] = {
    3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
    35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
static const uint8_t k_len_xb[29] = {
    0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
    3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
static const uint16_t k_pos_base[30] = {
    1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385,
    513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
static const uint8_t k_pos_xb[30] = {
    0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7,
    8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

/* squeeze.h:151-172 builds len_index[]/pos_index[] lookup tables; the same
 * mapping by search (index = last base <= value).  len 227..257 -> 27. */
static int len_code(uint32_t len) {
    int c = 0;
    while (c + 1 < 29 && k_len_base[c + 1] <= len) { c++; }
    return c;
}

static int pos_code(uint32_t pos) { /* pos_index[0] == 0 in the reference */
    int c = 0;
    while (c + 1 < 30 && k_pos_base[c + 1] <= pos) { c++; }
    return c;
}

/* ------------------------------------------------------------------ */
/* bit sink: bitstream.h:28-63,112-114 (memory mode)                    */
typedef struct {
    uint8_t* data;
    uint64_t capacity;
    uint64_t bytes;
    uint64_t acc;   /* bits accumulate at the low end, first bit ends up MSB */
    int      fill;  /* 0..63 */
    int      error; /* sticky */
} bit_sink;

static void sink_word(bit_sink* w) { /* bitstream.h:33-51 */
    for (int k = 0; k < 8 && w->error == 0; k++) {
        if (w->bytes == w->capacity) {
            w->error = E2BIG;
        } else {
            w->data[w->bytes++] = (uint8_t)(w->acc >> (56 - 8 * k));
        }
    }
    w->acc = 0;
    w->fill = 0;
}

/* value goes out LSB first (bitstream.h:55-63) */
static void sink_put(bit_sink* w, uint64_t value, int nbits) {
    for (int b = 0; b < nbits && w->error == 0; b++) {
        w->acc = (w->acc << 1) | ((value >> b) & 1u);
        if (++w->fill == 64) { sink_word(w); }
    }
}

static void sink_flush(bit_sink* w) { /* bitstream.h:112-114 */
    if (w->fill > 0 && w->error == 0) {
        w->acc <<= (64 - w->fill);
        sink_word(w);
    }
}

/* bit source: bitstream.h:65-103 (memory mode) */
typedef struct {
    const uint8_t* data;
    uint64_t bytes;
    uint64_t read;
    uint64_t acc;
    int      left;
    int      error;
} bit_source;

static int source_bit(bit_source* r) {
    if (r->error != 0) { return 0; }
    if (r->left == 0) {
        r->acc = 0;
        for (int k = 0; k < 8 && r->error == 0; k++) { /* bitstream.h:72-80 */
            if (r->read == r->bytes) {
                r->error = E2BIG;
            } else {
                r->acc |= (uint64_t)r->data[r->read++] << (56 - 8 * k);
            }
        }
        r->left = 64;
    }
    int bit = (int)(r->acc >> 63);
    r->acc <<= 1;
    r->left--;
    return bit;
}

static uint64_t source_get(bit_source* r, int nbits) { /* bitstream.h:95-103 */
    uint64_t v = 0;
    for (int b = 0; b < nbits && r->error == 0; b++) {
        v |= (uint64_t)source_bit(r) << b;
    }
    return v;
}

/* ------------------------------------------------------------------ */
/* adaptive Huffman tree: huffman.h:13-34 as struct-of-arrays           */
typedef struct {
    uint64_t freq[MAX_NODES];
    uint64_t path[MAX_NODES];
    int32_t  bits[MAX_NODES];
    int32_t  up[MAX_NODES];    /* pix */
    int32_t  lo[MAX_NODES];    /* lix */
    int32_t  hi[MAX_NODES];    /* rix */
    int32_t  n, next, depth, complete;
    /* huffman.h:29-33 `stats`: update_paths calls (one per node visited, :42), sibling
     * exchanges carried out (:76), promotions carried out (:111) */
    uint64_t st_updates, st_swaps, st_moves;
    /* explicit stacks standing in for the reference's recursion */
    int32_t  walk[MAX_NODES];
    int32_t  pend_parent[2 * MAX_NODES];
    int32_t  pend_child[2 * MAX_NODES];
} tree;

static void tree_init(tree* t, int32_t n) { /* huffman.h:251-269 */
    const int32_t m = 2 * n - 1;
    t->n = n;
    t->next = m - 1;
    t->depth = 0;
    t->complete = 0;
    t->st_updates = t->st_swaps = t->st_moves = 0;
    for (int32_t i = 0; i < m; i++) {
        t->freq[i] = 0; t->path[i] = 0; t->bits[i] = 0;
        t->up[i] = -1; t->lo[i] = -1; t->hi[i] = -1;
    }
}

/* huffman.h:41-62: relabel the subtree under `top`; depth is a high-water
 * mark, reset only when the relabel starts at the root. */
static void tree_relabel(tree* t, int32_t top) {
    if (top == 2 * t->n - 2) { t->depth = 0; }
    int sp = 0;
    t->walk[sp++] = top;
    while (sp > 0) {
        const int32_t v = t->walk[--sp];
        const int32_t b = t->bits[v];
        const uint64_t p = t->path[v];
        t->st_updates++;                            /* huffman.h:42 */
        if (b > t->depth) { t->depth = b; }
        const int32_t l = t->lo[v], r = t->hi[v];
        if (r != -1) {
            t->bits[r] = b + 1;
            t->path[r] = p | (1ULL << b);
            t->walk[sp++] = r;
        }
        if (l != -1) {
            t->bits[l] = b + 1;
            t->path[l] = p;
            t->walk[sp++] = l;
        }
    }
}

/* huffman.h:64-86: keep freq[left] <= freq[right]; after an exchange the
 * caller continues with the OTHER sibling's index (:81). */
static int32_t tree_order_pair(tree* t, int32_t i) {
    const int32_t p = t->up[i];
    if (p < 0) { return i; } /* root: :68 */
    const int32_t l = t->lo[p], r = t->hi[p];
    if (l >= 0 && r >= 0 && t->freq[l] > t->freq[r]) {
        t->st_swaps++;                              /* huffman.h:76 */
        t->lo[p] = r;
        t->hi[p] = l;
        tree_relabel(t, p);
        return i == l ? r : l;
    }
    return i;
}

static void tree_sum(tree* t, int32_t i) { /* huffman.h:90-96 */
    const int32_t l = t->lo[i], r = t->hi[i];
    t->freq[i] = (l >= 0 ? t->freq[l] : 0) + (r >= 0 ? t->freq[r] : 0);
}

/* huffman.h:130-147 + :98-128.  The reference recurses
 *   frequency_changed(i) -> frequency_changed(parent) ... then, on the way
 *   back, move_up(i) -> frequency_changed(grandparent) -> ...
 * Both inner calls are tail positions, so a LIFO of (parent, child) pairs
 * reproduces the evaluation order: climbing pushes bottom-up, popping
 * handles the highest level first, and a nested climb started by a
 * promotion is pushed above whatever is still pending. */
static void tree_changed(tree* t, int32_t start) {
    int sp = 0;
    int32_t i = start;
    for (;;) { /* climb: :132-142 */
        const int32_t p = t->up[i];
        if (p < 0) { tree_sum(t, i); break; }
        tree_sum(t, p);
        i = tree_order_pair(t, i);
        t->pend_parent[sp] = p; t->pend_child[sp] = i; sp++;
        i = p;
    }
    while (sp > 0) {
        sp--;
        const int32_t p = t->pend_parent[sp];
        const int32_t c = t->pend_child[sp];
        if (t->up[p] < 0 || c != t->hi[p]) { continue; } /* :143 */
        /* move_up(c): :98-128 */
        const int32_t par = t->up[c];
        const int32_t g = t->up[par];
        const int par_is_left = (par == t->lo[g]);
        const int32_t uncle = par_is_left ? t->hi[g] : t->lo[g];
        if (!(t->freq[c] > t->freq[uncle])) { continue; }
        t->st_moves++;                              /* huffman.h:111 */
        t->up[c] = g;
        if (par_is_left) { t->hi[g] = c; } else { t->lo[g] = c; }
        t->hi[par] = uncle;
        t->up[uncle] = par;
        tree_sum(t, par);
        tree_sum(t, g);
        (void)tree_order_pair(t, c);
        (void)tree_order_pair(t, uncle);
        (void)tree_order_pair(t, par);
        tree_relabel(t, g);
        i = g; /* frequency_changed(g): :126 */
        for (;;) {
            const int32_t q = t->up[i];
            if (q < 0) { tree_sum(t, i); break; }
            tree_sum(t, q);
            i = tree_order_pair(t, i);
            t->pend_parent[sp] = q; t->pend_child[sp] = i; sp++;
            i = q;
        }
    }
}

/* huffman.h:149-216; returns 0 when the internal-node pool is exhausted */
static int tree_insert(tree* t, int32_t i) {
    int ok = 1;
    int32_t at = 2 * t->n - 2; /* root */
    t->freq[i] = 1;
    while (at >= t->n) { /* :156-170 */
        if (t->hi[at] == -1) { t->hi[at] = i; t->up[i] = at; break; }
        if (t->lo[at] == -1) { t->lo[at] = i; t->up[i] = at; break; }
        at = t->lo[at];
    }
    if (at >= t->n) { /* hung under an internal node: :171-173 */
        t->freq[at]++;
        i = tree_order_pair(t, i);
    } else if (t->next == t->n) { /* :180-182 */
        ok = 0;
        t->complete = 1;
    } else { /* split leaf `at`: :184-209 */
        const int32_t fresh = --t->next;
        const int32_t above = t->up[at];
        t->freq[fresh] = t->freq[at];
        t->lo[fresh] = at;
        t->hi[fresh] = i;
        t->up[fresh] = above;
        t->bits[fresh] = t->bits[at];
        t->path[fresh] = t->path[at];
        if (above != -1) {
            if (t->lo[above] == at) { t->lo[above] = fresh; }
            else                    { t->hi[above] = fresh; }
        }
        t->up[at] = fresh;
        t->bits[at]++;
        t->up[i] = fresh;
        t->bits[i] = t->bits[fresh] + 1;
        t->path[i] = t->path[fresh] | (1ULL << t->bits[fresh]);
        tree_sum(t, fresh);
        at = fresh;
    }
    tree_changed(t, i); /* :212 */
    tree_relabel(t, at); /* :213 */
    return ok;
}

#ifdef SQZO_CHECK_INVARIANTS
/* Opt-in self-check (make -C oracle check-invariants): two properties the GPU kernels'
 * batched update relies on, asserted after EVERY update on whatever input is run:
 *   1. at rest every sibling pair is ordered, freq[lo] <= freq[hi] (the climb of
 *      huffman_frequency_changed re-orders each level it passes, huffman.h:132-142);
 *   2. a node that hangs below the root has a sibling unless the tree holds one leaf. */
#include <stdlib.h>
static void tree_check(const tree* t) {
    const int32_t m = 2 * t->n - 1, root = m - 1;
    int32_t leaves = 0;
    for (int32_t v = 0; v < t->n; v++) { if (t->up[v] >= 0) { leaves++; } }
    for (int32_t v = t->n; v < m; v++) {
        if (v != root && t->up[v] < 0) { continue; }            /* unused internal id */
        const int32_t l = t->lo[v], h = t->hi[v];
        if (l >= 0 && h >= 0 && t->freq[l] > t->freq[h]) {
            fprintf(stderr, "invariant 1 broken at node %d: lo %llu > hi %llu\n", v,
                    (unsigned long long)t->freq[l], (unsigned long long)t->freq[h]);
            abort();
        }
        if (leaves >= 2 && (l < 0 || h < 0)) {
            fprintf(stderr, "invariant 2 broken at node %d: a child without a sibling (%d leaves)\n", v, leaves);
            abort();
        }
    }
}
#define TREE_CHECK(t) tree_check(t)
#else
#define TREE_CHECK(t) ((void)0)
#endif

static void tree_bump(tree* t, int32_t i) { /* huffman.h:218-235 */
    if (t->up[i] == -1) {
        (void)tree_insert(t, i);
    } else if (!t->complete && t->depth < SQZO_FREEZE_DEPTH &&
               t->freq[i] < UINT64_MAX - 1) {
        t->freq[i]++;
        tree_changed(t, i);
    } else {
        t->complete = 1;
    }
    TREE_CHECK(t);
}

/* ------------------------------------------------------------------ */
/* codec state: squeeze.h:81-92                                         */
typedef struct {
    tree lit;
    tree pos;
    int  error;
} codec;

static codec* codec_new(void) { /* squeeze.h:174-222, map_bits == 0 */
    codec* c = (codec*)malloc(sizeof(codec));
    if (c != NULL) {
        tree_init(&c->lit, LIT_LEAVES);
        tree_init(&c->pos, POS_LEAVES);
        c->error = 0;
    }
    return c;
}

/* squeeze.h:239-246: code first, THEN the frequency update */
static void put_symbol(codec* c, bit_sink* w, tree* t, int32_t s) {
    if (c->error == 0) {
        sink_put(w, t->path[s], t->bits[s]);
        c->error = w->error;
    }
    tree_bump(t, s);
}

static void put_raw(codec* c, bit_sink* w, uint64_t v, int nbits) {
    if (c->error == 0) { /* squeeze.h:231-237 */
        sink_put(w, v, nbits);
        c->error = w->error;
    }
}

static void put_lit(codec* c, bit_sink* w, int32_t s) { /* squeeze.h:278-288 */
    if (c->lit.bits[s] == 0) {
        put_symbol(c, w, &c->lit, LIT_NYT);
        put_raw(c, w, (uint64_t)s, 9);
        if (!tree_insert(&c->lit, s)) { c->error = E2BIG; }
    } else {
        put_symbol(c, w, &c->lit, s);
    }
}

static void put_len(codec* c, bit_sink* w, uint32_t len) { /* squeeze.h:290-298 */
    const int k = len_code(len);
    put_lit(c, w, SYM_LEN0 + k);
    if (k_len_xb[k] > 0) { put_raw(c, w, len - k_len_base[k], k_len_xb[k]); }
}

static void put_pos(codec* c, bit_sink* w, uint32_t pos) { /* squeeze.h:300-315 */
    const int k = pos_code(pos);
    if (c->pos.bits[k] == 0) {
        put_symbol(c, w, &c->pos, POS_NYT);
        put_raw(c, w, (uint64_t)k, 5);
        if (!tree_insert(&c->pos, k)) { c->error = E2BIG; }
    } else {
        put_symbol(c, w, &c->pos, k);
    }
    if (k_pos_xb[k] > 0) { put_raw(c, w, pos - k_pos_base[k], k_pos_xb[k]); }
}

/* ------------------------------------------------------------------ */
/* LZ77 brute-force finder: squeeze.h:338-358                           */
void sqzo_match_at(const uint8_t* data, uint64_t bytes, uint64_t i,
                   uint32_t window, uint32_t* len, uint32_t* dist) {
    uint32_t best = 0, where = 0;
    if (i >= 1) {
        const uint64_t reach = (i >= window) ? (uint64_t)window - 1 : i;
        const uint64_t room = bytes - i;
        const uint64_t cap = room < LEN_MAX ? room : LEN_MAX;
        for (uint64_t d = 1; d <= reach; d++) {
            const uint8_t* a = data + i - d;
            const uint8_t* b = data + i;
            uint64_t k = 0;
            while (k < cap && a[k] == b[k]) { k++; }
            if (k >= LEN_MIN && k > best) {
                best = (uint32_t)k;
                where = (uint32_t)d;
                if (best == LEN_MAX) { break; }
            }
        }
    }
    *len = best;
    *dist = where;
}

static int window_ok(uint32_t window) { return window >= 2 && window <= 32768; }

int sqzo_tokens(const uint8_t* data, uint64_t bytes, uint32_t window,
                uint32_t* tokens, uint64_t capacity, uint64_t* count) {
    if (!window_ok(window)) { return EINVAL; }
    uint64_t i = 0, n = 0;
    while (i < bytes) {
        uint32_t len, dist;
        sqzo_match_at(data, bytes, i, window, &len, &dist);
        if (n == capacity) { *count = n; return E2BIG; }
        if (len >= LEN_MIN) {
            tokens[n++] = SQZO_TOKEN_MATCH | (len << 16) | dist;
            i += len;
        } else {
            tokens[n++] = data[i];
            i++;
        }
    }
    *count = n;
    return 0;
}

/* squeeze.h:255-265 */
static void put_header(bit_sink* w, uint64_t bytes, int win_bits) {
    if (win_bits < 10 || win_bits > 15) {
        w->error = EINVAL;
    } else {
        sink_put(w, bytes, 64);
        sink_put(w, (uint64_t)win_bits, 8);
    }
}

/* huffman.h:237-249 */
static double tree_entropy(const tree* t) {
    double total = 0.0, e = 0.0;
    for (int32_t i = 0; i < t->n; i++) { total += (double)t->freq[i]; }
    for (int32_t i = 0; i < t->n; i++) {
        if (t->freq[i] > 0) {
            const double p = (double)t->freq[i] / total;
            e += p * log2(p);
        }
    }
    return -e;
}

static void stats_of(const codec* c, uint64_t li, uint64_t br, sqzo_stats* st) {
    if (st == NULL) { return; }
    st->lit_updates = c->lit.st_updates; st->lit_swaps = c->lit.st_swaps; st->lit_moves = c->lit.st_moves;
    st->pos_updates = c->pos.st_updates; st->pos_swaps = c->pos.st_swaps; st->pos_moves = c->pos.st_moves;
    st->literal_bytes = li; st->backref_bytes = br;
    st->lit_entropy = tree_entropy(&c->lit); st->pos_entropy = tree_entropy(&c->pos);
    st->lit_depth = c->lit.depth; st->pos_depth = c->pos.depth;
}

int sqzo_encode_stats(const uint8_t* data, uint64_t bytes, uint32_t window,
                      int header_win_bits, uint8_t* out, uint64_t capacity,
                      uint64_t* out_bytes, sqzo_stats* st) {
    *out_bytes = 0;
    if (!window_ok(window)) { return EINVAL; }
    bit_sink w = { out, capacity, 0, 0, 0, 0 };
    if (header_win_bits != 0) {
        put_header(&w, bytes, header_win_bits);
        if (w.error != 0) { return w.error; }
    }
    codec* c = codec_new();
    if (c == NULL) { return ENOMEM; }
    /* squeeze.h:333-334 */
    if (!tree_insert(&c->lit, LIT_NYT)) { c->error = EINVAL; }
    if (!tree_insert(&c->pos, POS_NYT)) { c->error = EINVAL; }
    uint64_t i = 0, li = 0, br = 0;                /* squeeze.h:327-328 li_bytes / br_bytes */
    while (i < bytes && c->error == 0) { /* squeeze.h:337-395 */
        uint32_t len, dist;
        sqzo_match_at(data, bytes, i, window, &len, &dist);
        if (len >= LEN_MIN) {
            put_len(c, &w, len);
            put_pos(c, &w, dist);
            i += len;
            br += len;
        } else {
            put_lit(c, &w, data[i]);
            i++;
            li++;
        }
    }
    if (c->error == 0) { /* squeeze.h:248-253 */
        sink_flush(&w);
        c->error = w.error;
    }
    const int r = c->error;
    stats_of(c, li, br, st);
    free(c);
    *out_bytes = w.bytes;
    return r;
}

int sqzo_encode(const uint8_t* data, uint64_t bytes, uint32_t window,
                int header_win_bits, uint8_t* out, uint64_t capacity,
                uint64_t* out_bytes) {
    return sqzo_encode_stats(data, bytes, window, header_win_bits, out, capacity, out_bytes, NULL);
}

/* squeeze.h:377-394 + :278-315 driven by a caller-made token sequence (payload only): what
 * stage 2 of the HIP path (sqz_hip_huffman_blocks) is held against when a test needs a
 * symbol sequence no input text produces -- very deep trees, or a distance the decoder must
 * refuse.  Lengths 3..258 and distances 1..32768 are encodable by the code tables
 * (squeeze.h:29-79); nothing checks that the tokens describe a consistent text. */
int sqzo_encode_tokens(const uint32_t* tokens, uint64_t count, uint8_t* out,
                       uint64_t capacity, uint64_t* out_bytes, sqzo_stats* st) {
    *out_bytes = 0;
    bit_sink w = { out, capacity, 0, 0, 0, 0 };
    codec* c = codec_new();
    if (c == NULL) { return ENOMEM; }
    if (!tree_insert(&c->lit, LIT_NYT)) { c->error = EINVAL; }
    if (!tree_insert(&c->pos, POS_NYT)) { c->error = EINVAL; }
    uint64_t li = 0, br = 0;
    for (uint64_t k = 0; k < count && c->error == 0; k++) {
        const uint32_t t = tokens[k];
        if ((t & SQZO_TOKEN_MATCH) != 0) {
            const uint32_t len = (t >> 16) & 0x1FFu, dist = t & 0xFFFFu;
            if (len < LEN_MIN || len > 258 || dist < 1 || dist > 32768) { c->error = EINVAL; break; }
            put_len(c, &w, len);
            put_pos(c, &w, dist);
            br += len;
        } else {
            if (t > 0xFFu) { c->error = EINVAL; break; }
            put_lit(c, &w, (int32_t)t);
            li++;
        }
    }
    if (c->error == 0) {
        sink_flush(&w);
        c->error = w.error;
    }
    const int r = c->error;
    stats_of(c, li, br, st);
    free(c);
    *out_bytes = w.bytes;
    return r;
}

/* ------------------------------------------------------------------ */
/* decode: squeeze.h:429-442,458-551                                    */
static int32_t get_symbol(codec* c, bit_source* r, tree* t) {
    const int32_t root = 2 * t->n - 2;
    int32_t i = root;
    for (;;) {
        const int bit = source_bit(r);
        if (r->error != 0) { c->error = r->error; return -1; }
        i = bit ? t->hi[i] : t->lo[i];
        /* the reference indexes node[-1] here on a malformed stream
         * (squeeze.h:434-435, assert only); hardened: EINVAL */
        if (i < 0) { c->error = EINVAL; return -1; }
        if (t->lo[i] < 0 && t->hi[i] < 0) { break; }
    }
    tree_bump(t, i);
    return i;
}

int sqzo_decode(const uint8_t* in, uint64_t in_bytes, int with_header,
                uint8_t* data, uint64_t capacity, uint64_t* bytes_io,
                int* win_bits) {
    bit_source r = { in, in_bytes, 0, 0, 0, 0 };
    uint64_t bytes = *bytes_io;
    if (with_header) { /* squeeze.h:444-456 */
        const uint64_t b = source_get(&r, 64);
        const uint64_t w = source_get(&r, 8);
        if (r.error != 0) { return r.error; }
        if (w < 10 || w > 15) { return EINVAL; }
        bytes = b;
        if (win_bits != NULL) { *win_bits = (int)w; }
    }
    *bytes_io = bytes;
    if (bytes > capacity) { return E2BIG; }
    codec* c = codec_new();
    if (c == NULL) { return ENOMEM; }
    if (!tree_insert(&c->lit, LIT_NYT)) { c->error = EINVAL; }
    if (!tree_insert(&c->pos, POS_NYT)) { c->error = EINVAL; }
    uint64_t i = 0;
    while (i < bytes && c->error == 0) {
        int32_t s = get_symbol(c, &r, &c->lit);
        if (c->error != 0) { break; }
        if (s == LIT_NYT) { /* squeeze.h:512-520 */
            s = (int32_t)source_get(&r, 9);
            if (r.error != 0) { c->error = r.error; break; }
            /* reference inserts any 9-bit value and fails afterwards
             * (:521-548); a value that is already attached trips only an
             * assert there.  Hardened: both are EINVAL before the insert. */
            if (s == 256 || s >= LIT_NYT || c->lit.up[s] != -1) {
                c->error = EINVAL; break;
            }
            if (!tree_insert(&c->lit, s)) { c->error = E2BIG; break; }
        }
        if (s <= 0xFF) {
            data[i++] = (uint8_t)s;
            continue;
        }
        /* squeeze.h:458-474 */
        const int k = s - SYM_LEN0;
        if (k < 0 || k >= 29) { c->error = EINVAL; break; }
        uint32_t len = k_len_base[k];
        if (k_len_xb[k] != 0) {
            len += (uint32_t)source_get(&r, k_len_xb[k]);
            if (r.error != 0) { c->error = r.error; break; }
        }
        if (len < LEN_MIN || len > LEN_MAX) { c->error = EINVAL; break; }
        /* squeeze.h:476-500 */
        int32_t pk = get_symbol(c, &r, &c->pos);
        if (c->error != 0) { break; }
        if (pk == POS_NYT) {
            pk = (int32_t)source_get(&r, 5);
            if (r.error != 0) { c->error = r.error; break; }
            if (pk >= POS_NYT || c->pos.up[pk] != -1) { c->error = EINVAL; break; }
            if (!tree_insert(&c->pos, pk)) { c->error = E2BIG; break; }
        }
        if (pk >= 30) { c->error = EINVAL; break; }
        uint32_t dist = k_pos_base[pk];
        if (k_pos_xb[pk] != 0) {
            dist += (uint32_t)source_get(&r, k_pos_xb[pk]);
            if (r.error != 0) { c->error = r.error; break; }
        }
        /* reference checks only 0 < pos <= 0x7FFF (squeeze.h:534) and would
         * read before / write past the buffer; hardened: EINVAL */
        if (dist == 0 || dist > 0x7FFF || dist > i || len > bytes - i) {
            c->error = EINVAL; break;
        }
        for (uint32_t k2 = 0; k2 < len; k2++, i++) { /* squeeze.h:537-539 */
            data[i] = data[i - dist];
        }
    }
    const int rc = c->error;
    free(c);
    return rc;
}

/* ------------------------------------------------------------------ */
int sqzo_tree_run(int32_t n, const int32_t* symbols, uint64_t count,
                  uint64_t* freq, uint64_t* path, int32_t* bits,
                  int32_t* pix, int32_t* lix, int32_t* rix,
                  sqzo_tree_dump* info) {
    if (n < 8 || n > MAX_LEAVES || (n & (n - 1)) != 0) { return EINVAL; }
    tree* t = (tree*)malloc(sizeof(tree));
    if (t == NULL) { return ENOMEM; }
    tree_init(t, n);
    for (uint64_t k = 0; k < count; k++) {
        if (symbols[k] < 0 || symbols[k] >= n) { free(t); return EINVAL; }
        tree_bump(t, symbols[k]);
    }
    const int32_t m = 2 * n - 1;
    for (int32_t i = 0; i < m; i++) {
        freq[i] = t->freq[i]; path[i] = t->path[i]; bits[i] = t->bits[i];
        pix[i] = t->up[i]; lix[i] = t->lo[i]; rix[i] = t->hi[i];
    }
    info->n = t->n; info->next = t->next;
    info->depth = t->depth; info->complete = t->complete;
    free(t);
    return 0;
}

/* the same drive, returning only huffman.h:29-33's counters: updates, swaps, moves */
int sqzo_tree_counters(int32_t n, const int32_t* symbols, uint64_t count, uint64_t out[3]) {
    if (n < 8 || n > MAX_LEAVES || (n & (n - 1)) != 0) { return EINVAL; }
    tree* t = (tree*)malloc(sizeof(tree));
    if (t == NULL) { return ENOMEM; }
    tree_init(t, n);
    for (uint64_t k = 0; k < count; k++) {
        if (symbols[k] < 0 || symbols[k] >= n) { free(t); return EINVAL; }
        tree_bump(t, symbols[k]);
    }
    out[0] = t->st_updates; out[1] = t->st_swaps; out[2] = t->st_moves;
    free(t);
    return 0;
}

uint64_t sqzo_fnv1a64(const uint8_t* p, uint64_t n) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (uint64_t i = 0; i < n; i++) { h = (h ^ p[i]) * 0x100000001b3ULL; }
    return h;
}

/* ------------------------------------------------------------------ */
/* Zipf(s=1) generator of SURVEY.md section 8d.  The table below is the
 * authoritative CDF (computed once in IEEE double as the survey prescribes:
 * H = sum r^-1, acc += r^-1 / H, cdf = floor(acc * 2^32) clamped). */
#include "zipf_cdf.inc"

const uint32_t* sqzo_zipf_cdf(void) { return k_zipf_cdf; }

void sqzo_zipf_block(uint64_t block_index, uint8_t* out, uint64_t bytes) {
    uint64_t state = 0x5A17C0DEULL + block_index;
    for (uint64_t i = 0; i < bytes; i++) {
        state += 0x9E3779B97F4A7C15ULL;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        const uint32_t u = (uint32_t)(z >> 32);
        int lo = 0, hi = 255; /* smallest idx with u <= cdf[idx] */
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (u <= k_zipf_cdf[mid]) { hi = mid; } else { lo = mid + 1; }
        }
        out[i] = (uint8_t)lo;
    }
}
