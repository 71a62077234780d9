/* oracle/asan_driver.c -- TEST INFRASTRUCTURE ONLY.
 *
 * The CPU restatement (sqz_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer
 * (`make -C oracle asan-driver`; SURVEY.md section 5 asked for the restatement to run under the
 * sanitizers in the CPU suite).  A plain executable, so that no sanitizer runtime has to be
 * preloaded into the test runner:
 *
 *   asan_driver FILE WIN_BITS EXPECT_BYTES EXPECT_FNV    encode FILE with the H0 header, check size and
 *                                                        FNV-1a-64 of the stream, decode it back, then
 *                                                        decode 200 corrupted copies (any errno is fine,
 *                                                        a sanitizer report is not), the token-driven
 *                                                        encoder and a tree run over the same bytes
 * exit code 0 = everything matched and the sanitizers stayed silent.
 */
#include "sqz_oracle.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}

int main(int argc, char** argv) {
    if (argc != 5) { fprintf(stderr, "usage: %s FILE WIN_BITS EXPECT_BYTES EXPECT_FNV\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (f == NULL) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* data = (uint8_t*)malloc((size_t)n + 1);
    if (fread(data, 1, (size_t)n, f) != (size_t)n) { return 2; }
    fclose(f);
    const int wb = atoi(argv[2]);
    const uint64_t want_bytes = strtoull(argv[3], NULL, 10), want_fnv = strtoull(argv[4], NULL, 16);

    const uint64_t cap = 2 * (uint64_t)n + 1088;
    uint8_t* out = (uint8_t*)malloc(cap);
    uint64_t nb = 0;
    int e = sqzo_encode(data, (uint64_t)n, 1u << wb, wb, out, cap, &nb);
    if (e != 0 || nb != want_bytes || sqzo_fnv1a64(out, nb) != want_fnv) {
        fprintf(stderr, "encode: errno %d, %llu bytes, fnv %016llx\n", e, (unsigned long long)nb,
                (unsigned long long)sqzo_fnv1a64(out, nb));
        return 1;
    }
    uint8_t* back = (uint8_t*)malloc((size_t)n + 1);
    uint64_t got = 0;
    int win = 0;
    e = sqzo_decode(out, nb, 1, back, (uint64_t)n, &got, &win);
    if (e != 0 || got != (uint64_t)n || win != wb || memcmp(back, data, (size_t)n) != 0) { fprintf(stderr, "decode: errno %d\n", e); return 1; }

    /* exact-size buffers on the heap, so that an overrun by one byte is a report */
    for (int round = 0; round < 200; round++) {
        const uint64_t keep = 9 + rnd() % (nb - 8);
        uint8_t* bad = (uint8_t*)malloc(keep);
        memcpy(bad, out, keep);
        for (int k = 0; k < 1 + round % 5; k++) { bad[9 + rnd() % (keep - 9)] ^= (uint8_t)(1u << (rnd() & 7)); }
        uint8_t* dst = (uint8_t*)malloc((size_t)n);
        uint64_t m = 0;
        (void)sqzo_decode(bad, keep, 1, dst, (uint64_t)n, &m, &win);
        free(dst);
        free(bad);
    }

    /* stage 1 + stage 2 apart, and a short output buffer (E2BIG must stop at the capacity) */
    uint32_t* toks = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    uint64_t nt = 0;
    e = sqzo_tokens(data, (uint64_t)n, 1u << wb, toks, (uint64_t)n, &nt);
    sqzo_stats st;
    uint64_t nb2 = 0;
    uint8_t* out2 = (uint8_t*)malloc(cap);
    if (e != 0 || sqzo_encode_tokens(toks, nt, out2, cap, &nb2, &st) != 0 || st.literal_bytes + st.backref_bytes != (uint64_t)n) { return 1; }
    for (uint64_t small = 0; small < 64 && small < nb2; small += 7) {
        uint8_t* tight = (uint8_t*)malloc(small + 1);
        uint64_t m = 0;
        if (sqzo_encode_tokens(toks, nt, tight, small, &m, NULL) != E2BIG || m > small) { return 1; }
        free(tight);
    }

    /* the tree alone, driven by the bytes as symbols */
    int32_t* syms = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (long k = 0; k < n; k++) { syms[k] = data[k] % 31; }
    enum { M = 63 };
    uint64_t freq[M], path[M];
    int32_t bits[M], pix[M], lix[M], rix[M];
    sqzo_tree_dump info;
    if (sqzo_tree_run(32, syms, (uint64_t)n, freq, path, bits, pix, lix, rix, &info) != 0) { return 1; }

    free(syms); free(out2); free(toks); free(back); free(out); free(data);
    printf("ok %s w%d %llu -> %llu\n", argv[1], wb, (unsigned long long)n, (unsigned long long)nb);
    return 0;
}
