/* tests/harness/sqz_boundary.c -- C99 checks of the boundary residue (include/sqz/sqz.h):
 *
 *   1. the H0 header's public names compile and have the reference's values
 *      (/root/reference/attic/map_experiment/squeeze.h:9-25), and `init_with` takes a caller
 *      block of squeeze_sizeof(map_bits) bytes (squeeze.h:94-107,191-199);
 *   2. a compressed stream FOLLOWED BY OTHER DATA in the same file, read through the `.input`
 *      callback (bitstream.h:81-85): the decode is right, bs.read is the reference's figure (the
 *      words the stream holds x 8), and the shim's documented over-read -- at most
 *      max(4, 2 x the stream's words) words pulled -- is observed; the caller re-positions the
 *      file from bs.read and finds its trailer;
 *   3. two threads compressing at once (each call stages through a lane of its own: no shared
 *      lock or stream across the device work) produce the single-threaded bytes.
 *
 * Build:  gcc -std=c99 -O2 -Iinclude tests/harness/sqz_boundary.c -Lsqz_amd/lib -lsqz_amd -lpthread
 * Run:    tests/harness/sqz_boundary      (needs an MI355X; ENODEV otherwise)
 */
#define _POSIX_C_SOURCE 200809L
#include <errno.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sqz/sqz.h>

typedef char check_sym_min[squeeze_deflate_sym_min == 257 ? 1 : -1];
typedef char check_sym_max[squeeze_deflate_sym_max == 284 ? 1 : -1];
typedef char check_pos_max[squeeze_deflate_pos_max == 29 ? 1 : -1];
typedef char check_len[(squeeze_deflate_len_min == 3 && squeeze_deflate_len_max == 257) ? 1 : -1];
typedef char check_win[(squeeze_min_win_bits == 10 && squeeze_max_win_bits == 15) ? 1 : -1];
typedef char check_map[(squeeze_min_map_bits == 16 && squeeze_max_map_bits == 28) ? 1 : -1];
typedef char check_nyt[(squeeze_lit_nyt == 285 && squeeze_pos_nyt == 30) ? 1 : -1];

static long pulled;

static int write_file(bitstream* bs) { return fwrite(&bs->b64, 8, 1, (FILE*)bs->stream) == 1 ? 0 : EIO; }
static int read_file(bitstream* bs) {
    pulled++;
    return fread(&bs->b64, 8, 1, (FILE*)bs->stream) == 1 ? 0 : EIO;
}

#define CHECK(cond) do { if (!(cond)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

static void fill(uint8_t* d, size_t n, uint32_t seed) {         /* compressible, not trivial */
    for (size_t i = 0; i < n; i++) {
        seed = seed * 1664525u + 1013904223u;
        d[i] = (uint8_t)("abcdefgh  the quick brown fox"[(seed >> 24) % 29]);
    }
}

static int trailer_case(const uint8_t* data, size_t bytes, int win_bits, const char* path) {
    static const char trailer[64] = "NEXT RECORD: sixty-four bytes that are not part of the stream..";
    FILE* f = fopen(path, "wb");
    CHECK(f != NULL);
    /* caller-owned state block, sized the reference's way (squeeze.h:176-183 does this in alloc) */
    void* mem = calloc(1, squeeze_sizeof(0));
    CHECK(mem != NULL);
    squeeze_type* s = (squeeze_type*)mem;
    CHECK(squeeze.init_with(s, mem, squeeze_sizeof(0), 0) == 0);
    CHECK(squeeze.init_with(s, mem, squeeze_sizeof(0) - 1, 0) == EINVAL);
    CHECK(squeeze.init_with(s, mem, squeeze_sizeof(0), squeeze_min_map_bits) == EINVAL);   /* no map here */
    CHECK(squeeze.init_with(s, mem, squeeze_sizeof(0), 0) == 0);
    bitstream w = { .stream = f, .output = write_file };
    squeeze.write_header(&w, bytes, (uint8_t)win_bits);
    squeeze.compress(s, &w, data, bytes, (uint16_t)(1u << win_bits));
    CHECK(s->error == 0 && w.error == 0);
    const uint64_t stream_bytes = w.bytes;
    CHECK(fwrite(trailer, 1, sizeof(trailer), f) == sizeof(trailer));
    CHECK(fclose(f) == 0);

    f = fopen(path, "rb");
    CHECK(f != NULL);
    pulled = 0;
    bitstream r = { .stream = f, .input = read_file };
    uint64_t n = 0; uint8_t wb = 0;
    squeeze.read_header(&r, &n, &wb);
    CHECK(r.error == 0 && n == bytes && wb == win_bits);
    uint8_t* back = (uint8_t*)calloc(1, bytes + 1);
    CHECK(back != NULL);
    CHECK(squeeze.init_with(s, mem, squeeze_sizeof(0), 0) == 0);
    squeeze.decompress(s, &r, back, bytes);
    CHECK(s->error == 0 && r.error == 0);
    CHECK(memcmp(back, data, bytes) == 0);
    /* the reference's reader has fetched exactly the stream's words (bitstream.h:81-85) */
    CHECK(r.read == stream_bytes);
    /* the shim pulled ahead: the documented bound (the header's two words come on top of the first four) */
    const long words = (long)(stream_bytes / 8);
    CHECK(pulled >= words);
    CHECK(pulled <= (2 * words > 6 ? 2 * words : 6));
    /* a caller with more data behind the stream re-positions its source from bs.read */
    char got[64];
    CHECK(fseek(f, (long)r.read, SEEK_SET) == 0);
    CHECK(fread(got, 1, sizeof(got), f) == sizeof(got));
    CHECK(memcmp(got, trailer, sizeof(trailer)) == 0);
    fclose(f);
    free(back);
    free(mem);
    printf("trailer case: %zu -> %llu bytes, %ld words pulled for %ld\n", bytes,
           (unsigned long long)stream_bytes, pulled, words);
    return 0;
}

struct job { const uint8_t* data; size_t bytes; uint8_t* out; uint64_t cap, produced; int error; };

static void* compress_job(void* arg) {
    struct job* j = (struct job*)arg;
    for (int rep = 0; rep < 4; rep++) {
        bitstream bs = { .data = j->out, .capacity = j->cap };
        struct sqz s;
        sqz_init(&s);
        sqz_write_header_h0(&bs, j->bytes, 12);
        sqz_compress(&s, &bs, j->data, j->bytes, 1u << 12);
        j->error = s.error;
        j->produced = bs.bytes;
        if (s.error != 0) { break; }
    }
    return NULL;
}

int main(void) {
    char name[128]; int cus = 0; uint64_t lds = 0;
    const int e = sqz_hip_device_info(name, sizeof(name), &cus, &lds);
    if (e != 0) { printf("no gfx950 device: %s\n", strerror(e)); return e; }

    enum { N = 40000 };
    static uint8_t text[N];
    fill(text, N, 1u);
    if (trailer_case(text, N, 12, "~boundary~.bin") != 0) { return 1; }
    /* a stream that compresses far better than 8:1: the old first pull (a quarter of the OUTPUT size)
       read many times what the stream holds */
    static uint8_t zeros[262144];
    if (trailer_case(zeros, sizeof(zeros), 15, "~boundary~.bin") != 0) { return 1; }
    if (trailer_case((const uint8_t*)"a", 1, 10, "~boundary~.bin") != 0) { return 1; }
    (void)remove("~boundary~.bin");

    /* two threads at once, different inputs; then each alone: same bytes */
    static uint8_t a[N], b[N];
    fill(a, N, 7u);
    fill(b, N, 99u);
    struct job ja = { a, N, malloc(sqz_bound(N) + 16), sqz_bound(N) + 16, 0, 0 };
    struct job jb = { b, N, malloc(sqz_bound(N) + 16), sqz_bound(N) + 16, 0, 0 };
    struct job sa = ja, sb = jb;
    sa.out = malloc(sa.cap); sb.out = malloc(sb.cap);
    CHECK(ja.out && jb.out && sa.out && sb.out);
    pthread_t ta, tb;
    CHECK(pthread_create(&ta, NULL, compress_job, &ja) == 0);
    CHECK(pthread_create(&tb, NULL, compress_job, &jb) == 0);
    pthread_join(ta, NULL);
    pthread_join(tb, NULL);
    compress_job(&sa);
    compress_job(&sb);
    CHECK(ja.error == 0 && jb.error == 0 && sa.error == 0 && sb.error == 0);
    CHECK(ja.produced == sa.produced && memcmp(ja.out, sa.out, sa.produced) == 0);
    CHECK(jb.produced == sb.produced && memcmp(jb.out, sb.out, sb.produced) == 0);
    printf("two threads: %llu and %llu bytes, equal to the single-threaded streams\n",
           (unsigned long long)ja.produced, (unsigned long long)jb.produced);
    printf("ok\n");
    return 0;
}
