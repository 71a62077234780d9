/* tests/harness/sqz_harness.c -- C99 host over the C ABI of libsqz_amd.so.
 *
 * POSIX re-creation of the reference's H0 harness behaviour
 * (/root/reference/attic/map_experiment/test.c): every input is compressed through
 * the `squeeze` vtable (test.c:54-61), decompressed again (test.c:114-134) and compared
 * with memcmp (test.c:138); the cases and their order follow main() (test.c:195-236):
 * 4 KB of zeros, 4 KB of 01 02 03 04, "Hello World Hello.World Hello World", this
 * source file, the executable, then every corpus file that exists (missing files are
 * skipped like file_exist() does, test.c:231).  The reference harness itself needs the
 * MSVC CRT (fopen_s, errno_t) and does not build on Linux.
 *
 * Differences: the reference streams 8-byte words to a FILE through a callback
 * (test.c:39-42,98-101); the device cannot call back, so the stream goes to a host
 * buffer (memory mode, bitstream.h:34-43) -- see INTEGRATION.md for the replay loop.
 *
 * Build:  gcc -std=c99 -O2 -Iinclude tests/harness/sqz_harness.c -Lsqz_amd/lib -lsqz_amd
 *         -Wl,-rpath,$PWD/sqz_amd/lib -o tests/harness/sqz_harness
 * Run:    tests/harness/sqz_harness [win_bits] [corpus_dir]     (needs an MI355X)
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <sqz/sqz.h>

static int bits_win = 10;   /* test.c:31: default configuration */

static int read_fully(const char* fn, uint8_t** data, size_t* bytes) {
    FILE* f = fopen(fn, "rb");
    if (f == NULL) { return errno; }
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return errno; }
    const long n = ftell(f);
    if (n < 0 || fseek(f, 0, SEEK_SET) != 0) { fclose(f); return errno; }
    *data = (uint8_t*)malloc((size_t)n + 1);
    if (*data == NULL) { fclose(f); return ENOMEM; }
    *bytes = fread(*data, 1, (size_t)n, f);
    fclose(f);
    return *bytes == (size_t)n ? 0 : EIO;
}

/* test.c:44-96 compress + test.c:103-162 verify, in memory */
static int test(const char* name, const uint8_t* data, size_t bytes) {
    const uint64_t capacity = sqz_bound(bytes) + 16;
    uint8_t* comp = (uint8_t*)malloc(capacity);
    uint8_t* back = (uint8_t*)malloc(bytes + 1);
    if (comp == NULL || back == NULL) { free(comp); free(back); return ENOMEM; }
    int r = 0;
    bitstream bs = { .data = comp, .capacity = capacity };
    squeeze.write_header(&bs, bytes, (uint8_t)bits_win);               /* test.c:54 */
    if (bs.error != 0) { r = bs.error; }
    squeeze_type* s = NULL;
    if (r == 0) {
        s = squeeze.alloc(0);                                          /* test.c:59 */
        if (s == NULL) { r = ENOMEM; }
    }
    if (r == 0) {
        squeeze.compress(s, &bs, data, bytes, (uint16_t)(1u << bits_win));   /* test.c:61 */
        r = s->error;
    }
    if (s != NULL) { squeeze.free(s); s = NULL; }
    if (r != 0) {
        printf("Failed to compress: %s\n", strerror(r));
    } else {
        const double percent = bytes > 0 ? bs.bytes * 100.0 / (double)bytes : 0.0;
        if (name != NULL) {                                            /* test.c:85-88 */
            printf("%7lld -> %7lld %5.1f%% of \"%s\"\n", (long long)bytes, (long long)bs.bytes, percent, name);
        } else {
            printf("%7lld -> %7lld %5.1f%%\n", (long long)bytes, (long long)bs.bytes, percent);
        }
        bitstream rd = { .data = comp, .bytes = bs.bytes };            /* test.c:110 */
        uint64_t n = 0; uint8_t win_bits = 0;
        squeeze.read_header(&rd, &n, &win_bits);                       /* test.c:114 */
        if (rd.error != 0 || n != bytes || win_bits != bits_win) {
            printf("Failed to read header\n");
            r = rd.error != 0 ? rd.error : EINVAL;
        } else {
            s = squeeze.alloc(0);                                      /* test.c:121 */
            if (s == NULL) { r = ENOMEM; }
            else {
                squeeze.decompress(s, &rd, back, (size_t)n);           /* test.c:134 */
                r = s->error;
                squeeze.free(s);
            }
            if (r == 0 && memcmp(data, back, bytes) != 0) {            /* test.c:138-144 */
                size_t k = 0;
                while (k < bytes && data[k] == back[k]) { k++; }
                printf("Decompressed data does not match input at offset %lld\n", (long long)k);
                r = EINVAL;
            }
        }
    }
    free(comp);
    free(back);
    return r;
}

static int test_file(const char* dir, const char* fn) {
    char path[1024];
    snprintf(path, sizeof(path), "%s/%s", dir, fn);
    uint8_t* data = NULL; size_t bytes = 0;
    int r = read_fully(path, &data, &bytes);
    if (r == ENOENT) { return 0; }                                     /* file_exist(): skip */
    if (r == 0) { r = test(fn, data, bytes); }
    free(data);
    return r;
}

int main(int argc, const char* argv[]) {
    if (argc > 1) { bits_win = atoi(argv[1]); }
    const char* corpus = argc > 2 ? argv[2] : "tests/corpus";
    char name[128]; int cus = 0; uint64_t lds = 0;
    int r = sqz_hip_device_info(name, sizeof(name), &cus, &lds);
    if (r != 0) { printf("no gfx950 device: %s\n", strerror(r)); return r; }
    printf("%s on %s (%d CUs), win_bits=%d\n", sqz_version(), name, cus, bits_win);
    static uint8_t data[4 * 1024];
    r = test(NULL, data, sizeof(data));                                /* test.c:199-200 */
    if (r == 0) {
        for (size_t i = 0; i < sizeof(data); i += 4) { memcpy(data + i, "\x01\x02\x03\x04", 4); }
        r = test(NULL, data, sizeof(data));                            /* test.c:202-205 */
    }
    if (r == 0) {
        const char* hello = "Hello World Hello.World Hello World";     /* test.c:208-210 */
        r = test(NULL, (const uint8_t*)hello, strlen(hello));
    }
    if (r == 0) { r = test_file(".", __FILE__); }                      /* test.c:212-214 */
    if (r == 0) { r = test_file(".", argv[0]); }                       /* test.c:216-218 */
    static const char* files[] = {                                     /* test.c:219-229 */
        "bible.txt", "hhgttg.txt", "confucius.txt", "laozi.txt", "sqlite3.c",
        "arm64.elf", "x64.elf", "mandrill.bmp", "mandrill.png" };
    for (size_t i = 0; i < sizeof(files) / sizeof(files[0]) && r == 0; i++) {
        r = test_file(corpus, files[i]);
    }
    printf(r == 0 ? "ok\n" : "FAILED: %s\n", strerror(r));
    return r;
}
