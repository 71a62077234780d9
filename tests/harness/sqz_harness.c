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
 * File mode: like the reference, every case is compressed INTO A FILE ("~compressed~.bin",
 * test.c:163) and verified FROM that file, and the file holds what the reference's harness
 * writes: the 64-bit words of the stream in host byte order (fwrite(&b64, 8, 1), test.c:39-42,
 * read back with fread, test.c:98-101).  The device cannot call back per word, so the stream
 * is produced in a host buffer (memory mode, bitstream.h:34-43) and sqz_file_words() turns it
 * into the file image (the .file images under tests/golden are the reference's own files).
 * locate_test_folder() (test.c:183-193) walks up until the corpus is found.
 * SQZ_HARNESS_KEEP=<dir> additionally keeps each named case as <dir>/<name>.w<bits>.file.
 *
 * Build:  gcc -std=c99 -O2 -Iinclude tests/harness/sqz_harness.c -Lsqz_amd/lib -lsqz_amd
 *         -Wl,-rpath,$PWD/sqz_amd/lib -o tests/harness/sqz_harness
 * Run:    tests/harness/sqz_harness [win_bits] [corpus_dir]     (needs an MI355X)
 */
#define _POSIX_C_SOURCE 200809L   /* getcwd, chdir */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

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

static const char* compressed = "~compressed~.bin";                    /* test.c:163 */

static int write_image(const char* to, const uint8_t* image, uint64_t bytes) {
    FILE* out = fopen(to, "wb");                                       /* test.c:47 */
    if (out == NULL) { printf("Failed to create \"%s\": %s\n", to, strerror(errno)); return errno; }
    int r = fwrite(image, 1, (size_t)bytes, out) == (size_t)bytes ? 0 : (errno != 0 ? errno : EIO);
    if (fclose(out) != 0 && r == 0) {                                  /* test.c:72-76 */
        r = errno;
        printf("Failed to flush on file close: %s\n", strerror(r));
    }
    return r;
}

/* test.c:44-96: header + payload through the vtable, then the file the reference writes */
static int compress(const char* from, const char* to, const uint8_t* data, size_t bytes) {
    const uint64_t capacity = sqz_bound(bytes) + 16;
    uint8_t* comp = (uint8_t*)malloc(capacity);
    if (comp == NULL) { return ENOMEM; }
    int r = 0;
    bitstream bs = { .data = comp, .capacity = capacity };
    squeeze.write_header(&bs, bytes, (uint8_t)bits_win);               /* test.c:54 */
    if (bs.error != 0) { r = bs.error; }
    squeeze_type* s = NULL;
    if (r == 0) {
        s = squeeze.alloc(0);                                          /* test.c:59 */
        if (s == NULL) { r = ENOMEM; printf("squeeze_new() failed.\n"); }
    }
    if (r == 0) {
        squeeze.compress(s, &bs, data, bytes, (uint16_t)(1u << bits_win));   /* test.c:61 */
        r = s->error;
    }
    if (s != NULL) { squeeze.free(s); s = NULL; }
    if (r != 0) {
        printf("Failed to compress: %s\n", strerror(r));
    } else {
        r = sqz_file_words(comp, bs.bytes, comp);                      /* words in host order, in place */
        if (r == 0) { r = write_image(to, comp, bs.bytes); }
        const char* keep = getenv("SQZ_HARNESS_KEEP");
        if (r == 0 && keep != NULL && from != NULL) {
            const char* fn = strrchr(from, '/');                       /* basename, test.c:81-83 */
            fn = fn != NULL ? fn + 1 : from;
            char path[1024];
            snprintf(path, sizeof(path), "%s/%s.w%d.file", keep, fn, bits_win);
            r = write_image(path, comp, bs.bytes);
        }
    }
    if (r == 0) {
        const char* fn = from == NULL ? NULL : strrchr(from, '/');
        fn = fn != NULL ? fn + 1 : from;
        const double percent = bytes > 0 ? bs.bytes * 100.0 / (double)bytes : 0.0;
        if (from != NULL) {                                            /* test.c:85-88 */
            printf("%7lld -> %7lld %5.1f%% of \"%s\"\n", (long long)bytes, (long long)bs.bytes, percent, fn);
        } else {
            printf("%7lld -> %7lld %5.1f%%\n", (long long)bytes, (long long)bs.bytes, percent);
        }
    }
    free(comp);
    return r;
}

/* test.c:103-162: read the file back, decompress, compare */
static int verify(const char* fn, const uint8_t* input, size_t size) {
    uint8_t* image = NULL; size_t n_image = 0;
    int r = read_fully(fn, &image, &n_image);
    if (r != 0) { printf("Failed to open \"%s\"\n", fn); return r; }
    uint8_t* back = (uint8_t*)malloc(size + 1);
    if (back == NULL) { free(image); return ENOMEM; }
    r = sqz_file_words(image, n_image, image);                         /* file words -> stream */
    bitstream rd = { .data = image, .bytes = n_image };                /* test.c:110 */
    uint64_t n = 0; uint8_t win_bits = 0;
    if (r == 0) {
        squeeze.read_header(&rd, &n, &win_bits);                       /* test.c:114 */
        if (rd.error != 0 || n != size || win_bits != bits_win) {
            printf("Failed to read header\n");
            r = rd.error != 0 ? rd.error : EINVAL;
        }
    }
    if (r == 0) {
        squeeze_type* s = squeeze.alloc(0);                            /* test.c:121 */
        if (s == NULL) { r = ENOMEM; }
        else {
            squeeze.decompress(s, &rd, back, (size_t)n);               /* test.c:134 */
            r = s->error;
            squeeze.free(s);
        }
        if (r == 0 && memcmp(input, back, size) != 0) {                /* test.c:138-144 */
            size_t k = 0;
            while (k < size && input[k] == back[k]) { k++; }
            printf("Decompressed data does not match input at offset %lld\n", (long long)k);
            r = EINVAL;
        }
    }
    free(back);
    free(image);
    return r;
}

static int test(const char* fn, const uint8_t* data, size_t bytes) {   /* test.c:165-172 */
    int r = compress(fn, compressed, data, bytes);
    if (r == 0) { r = verify(compressed, data, bytes); }
    (void)remove(compressed);
    return r;
}

/* test.c:183-193: walk up until the test files are found (the snapshot's corpus lives in
 * tests/corpus; the reference looks for test/bible.txt) */
static int locate_test_folder(char* dir, size_t cap) {
    for (int up = 0; up < 16; up++) {
        FILE* f = fopen("tests/corpus/laozi.txt", "rb");
        if (f != NULL) {
            fclose(f);
            return getcwd(dir, cap) != NULL ? 0 : errno;
        }
        if (chdir("..") != 0) { return errno; }
    }
    return ENOENT;
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
    static char root[1024];
    const char* corpus = argc > 2 ? argv[2] : "tests/corpus";
    if (argc <= 2) {                                                   /* test.c:197 */
        const int lr = locate_test_folder(root, sizeof(root));
        if (lr != 0) { printf("test files not found: %s\n", strerror(lr)); return lr; }
    }
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
