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
 * File mode, exactly as the reference does it: every case is compressed INTO A FILE
 * ("~compressed~.bin", test.c:163) through a callback-mode bit stream
 * `{ .stream = FILE*, .output = write_file }` whose callback fwrite()s `b64` in host byte
 * order (test.c:39-42,53), and verified FROM that file through `{ .stream, .input = read_file }`
 * (test.c:98-101,110).  The shim replays the device-produced stream through `.output` word by
 * word and pulls words through `.input` (include/sqz/sqz.h); the .file images under
 * tests/golden are the files the compiled reference wrote through the same callbacks.
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

static int write_file(bitstream* bs) {                                 /* test.c:39-42 */
    return fwrite(&bs->b64, 8, 1, (FILE*)bs->stream) == 1 ? 0 : (errno != 0 ? errno : EIO);
}

static int read_file(bitstream* bs) {                                  /* test.c:98-101 */
    return fread(&bs->b64, 8, 1, (FILE*)bs->stream) == 1 ? 0 : (errno != 0 ? errno : EIO);
}

static int copy_file(const char* from, const char* to) {
    uint8_t* image = NULL; size_t n = 0;
    int r = read_fully(from, &image, &n);
    if (r != 0) { return r; }
    FILE* out = fopen(to, "wb");
    if (out == NULL) { r = errno; }
    else {
        if (fwrite(image, 1, n, out) != n) { r = errno != 0 ? errno : EIO; }
        if (fclose(out) != 0 && r == 0) { r = errno; }
    }
    free(image);
    return r;
}

/* test.c:44-96: header + payload through the vtable into the file, word by word */
static int compress(const char* from, const char* to, const uint8_t* data, size_t bytes) {
    FILE* out = fopen(to, "wb");                                       /* test.c:46-51 */
    if (out == NULL) {
        const int e = errno;
        printf("Failed to create \"%s\": %s\n", to, strerror(e));
        return e;
    }
    int r = 0;
    squeeze_type* s = NULL;
    bitstream bs = { .stream = out, .output = write_file };            /* test.c:53 */
    squeeze.write_header(&bs, bytes, (uint8_t)bits_win);               /* test.c:54 */
    if (bs.error != 0) {
        r = bs.error;
        printf("Failed to create \"%s\": %s\n", to, strerror(r));
    } else {
        s = squeeze.alloc(0);                                          /* test.c:59 */
        if (s != NULL) {
            squeeze.compress(s, &bs, data, bytes, (uint16_t)(1u << bits_win));   /* test.c:61 */
        } else {
            r = ENOMEM;
            printf("squeeze_new() failed.\n");
        }
    }
    const int rc = fclose(out) == 0 ? 0 : errno;                       /* test.c:70-76 */
    if (rc != 0) {
        printf("Failed to flush on file close: %s\n", strerror(rc));
        if (r == 0) { r = rc; }
    }
    if (r == 0) {
        r = s->error;                                                  /* test.c:77-78 */
        if (r != 0) {
            printf("Failed to compress: %s\n", strerror(r));
        } else {
            const char* fn = from == NULL ? NULL : strrchr(from, '/'); /* basename, test.c:81-83 */
            fn = fn != NULL ? fn + 1 : from;
            const uint64_t written = s->bs->bytes;                     /* test.c:84 */
            const double percent = bytes > 0 ? written * 100.0 / (double)bytes : 0.0;
            if (from != NULL) {                                        /* test.c:85-88 */
                printf("%7lld -> %7lld %5.1f%% of \"%s\"\n", (long long)bytes, (long long)written, percent, fn);
            } else {
                printf("%7lld -> %7lld %5.1f%%\n", (long long)bytes, (long long)written, percent);
            }
            const char* keep = getenv("SQZ_HARNESS_KEEP");
            if (keep != NULL && from != NULL) {
                char path[1024];
                snprintf(path, sizeof(path), "%s/%s.w%d.file", keep, fn, bits_win);
                r = copy_file(to, path);
            }
        }
    }
    if (s != NULL) { squeeze.free(s); s = NULL; }                      /* test.c:92-94 */
    return r;
}

/* test.c:103-162: read the file back through the .input callback, decompress, compare */
static int verify(const char* fn, const uint8_t* input, size_t size) {
    FILE* in = fopen(fn, "rb");                                        /* test.c:105-109 */
    if (in == NULL) { printf("Failed to open \"%s\"\n", fn); return errno; }
    int r = 0;
    bitstream bs = { .stream = in, .input = read_file };               /* test.c:110 */
    uint64_t bytes = 0; uint8_t win_bits = 0;
    squeeze.read_header(&bs, &bytes, &win_bits);                       /* test.c:114 */
    if (bs.error != 0 || bytes != size || win_bits != bits_win) {
        printf("Failed to read header from \"%s\"\n", fn);
        r = bs.error != 0 ? bs.error : EINVAL;
    }
    uint8_t* back = NULL;
    if (r == 0) {
        back = (uint8_t*)calloc(1, size + 1);                          /* test.c:127 */
        if (back == NULL) { r = ENOMEM; }
    }
    if (r == 0) {
        squeeze_type* s = squeeze.alloc(0);                            /* test.c:121 */
        if (s == NULL) { r = ENOMEM; printf("squeeze_new() failed.\n"); }
        else {
            squeeze.decompress(s, &bs, back, (size_t)bytes);           /* test.c:134 */
            r = s->error;
            squeeze.free(s);
        }
        if (r == 0 && memcmp(input, back, size) != 0) {                /* test.c:138-144 */
            size_t k = 0;
            while (k < size && input[k] == back[k]) { k++; }
            printf("compress() and decompress() are not the same @%lld\n", (long long)k);
            r = ENODATA;
        }
        if (r != 0) { printf("Failed to decompress\n"); }
    }
    fclose(in);
    free(back);
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
