/* tests/harness/sqz_rc_harness.c -- C99 host over the R-era entry points of libsqz_amd.so.
 *
 * POSIX re-creation of what the reference's HEAD harness does (/root/reference/test.c): every input is
 * compressed INTO A FILE ("~compressed~.bin", test.c:163) -- the 8 bytes "squeeze4", the input's size as
 * a host-order uint64 (write_header, test.c:41-46), then the range coder's bytes, one rc.write call each
 * (put, test.c:48-55: the callback finds its file through `that` behind the rc-is-first-field cast and
 * carries the file's error into rc.error) -- and verified FROM that file: header back (read_header,
 * test.c:104-111: EILSEQ on a foreign id), bytes pulled one rc.read call each (get, test.c:113-122),
 * memcmp (test.c:150-162).  The lines printed per file are the reference's ("bps: %4.1f " and
 * "%7lld -> %7lld %6.2f%% of \"%s\"", test.c:83-91).  main() tests the files of test.c:241-251 that
 * exist (bible, hhgttg, confucius, laozi, sqlite3.c: the others are commented out there) -- and, an
 * addition, any file names given on the command line after the corpus directory.
 * The reference's own harness needs its rt/ headers and C23; this one is written
 * against <sqz/sqz_rc.h> with SQZ_RC_REFERENCE_NAMES, i.e. with the reference's spellings
 * (struct sqz, sqz_init, sqz_compress, sqz_decompress), as a HEAD caller would be compiled.
 * SQZ_HARNESS_KEEP=<dir> keeps each file's image as <dir>/<name>.rc.file.
 *
 * Build:  gcc -std=c99 -O2 -Iinclude tests/harness/sqz_rc_harness.c -Lsqz_amd/lib -lsqz_amd
 *         -Wl,-rpath,$PWD/sqz_amd/lib -o tests/harness/sqz_rc_harness
 * Run:    tests/harness/sqz_rc_harness [corpus_dir [file ...]]     (needs an MI355X)
 */
#define _POSIX_C_SOURCE 200809L   /* getcwd, chdir */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define SQZ_RC_REFERENCE_NAMES
#include <sqz/sqz_rc.h>

enum { window_bits = 11 };                                             /* test.c:18 (unused by HEAD's coder) */

/* the part of the reference's `struct io` (inc/rt/fileio.h) the harness uses: a file, a sticky error,
 * the count of bytes written */
struct io { FILE* file; int error; uint64_t written; };

static void io_write(struct io* io, const void* data, size_t bytes) {  /* fileio.h:180-194 */
    io->error = fwrite(data, bytes, 1, io->file) == 0 ? (errno != 0 ? errno : EIO) : 0;
    if (io->error == 0) { io->written += bytes; }
}

static void io_read(struct io* io, void* data, size_t bytes) {         /* fileio.h:196-210: a file that just */
    io->error = fread(data, bytes, 1, io->file) == 0 ? errno : 0;      /* ends leaves errno, usually 0        */
}

static void io_put(struct io* io, uint8_t b) { if (io->error == 0) { io_write(io, &b, 1); } }   /* fileio.h:212-219 */

static uint8_t io_get(struct io* io) {                                 /* fileio.h:221-230 */
    uint8_t b = 0;
    if (io->error == 0) { io_read(io, &b, 1); }
    return b;
}

static const uint8_t squeeze_id[8] = { 's', 'q', 'u', 'e', 'e', 'z', 'e', '4' };   /* test.c:41 */

static void write_header(struct io* io, uint64_t bytes) {              /* test.c:43-46 */
    io_write(io, squeeze_id, sizeof(squeeze_id));
    if (io->error == 0) { io_write(io, &bytes, sizeof(bytes)); }
}

static void read_header(struct io* io, uint64_t* bytes) {              /* test.c:104-111 */
    uint8_t id[8] = {0};
    io_read(io, id, sizeof(id));
    if (io->error == 0) { io_read(io, bytes, sizeof(*bytes)); }
    if (io->error == 0 && memcmp(id, squeeze_id, sizeof(id)) != 0) { io->error = EILSEQ; }
}

static void put(struct range_coder* rc, uint8_t b) {                   /* test.c:48-55 */
    struct sqz* s = (struct sqz*)rc;
    struct io* io = (struct io*)s->that;
    if (rc->error == 0) {
        io_put(io, b);
        rc->error = io->error;
    }
}

static uint8_t get(struct range_coder* rc) {                           /* test.c:113-122 */
    struct sqz* s = (struct sqz*)rc;
    struct io* io = (struct io*)s->that;
    uint8_t b = 0;
    if (rc->error == 0) {
        b = io_get(io);
        rc->error = io->error;
    }
    return b;
}

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

static int copy_file(const char* from, const char* to) {
    uint8_t* image = NULL; size_t n = 0;
    int r = read_fully(from, &image, &n);
    if (r != 0) { return r; }
    FILE* out = fopen(to, "wb");
    if (out == NULL) { r = errno; }
    else {
        if (n > 0 && fwrite(image, 1, n, out) != n) { r = errno != 0 ? errno : EIO; }
        if (fclose(out) != 0 && r == 0) { r = errno; }
    }
    free(image);
    return r;
}

static const char* compressed = "~compressed~.bin";                    /* test.c:163 */

static int compress(const char* from, const char* to, const uint8_t* data, size_t bytes) {   /* test.c:57-102 */
    struct io out = { NULL, 0, 0 };
    out.file = fopen(to, "wb");
    if (out.file == NULL) {
        out.error = errno;
        printf("Failed to create \"%s\": %s\n", to, strerror(out.error));
        return out.error;
    }
    static struct sqz encoder;                                         /* test.c:64 */
    encoder.that = &out;
    sqz_init(&encoder, NULL, 0);                                       /* (HEAD hands 32 M map entries: unused there too) */
    encoder.that = &out;
    encoder.rc.write = put;
    write_header(&out, bytes);
    if (out.error != 0) { encoder.rc.error = out.error; }
    if (encoder.rc.error != 0) {
        printf("io_create(\"%s\") failed: %s\n", to, strerror(encoder.rc.error));
    } else {
        sqz_compress(&encoder, data, bytes, 1u << window_bits);        /* test.c:74 */
        if (encoder.rc.error != 0) { printf("Failed to compress: %s\n", strerror(encoder.rc.error)); }
    }
    if (fclose(out.file) != 0 && encoder.rc.error == 0) {              /* test.c:80-84 */
        out.error = errno;
        printf("io_close(\"%s\") failed: %s\n", to, strerror(out.error));
        encoder.rc.error = out.error;
    }
    if (encoder.rc.error == 0) {
        const char* fn = from == NULL ? NULL : strrchr(from, '/');     /* basename, test.c:86-88 */
        fn = fn != NULL ? fn + 1 : from;
        const double pc = bytes > 0 ? out.written * 100.0 / (double)bytes : 0.0;
        const double bps = bytes > 0 ? out.written * 8.0 / (double)bytes : 0.0;
        printf("bps: %4.1f ", bps);                                    /* test.c:91-98 */
        if (from != NULL) {
            printf("%7lld -> %7lld %6.2f%% of \"%s\"\n\n", (long long)bytes, (long long)out.written, pc, fn);
        } else {
            printf("%7lld -> %7lld %6.2f%%\n\n", (long long)bytes, (long long)out.written, pc);
        }
        const char* keep = getenv("SQZ_HARNESS_KEEP");
        if (keep != NULL && fn != NULL) {
            char path[1024];
            snprintf(path, sizeof(path), "%s/%s.rc.file", keep, fn);
            const int cr = copy_file(to, path);
            if (cr != 0) { return cr; }
        }
    }
    return encoder.rc.error;
}

static int verify(const char* fn, const uint8_t* input, size_t size) { /* test.c:124-182 */
    struct io in = { NULL, 0, 0 };
    in.file = fopen(fn, "rb");
    if (in.file == NULL) { printf("Failed to open \"%s\"\n", fn); return errno; }
    uint64_t bytes = 0;
    static struct sqz decoder;
    sqz_init(&decoder, NULL, 0);                                       /* test.c:135 */
    decoder.that = &in;
    decoder.rc.read = get;
    read_header(&in, &bytes);
    if (in.error != 0) {
        printf("Failed to read header from \"%s\"\n", fn);
        decoder.rc.error = in.error;
    }
    uint8_t* back = NULL;
    if (decoder.rc.error == 0) {
        back = (uint8_t*)calloc(1, (size_t)bytes + 1);
        if (back == NULL) { decoder.rc.error = ENOMEM; }
        else if (bytes > size) { decoder.rc.error = E2BIG; }           /* test.c:153 */
    }
    if (decoder.rc.error == 0) {
        const uint64_t produced = sqz_decompress(&decoder, back, (size_t)bytes);   /* test.c:160 */
        if (decoder.rc.error == 0) {
            const int same = size == bytes && produced == bytes && memcmp(input, back, (size_t)bytes) == 0;
            if (!same) {
                long long k = -1;
                for (size_t i = 0; i < (bytes < size ? bytes : size) && k < 0; i++) {
                    if (input[i] != back[i]) { k = (long long)i; }
                }
                printf("compress() and decompress() differ @%d\n", (int)k);
                decoder.rc.error = ENODATA;
            }
        } else {
            printf("Failed to decompress: %s\n", strerror(decoder.rc.error));
        }
    }
    free(back);
    fclose(in.file);
    return decoder.rc.error;
}

static int test(const char* fn, const uint8_t* data, size_t bytes) {   /* test.c:165-172 */
    int r = compress(fn, compressed, data, bytes);
    if (r == 0) { r = verify(compressed, data, bytes); }
    (void)remove(compressed);
    return r;
}

/* test.c:200-211: walk up until the test files are found (the snapshot's corpus lives in tests/corpus;
 * the reference looks for test/bible.txt and never gives up) */
static int locate_test_folder(void) {
    for (int up = 0; up < 16; up++) {
        FILE* f = fopen("tests/corpus/laozi.txt", "rb");
        if (f != NULL) { fclose(f); return 0; }
        if (chdir("..") != 0) { return errno; }
    }
    return ENOENT;
}

static int test_file(const char* dir, const char* fn) {                /* test_compression, test.c:174-180 */
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
    printf("Window: 2^%d %d sizeof(size_t): %d sizeof(int): %d\n",    /* test.c:218-219 */
           (int)window_bits, (int)(1u << window_bits), (int)sizeof(size_t), (int)sizeof(int));
    const char* corpus = argc > 1 ? argv[1] : "tests/corpus";
    int r = 0;
    if (argc <= 1) {
        r = locate_test_folder();
        if (r != 0) { printf("test files not found: %s\n", strerror(r)); return r; }
    }
    {   /* loud without a device: the coder runs on the GPU, there is no CPU path behind these names */
        static struct sqz probe;
        static uint8_t sink[64];
        struct io none = { NULL, 0, 0 };
        none.file = fopen("/dev/null", "wb");
        if (none.file == NULL) { return errno; }
        sqz_init(&probe, NULL, 0);
        probe.that = &none;
        probe.rc.write = put;
        sqz_compress(&probe, sink, sizeof(sink), 1u << window_bits);
        fclose(none.file);
        if (probe.rc.error != 0) { printf("no gfx950 device: %s\n", strerror(probe.rc.error)); return probe.rc.error; }
    }
    static const char* files[] = {                                     /* test.c:241-251 */
        "bible.txt", "hhgttg.txt", "confucius.txt", "laozi.txt", "sqlite3.c" };
    for (size_t i = 0; i < sizeof(files) / sizeof(files[0]) && r == 0; i++) { r = test_file(corpus, files[i]); }
    for (int i = 2; i < argc && r == 0; i++) { r = test_file(corpus, argv[i]); }
    printf(r == 0 ? "ok\n" : "FAILED: %s\n", strerror(r));
    return r;
}
