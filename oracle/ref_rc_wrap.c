/* oracle/ref_rc_wrap.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Wrapper translation unit around the reference's HEAD codec: inc/sqz/sqz.h + src/sqz.c, compiled
 * WHERE THEY LIE under /root/reference (`make -C oracle ref-rc`: the reference's own src/sqz.c is
 * given to the compiler by path next to this file; clang -std=c23 -include errno.h as SURVEY.md
 * section 8c records).  Nothing of the reference is copied here: this file implements the
 * reference's runtime header the way its own mains do (`rt/ustd.h` without
 * UNSTD_NO_RT_IMPLEMENTATION, test.c / shl.c) and calls sqz_init / sqz_compress / sqz_decompress
 * with in-memory rc.write / rc.read callbacks (inc/sqz/sqz.h:49-50).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <errno.h>

#include "rt/ustd.h"           /* the reference's rt_printf / rt_swear implementations */
#include "sqz/sqz.h"

struct mem_io { uint8_t* out; uint64_t cap, written; const uint8_t* in; uint64_t avail, consumed; int dry_error; };

static void put(struct range_coder* rc, uint8_t b) {
    struct mem_io* io = (struct mem_io*)((struct sqz*)rc)->that;
    if (io->written < io->cap) { io->out[io->written] = b; } else { rc->error = ENOBUFS; }
    io->written++;
}

static uint8_t get(struct range_coder* rc) {
    struct mem_io* io = (struct mem_io*)((struct sqz*)rc)->that;
    uint8_t b = 0;
    /* a source that fails at its end reports it the way the reference's own callback does (test.c:112-121) */
    if (io->dry_error != 0 && io->consumed >= io->avail && rc->error == 0) { rc->error = io->dry_error; }
    if (io->consumed < io->avail) { b = io->in[io->consumed]; }
    io->consumed++;
    return b;
}

/* returns rc.error; *out_bytes = bytes written through rc.write */
int sqz_ref_rc_compress(const uint8_t* data, uint64_t bytes, uint32_t window, uint8_t* out, uint64_t cap,
                        uint64_t* out_bytes) {
    struct sqz* s = (struct sqz*)calloc(1, sizeof(struct sqz));
    if (s == NULL) { return ENOMEM; }
    struct mem_io io = { out, cap, 0, NULL, 0, 0, 0 };
    s->that = &io;
    sqz_init(s, NULL, 0);
    s->rc.write = put;
    /* the reference prints its statistics on every call (SQUEEZE_MAP_STATS is forced on, src/sqz.c:567):
     * keep them off this process's stdout */
    fflush(stdout);
    FILE* keep = stdout;
    stdout = fopen("/dev/null", "w");
    sqz_compress(s, data, (size_t)bytes, window);
    if (stdout != NULL) { fclose(stdout); }
    stdout = keep;
    const int e = s->rc.error;
    *out_bytes = io.written;
    free(s);
    return e;
}

int sqz_ref_rc_decompress_dry(const uint8_t* in, uint64_t in_bytes, uint8_t* data, uint64_t cap,
                              uint64_t* out_bytes, uint64_t* consumed, int dry_error);

int sqz_ref_rc_decompress(const uint8_t* in, uint64_t in_bytes, uint8_t* data, uint64_t cap,
                          uint64_t* out_bytes, uint64_t* consumed) {
    return sqz_ref_rc_decompress_dry(in, in_bytes, data, cap, out_bytes, consumed, 0);
}

int sqz_ref_rc_decompress_dry(const uint8_t* in, uint64_t in_bytes, uint8_t* data, uint64_t cap,
                              uint64_t* out_bytes, uint64_t* consumed, int dry_error) {
    struct sqz* s = (struct sqz*)calloc(1, sizeof(struct sqz));
    if (s == NULL) { return ENOMEM; }
    struct mem_io io = { NULL, 0, 0, in, in_bytes, 0, dry_error };
    s->that = &io;
    sqz_init(s, NULL, 0);
    s->rc.read = get;
    *out_bytes = sqz_decompress(s, data, (size_t)cap);
    if (consumed != NULL) { *consumed = io.consumed; }
    const int e = s->rc.error;
    free(s);
    return e;
}
