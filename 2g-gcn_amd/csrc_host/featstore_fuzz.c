// Mutation driver for the frame decoder, built with -fsanitize=address,undefined (CPU only; see the Makefile).
// usage: featstore_fuzz <frame file> <iterations> <seed>
// Decodes the pristine frame (must succeed), then `iterations` corrupted copies (bit flips, byte stores, truncations,
// header field edits) into an exactly-sized heap buffer: any out-of-bounds access aborts under the sanitizer; a
// corrupted frame may decode or fail, it must never crash. Prints "ok <decoded> <rejected>".
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/twog_featstore.h"

static uint64_t rng_state;
static uint32_t rnd(void) {  // xorshift64*
    rng_state ^= rng_state >> 12;
    rng_state ^= rng_state << 25;
    rng_state ^= rng_state >> 27;
    return (uint32_t)((rng_state * 2685821657736338717ull) >> 32);
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* frame = (uint8_t*)malloc((size_t)n);
    if (fread(frame, 1, (size_t)n, f) != (size_t)n) return 2;
    fclose(f);
    const int iters = atoi(argv[2]);
    rng_state = 0x9e3779b97f4a7c15ull ^ (uint64_t)atoll(argv[3]);

    twog_blosc_info_t h;
    if (twog_blosc_info(frame, n, &h) != 0) return 3;
    uint8_t* out = (uint8_t*)malloc(h.nbytes ? (size_t)h.nbytes : 1);
    if (twog_blosc_decode(frame, n, out, h.nbytes, 1) != h.nbytes) return 4;
    if (twog_blosc_decode(frame, n, out, h.nbytes, 4) != h.nbytes) return 5;

    long decoded = 0, rejected = 0;
    for (int it = 0; it < iters; ++it) {
        long len = n;
        uint8_t* m = (uint8_t*)malloc((size_t)n);  // exact size: reads past a truncated copy are caught
        memcpy(m, frame, (size_t)n);
        const int kind = rnd() % 5;
        if (kind == 0) {  // bit flips anywhere
            const int k = 1 + rnd() % 4;
            for (int i = 0; i < k; ++i) m[rnd() % n] ^= (uint8_t)(1u << (rnd() % 8));
        } else if (kind == 1) {  // header / block-table bytes
            const long span = n < 64 ? n : 64;
            m[rnd() % span] = (uint8_t)rnd();
        } else if (kind == 2) {  // truncation
            len = rnd() % n;
        } else if (kind == 3) {  // a run of 0xff (long length codes) or 0x00 (zero offsets)
            const long at = rnd() % n, run = 1 + rnd() % 16;
            for (long i = at; i < n && i < at + run; ++i) m[i] = (rnd() & 1) ? 0xff : 0x00;
        } else {  // stream length fields: write a random le32 at a random position
            const long at = rnd() % (n > 4 ? n - 4 : 1);
            const uint32_t v = rnd() >> (rnd() % 32);
            memcpy(m + at, &v, n >= 4 ? 4 : (size_t)n);
        }
        uint8_t* shrunk = (uint8_t*)malloc(len ? (size_t)len : 1);  // exactly `len` readable bytes
        memcpy(shrunk, m, (size_t)len);
        const int64_t rc = twog_blosc_decode(shrunk, len, out, h.nbytes, (it & 1) ? 3 : 1);
        if (rc >= 0) ++decoded; else ++rejected;
        free(shrunk);
        free(m);
    }
    printf("ok %ld %ld\n", decoded, rejected);
    free(out);
    free(frame);
    return 0;
}
