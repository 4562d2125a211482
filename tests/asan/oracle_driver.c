/*
 * oracle_driver.c -- TEST INFRASTRUCTURE: the oracle's C restatements (the oracle directory's .c files, compiled into this program with
 * -fsanitize=address,undefined -fno-sanitize-recover) on the inputs where undefined behaviour would hide: full-range and
 * +-32768 data through the int16-wrapping SATD stages, the truncating transform stores, the clipping inverses, the search
 * harness at frame edges.  Prints one checksum per function; tests/test_sanitizers.py compares them with what the same
 * driver prints when built WITHOUT the sanitizers (the instrumented build must compute the same numbers).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/x266_oracle.h"

static unsigned long long fnv(const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

int main(void)
{
    enum { NB = 300 };
    int16_t *x = malloc(sizeof(int16_t) * NB * 1024), *y = malloc(sizeof(int16_t) * NB * 1024), *z = malloc(sizeof(int16_t) * NB * 1024);
    /* a third 9-bit residuals, a third full-range, a third extremes */
    orc_fill_residual(x, (size_t)100 * 1024, 0x266, 0);
    { unsigned long long r = 0x267; for (int i = 0; i < 100 * 1024; i++) { r = r * 6364136223846793005ull + 1442695040888963407ull; x[100 * 1024 + i] = (int16_t)(r >> 40); } }
    for (int i = 0; i < 100 * 1024; i++) x[200 * 1024 + i] = (i * 2654435761u >> 7) & 1 ? 32767 : -32768;
    orc_dct32_fwd(x, y, NB);
    printf("dct32_fwd %016llx\n", fnv(y, sizeof(int16_t) * NB * 1024));
    orc_dct32_inv(x, z, NB);                                         /* arbitrary int16 input: exercises the clipping */
    printf("dct32_inv %016llx\n", fnv(z, sizeof(int16_t) * NB * 1024));
    uint32_t *s = malloc(sizeof(uint32_t) * NB * 16);
    orc_satd8x8_batch(x, s, (size_t)NB * 16);
    printf("satd8x8 %016llx\n", fnv(s, sizeof(uint32_t) * NB * 16));
    for (int type = 0; type < 4; type++)
        for (int n = 4; n <= 16; n *= 2) {
            const size_t nb = (size_t)NB * 1024 / (size_t)(n * n);
            if (orc_transform_fwd(type, n, x, y, nb) || orc_transform_inv(type, n, x, z, nb)) return 1;
            printf("transform_%d_%d %016llx %016llx\n", type, n, fnv(y, sizeof(int16_t) * NB * 1024), fnv(z, sizeof(int16_t) * NB * 1024));
        }
    {   /* caller-supplied matrices with the int8 extremes */
        int16_t m[64];
        for (int i = 0; i < 64; i++) m[i] = (int16_t)((i * 37) % 256 - 128);
        if (orc_transform_fwd_matrix(m, m, 8, x, y, (size_t)NB * 16) || orc_transform_inv_matrix(m, m, 8, x, z, (size_t)NB * 16)) return 1;
        printf("transform_matrix %016llx %016llx\n", fnv(y, sizeof(int16_t) * NB * 1024), fnv(z, sizeof(int16_t) * NB * 1024));
    }
    {   /* search harness on a small frame whose windows hang over every edge of the padded reference */
        enum { W = 40, H = 24, R = 9, PAD = 9 };
        unsigned char cur[W * H], ref[(W + 2 * PAD) * (H + 2 * PAD)];
        for (int i = 0; i < W * H; i++) cur[i] = (unsigned char)(x[i] & 0xFF);
        for (int i = 0; i < (W + 2 * PAD) * (H + 2 * PAD); i++) ref[i] = (unsigned char)(x[5000 + i] >> 3);
        int16_t mv[(W / 8) * (H / 8) * 2];
        uint32_t cost[(W / 8) * (H / 8)];
        uint32_t *costs = malloc(sizeof(uint32_t) * (W / 8) * (H / 8) * (2 * R + 1) * (2 * R + 1));
        orc_satd8x8_search(cur, W, ref + PAD * (W + 2 * PAD) + PAD, W + 2 * PAD, W, H, R, mv, cost, costs, 3);
        printf("satd_search %016llx %016llx\n", fnv(mv, sizeof mv), fnv(costs, sizeof(uint32_t) * (W / 8) * (H / 8) * (2 * R + 1) * (2 * R + 1)));
        orc_sad8x8_search(cur, W, ref + PAD * (W + 2 * PAD) + PAD, W + 2 * PAD, W, H, R, mv, cost, costs, 2);
        printf("sad_search %016llx %016llx\n", fnv(mv, sizeof mv), fnv(cost, sizeof cost));
        free(costs);
    }
    free(x); free(y); free(z); free(s);
    return 0;
}
