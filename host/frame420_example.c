/*
 * frame420_example.c -- a plain-C host taking one 4:2:0 frame pair through the tile stage of include/x266hip.h the way an
 * encoder's frame loop would, with no HIP headers:
 *
 *   planar Y, U, V of the current and the predicted frame (host memory)
 *     -> xConvInputFmtDev                    (xConvInputFmt, src/x266.cpp:415-453: 512-byte ref_block_t tiles)
 *     -> xDct32FwdCtuFromTilesDev            one launch: per 64x64 CTU the coefficients of Y0 Y1 Y2 Y3 U V (12 KiB)
 *     -> xSatd8x8FromTilesDev / xSatd8x8ChromaFromTilesDev   the 8x8 costs of the same residual, luma and chroma
 *
 * and checks every output against the two-step calls it fuses (residual formed in HBM by xResidualLumaDev / xResidualChromaDev,
 * then the pinned batch kernels) -- on the host, block by block.
 *
 *   usage: frame420_example [width height]      (multiples of 64; default 1920 1088)      exit code 0 on success
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/x266hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != X266HIP_OK) { fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, xHipLastError(hip)); return 2; } } while (0)

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(int argc, char **argv)
{
    const int w = argc > 2 ? atoi(argv[1]) : 1920, h = argc > 2 ? atoi(argv[2]) : 1088;
    if (w <= 0 || h <= 0 || (w & 63) || (h & 63)) { fprintf(stderr, "width and height must be multiples of 64\n"); return 2; }
    x266hip_ctx *hip = NULL;
    if (xHipCodecInit(&hip, 0) != X266HIP_OK) { fprintf(stderr, "no gfx950 device\n"); return 2; }
    const size_t npx = (size_t)w * h, nchroma = npx / 4, n_tiles = npx / 256, n_ctu = npx / 4096;

    /* a smooth current frame and a prediction that is the same picture moved and dimmed a little */
    uint8_t *planes[2][3];
    uint32_t seed = 0x266;
    for (int f = 0; f < 2; f++)
        for (int p = 0; p < 3; p++) planes[f][p] = malloc(p ? nchroma : npx);
    for (int p = 0; p < 3; p++) {
        const int pw = p ? w / 2 : w, ph = p ? h / 2 : h;
        for (int y = 0; y < ph; y++)
            for (int x = 0; x < pw; x++) {
                const int v = 128 + (int)(96.0 * ((x * 7 + y * 3 + 13 * p) % 97) / 97.0) - 48 + (int)(lcg(&seed) % 7) - 3;
                planes[0][p][(size_t)y * pw + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
                const int u = 126 + (int)(96.0 * (((x + 2) * 7 + (y + 1) * 3 + 13 * p) % 97) / 97.0) - 48;
                planes[1][p][(size_t)y * pw + x] = (uint8_t)(u < 0 ? 0 : u > 255 ? 255 : u);
            }
    }

    /* upload the planes, pack both frames into tiles on the device */
    void *d_plane[2][3], *d_tiles[2];
    for (int f = 0; f < 2; f++) {
        for (int p = 0; p < 3; p++) {
            CHECK(xHipMalloc(hip, &d_plane[f][p], p ? nchroma : npx));
            CHECK(xHipMemcpyH2D(hip, d_plane[f][p], planes[f][p], p ? nchroma : npx));
        }
        CHECK(xHipMalloc(hip, &d_tiles[f], n_tiles * sizeof(x266_ref_block_t)));
        CHECK(xConvInputFmtDev(hip, (x266_ref_block_t *)d_tiles[f], d_plane[f][0], d_plane[f][1], d_plane[f][2], w, w, h, NULL));
    }
    const x266_ref_block_t *cur = d_tiles[0], *pred = d_tiles[1];

    /* the fused forms: one launch per output */
    void *d_ctu, *d_cost_y, *d_cost_c;
    CHECK(xHipMalloc(hip, &d_ctu, n_ctu * 6 * 2048));
    CHECK(xHipMalloc(hip, &d_cost_y, npx / 64 * 4));
    CHECK(xHipMalloc(hip, &d_cost_c, n_tiles * 2 * 4));
    CHECK(xDct32FwdCtuFromTilesDev(hip, cur, pred, w, h, (int16_t *)d_ctu, NULL));
    CHECK(xSatd8x8FromTilesDev(hip, cur, pred, w, h, (uint32_t *)d_cost_y, NULL));
    CHECK(xSatd8x8ChromaFromTilesDev(hip, cur, pred, w, h, (uint32_t *)d_cost_c, (uint32_t *)d_cost_c + 1, 2, NULL));   /* (U, V) pairs per tile */

    /* the two-step calls: residual in HBM, then the pinned batch kernels */
    void *d_res_y, *d_res_c, *d_coef_y, *d_coef_c, *d_res8_y, *d_res8_c, *d_cost2_y, *d_cost2_c;
    CHECK(xHipMalloc(hip, &d_res_y, npx * 2)); CHECK(xHipMalloc(hip, &d_coef_y, npx * 2));
    CHECK(xHipMalloc(hip, &d_res_c, nchroma * 4)); CHECK(xHipMalloc(hip, &d_coef_c, nchroma * 4));
    CHECK(xHipMalloc(hip, &d_res8_y, npx * 2)); CHECK(xHipMalloc(hip, &d_res8_c, nchroma * 4));
    CHECK(xHipMalloc(hip, &d_cost2_y, npx / 64 * 4)); CHECK(xHipMalloc(hip, &d_cost2_c, n_tiles * 2 * 4));
    CHECK(xResidualLumaDev(hip, cur, pred, w, h, 32, (int16_t *)d_res_y, NULL));
    CHECK(xDct32FwdBatchDev(hip, (const int16_t *)d_res_y, (int16_t *)d_coef_y, npx / 1024, NULL));
    /* chroma, CTU order: U0 V0 U1 V1 ... (block_pitch 2, V one block behind U) -> one batch call transforms both planes */
    CHECK(xResidualChromaDev(hip, cur, pred, w, h, 32, (int16_t *)d_res_c, (int16_t *)d_res_c + 1024, 2, NULL));
    CHECK(xDct32FwdBatchDev(hip, (const int16_t *)d_res_c, (int16_t *)d_coef_c, n_ctu * 2, NULL));
    CHECK(xResidualLumaDev(hip, cur, pred, w, h, 8, (int16_t *)d_res8_y, NULL));
    CHECK(xSatd8x8BatchDev(hip, (const int16_t *)d_res8_y, (uint32_t *)d_cost2_y, npx / 64, NULL));
    CHECK(xResidualChromaDev(hip, cur, pred, w, h, 8, (int16_t *)d_res8_c, (int16_t *)d_res8_c + 64, 2, NULL));
    CHECK(xSatd8x8BatchDev(hip, (const int16_t *)d_res8_c, (uint32_t *)d_cost2_c, n_tiles * 2, NULL));
    CHECK(xHipStreamSync(hip, NULL));

    int16_t *ctu = malloc(n_ctu * 6 * 2048), *coef_y = malloc(npx * 2), *coef_c = malloc(nchroma * 4), *res_y = malloc(npx * 2);
    uint32_t *cost_y = malloc(npx / 64 * 4), *cost2_y = malloc(npx / 64 * 4), *cost_c = malloc(n_tiles * 8), *cost2_c = malloc(n_tiles * 8);
    CHECK(xHipMemcpyD2H(hip, ctu, d_ctu, n_ctu * 6 * 2048)); CHECK(xHipMemcpyD2H(hip, coef_y, d_coef_y, npx * 2));
    CHECK(xHipMemcpyD2H(hip, coef_c, d_coef_c, nchroma * 4)); CHECK(xHipMemcpyD2H(hip, res_y, d_res_y, npx * 2));
    CHECK(xHipMemcpyD2H(hip, cost_y, d_cost_y, npx / 64 * 4)); CHECK(xHipMemcpyD2H(hip, cost2_y, d_cost2_y, npx / 64 * 4));
    CHECK(xHipMemcpyD2H(hip, cost_c, d_cost_c, n_tiles * 8)); CHECK(xHipMemcpyD2H(hip, cost2_c, d_cost2_c, n_tiles * 8));

    /* residual definition on the host: block (by, bx) of the luma plane, row-major */
    const int bxn = w / 32;
    size_t bad = 0;
    for (size_t b = 0; b < npx / 1024 && bad == 0; b += 97) {
        const int by = (int)(b / bxn), bx = (int)(b % bxn);
        for (int r = 0; r < 32; r++)
            for (int c = 0; c < 32; c++) {
                const size_t px = (size_t)(by * 32 + r) * w + bx * 32 + c;
                if (res_y[b * 1024 + r * 32 + c] != (int16_t)((int)planes[0][0][px] - (int)planes[1][0][px])) bad++;
            }
    }
    /* CTU order against frame-raster luma and the CTU-ordered chroma pairs */
    const int cxn = w / 64;
    for (size_t c = 0; c < n_ctu; c++) {
        const size_t cy = c / cxn, cx = c % cxn;
        for (int q = 0; q < 4; q++) {
            const size_t blk = (cy * 2 + (q >> 1)) * (size_t)bxn + cx * 2 + (q & 1);
            if (memcmp(ctu + (c * 6 + q) * 1024, coef_y + blk * 1024, 2048)) bad++;
        }
        if (memcmp(ctu + (c * 6 + 4) * 1024, coef_c + c * 2048, 4096)) bad++;          /* U then V */
    }
    if (memcmp(cost_y, cost2_y, npx / 64 * 4)) bad++;
    if (memcmp(cost_c, cost2_c, n_tiles * 8)) bad++;
    unsigned long long sum_y = 0, sum_c = 0;
    for (size_t i = 0; i < npx / 64; i++) sum_y += cost_y[i];
    for (size_t i = 0; i < n_tiles * 2; i++) sum_c += cost_c[i];
    printf("{\"frame\": \"%dx%d 4:2:0\", \"ctus\": %zu, \"coefficient_bytes\": %zu, \"luma_satd_sum\": %llu, \"chroma_satd_sum\": %llu, "
           "\"fused_equals_two_step\": %s}\n", w, h, n_ctu, n_ctu * 6 * 2048, sum_y, sum_c, bad ? "false" : "true");
    xHipCodecFree(hip);
    return bad ? 1 : 0;
}
