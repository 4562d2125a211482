/*
 * batch_example.c -- a plain-C host using the batch API of include/x266hip.h the way an encoder
 * would, with no HIP headers: device memory through xHipMalloc,
 * device-pointer entry points on the default stream.  It also cross-checks the two surfaces of
 * the library against each other: the coefficients served by the BDPI shims for a block must be
 * the ones the batch entry point computes for the same block.
 *
 *   usage: batch_example [n_blocks]        exit code 0 on success
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/x266hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != X266HIP_OK) { fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, xHipLastError(hip)); return -1; } } while (0)

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 4096;
    x266hip_ctx *hip = NULL;
    if (xHipCodecInit(&hip, 0) != X266HIP_OK) { fprintf(stderr, "no gfx950 device\n"); return -1; }
    char name[128]; int cus = 0, mhz = 0; size_t mem = 0;
    CHECK(xHipDeviceInfo(hip, name, sizeof name, &cus, &mhz, &mem));
    printf("device: %s, %d CUs, %d MHz, %.0f GB\n", name, cus, mhz, mem / 1e9);

    /* 1. BDPI block -> same block through the batch API */
    unsigned int words[32];
    int16_t block[1024], coef[1024];
    dct32_genNew();
    for (int i = 0; i < 16; i++) {
        dct32_getDiff(words);
        for (int w = 0; w < 32; w++) {               /* unpack Vector#(2,Vector#(32,Bit#(16))), dct32.c:205-220 */
            block[(2 * i + w / 16) * 32 + 2 * (w % 16)]     = (int16_t)(words[w] & 0xFFFF);
            block[(2 * i + w / 16) * 32 + 2 * (w % 16) + 1] = (int16_t)(words[w] >> 16);
        }
    }
    CHECK(xDct32FwdBatch(hip, block, coef, 1));
    for (int idx = 0; idx < 1024; idx += 4)
        if (dct32_getDct() != xDct32PackDctWord(coef, idx)) { fprintf(stderr, "BDPI / batch mismatch at %d\n", idx); return -1; }
    printf("BDPI surface and batch surface agree on the rand() block\n");

    /* 2. device-resident batch: fill -> forward -> inverse, round-trip error */
    void *d_x = NULL, *d_z = NULL, *d_r = NULL;
    CHECK(xHipMalloc(hip, &d_x, n * 2048)); CHECK(xHipMalloc(hip, &d_z, n * 2048)); CHECK(xHipMalloc(hip, &d_r, n * 2048));
    CHECK(xFillResidualDev(hip, (int16_t *)d_x, n * 1024, 0x266, 0, NULL));
    CHECK(xDct32FwdBatchDev(hip, (const int16_t *)d_x, (int16_t *)d_z, n, NULL));
    CHECK(xDct32InvBatchDev(hip, (const int16_t *)d_z, (int16_t *)d_r, n, NULL));
    CHECK(xHipStreamSync(hip, NULL));
    int16_t *x = malloc(n * 2048), *r = malloc(n * 2048);
    CHECK(xHipMemcpyD2H(hip, x, d_x, n * 2048)); CHECK(xHipMemcpyD2H(hip, r, d_r, n * 2048));
    int worst = 0;
    for (size_t i = 0; i < n * 1024; i++) { int e = abs((int)x[i] - (int)r[i]); if (e > worst) worst = e; }
    printf("%zu blocks forward + inverse on the device: max reconstruction error %d\n", n, worst);
    /* 3. the literal drop-in path: host pointers in and out (chunks of 16 MiB over three staging slots, uploads, kernels and
     *    downloads overlapped -- include/x266hip.h); must equal what the device-pointer call left in d_z */
    int16_t *zh = malloc(n * 2048), *zd = malloc(n * 2048);
    uint32_t *cost = malloc((n * 16) * sizeof *cost);
    CHECK(xDct32FwdBatch(hip, x, zh, n));
    CHECK(xHipMemcpyD2H(hip, zd, d_z, n * 2048));
    if (memcmp(zh, zd, n * 2048) != 0) { fprintf(stderr, "host-pointer and device-pointer forward transforms differ\n"); return -1; }
    CHECK(xSatd8x8Batch(hip, x, cost, n * 16));               /* the same samples as 8x8 blocks */
    CHECK(xDct32InvBatch(hip, zh, zd, n));
    if (memcmp(zd, r, n * 2048) != 0) { fprintf(stderr, "host-pointer and device-pointer inverse transforms differ\n"); return -1; }
    printf("host-pointer calls agree with the device-pointer calls on %zu blocks (first SATD cost %u)\n", n, cost[0]);
    free(zh); free(zd); free(cost);
    double ms = 0;
    CHECK(xHipTimeKernel(hip, 0, d_x, d_z, n, 20, NULL, &ms));
    printf("forward: %.3f ms per launch, %.3e blocks/s\n", ms, n / ms * 1e3);
    free(x); free(r);
    CHECK(xHipFree(hip, d_x)); CHECK(xHipFree(hip, d_z)); CHECK(xHipFree(hip, d_r));
    xHipCodecFree(hip);
    return worst <= 6 ? 0 : -1;
}
