/*
 * plan_driver.c -- TEST INFRASTRUCTURE: the host-only parts of libx266hip under AddressSanitizer + UBSan (no device needed).
 * Links tests/../x266_amd/libx266hip_asan.so (make -C x266_amd/csrc asan).  Exercises the planning functions of the node
 * layer over their whole argument space, the BDPI packing helpers, the table query, and the error paths a host without a GPU
 * takes (context and node creation must fail cleanly, not crash or leak into undefined behaviour).
 * Prints "plan_driver ok <checksum>"; any sanitizer report aborts (-fno-sanitize-recover).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/x266hip.h"

static unsigned long long mix(unsigned long long h, unsigned long long v) { return (h ^ v) * 0x9E3779B97F4A7C15ull + (h >> 29); }

int main(void)
{
    unsigned long long h = 0x266;
    /* xShardRange: shards are contiguous, cover [0, n), differ by at most one unit; bad arguments are refused */
    const size_t sizes[] = {0, 1, 2, 7, 8, 9, 63, 64, 65, 1000, 32400, 518400, (size_t)1 << 20, ((size_t)1 << 32) + 5};
    for (unsigned si = 0; si < sizeof sizes / sizeof sizes[0]; si++)
        for (int world = 1; world <= 16; world++) {
            size_t prev = 0, lo = (size_t)-1, hi = 0;
            for (int r = 0; r < world; r++) {
                size_t b = 123, e = 456;
                if (xShardRange(sizes[si], r, world, &b, &e) != 0 || b != prev || e < b) { fprintf(stderr, "xShardRange(%zu, %d, %d)\n", sizes[si], r, world); return 1; }
                prev = e;
                if (e - b < lo) lo = e - b;
                if (e - b > hi) hi = e - b;
                h = mix(h, b * 31 + e);
            }
            if (prev != sizes[si] || hi - lo > 1) { fprintf(stderr, "xShardRange does not tile %zu over %d\n", sizes[si], world); return 1; }
        }
    {
        size_t b, e;
        if (xShardRange(10, -1, 4, &b, &e) == 0 || xShardRange(10, 4, 4, &b, &e) == 0 || xShardRange(10, 0, 0, &b, &e) == 0) { fprintf(stderr, "xShardRange accepted bad ranks\n"); return 1; }
        (void)xShardRange(10, 1, 4, NULL, NULL);                       /* NULL outputs: must not be written through */
    }
    /* xMeStripePlan: stripes tile the block rows; halo = +-range; bad arguments refused */
    const int heights[] = {8, 16, 64, 136, 544, 1080 - 1080 % 8, 2160, 4320};
    for (unsigned hi_ = 0; hi_ < sizeof heights / sizeof heights[0]; hi_++)
        for (int range = 0; range <= 64; range += 8)
            for (int n = 1; n <= 24; n++) {
                int prev = 0;
                for (int s = 0; s < n; s++) {
                    int b0, b1, r0, r1;
                    if (xMeStripePlan(heights[hi_], range, s, n, &b0, &b1, &r0, &r1) != 0 || b0 != prev || b1 < b0 || r0 != b0 * 8 - range || r1 != b1 * 8 + range) {
                        fprintf(stderr, "xMeStripePlan(%d, %d, %d, %d)\n", heights[hi_], range, s, n); return 1; }
                    prev = b1;
                    h = mix(h, (unsigned)(b0 * 131 + b1 * 17 + r0 + r1));
                }
                if (prev != heights[hi_] / 8) { fprintf(stderr, "stripes do not tile %d rows\n", heights[hi_]); return 1; }
            }
    {
        int a, b, c, d;
        if (!xMeStripePlan(12, 4, 0, 1, &a, &b, &c, &d) || !xMeStripePlan(16, -1, 0, 1, &a, &b, &c, &d) || !xMeStripePlan(16, 4, 2, 2, &a, &b, &c, &d) ||
            !xMeStripePlan(16, 4, 0, 0, &a, &b, &c, &d)) { fprintf(stderr, "xMeStripePlan accepted bad arguments\n"); return 1; }
        (void)xMeStripePlan(16, 4, 0, 1, NULL, NULL, NULL, NULL);
    }
    /* tables and BDPI packing (host only) */
    for (int type = 0; type < 2; type++)
        for (int n = 4; n <= 32; n *= 2) {
            int16_t m[32 * 32];
            const int rc = xTransformMatrix(type, n, m);
            if ((rc == 0) != (n < 32 || type == 0)) { fprintf(stderr, "xTransformMatrix(%d, %d) = %d\n", type, n, rc); return 1; }
            if (rc == 0) for (int i = 0; i < n * n; i++) h = mix(h, (unsigned short)m[i]);
        }
    {
        int16_t blk[1024];
        for (int i = 0; i < 1024; i++) blk[i] = (int16_t)(i * 37 - 20000);
        unsigned int res[32];
        for (int row = 0; row < 32; row += 2) { xDct32PackDiffRows(blk, row, res); for (int i = 0; i < 32; i++) h = mix(h, res[i]); }
        for (int idx = 0; idx < 1024; idx += 4) h = mix(h, xDct32PackDctWord(blk, idx));
    }
    /* no device: creation fails cleanly, NULL handles are tolerated everywhere they are documented to be */
    if (xHipDeviceCount() == 0) {
        x266hip_ctx *ctx = (x266hip_ctx *)0x1;
        if (xHipCodecInit(&ctx, 0) == 0 || ctx != NULL) { fprintf(stderr, "xHipCodecInit succeeded without a device\n"); return 1; }
        x266hip_node *node = (x266hip_node *)0x1;
        if (xHipNodeInit(&node, NULL, 1) == 0) { fprintf(stderr, "xHipNodeInit succeeded without a device\n"); return 1; }
    }
    xHipCodecFree(NULL);
    xHipNodeFree(NULL);
    xNodeStreamFree(NULL);
    if (xHipSetOption(NULL, "nontemporal", 1) == 0 || xNodeStreamFlush(NULL) == 0 || xHipNodeSelfTest(NULL) == 0) { fprintf(stderr, "NULL handle accepted\n"); return 1; }
    if (strlen(xHipLastError(NULL)) == 0 || !xHipVersion()) return 1;
    printf("plan_driver ok %016llx\n", h);
    return 0;
}
