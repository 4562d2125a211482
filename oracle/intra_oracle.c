/*
 * intra_oracle.c -- CPU restatement of 32x32 intra prediction.
 * TEST INFRASTRUCTURE ONLY (see x266_oracle.h).
 *
 * PARITY UNPINNED.  The reference holds only a work-in-progress RTL sketch of this stage
 * (src/mkIntra32-wip.bsv): no C model, a stub testbench (:525-551), half of the datapath under
 * `ifdef XXX` (:419-505).  What that file does pin is WHICH predictor is meant: the HEVC 35-mode
 * scheme on 32x32 luma blocks --
 *   - IntraRef_t {left[64], top[65]} (:36-39), top[0] being the corner sample (getRefPixels, mode
 *     16 = HEVC mode 18, reads xT[0] between the left and top runs, :237-246);
 *   - mapTbl / facTbl (:75-112) are ((k+1)*intraPredAngle) >> 5 (re-based to be non-negative) and
 *     ((k+1)*intraPredAngle) & 31 for the HEVC angles 32,26,21,17,13,9,5,2,0,-2,...,-32;
 *   - the projected side references listed in getRefPixels (:150-316) are HEVC's
 *     ((x*invAngle + 128) >> 8) with the inverse angles of Table 8-5;
 *   - two-tap interpolation (32-f)*a + f*b + 16 (:358-361), DC from 32 top + 32 left samples
 *     (:380-384, :509-513).
 * So the oracle follows the HEVC text those tables come from (ITU-T H.265 8.4.4.2.4 - 8.4.4.2.6,
 * nTbS = 32, 8-bit, cIdx = 0: no edge filters apply at this size) and tests/test_intra_tables.py
 * checks its closed forms against the reference's tables where the file is present.  Known
 * slips of the WIP file that are NOT reproduced: `tmp >> 6` instead of >> 5 at :359 (the ifdef'ed
 * twin uses roundN(...,5), :500), DC without the +32 rounding term (:384, :513), and two
 * swapped entries in mapTbl's "Mode 16" row (:90).
 */
#include "x266_oracle.h"

#include <string.h>

/* H.265 Table 8-4: intraPredAngle for predModeIntra 2..34 */
static const int8_t k_angle[33] = {32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26,
                                   -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32};
/* H.265 Table 8-5: invAngle for predModeIntra 11..25 */
static const int16_t k_inv_angle[15] = {-4096, -1638, -910, -630, -482, -390, -315, -256,
                                        -315, -390, -482, -630, -910, -1638, -4096};

int orc_intra_angle(int mode) { return (mode >= 2 && mode <= 34) ? k_angle[mode - 2] : 0; }
int orc_intra_inv_angle(int mode) { return (mode >= 11 && mode <= 25) ? k_inv_angle[mode - 11] : 0; }

/* left[y] = p[-1][y], y = 0..63; top[0] = p[-1][-1], top[1+x] = p[x][-1], x = 0..63.
 * pred[y*32 + x], 8-bit.  Returns 0, or -1 for a mode outside 0..34. */
int orc_intra32_predict(const uint8_t left[64], const uint8_t top[65], int mode, uint8_t pred[1024])
{
    enum { N = 32 };
    if (mode < 0 || mode > 34) return -1;
    if (mode == 0) {                                        /* 8.4.4.2.4 planar */
        const int tr = top[1 + N], bl = left[N];            /* p[N][-1], p[-1][N] */
        for (int y = 0; y < N; ++y)
            for (int x = 0; x < N; ++x)
                pred[y * N + x] = (uint8_t)(((N - 1 - x) * left[y] + (x + 1) * tr + (N - 1 - y) * top[1 + x] + (y + 1) * bl + N) >> 6);
        return 0;
    }
    if (mode == 1) {                                        /* 8.4.4.2.5 DC (no edge smoothing at nTbS = 32) */
        int sum = N;
        for (int i = 0; i < N; ++i) sum += top[1 + i] + left[i];
        memset(pred, sum >> 6, 1024);
        return 0;
    }
    /* 8.4.4.2.6 angular.  ref[] is indexed -N .. 2N; main = the side the mode family reads along. */
    const int angle = k_angle[mode - 2];
    const int vertical = mode >= 18;
    int refbuf[3 * N + 1];
    int *ref = refbuf + N;
    /* main[x] = p[-1+x][-1] (vertical family) or p[-1][-1+x] (horizontal family), x = 0..2N; side likewise swapped */
    for (int x = 0; x <= 2 * N; ++x) ref[x] = vertical ? top[x] : (x == 0 ? top[0] : left[x - 1]);
    if (angle < 0) {
        const int last = (N * angle) >> 5;
        if (last < -1) {
            const int inv = k_inv_angle[mode - 11];
            for (int x = -1; x >= last; --x) {
                const int s = -1 + ((x * inv + 128) >> 8);  /* side coordinate, -1 = corner */
                ref[x] = vertical ? (s < 0 ? top[0] : left[s]) : (s < 0 ? top[0] : top[1 + s]);
            }
        }
    }
    for (int k = 0; k < N; ++k) {                           /* k = y (vertical family) or x (horizontal family) */
        const int idx = ((k + 1) * angle) >> 5, fact = ((k + 1) * angle) & 31;
        for (int j = 0; j < N; ++j) {                       /* j = x (vertical) or y (horizontal) */
            const int a = ref[j + idx + 1];
            const int v = fact ? ((32 - fact) * a + fact * ref[j + idx + 2] + 16) >> 5 : a;
            if (vertical) pred[k * N + j] = (uint8_t)v;
            else          pred[j * N + k] = (uint8_t)v;
        }
    }
    return 0;
}

/* refs: n x 129 bytes (left[64] then top[65]); modes[i] and ref_index[i] (NULL: i) select what
 * output block i holds. */
int orc_intra32_predict_batch(const uint8_t *refs, const uint8_t *modes, const uint32_t *ref_index, uint8_t *pred, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        const uint8_t *r = refs + (size_t)(ref_index ? ref_index[i] : i) * 129;
        if (orc_intra32_predict(r, r + 64, modes[i], pred + i * 1024)) return -1;
    }
    return 0;
}

/* Mode decision (the RTL sketch's "Decide" channel, IntraChannel_t :41-44): for every block the
 * SATD cost of each of the 35 predictions against the source block, cost = sum over the sixteen
 * 8x8 sub-blocks of satd8x8(src - pred) (src_tb/satd.c:31-118 per sub-block).  costs[b*35 + mode];
 * best_mode[b] = the cheapest mode, lowest index on ties (may be NULL). */
int orc_intra32_costs(const uint8_t *refs /* n x 129 */, const uint8_t *src /* n x 1024 */, size_t n,
                      uint32_t *costs, uint8_t *best_mode)
{
    uint8_t pred[1024];
    int16_t diff[64];
    for (size_t b = 0; b < n; ++b) {
        const uint8_t *r = refs + b * 129, *s = src + b * 1024;
        uint32_t best = 0xFFFFFFFFu;
        int best_m = 0;
        for (int m = 0; m < 35; ++m) {
            if (orc_intra32_predict(r, r + 64, m, pred)) return -1;
            uint32_t c = 0;
            for (int sy = 0; sy < 4; ++sy)
                for (int sx = 0; sx < 4; ++sx) {
                    for (int y = 0; y < 8; ++y)
                        for (int x = 0; x < 8; ++x) {
                            const int o = (8 * sy + y) * 32 + 8 * sx + x;
                            diff[y * 8 + x] = (int16_t)((int)s[o] - (int)pred[o]);
                        }
                    c += orc_satd8x8(diff);
                }
            costs[b * 35 + m] = c;
            if (c < best) { best = c; best_m = m; }
        }
        if (best_mode) best_mode[b] = (uint8_t)best_m;
    }
    return 0;
}
