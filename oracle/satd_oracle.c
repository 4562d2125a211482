/*
 * satd_oracle.c -- CPU restatement of the reference's 8x8 Hadamard SATD.
 *
 * TEST INFRASTRUCTURE ONLY (see x266_oracle.h).  Parity: PINNED against
 * oracle/_ref (src_tb/satd.c compiled in place) and tests/golden vectors.
 *
 * Reference: satd8x8(), src_tb/satd.c:31-118 -- three radix-2 butterfly
 * stages with pairing distance 4, 2, 1 along rows (:38-70), the same along
 * columns (:73-103), every intermediate stored to int16 (wraps), sum of
 * absolute values in int32 (:105-111), result (sum + 2) >> 2 (:113).
 * RTL twin: satd_1d / mkSatd8, src/mkSatd.bsv:44-80, 121-159.
 */
#include "x266_oracle.h"

#include <pthread.h>
#include <stdlib.h>

/* One in-place 8-point Hadamard over elements v[0], v[s], ..., v[7*s], with the
 * reference's stage order (distance 4, then 2, then 1) and int16 wraparound. */
static void hadamard8_i16(int16_t *v, int s)
{
    for (int dist = 4; dist >= 1; dist >>= 1) {
        int16_t t[8];
        for (int i = 0; i < 8; i++) {
            /* element i pairs with i^dist; the lower index gets the sum */
            const int lo = i & ~dist, hi = i | dist;
            const int a = v[lo * s], b = v[hi * s];
            t[i] = (int16_t)((i & dist) ? a - b : a + b);
        }
        for (int i = 0; i < 8; i++) v[i * s] = t[i];
    }
}

uint32_t orc_satd8x8(const int16_t diff[64])
{
    int16_t m[64];
    for (int i = 0; i < 64; i++) m[i] = diff[i];
    for (int r = 0; r < 8; r++) hadamard8_i16(m + 8 * r, 1);   /* horizontal */
    for (int c = 0; c < 8; c++) hadamard8_i16(m + c, 8);       /* vertical   */
    int32_t sum = 0;
    for (int i = 0; i < 64; i++) sum += abs((int)m[i]);
    return (uint32_t)((sum + 2) >> 2);
}

void orc_satd8x8_batch(const int16_t *diff, uint32_t *out, size_t n_blocks)
{
    for (size_t b = 0; b < n_blocks; b++) out[b] = orc_satd8x8(diff + 64 * b);
}

typedef struct { const int16_t *in; uint32_t *out; size_t n; } satd_job_t;

static void *satd_worker(void *p)
{
    satd_job_t *j = (satd_job_t *)p;
    orc_satd8x8_batch(j->in, j->out, j->n);
    return NULL;
}

void orc_satd8x8_batch_mt(const int16_t *diff, uint32_t *out, size_t n, int threads)
{
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    satd_job_t *job = (satd_job_t *)malloc(sizeof(satd_job_t) * threads);
    size_t done = 0;
    for (int t = 0; t < threads; t++) {
        size_t cnt = n / threads + ((size_t)t < n % threads ? 1 : 0);
        job[t].in = diff + done * 64; job[t].out = out + done; job[t].n = cnt;
        done += cnt;
        pthread_create(&tid[t], NULL, satd_worker, &job[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    free(tid); free(job);
}

/* ------------------------------------------------------------------------- */
/* Full-search harness around satd8x8 (BASELINE configs[2]).  The per-candidate */
/* cost is the reference's satd8x8 on the 9-bit difference block; the harness   */
/* itself (raster candidate order dy-major, first minimum wins, reference frame */
/* padded by `range`) is this repository's definition -- UNPINNED upstream, see */
/* include/x266hip.h: xSatd8x8SearchDev.                                        */
/* ------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *cur, *ref;
    ptrdiff_t cs, rs;
    int w, h, range, by0, by1;
    int16_t *mv;
    uint32_t *cost, *costs;
    int metric;                       /* 0 = satd8x8, 1 = sum of absolute differences (sad.c:28-39 on the 8x8 block) */
} me_job_t;

static void *me_worker(void *p)
{
    me_job_t *j = (me_job_t *)p;
    const int bxn = j->w / 8, span = 2 * j->range + 1;
    for (int by = j->by0; by < j->by1; by++)
        for (int bx = 0; bx < bxn; bx++) {
            uint32_t best = 0xFFFFFFFFu;
            int bdx = 0, bdy = 0;
            for (int dy = -j->range; dy <= j->range; dy++)
                for (int dx = -j->range; dx <= j->range; dx++) {
                    int16_t d[64];
                    for (int y = 0; y < 8; y++)
                        for (int x = 0; x < 8; x++)
                            d[8 * y + x] = (int16_t)((int)j->cur[(by * 8 + y) * j->cs + bx * 8 + x] -
                                                     (int)j->ref[(by * 8 + y + dy) * j->rs + bx * 8 + x + dx]);
                    uint32_t c;
                    if (j->metric == 0) {
                        c = orc_satd8x8(d);
                    } else {
                        c = 0;
                        for (int k = 0; k < 64; k++) c += (uint32_t)(d[k] < 0 ? -d[k] : d[k]);
                    }
                    if (j->costs)
                        j->costs[((size_t)by * bxn + bx) * (size_t)(span * span) + (size_t)(dy + j->range) * span + (dx + j->range)] = c;
                    if (c < best) { best = c; bdx = dx; bdy = dy; }
                }
            j->mv[2 * (by * bxn + bx)] = (int16_t)bdx;
            j->mv[2 * (by * bxn + bx) + 1] = (int16_t)bdy;
            j->cost[by * bxn + bx] = best;
        }
    return NULL;
}

static void me_search(int metric, const uint8_t *cur, ptrdiff_t cur_stride, const uint8_t *ref, ptrdiff_t ref_stride,
                      int width, int height, int range, int16_t *best_mv, uint32_t *best_cost,
                      uint32_t *costs, int threads)
{
    const int byn = height / 8;
    if (threads < 1) threads = 1;
    if (threads > byn) threads = byn;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    me_job_t *job = (me_job_t *)malloc(sizeof(me_job_t) * threads);
    int done = 0;
    for (int t = 0; t < threads; t++) {
        const int cnt = byn / threads + (t < byn % threads ? 1 : 0);
        job[t] = (me_job_t){cur, ref, cur_stride, ref_stride, width, height, range, done, done + cnt, best_mv, best_cost, costs, metric};
        done += cnt;
        pthread_create(&tid[t], NULL, me_worker, &job[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    free(tid); free(job);
}

void orc_satd8x8_search(const uint8_t *cur, ptrdiff_t cur_stride, const uint8_t *ref, ptrdiff_t ref_stride,
                        int width, int height, int range, int16_t *best_mv, uint32_t *best_cost,
                        uint32_t *costs, int threads)
{
    me_search(0, cur, cur_stride, ref, ref_stride, width, height, range, best_mv, best_cost, costs, threads);
}

/* Same harness, cheaper metric: cost = sum |cur - ref| over the 8x8 block, i.e. sad() of
 * riscv/programs/benchmarks/sad/sad.c:28-39 at n = 8. */
void orc_sad8x8_search(const uint8_t *cur, ptrdiff_t cur_stride, const uint8_t *ref, ptrdiff_t ref_stride,
                       int width, int height, int range, int16_t *best_mv, uint32_t *best_cost,
                       uint32_t *costs, int threads)
{
    me_search(1, cur, cur_stride, ref, ref_stride, width, height, range, best_mv, best_cost, costs, threads);
}
