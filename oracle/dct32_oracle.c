/*
 * dct32_oracle.c -- CPU restatement of the reference's 32x32 integer DCT-II.
 *
 * TEST INFRASTRUCTURE ONLY (see x266_oracle.h).  Plain scalar C that keeps the
 * reference's algorithm (even/odd partial butterfly, int32 accumulate,
 * round-half-up shift, truncating int16 store, transposed write), so that it
 * can double as the "reference C path" CPU baseline in bench.py.
 *
 * Parity: PINNED for the forward path -- tests/test_oracle_vs_ref.py compares
 * it bit-for-bit with oracle/_ref (the real src_tb/dct32.c compiled in place)
 * and tests/test_oracle_golden.py with the committed tests/golden vectors.
 * The inverse is UNPINNED (no inverse exists anywhere in the reference).
 */
#include "x266_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* ------------------------------------------------------------------------- */
/* Coefficient table.  Reference: const int16_t g_t32[32][32],               */
/* src_tb/dct32.c:30-64 (half-table twin: src/mkDct32.bsv:39-73).            */
/* The 32x32 matrix is round(64*sqrt(2)*cos((2n+1)k*pi/64)) hand-tuned by    */
/* the HEVC/VVC standard; all 1024 entries are +/- one of the 32 magnitudes   */
/* of column 0, selected by folding the angle index (2n+1)k mod 128 into the  */
/* first quadrant.                                                            */
/* ------------------------------------------------------------------------- */
static const int16_t k_col0[33] = {
    64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67,
    64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0
};

static int16_t g_tab[32 * 32];
static pthread_once_t g_tab_once = PTHREAD_ONCE_INIT;

static void build_table(void)
{
    for (int k = 0; k < 32; k++) {
        for (int n = 0; n < 32; n++) {
            int v;
            if (k == 0) {
                v = k_col0[0];
            } else {
                int a = ((2 * n + 1) * k) & 127;   /* angle in units of pi/64 */
                if (a <= 32)       v =  k_col0[a];
                else if (a <= 64)  v = -k_col0[64 - a];
                else if (a <= 96)  v = -k_col0[a - 64];
                else               v =  k_col0[128 - a];
            }
            g_tab[k * 32 + n] = (int16_t)v;
        }
    }
}

const int16_t *orc_dct32_table(void)
{
    pthread_once(&g_tab_once, build_table);
    return g_tab;
}

/* ------------------------------------------------------------------------- */
/* Forward 1-D pass.  Reference: partialButterfly32(), src_tb/dct32.c:66-170. */
/* The reference unrolls four fold levels by hand (E/O :78-82, EE/EO :109-113,*/
/* EEE/EEO :116-120, EEEE/EEEO :123-126) and then writes the outputs level by */
/* level (:128-152).  Here the same decomposition is one loop: at each level  */
/* the current vector of length `len` is folded into a symmetric half (kept)  */
/* and an antisymmetric half, which yields the rows k = step, 3*step, ...     */
/* ------------------------------------------------------------------------- */
void orc_dct32_pass(const int16_t *src, int16_t *dst, int shift, int line)
{
    const int16_t *g = orc_dct32_table();
    const int rnd = 1 << (shift - 1);

    for (int j = 0; j < line; j++) {
        int cur[32], sym[16], asym[16];
        for (int n = 0; n < 32; n++) cur[n] = src[32 * j + n];

        int len = 32, step = 1;
        while (len > 2) {
            const int half = len >> 1;
            for (int i = 0; i < half; i++) {
                sym[i]  = cur[i] + cur[len - 1 - i];
                asym[i] = cur[i] - cur[len - 1 - i];
            }
            for (int k = step; k < 32; k += 2 * step) {
                int acc = rnd;
                for (int i = 0; i < half; i++) acc += g[k * 32 + i] * asym[i];
                dst[k * line + j] = (int16_t)(acc >> shift);
            }
            memcpy(cur, sym, sizeof(int) * half);
            len = half;
            step <<= 1;
        }
        /* len == 2: rows 0 and 16 (dct32.c:128-129) */
        dst[0 * line + j]  = (int16_t)((g[0 * 32 + 0]  * cur[0] + g[0 * 32 + 1]  * cur[1] + rnd) >> shift);
        dst[16 * line + j] = (int16_t)((g[16 * 32 + 0] * cur[0] + g[16 * 32 + 1] * cur[1] + rnd) >> shift);
    }
}

/* Dense form of the same contract (bit-identical). */
void orc_dct32_pass_dense(const int16_t *src, int16_t *dst, int shift, int line)
{
    const int16_t *g = orc_dct32_table();
    const int rnd = 1 << (shift - 1);
    for (int j = 0; j < line; j++)
        for (int k = 0; k < 32; k++) {
            int acc = rnd;
            for (int n = 0; n < 32; n++) acc += g[k * 32 + n] * src[32 * j + n];
            dst[k * line + j] = (int16_t)(acc >> shift);
        }
}

/* 2-D forward.  Reference: dct32_genNew(), src_tb/dct32.c:180-181,197-198:
 * rows with shift 4 into a transposed temporary, then shift 11. */
void orc_dct32_fwd(const int16_t *in, int16_t *out, size_t n_blocks)
{
    int16_t tmp[1024];
    for (size_t b = 0; b < n_blocks; b++) {
        orc_dct32_pass(in + b * 1024, tmp, 4, 32);
        orc_dct32_pass(tmp, out + b * 1024, 11, 32);
    }
}

/* ------------------------------------------------------------------------- */
/* Inverse (UNPINNED -- this repository's definition). */
/*   pass(src,dst,shift): dst[j*32+n] = clip16((sum_k g[k][n]*src[k*32+j]     */
/*                                              + (1<<(shift-1))) >> shift)   */
/*   inverse 2-D = pass(shift 7) then pass(shift 12)                          */
/* i.e. the standard HEVC/VVC inverse for 8-bit video.  Computed with the     */
/* even/odd recomposition that mirrors the forward fold.                      */
/* ------------------------------------------------------------------------- */
static inline int16_t clip16(int v)
{
    return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
}

/* out[n], n < len: sum over the rows k = 0, step, 2*step, ... (len of them). */
static void inv_recompose(const int16_t *g, const int *coef /*[32], indexed by k*/,
                          int len, int step, int *out)
{
    if (len == 2) {
        out[0] = g[0 * 32 + 0] * coef[0] + g[16 * 32 + 0] * coef[16];
        out[1] = g[0 * 32 + 1] * coef[0] + g[16 * 32 + 1] * coef[16];
        return;
    }
    const int half = len >> 1;
    int ev[16], od[16];
    inv_recompose(g, coef, half, step * 2, ev);
    for (int i = 0; i < half; i++) {
        int acc = 0;
        for (int k = step; k < 32; k += 2 * step) acc += g[k * 32 + i] * coef[k];
        od[i] = acc;
    }
    for (int i = 0; i < half; i++) {
        out[i]           = ev[i] + od[i];
        out[len - 1 - i] = ev[i] - od[i];
    }
}

static void inv_pass(const int16_t *src, int16_t *dst, int shift)
{
    const int16_t *g = orc_dct32_table();
    const int rnd = 1 << (shift - 1);
    for (int j = 0; j < 32; j++) {
        int coef[32], res[32];
        for (int k = 0; k < 32; k++) coef[k] = src[k * 32 + j];
        inv_recompose(g, coef, 32, 1, res);
        for (int n = 0; n < 32; n++) dst[j * 32 + n] = clip16((res[n] + rnd) >> shift);
    }
}

void orc_dct32_inv(const int16_t *in, int16_t *out, size_t n_blocks)
{
    int16_t tmp[1024];
    for (size_t b = 0; b < n_blocks; b++) {
        inv_pass(in + b * 1024, tmp, 7);
        inv_pass(tmp, out + b * 1024, 12);
    }
}

/* ------------------------------------------------------------------------- */
/* BDPI word packing.  Reference: dct32_getDiff() src_tb/dct32.c:205-220 and  */
/* dct32_getDct() :223-246.                                                   */
/* ------------------------------------------------------------------------- */
void orc_pack_diff_rows(const int16_t *mat, int first_row, uint32_t res[32])
{
    for (int w = 0; w < 32; w++) {
        const int r = first_row + (w >> 4), c = (w & 15) * 2;
        const uint32_t lo = (uint16_t)mat[r * 32 + c], hi = (uint16_t)mat[r * 32 + c + 1];
        res[w] = (hi << 16) + lo;
    }
}

uint64_t orc_pack_dct_word(const int16_t *dct, int idx)
{
    const int col = idx >> 5, row = idx & 31;
    uint64_t w = 0;
    for (int i = 0; i < 4; i++)
        w |= (uint64_t)(uint16_t)dct[(row + i) * 32 + col] << (16 * i);
    return w;
}

/* ------------------------------------------------------------------------- */
/* Threaded drivers (contiguous shard per thread).     */
/* ------------------------------------------------------------------------- */
typedef struct {
    void (*fn)(const int16_t *, int16_t *, size_t);
    const int16_t *in;
    int16_t *out;
    size_t n;
} dct_job_t;

static void *dct_worker(void *p)
{
    dct_job_t *j = (dct_job_t *)p;
    j->fn(j->in, j->out, j->n);
    return NULL;
}

static void run_mt(void (*fn)(const int16_t *, int16_t *, size_t),
                   const int16_t *in, int16_t *out, size_t n, int threads)
{
    if (threads < 1) threads = 1;
    if ((size_t)threads > n) threads = n ? (int)n : 1;
    pthread_t *tid = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    dct_job_t *job = (dct_job_t *)malloc(sizeof(dct_job_t) * threads);
    size_t done = 0;
    for (int t = 0; t < threads; t++) {
        size_t cnt = n / threads + ((size_t)t < n % threads ? 1 : 0);
        job[t].fn = fn; job[t].in = in + done * 1024; job[t].out = out + done * 1024; job[t].n = cnt;
        done += cnt;
        pthread_create(&tid[t], NULL, dct_worker, &job[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
    free(tid); free(job);
}

void orc_dct32_fwd_mt(const int16_t *in, int16_t *out, size_t n, int threads) { run_mt(orc_dct32_fwd, in, out, n, threads); }
void orc_dct32_inv_mt(const int16_t *in, int16_t *out, size_t n, int threads) { run_mt(orc_dct32_inv, in, out, n, threads); }

int orc_hw_threads(void)
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (int)n : 1;
}
