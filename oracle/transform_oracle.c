/*
 * transform_oracle.c -- CPU statement of the mixed transform set of BASELINE
 * configs[3]: forward 2-D DCT-II N = 4, 8, 16, 32 and closed-form DST-VII N = 4, 8, 16
 * (N = 4 is the H.266 table; N = 8 / 16 are NOT claimed to be the standard's integers),
 * plus the same passes with caller-supplied matrices.
 *
 * TEST INFRASTRUCTURE ONLY (see x266_oracle.h).
 * Parity: (DCT-II, 32) is PINNED -- it must equal orc_dct32_fwd, which is pinned to
 * the real src_tb/dct32.c.  Everything else is UNPINNED: upstream has no C model
 * for other sizes and no DST-VII at all.  What upstream does fix is
 *   - the N-point DCT-II matrix = rows 0, 32/N, 2*32/N ... of g_t32, first N columns
 *     (the RTL taps them at src/mkDct32.bsv:132-141), and
 *   - the pass structure (src_tb/dct32.c:66-170): dst[k*N + j] =
 *     (int16)((sum_n M[k][n]*src[j*N + n] + (1 << (shift-1))) >> shift), twice,
 *     with shifts log2N - 1 and log2N + 6 (4 and 11 for N = 32, dct32.c:180-181).
 * DST-VII: M[k][n] = round(64*sqrt(N)*sqrt(4/(2N+1))*sin(pi(2k+1)(n+1)/(2N+1))); the N = 4
 * instance is the VVC table.  Literal tables below; tests/test_oracle_props.py recomputes
 * them from the closed form.
 */
#include "x266_oracle.h"

static const signed char k_dst7_4[16] = {
     29,  55,  74,  84,
     74,  74,   0, -74,
     84, -29, -74,  55,
     55, -84,  74, -29,
};
static const signed char k_dst7_8[64] = {
     16,  32,  46,  59,  70,  79,  84,  87,
     46,  79,  87,  70,  32, -16, -59, -84,
     70,  84,  32, -46, -87, -59,  16,  79,
     84,  46, -59, -79,  16,  87,  32, -70,
     87, -16, -84,  32,  79, -46, -70,  59,
     79, -70, -16,  84, -59, -32,  87, -46,
     59, -87,  70, -16, -46,  84, -79,  32,
     32, -59,  79, -87,  84, -70,  46, -16,
};
static const signed char k_dst7_16[256] = {
      8,  17,  25,  33,  41,  48,  55,  62,  67,  73,  77,  81,  84,  87,  88,  89,
     25,  48,  67,  81,  88,  88,  81,  67,  48,  25,   0, -25, -48, -67, -81, -88,
     41,  73,  88,  84,  62,  25, -17, -55, -81, -89, -77, -48,  -8,  33,  67,  87,
     55,  87,  81,  41, -17, -67, -89, -73, -25,  33,  77,  88,  62,   8, -48, -84,
     67,  88,  48, -25, -81, -81, -25,  48,  88,  67,   0, -67, -88, -48,  25,  81,
     77,  77,   0, -77, -77,   0,  77,  77,   0, -77, -77,   0,  77,  77,   0, -77,
     84,  55, -48, -87,  -8,  81,  62, -41, -88, -17,  77,  67, -33, -89, -25,  73,
     88,  25, -81, -48,  67,  67, -48, -81,  25,  88,   0, -88, -25,  81,  48, -67,
     89,  -8, -88,  17,  87, -25, -84,  33,  81, -41, -77,  48,  73, -55, -67,  62,
     87, -41, -67,  73,  33, -88,   8,  84, -48, -62,  77,  25, -89,  17,  81, -55,
     81, -67, -25,  88, -48, -48,  88, -25, -67,  81,   0, -81,  67,  25, -88,  48,
     73, -84,  25,  55, -89,  48,  33, -87,  67,   8, -77,  81, -17, -62,  88, -41,
     62, -89,  67,  -8, -55,  88, -73,  17,  48, -87,  77, -25, -41,  84, -81,  33,
     48, -81,  88, -67,  25,  25, -67,  88, -81,  48,   0, -48,  81, -88,  67, -25,
     33, -62,  81, -89,  84, -67,  41,  -8, -25,  55, -77,  88, -87,  73, -48,  17,
     17, -33,  48, -62,  73, -81,  87, -89,  88, -84,  77, -67,  55, -41,  25,  -8,
};

int orc_transform_matrix(int type, int n, int16_t *m /* n*n */)
{
    if (!(n == 4 || n == 8 || n == 16 || (n == 32 && type == ORC_TR_DCT2))) return -1;
    const int16_t *g = orc_dct32_table();
    for (int k = 0; k < n; k++)
        for (int c = 0; c < n; c++) {
            if (type == ORC_TR_DCT2) m[k * n + c] = g[(k * (32 / n)) * 32 + c];
            else if (type == ORC_TR_DST7) m[k * n + c] = n == 4 ? k_dst7_4[k * 4 + c] : (n == 8 ? k_dst7_8[k * 8 + c] : k_dst7_16[k * 16 + c]);
            else return -1;
        }
    return 0;
}

static void tr_pass(const int16_t *m, int n, const int16_t *src, int16_t *dst, int shift)
{
    const int rnd = 1 << (shift - 1);
    for (int j = 0; j < n; j++)
        for (int k = 0; k < n; k++) {
            int acc = rnd;
            for (int c = 0; c < n; c++) acc += m[k * n + c] * src[j * n + c];
            dst[k * n + j] = (int16_t)(acc >> shift);
        }
}

/* type codes of the API: 0 DCT-II both ways, 1 DST-VII both ways, 2 DST-VII along rows (horizontal) + DCT-II
 * vertically, 3 the other way round.  Pass 1 of the forward transform runs along rows. */
static int htype_of(int type) { return (type == 1 || type == 2) ? ORC_TR_DST7 : ORC_TR_DCT2; }
static int vtype_of(int type) { return (type == 1 || type == 3) ? ORC_TR_DST7 : ORC_TR_DCT2; }

/* The same two passes with CALLER-SUPPLIED matrices (mh along rows, mv vertically; row k = basis function): what
 * xTransformSetMatrix installs in the product.  The pass structure is the pinned one (src_tb/dct32.c:66-170); the
 * matrices are whatever the caller says they are. */
int orc_transform_fwd_matrix(const int16_t *mh, const int16_t *mv, int n, const int16_t *in, int16_t *out, size_t n_blocks)
{
    int16_t tmp[32 * 32];
    if (!(n == 4 || n == 8 || n == 16 || n == 32)) return -1;
    int log2n = 0;
    while ((1 << log2n) < n) log2n++;
    for (size_t b = 0; b < n_blocks; b++) {
        tr_pass(mh, n, in + b * n * n, tmp, log2n - 1);
        tr_pass(mv, n, tmp, out + b * n * n, log2n + 6);
    }
    return 0;
}

int orc_transform_fwd(int type, int n, const int16_t *in, int16_t *out, size_t n_blocks)
{
    int16_t m[32 * 32], mv[32 * 32];
    if (type < 0 || type > 3 || orc_transform_matrix(htype_of(type), n, m) || orc_transform_matrix(vtype_of(type), n, mv)) return -1;
    return orc_transform_fwd_matrix(m, mv, n, in, out, n_blocks);
}

/* Inverse of the above (UNPINNED): columns first, dst[j*n + c] = clip16((sum_k M[k][c]*src[k*n + j]
 * + (1 << (shift-1))) >> shift), shifts 7 then 12 (8-bit video; size independent as in HEVC/VVC).
 * For (DCT-II, 32) this is orc_dct32_inv. */
static int16_t tr_clip16(int v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

static void tr_inv_pass(const int16_t *m, int n, const int16_t *src, int16_t *dst, int shift)
{
    const int rnd = 1 << (shift - 1);
    for (int j = 0; j < n; j++)
        for (int c = 0; c < n; c++) {
            int acc = rnd;
            for (int k = 0; k < n; k++) acc += m[k * n + c] * src[k * n + j];
            dst[j * n + c] = tr_clip16(acc >> shift);
        }
}

int orc_transform_inv_matrix(const int16_t *mh, const int16_t *mv, int n, const int16_t *in, int16_t *out, size_t n_blocks)
{
    int16_t tmp[32 * 32];
    if (!(n == 4 || n == 8 || n == 16 || n == 32)) return -1;
    for (size_t b = 0; b < n_blocks; b++) {
        tr_inv_pass(mv, n, in + b * n * n, tmp, 7);                   /* columns first: the vertical inverse */
        tr_inv_pass(mh, n, tmp, out + b * n * n, 12);
    }
    return 0;
}

int orc_transform_inv(int type, int n, const int16_t *in, int16_t *out, size_t n_blocks)
{
    int16_t m[32 * 32], mv[32 * 32];
    if (type < 0 || type > 3 || orc_transform_matrix(htype_of(type), n, m) || orc_transform_matrix(vtype_of(type), n, mv)) return -1;
    return orc_transform_inv_matrix(m, mv, n, in, out, n_blocks);
}
