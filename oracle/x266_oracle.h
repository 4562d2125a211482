/*
 * x266_oracle.h -- CPU restatement of the x266 DCT32 / SATD hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.  The
 * shipped path (libx266hip.so) never links or calls it.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference).  Parity status is stated per function:
 *   PINNED    -- checked bit-for-bit against the real reference objects
 *                (oracle/_ref, built from /root/reference/src_tb/{dct32,satd}.c)
 *                and against tests/golden/ vectors generated from them.
 *   UNPINNED  -- the reference holds no implementation; semantics are this
 *                repository's own definition (documented in DESIGN.md).
 */
#ifndef X266_ORACLE_H
#define X266_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- coefficient table (src_tb/dct32.c:30-64) ---------------------- PINNED */
/* Row k = frequency, column n = sample.  Regenerated from the 32 magnitudes of
 * the first column by cosine index folding, not pasted. */
const int16_t *orc_dct32_table(void);            /* 32*32 int16, row-major   */

/* ---- 1-D pass with transposed store (src_tb/dct32.c:66-170) -------- PINNED */
/* dst[k*line + j] = (int16)((sum_n g[k][n]*src[32*j+n] + (1<<(shift-1))) >> shift)
 * computed with the reference's even/odd partial-butterfly decomposition. */
void orc_dct32_pass(const int16_t *src, int16_t *dst, int shift, int line);
/* Same contract, computed as the dense 32-tap contraction (cross-check). */
void orc_dct32_pass_dense(const int16_t *src, int16_t *dst, int shift, int line);

/* ---- 2-D forward transform (src_tb/dct32.c:178-198) ---------------- PINNED */
/* n blocks of 32x32 int16 row-major, blocks contiguous; shifts 4 then 11.
 * out[v*32+u]: v = vertical frequency, u = horizontal frequency. */
void orc_dct32_fwd(const int16_t *in, int16_t *out, size_t n_blocks);
void orc_dct32_fwd_mt(const int16_t *in, int16_t *out, size_t n_blocks, int threads);

/* ---- 2-D inverse transform ---------------------------------------- UNPINNED */
/* HEVC/VVC-style inverse of the above for 8-bit video: column pass shift 7,
 * row pass shift 12, each output clipped to int16. */
void orc_dct32_inv(const int16_t *in, int16_t *out, size_t n_blocks);
void orc_dct32_inv_mt(const int16_t *in, int16_t *out, size_t n_blocks, int threads);

/* ---- 8x8 Hadamard SATD (src_tb/satd.c:31-118) ---------------------- PINNED */
uint32_t orc_satd8x8(const int16_t diff[64]);
void orc_satd8x8_batch(const int16_t *diff, uint32_t *out, size_t n_blocks);
void orc_satd8x8_batch_mt(const int16_t *diff, uint32_t *out, size_t n_blocks, int threads);

/* ---- mixed transform set (BASELINE configs[3]): DCT-II + closed-form DST-VII -------- (DCT-II,32) PINNED, rest UNPINNED */
#define ORC_TR_DCT2 0
#define ORC_TR_DST7 1
int orc_transform_matrix(int type, int n, int16_t *m /* n*n, row = frequency */);
int orc_transform_fwd(int type, int n, const int16_t *in, int16_t *out, size_t n_blocks);
int orc_transform_inv(int type, int n, const int16_t *in, int16_t *out, size_t n_blocks);   /* UNPINNED */
/* the same passes with caller-supplied matrices (mh along rows, mv vertically; n*n int16, row k = basis function) */
int orc_transform_fwd_matrix(const int16_t *mh, const int16_t *mv, int n, const int16_t *in, int16_t *out, size_t n_blocks);
int orc_transform_inv_matrix(const int16_t *mh, const int16_t *mv, int n, const int16_t *in, int16_t *out, size_t n_blocks);

/* ---- full-search harness around satd8x8 (BASELINE configs[2]) ---- cost PINNED, harness UNPINNED */
/* ref points at pixel (0,0) of a frame padded by >= range; candidates in raster
 * order (dy-major), first minimum wins; best_mv[2*b] = dx, [2*b+1] = dy. */
void orc_satd8x8_search(const uint8_t *cur, ptrdiff_t cur_stride, const uint8_t *ref, ptrdiff_t ref_stride,
                        int width, int height, int range, int16_t *best_mv, uint32_t *best_cost,
                        uint32_t *costs /* NULL or [blocks][(2R+1)^2] */, int threads);

/* same harness with the cheaper metric of SURVEY 8 f3: cost = sum |cur - ref| (sad.c:28-39 at n = 8) */
void orc_sad8x8_search(const uint8_t *cur, ptrdiff_t cur_stride, const uint8_t *ref, ptrdiff_t ref_stride,
                       int width, int height, int range, int16_t *best_mv, uint32_t *best_cost,
                       uint32_t *costs, int threads);

/* ---- frame container (src/x266.cpp:56-63, 415-492) -------- restated, not executed; residual UNPINNED */
void orc_conv_input_fmt(uint8_t *tiles, const uint8_t *y, const uint8_t *u, const uint8_t *v,
                        ptrdiff_t strd_y, int width, int height);
void orc_conv_output_420(const uint8_t *tiles, uint8_t *y, ptrdiff_t strd_y, uint8_t *u, uint8_t *v,
                         ptrdiff_t strd_c, int width, int height);
void orc_residual_luma(const uint8_t *cur_tiles, const uint8_t *pred_tiles, int width, int height, int edge,
                       int16_t *res);
void orc_residual_chroma(const uint8_t *cur_tiles, const uint8_t *pred_tiles, int width, int height, int edge,
                         int16_t *res_u, int16_t *res_v, size_t block_pitch);

/* ---- 32x32 intra prediction (src/mkIntra32-wip.bsv is a WIP sketch) ---- UNPINNED */
/* HEVC 35-mode scheme, nTbS = 32 (H.265 8.4.4.2.4-6): mode 0 planar, 1 DC, 2..34 angular, references
 * used as given.  left[y] = p[-1][y]; top[0] = corner, top[1+x] = p[x][-1] (IntraRef_t, :36-39). */
int orc_intra_angle(int mode);
int orc_intra_inv_angle(int mode);
int orc_intra32_predict(const uint8_t left[64], const uint8_t top[65], int mode, uint8_t pred[1024]);
int orc_intra32_predict_batch(const uint8_t *refs /* n_refs x 129 */, const uint8_t *modes, const uint32_t *ref_index /* or NULL */,
                              uint8_t *pred /* n x 1024 */, size_t n);

/* mode decision: costs[b*35 + m] = sum of satd8x8 over the 16 sub-blocks of (src - prediction m) */
int orc_intra32_costs(const uint8_t *refs /* n x 129 */, const uint8_t *src /* n x 1024 */, size_t n,
                      uint32_t *costs /* n x 35 */, uint8_t *best_mode /* n or NULL */);

/* ---- BDPI word packing (src_tb/dct32.c:205-246, satd.c:143-147) ---- PINNED */
void     orc_pack_diff_rows(const int16_t *mat, int first_row, uint32_t res[32]);
uint64_t orc_pack_dct_word(const int16_t *dct, int idx);

/* ---- synthetic residual stream (distribution of dct32.c:191-193) ---------- */
/* Counter-based SplitMix64: sample i of stream `seed` = a - b with a,b the two
 * low bytes of mix(seed + (i+1)*0x9E3779B97F4A7C15).  Same generator as the
 * device-side xFillResidual kernel. */
void orc_fill_residual(int16_t *dst, size_t n_samples, uint64_t seed, uint64_t first_index);

/* ---- helpers --------------------------------------------------------------- */
uint64_t orc_checksum64(const void *data, size_t n_bytes);  /* FNV-1a over 8-byte words, order dependent */
uint64_t orc_sum_u16(const int16_t *data, size_t n);        /* order independent: sum of (uint16) values */
int      orc_hw_threads(void);

#ifdef __cplusplus
}
#endif
#endif
