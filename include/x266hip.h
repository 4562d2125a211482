/*
 * x266hip.h -- C ABI of libx266hip.so: the MI355X (gfx950) implementation of
 * x266's block-transform / motion-cost hot path.
 *
 * Plain C, plain pointers and sizes; no HIP, torch or C++ types cross this
 * boundary (a stream is passed as an opaque `void *` that holds a hipStream_t).
 * Reference paths below are relative to the upstream x266 tree.
 *
 * Four groups of entry points:
 *
 *  1. The six BDPI symbols the Bluespec testbenches import
 *     (src/mkDct32.bsv:409-411, src/mkSatd.bsv:204-206) and that upstream
 *     defines in src_tb/dct32.c:178-246 and src_tb/satd.c:124-152.  Signatures,
 *     packing, call order and statefulness are identical, so this library can
 *     replace src_tb/{dct32,satd}.c on the bsc link line that names every src_tb C file
 *     link line (build/Makefile:65-67); the golden values the DUT is compared
 *     with are then computed by the GPU kernels.
 *
 *  2. Batch entry points (no upstream counterpart -- upstream is one block at
 *     a time).  Conventions follow src/x266.cpp: `x` prefix, context first,
 *     caller-allocated buffers, int 0 / negative return (x266.cpp:494-513).
 *     Block layout is the reference's: row-major int16, blocks contiguous
 *     (32x32 = 2048 B per DCT block, 8x8 = 128 B per SATD block).
 *
 *  3. Small host utilities (word packing, device memory, streams, events,
 *     graphs) so that a pure-C host can drive the device-pointer API without
 *     HIP headers.
 *
 *  4. One node, several GPUs (BASELINE configs[4]): shards of a batch, a
 *     pipelined frame stream and a striped motion search over RCCL send/recv
 *     groups.  A node and its streams are driven by one host thread at a time.
 *
 * The library NEVER falls back to a CPU implementation: without a usable
 * gfx950 device every compute entry point fails (negative return; the BDPI
 * shims print to stderr and abort()).
 */
#ifndef X266HIP_H
#define X266HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* return codes (x266.cpp style: 0 ok, negative failure)                     */
/* ------------------------------------------------------------------------ */
#define X266HIP_OK        0
#define X266HIP_EINVAL   (-1)   /* bad argument (NULL pointer with n > 0, ...)  */
#define X266HIP_EDEVICE  (-2)   /* no gfx950 device / HIP runtime error         */
#define X266HIP_ENOMEM   (-3)   /* device or host allocation failed             */

typedef struct x266hip_ctx x266hip_ctx;     /* opaque, one per (thread, device) */

/* ------------------------------------------------------------------------ */
/* context (cf. xCodecInit / xCodecFree, src/x266.cpp:494-524)               */
/* ------------------------------------------------------------------------ */
/* Creates a context on HIP device `device_id`, uploads the MFMA operand images
 * of the coefficient matrix (g_t32, src_tb/dct32.c:30-64).  Fails with
 * X266HIP_EDEVICE when the device is not a gfx950 part. */
int  xHipCodecInit(x266hip_ctx **ctx, int device_id);
/* Number of HIP devices visible to the process (0 when there is none or the runtime fails). */
int  xHipDeviceCount(void);
void xHipCodecFree(x266hip_ctx *ctx);
/* Last error text of this context (never NULL; "" when none). */
const char *xHipLastError(const x266hip_ctx *ctx);
/* Device facts for reports: name, CU count, max engine clock (MHz), HBM bytes. */
int  xHipDeviceInfo(const x266hip_ctx *ctx, char *name, size_t name_cap,
                    int *cu_count, int *clock_mhz, size_t *hbm_bytes);
/* Launch options, for A/B measurement (defaults are the measured optimum; results never depend on them; unknown keys and
 * out-of-range values return X266HIP_EINVAL).  The complete list (the kOptions table of x266_amd/csrc/x266hip_abi.hip):
 *   "dct32_variant"          0 matrix-core kernel, 2 the reference's even/odd butterfly on the vector ALU
 *   "satd_variant"           0 by batch size (staged kernel below 3 Mi blocks, LDS-DMA kernel from there on), 1 staged kernel,
 *                            2 radix-2 butterflies on the vector ALU, 3 LDS-DMA kernel
 *                            (2 = the one comparison variant per kernel family north_star asks for)
 *   "dct32_blocks_per_wave", "dct32_inv_blocks_per_wave", "dct32_fwdinv_blocks_per_wave" (1 / 2 / 0 = automatic: 2, or 8 when d_coef is NULL)
 *                            consecutive blocks (tiles, for the transform set's inverses) one wave loops over
 *   "dct32_wg_threads"       workgroup size of the DCT32 / transform-set kernels (64, 128, 192, 256; default 0 = the measured
 *                            best: 64, and 256 for the fused forward + inverse kernel)
 *   "satd_groups_per_wave", "satd_wg_threads", "satd_lds_bytes_per_wave"
 *                            SATD batch: 32-block groups per wave, workgroup size, LDS charged per wave (= cap on resident
 *                            waves per CU); 0 (default) = the chosen kernel's own: 2 / 128 / 6144 staged, 4 / 256 / 16384 LDS-DMA.
 *                            "satd_lds_bytes_per_wave" takes 0 .. 16384 in multiples of 16 (whole 16-byte rows; 16 KiB x the four waves of
 *                            the largest workgroup = the 64 KiB a launch may ask for; values below the kernel's own need are raised to it)
 *   "tile_tiles_per_wave"    xTransformTilesDev: consecutive tiles per wave (0 = 2)
 *   "adaptive_per_wave"      shrink the per-wave run on small batches (default 1)
 *   "me_tile_rows"           motion-search tile height in block rows (0, the default: chosen from the frame size and the CU count)
 *   "autotune"               0 (default) / 1: boxes differ in which launch shape a few kernels run fastest in (by up to 5 %).  With 1, the FIRST
 *                            large call of a family -- xDct32FwdInvBatchDev with and without d_coef (>= 2^18 blocks), xSatd8x8BatchDev
 *                            (>= 2^23 blocks), xSadBatchDev edge >= 8 (>= 128 MiB per input) -- times the family's candidate shapes on the
 *                            caller's own buffers and stream (that one call is synchronous and launches the kernel ~20 times; every
 *                            launch writes the same bytes) and the context keeps the fastest (the default unless beaten by > 2 %); xHipAutotuneReport shows what was
 *                            measured.  Skipped under stream capture, for overlapping buffers and when the family's own knobs are set.
 * Rounds 1-3 had sixteen more (cache-policy bits, LDS staging on / off, padding, per-kernel LDS charges, ...): the forms they
 * selected lost their A/Bs (profiles/r01_*.txt) and are gone. */
int  xHipSetOption(x266hip_ctx *ctx, const char *key, int value);
int  xHipGetOption(const x266hip_ctx *ctx, const char *key, int *value);
/* "autotune": one text line per tuned family -- "<family> choice <index> ms <per-candidate milliseconds, -1 = not launchable>" --
 * candidate 0 being the default shape (x266_amd/csrc/x266hip_abi.hip, k*Cands). */
int  xHipAutotuneReport(const x266hip_ctx *ctx, char *buf, size_t cap);

/* ------------------------------------------------------------------------ */
/* batch API, device pointers (inputs already resident in HBM)               */
/* `stream` holds a hipStream_t (NULL = the default stream).  Asynchronous:  */
/* returns after enqueueing; use xHipStreamSync or the caller's own stream   */
/* API to wait.  d_in / d_out must be 16-byte aligned and must not overlap.   */
/* ------------------------------------------------------------------------ */
/* 2-D forward 32x32 DCT-II, shifts 4 then 11, truncating int16 stores:
 * bit-exact with partialButterfly32 x2 as called by dct32_genNew
 * (src_tb/dct32.c:66-170,180-198).  out[v*32+u], v = vertical frequency. */
int xDct32FwdBatchDev(x266hip_ctx *ctx, const int16_t *d_in, int16_t *d_out,
                      size_t n_blocks, void *stream);
/* 2-D inverse (no upstream counterpart; HEVC/VVC inverse for 8-bit video:
 * column pass shift 7, row pass shift 12, int16 clipping after each pass). */
int xDct32InvBatchDev(x266hip_ctx *ctx, const int16_t *d_in, int16_t *d_out,
                      size_t n_blocks, void *stream);
/* The two lanes of a frame in ONE launch: xDct32FwdBatchDev of one batch and xSatd8x8BatchDev of another, bit-identical
 * to the two calls (BASELINE configs[4]: a 7680x4320 frame is 32 400 DCT32 blocks + 518 400 SATD blocks, ~15 us of kernel
 * each -- submissions, not arithmetic, pace a frame stream on one GPU).  Either count may be 0. */
int xDct32SatdFrameDev(x266hip_ctx *ctx, const int16_t *d_dct_in, int16_t *d_dct_out, size_t n_dct_blocks,
                       const int16_t *d_diff, uint32_t *d_satd_out, size_t n_satd_blocks, void *stream);
/* The 1-D pass by itself: partialButterfly32(src, dst, shift, line = 32) (src_tb/dct32.c:66-170; the RTL's first stage,
 * src/mkDct32.bsv:213-284) on every 32x32 block of the batch, i.e. dst[k*32 + j] = (int16)((sum_n g_t32[k][n] *
 * src[j*32 + n] + (1 << (shift-1))) >> shift) -- note the TRANSPOSED store.  xDct32PassDev(shift 4) followed by
 * xDct32PassDev(shift 11) is xDct32FwdBatchDev; on its own it lets a testbench compare the intermediate of a DUT.
 * Exact for every int16 input at shifts 1..15.  (A checking entry point: one block per wave, not a tuned kernel.) */
int xDct32PassDev(x266hip_ctx *ctx, const int16_t *d_in, int16_t *d_out, size_t n_blocks, int shift, void *stream);
/* Forward and inverse in one pass over the batch: d_coef = forward(d_in) (may be NULL when only
 * the reconstruction is wanted), d_recon = inverse(forward(d_in)) -- the transform half of an
 * encoder's reconstruction loop; SURVEY 8(d) "fused fwd+inv", 6144 bytes per block instead of
 * 8192.  Bit-identical to xDct32FwdBatchDev followed by xDct32InvBatchDev. */
int xDct32FwdInvBatchDev(x266hip_ctx *ctx, const int16_t *d_in, int16_t *d_coef, int16_t *d_recon,
                         size_t n_blocks, void *stream);
/* 8x8 Hadamard SATD of n residual blocks: bit-exact with satd8x8
 * (src_tb/satd.c:31-118), including its int16 wraparound.  d_out[n] uint32. */
int xSatd8x8BatchDev(x266hip_ctx *ctx, const int16_t *d_diff, uint32_t *d_out,
                     size_t n_blocks, void *stream);
/* The mixed transform set of BASELINE configs[3]: forward 2-D transforms of square N x N
 * int16 blocks (row-major, N*N samples each) built from two 1-D transform SLOTS: slot 0 with
 * N in {4, 8, 16, 32}, by default the DCT-II (sub-matrices of g_t32; N = 32 is pinned by the reference),
 * and slot 1 with N in {4, 8, 16}, by default the closed-form DST-VII
 * round(64 sqrt(N) sqrt(4/(2N+1)) sin(pi (2k+1)(n+1)/(2N+1))) -- for N = 4 the table of H.266, for N = 8 / 16
 * NOT claimed to be the standard's integers (those could not be checked offline; a host that holds the
 * normative tables, or wants DCT-VIII, installs them with xTransformSetMatrix below).
 * Two passes with partialButterfly32's structure
 * (src_tb/dct32.c:66-170): rows then columns, shifts log2N-1 and log2N+6, rounding
 * half up, truncating int16 stores; the DCT-II matrices are the sub-matrices of g_t32
 * that src/mkDct32.bsv:132-141 taps.  Only (DCT-II, 32) is pinned by upstream.
 * d_offsets == NULL: block b lives at sample offset b*N*N in both buffers.
 * d_offsets != NULL: block b lives at sample offset d_offsets[b] (a multiple of 8) in
 * both buffers -- the per-CTU mixed batches: one call per (type, size) class over a
 * shared residual / coefficient buffer pair. */
#define X266_TR_DCT2 0            /* slot 0 (DCT-II) horizontally and vertically */
#define X266_TR_DST7 1            /* slot 1 (DST-VII) horizontally and vertically */
#define X266_TR_DST7_DCT2 2       /* slot 1 horizontally (along rows), slot 0 vertically: N = 4, 8, 16 */
#define X266_TR_DCT2_DST7 3       /* slot 0 horizontally, slot 1 vertically */
int xTransformFwdBatchDev(x266hip_ctx *ctx, int type, int size, const int16_t *d_in, int16_t *d_out,
                          size_t n_blocks, const uint32_t *d_offsets, void *stream);
/* Caller-supplied 1-D transform matrices (the RTL re-uses one datapath for any tap set the same way,
 * src/mkDct32.bsv:132-141, 385-387).  slot 0 / 1 = the "DCT-II" / "DST-VII" slot of the type codes above,
 * size in {4, 8, 16}; m[k*size + n], row k = basis function, int8 (any values: the kernels only need the int8
 * operand images rebuilt); m == NULL restores the built-in matrix.  Affects every entry point of the set --
 * forward, inverse (which applies the transposes, columns first), the per-class and the one-launch calls --
 * of THIS context from the next call on.  Synchronises the device (launches in flight read the old tables);
 * not to be called concurrently with other calls on the context.  The 32-point DCT-II cannot be replaced. */
int xTransformSetMatrix(x266hip_ctx *ctx, int slot, int size, const int8_t *m);
int xTransformGetMatrix(const x266hip_ctx *ctx, int slot, int size, int8_t *m);
/* Named contents of slot 1 at all three sizes at once (slot 0 stays the DCT-II), through the same mechanism and with the
 * same all-or-nothing behaviour as xTransformSetMatrix:
 *   X266_PRESET_CLOSED_FORM  the built-in closed-form DST-VII (what a fresh context has)
 *   X266_PRESET_VTM_DST7     H.266's DST-VII integers as recalled from the VTM sources (DEFINE_DST7_P8/P16_MATRIX; N = 4 is the
 *                            closed form) -- marked "as recalled, unverified offline": the build environment holds neither the
 *                            standard nor VTM; a host that has the normative tables should install them with xTransformSetMatrix
 *   X266_PRESET_VTM_DCT8     DCT-VIII, H.266's third MTS kernel, as the flipped, sign-alternated DST-VII of the preset above:
 *                            T8[k][n] = (-1)^k T7[k][N-1-n]; with it the type codes X266_TR_DST7* mean DCT-VIII
 * xTransformPreset returns the preset slot 1 holds, or -1 after xTransformSetMatrix on slot 1. */
#define X266_PRESET_CLOSED_FORM 0
#define X266_PRESET_VTM_DST7    1
#define X266_PRESET_VTM_DCT8    2
int xTransformUsePreset(x266hip_ctx *ctx, int preset);
int xTransformPreset(const x266hip_ctx *ctx);
/* Inverse transforms of the same set (no upstream counterpart): columns first, shifts 7 and 12
 * (8-bit video), int16 clipping after each pass; (DCT-II, 32) contiguous is xDct32InvBatchDev.
 * d_offsets as in the forward call. */
int xTransformInvBatchDev(x266hip_ctx *ctx, int type, int size, const int16_t *d_in, int16_t *d_out,
                          size_t n_blocks, const uint32_t *d_offsets, void *stream);
/* The whole mixed set in ONE launch (BASELINE configs[3], "batched per CTU").  The buffers are sequences of
 * 32x32-sample regions ("tiles", 1024 samples); tile t is cut into (32/N)^2 blocks of one (type, size) class,
 * stored block-major, its class given by d_tile_class[t] = X266_TILE_CLASS(type, size) and its position by
 * d_tile_offsets[t] (sample offset, a multiple of 8; NULL: tile t at t * 1024, i.e. the buffer is the tiles
 * in order -- a CTU-ordered residual buffer whose 64x64 CTUs are four such tiles).  inverse = 0 forward,
 * 1 inverse; results identical to the per-class calls above. */
#define X266_TILE_CLASS(type, size) ((uint8_t)((type) * 4 + ((size) == 4 ? 0 : (size) == 8 ? 1 : (size) == 16 ? 2 : 3)))
int xTransformTilesDev(x266hip_ctx *ctx, int inverse, const int16_t *d_in, int16_t *d_out, size_t n_tiles,
                       const uint32_t *d_tile_offsets, const uint8_t *d_tile_class, void *stream);
/* Full-search motion estimation with the 8x8 SATD cost (BASELINE configs[2]).
 * For every 8x8 block of `cur` (block grid aligned to (0,0); width, height
 * multiples of 8) and every displacement (dx,dy) in [-range, range]^2,
 *     cost = satd8x8(cur_block - ref_block_at(x+dx, y+dy))      (src_tb/satd.c:31-118)
 * and d_best[by * (width/8) + bx] receives the minimum; among equal costs the
 * first candidate in raster order (dy ascending, then dx ascending) wins.
 * `ref` points at pixel (0,0) of a frame padded by at least `range` pixels on
 * every side (rows are ref_stride bytes apart; negative offsets are read).
 * 1 <= range <= 64.  d_costs may be NULL; otherwise it receives every cost,
 * d_costs[block * (2*range+1)^2 + (dy+range)*(2*range+1) + (dx+range)].
 * Strides follow src/x266.cpp:419 (intptr_t, in bytes). */
typedef struct x266_me_result_t {
    int16_t  mvx, mvy;      /* best displacement */
    uint32_t cost;          /* its SATD          */
} x266_me_result_t;
int xSatd8x8SearchDev(x266hip_ctx *ctx, const uint8_t *d_cur, intptr_t cur_stride,
                      const uint8_t *d_ref, intptr_t ref_stride, int width, int height,
                      int range, x266_me_result_t *d_best, uint32_t *d_costs, void *stream);
/* The same search with the cheaper metric: cost = sum |cur - ref| over the 8x8 block,
 * i.e. sad() of riscv/programs/benchmarks/sad/sad.c:28-39 at n = 8; same candidate order and
 * tie-break.  d_cur must be 4-byte aligned with cur_stride a multiple of 4. */
/* Allocates xSatd8x8SearchDev's scratch (128 bytes per 8x8 block of the frame) for searches of frames up to
 * width x height enqueued on `stream`, so that no launch path allocates: for stream captures and real-time loops.
 * The library keeps one buffer per stream, for at most 8 streams (the least recently used one is released after
 * its last search has finished); buffers handed out under a capture live as long as the context. */
int xHipMeScratchReserve(x266hip_ctx *ctx, void *stream, int width, int height);
int xSad8x8SearchDev(x266hip_ctx *ctx, const uint8_t *d_cur, intptr_t cur_stride, const uint8_t *d_ref,
                     intptr_t ref_stride, int width, int height, int range, x266_me_result_t *d_best,
                     uint32_t *d_costs, void *stream);
/* Frame container of the codec skeleton: ref_block_t, src/x266.cpp:56-63 -- a frame is a
 * raster of 512-byte tiles (16x16 luma, 8 rows of interleaved U,V pairs, 128 info bytes). */
typedef struct x266_ref_block_t {
    uint8_t m_Y[16 * 16];
    uint8_t m_C[2 * 8 * 8];
    uint8_t m_I[128];
} x266_ref_block_t;
/* xConvInputFmt (src/x266.cpp:415-453) on the device: planar YUV 4:2:0 -> tiles.  width and
 * height multiples of 16; chroma stride = strdY / 2 as upstream; rows 16-byte (luma) and
 * 8-byte (chroma) aligned.  m_I is left untouched, as upstream leaves it. */
int xConvInputFmtDev(x266hip_ctx *ctx, x266_ref_block_t *d_tiles, const uint8_t *d_y, const uint8_t *d_u,
                     const uint8_t *d_v, intptr_t strdY, int width, int height, void *stream);
/* xConvOutput420 (src/x266.cpp:455-492) on the device: tiles -> planar YUV 4:2:0. */
int xConvOutput420Dev(x266hip_ctx *ctx, const x266_ref_block_t *d_tiles, uint8_t *d_y, intptr_t strdY,
                      uint8_t *d_u, uint8_t *d_v, intptr_t strdC, int width, int height, void *stream);
/* Residual formation (no upstream counterpart: upstream stops before the residual stage):
 * luma of two tiled frames, residual = cur - pred as int16, emitted as row-major blocks in
 * raster order of blocks -- block_edge 32 feeds xDct32FwdBatchDev (width, height multiples
 * of 32), block_edge 8 feeds xSatd8x8BatchDev (multiples of 16). */
int xResidualLumaDev(x266hip_ctx *ctx, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred,
                     int width, int height, int block_edge, int16_t *d_residual, void *stream);
/* Fused residual formation + forward DCT32: d_coef[block] = DCT32(cur - pred) for every 32x32 luma
 * block of two tiled frames, blocks in raster order -- bit-identical to xResidualLumaDev(.., 32, ..)
 * followed by xDct32FwdBatchDev, without the residual ever touching HBM (half the traffic). */
int xDct32FwdFromTilesDev(x266hip_ctx *ctx, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred,
                          int width, int height, int16_t *d_coef, void *stream);
/* Fused residual formation + SATD: d_out[block] = satd8x8(cur - pred) for every 8x8 luma block of
 * two tiled frames (raster order of blocks) -- bit-identical to xResidualLumaDev(.., 8, ..) followed
 * by xSatd8x8BatchDev.  width, height multiples of 16. */
int xSatd8x8FromTilesDev(x266hip_ctx *ctx, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred,
                         int width, int height, uint32_t *d_out, void *stream);
/* The chroma half of the same stage.  A tile's chroma is m_C (src/x266.cpp:60): 8 rows of 8 interleaved (U, V) pairs, as
 * xConvInputFmt packs them (src/x266.cpp:441-449); the chroma planes of a width x height (luma) 4:2:0 frame are
 * (width / 2) x (height / 2).  block_edge 8: one 8x8 U and one 8x8 V block per tile (width, height multiples of 16),
 * the inputs of xSatd8x8BatchDev; block_edge 32: one 32x32 U and one 32x32 V block per 64x64 CTU (multiples of 64), the
 * inputs of xDct32FwdBatchDev.  Blocks are row-major int16, numbered in raster order within the chroma plane; block b of
 * U goes to d_res_u + b * block_pitch * edge^2, of V to d_res_v + b * block_pitch * edge^2 (block_pitch >= 1, in blocks):
 * block_pitch 1 with two buffers gives two planar block streams, block_pitch 2 with d_res_v = d_res_u + edge^2 gives
 * the CTU-ordered stream U0 V0 U1 V1 ...  (No upstream counterpart, as for luma.) */
int xResidualChromaDev(x266hip_ctx *ctx, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred,
                       int width, int height, int block_edge, int16_t *d_res_u, int16_t *d_res_v,
                       size_t block_pitch, void *stream);
/* Fused chroma residual + forward DCT32: both 32x32 chroma blocks of every 64x64 CTU, one wave per CTU -- bit-identical to
 * xResidualChromaDev(.., 32, ..) followed by xDct32FwdBatchDev; coefficients of CTU b's U block at
 * d_coef_u + b * block_pitch * 1024, V likewise.  width, height multiples of 64. */
int xDct32FwdChromaFromTilesDev(x266hip_ctx *ctx, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred,
                                int width, int height, int16_t *d_coef_u, int16_t *d_coef_v, size_t block_pitch,
                                void *stream);
/* A whole 4:2:0 CTU per unit of output, one launch: for every 64x64 CTU (raster order) 12 KiB of coefficients
 * d_coef[ctu * 6144 + q * 1024 ..]: q = 0..3 the forward DCT32 of its four 32x32 luma residual quadrants (top-left, top-right,
 * bottom-left, bottom-right), q = 4 of its 32x32 U residual, q = 5 of V -- bit-identical to xDct32FwdFromTilesDev and
 * xDct32FwdChromaFromTilesDev, whose frame-raster luma and separate chroma streams it re-orders into the order a per-CTU
 * encoder loop consumes (BASELINE configs[3]: "batched per-CTU").  width, height multiples of 64. */
int xDct32FwdCtuFromTilesDev(x266hip_ctx *ctx, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred,
                             int width, int height, int16_t *d_coef, void *stream);
/* Fused chroma residual + SATD: d_out_u[t * pitch] = satd8x8 of tile t's U residual, d_out_v[t * pitch] of its V residual
 * (tiles in raster order) -- bit-identical to xResidualChromaDev(.., 8, ..) followed by xSatd8x8BatchDev.  pitch 2 with
 * d_out_v = d_out_u + 1 interleaves the two costs.  width, height multiples of 16. */
int xSatd8x8ChromaFromTilesDev(x266hip_ctx *ctx, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred,
                               int width, int height, uint32_t *d_out_u, uint32_t *d_out_v, size_t pitch, void *stream);
/* Sum of absolute differences of n_blocks pairs of edge x edge 8-bit blocks (edge in
 * {4, 8, 16, 32, 64}; each block edge*edge contiguous bytes, row-major; buffers 16-byte
 * aligned): d_out[b] = sum |a - b|, exactly sad() of
 * riscv/programs/benchmarks/sad/sad.c:28-39 (whose 64 x 64 known answer 344807 the tests replay). */
int xSadBatchDev(x266hip_ctx *ctx, int edge, const uint8_t *d_a, const uint8_t *d_b, uint32_t *d_out,
                 size_t n_blocks, void *stream);
/* 32x32 intra prediction.  Upstream has only a work-in-progress RTL sketch of this
 * stage (src/mkIntra32-wip.bsv; no C model, so parity is UNPINNED): the HEVC 35-mode predictor
 * (H.265 8.4.4.2.4-6, nTbS = 32) -- mode 0 planar, 1 DC, 2..34 angular -- on references used as
 * given.  x266_intra_ref_t mirrors IntraRef_t (:36-39): left[y] = p[-1][y], top[0] = the corner
 * sample, top[1+x] = p[x][-1]; padded to 144 bytes so that sets are 16-byte aligned.
 * Output block i (1024 bytes, row-major) = mode d_modes[i] on set d_ref_index[i] (NULL: set i). */
typedef struct x266_intra_ref_t {
    uint8_t left[64];
    uint8_t top[65];
    uint8_t reserved[15];
} x266_intra_ref_t;
int xIntra32PredictDev(x266hip_ctx *ctx, const x266_intra_ref_t *d_refs, const uint8_t *d_modes,
                       const uint32_t *d_ref_index, uint8_t *d_pred, size_t n, void *stream);
/* The encoder loop's form of intra coding, in ONE kernel: d_coef[i] = forward DCT32 (xDct32FwdBatchDev's transform,
 * src_tb/dct32.c:66-170,180-198) of the residual d_src[i] - prediction(mode d_modes[i] on set d_ref_index[i] (NULL: set i)).
 * The prediction never reaches memory: 1 KiB of source samples (row-major 32x32 uint8) + 144 bytes of references in, 2 KiB of
 * coefficients out per block.  Bit-identical to xIntra32PredictDev -> residual -> xDct32FwdBatchDev.  Modes above 34 are
 * undefined input, as for xIntra32PredictDev. */
int xIntra32ResidualDct32Dev(x266hip_ctx *ctx, const x266_intra_ref_t *d_refs, const uint8_t *d_modes, const uint32_t *d_ref_index,
                             const uint8_t *d_src, int16_t *d_coef, size_t n, void *stream);
/* Intra mode decision (the sketch's "Decide" channel, IntraChannel_t :41-44): for block b with reference
 * set d_refs[b] and source samples d_src[b*1024 ..] (row-major 32x32, 16-byte aligned),
 * d_costs[b*35 + m] = sum over the sixteen 8x8 sub-blocks of satd8x8(src - prediction m)
 * (satd8x8 = src_tb/satd.c:31-118), m = 0..34; d_best_mode[b] (may be NULL) = the cheapest mode,
 * lowest index on ties.  The predictions are never written to memory. */
int xIntra32CostsDev(x266hip_ctx *ctx, const x266_intra_ref_t *d_refs, const uint8_t *d_src,
                     uint32_t *d_costs, uint8_t *d_best_mode, size_t n_blocks, void *stream);
/* Synthetic residual stream with the reference's stimulus distribution
 * ((rand()&0xFF)-(rand()&0xFF), src_tb/dct32.c:191-193) from a counter-based
 * SplitMix64: sample i = lo8(r) - lo8(r>>8), r = mix(seed+(first_index+i+1)*phi). */
int xFillResidualDev(x266hip_ctx *ctx, int16_t *d_dst, size_t n_samples,
                     uint64_t seed, uint64_t first_index, void *stream);

/* ------------------------------------------------------------------------ */
/* batch API, host pointers (caller-owned host buffers; staged through        */
/* internal device buffers in chunks, H2D / kernel / D2H overlapped).         */
/* Synchronous: results are in `out` on return.                               */
/* ------------------------------------------------------------------------ */
/* Rates over a PCIe 5.0 x16 link that gives 57 GB/s one way and 48.5 GB/s each way when both directions run: 43 GB/s each way
 * from PINNED host buffers (xHipHostAlloc below, hipHostMalloc, or memory the host registered), 27 GB/s each way from pageable ones
 * -- the runtime stages pageable copies through the calling thread, so uploads and downloads take turns
 * (profiles/r04_hostpipe.txt).  A host that re-uses its buffers should allocate them pinned. */
int xDct32FwdBatch(x266hip_ctx *ctx, const int16_t *in, int16_t *out, size_t n_blocks);
int xDct32InvBatch(x266hip_ctx *ctx, const int16_t *in, int16_t *out, size_t n_blocks);
int xSatd8x8Batch(x266hip_ctx *ctx, const int16_t *diff, uint32_t *out, size_t n_blocks);

/* ------------------------------------------------------------------------ */
/* device memory / stream helpers for hosts without HIP headers               */
/* ------------------------------------------------------------------------ */
int xHipMalloc(x266hip_ctx *ctx, void **d_ptr, size_t bytes);
/* Page-locked host memory (hipHostMalloc) for the host-pointer batch calls and xHipMemcpy*: copies from / to it are true DMA. */
int xHipHostAlloc(x266hip_ctx *ctx, void **h_ptr, size_t bytes);
int xHipHostFree(x266hip_ctx *ctx, void *h_ptr);
int xHipFree(x266hip_ctx *ctx, void *d_ptr);
int xHipMemcpyH2D(x266hip_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int xHipMemcpyD2H(x266hip_ctx *ctx, void *dst, const void *d_src, size_t bytes);
int xHipStreamSync(x266hip_ctx *ctx, void *stream);
int xHipStreamCreate(x266hip_ctx *ctx, void **stream);     /* a non-blocking hipStream_t */
int xHipStreamDestroy(x266hip_ctx *ctx, void *stream);
/* HIP graphs for launch-bound sequences (a frame's handful of small kernels costs more in launch
 * overhead than in execution): everything enqueued on `stream` (not the NULL stream) between
 * xHipGraphBegin and xHipGraphEnd -- any of the ...Dev calls above, in any number -- is recorded
 * instead of run, and xHipGraphLaunch replays the whole sequence with one submission.  The recorded
 * calls keep their pointer and size arguments, so a graph is replayed over the same buffers with new
 * contents.  xSatd8x8SearchDev sizes an internal scratch buffer per (stream, frame size) on first use, which is
 * illegal inside a capture: call xHipMeScratchReserve (below) or run one search on that stream beforehand. */
typedef struct x266hip_graph x266hip_graph;
int xHipGraphBegin(x266hip_ctx *ctx, void *stream);
int xHipGraphEnd(x266hip_ctx *ctx, void *stream, x266hip_graph **graph);
int xHipGraphLaunch(x266hip_ctx *ctx, x266hip_graph *graph, void *stream);
void xHipGraphFree(x266hip_ctx *ctx, x266hip_graph *graph);
/* Times `reps` back-to-back launches of one kernel with HIP events recorded on
 * `stream` itself; returns the mean milliseconds per launch in *ms_per_launch.
 * op: 0 = dct32 fwd, 1 = dct32 inv, 2 = satd8x8 (buffers as in the Dev calls); 3 .. 6 = xHipMemCeilingDev copy / read /
 * write / read probe of n_blocks * 2048 bytes. */
int xHipTimeKernel(x266hip_ctx *ctx, int op, const void *d_in, void *d_out,
                   size_t n_blocks, int reps, void *stream, double *ms_per_launch);
/* What THIS box's memory system gives the launch shape of the streaming kernels, with no arithmetic -- so that a report can
 * put "fraction of this box's copy / read rate" next to "fraction of the 8 TB/s spec" (boxes differ by 3-10 %).
 * kind X266_MEM_COPY: d_dst[0 .. bytes) = d_src[0 .. bytes) (16 bytes per lane, nontemporal 1 KiB-linear loads, "sc1 nt"
 * stores: the access pattern of the DCT / transform kernels); X266_MEM_READ: the same loads and nothing written but one
 * 32-bit XOR of the words of every 2 KiB piece, d_dst[piece] as uint32 (so d_dst holds 4 * ceil(bytes / 2048) bytes: the
 * pattern of the SATD / SAD kernels; the checksums make the stream checkable); X266_MEM_WRITE: nothing read (d_src may be
 * NULL), every 16-byte chunk c of d_dst = {(uint32)c, 0, 0, 0} with the same stores (the pattern of the intra predictor).
 * X266_MEM_READ_PROBE: X266_MEM_READ that stores a piece's XOR only where it equals X266_MEM_PROBE_MAGIC, i.e. practically
 * never: the rate of loads with nothing flowing back (a host that wants to see the loads happen plants a piece with that XOR).
 * bytes a multiple of 16, buffers 16-byte aligned.
 * Asynchronous on `stream`; time it with the event calls below or xHipTimeKernel's ops 3 / 4 / 5 / 6. */
#define X266_MEM_COPY 0
#define X266_MEM_READ 1
#define X266_MEM_WRITE 2
#define X266_MEM_READ_PROBE 3
#define X266_MEM_PROBE_MAGIC 0x12345678u
int xHipMemCeilingDev(x266hip_ctx *ctx, int kind, const void *d_src, void *d_dst, size_t bytes, void *stream);
/* HIP events for hosts without HIP headers, so that ANY sequence of the ...Dev calls can be timed on the
 * stream it is launched on (record an event before every launch and one after the last: consecutive
 * differences are per-launch durations).  xHipEventElapsedMs waits for `stop` and returns stop - start. */
int xHipEventCreate(x266hip_ctx *ctx, void **event);
int xHipEventDestroy(x266hip_ctx *ctx, void *event);
int xHipEventRecord(x266hip_ctx *ctx, void *event, void *stream);
int xHipEventElapsedMs(x266hip_ctx *ctx, void *start, void *stop, double *ms);

/* ------------------------------------------------------------------------ */
/* one node, several GPUs (BASELINE configs[4])                    */
/*                                                                          */
/* Blocks are independent (src_tb/dct32.c:75,167-168; satd8x8 is a pure      */
/* function, src_tb/satd.c:31-118), so the only multi-GPU traffic is moving   */
/* shards: the root rank (rank 0) owns the frame, every rank gets a           */
/* contiguous shard of each batch, transforms it, and returns the results.    */
/* Transport is RCCL point-to-point: ONE ncclGroupStart/End of               */
/* ncclSend/ncclRecv per step, so that all of the root's xGMI links and both  */
/* directions of each are busy at once; librccl.so.1 is opened on first use   */
/* (dlopen), so hosts that never create a node do not load it; a process that   */
/* has already loaded an RCCL (e.g. torch's bundled one) shares that copy.  The  */
/* environment variable X266HIP_RCCL_LIB, when set, names the library to open   */
/* INSTEAD (another RCCL build; the repository's tests point it at their        */
/* single-box model of RCCL's semantics) -- xHipNodeRcclInfo tells which        */
/* library and version a process ended up with.  A failed communication step    */
/* aborts the node's communicators (ncclCommAbort) so that no rank is left in a */
/* group that cannot complete; the node is then good for xHipNodeFree only and   */
/* every rank must treat the failure the same way.  Two process                  */
/* models, same calls afterwards:                                             */
/*   xHipNodeInit      one process drives n devices (ncclCommInitAll)         */
/*   xHipNodeInitRank  one process per GPU (ncclCommInitRank); every rank     */
/*                     makes the same sequence of xNode... calls              */
/* ------------------------------------------------------------------------ */
#define X266HIP_ECOMM    (-4)   /* RCCL unavailable or a communication call failed */
#define X266HIP_NODE_ID_BYTES 128
typedef struct x266hip_node x266hip_node;
typedef struct x266hip_nstream x266hip_nstream;

/* Host-only planning (no device needed; also what the CPU tests of the N > 1 logic call).
 * xShardRange: rank's contiguous [begin, end) of n_units; the first n_units % world ranks take one extra.
 * xMeStripePlan: motion search partition into horizontal stripes of 8-pixel block rows -- stripe's block
 * rows [*block_row_begin, *block_row_end) and the reference rows [*ref_row_begin, *ref_row_end) it reads
 * (frame row coordinates of the padded reference: may be negative / exceed height by up to `range`; the
 * halo is read-only input that travels with the scatter, there is no exchange step).  Returns 0 / EINVAL. */
int  xShardRange(size_t n_units, int rank, int world, size_t *begin, size_t *end);
int  xMeStripePlan(int height, int range, int stripe, int n_stripes, int *block_row_begin, int *block_row_end,
                   int *ref_row_begin, int *ref_row_end);

int  xHipNodeInit(x266hip_node **node, const int *devices, int n_devices);       /* devices NULL: 0..n-1 */
int  xHipNodeUniqueId(void *id /* X266HIP_NODE_ID_BYTES, from rank 0, handed to every rank by the host */);
int  xHipNodeInitRank(x266hip_node **node, int device, int rank, int world, const void *id);
void xHipNodeFree(x266hip_node *node);
int  xHipNodeWorld(const x266hip_node *node);
int  xHipNodeLocalCount(const x266hip_node *node);                               /* ranks driven by this process */
int  xHipNodeLocalRank(const x266hip_node *node, int local_index);               /* their global ranks */
x266hip_ctx *xHipNodeCtx(x266hip_node *node, int local_index);                   /* owned by the node */
const char *xHipNodeLastError(const x266hip_node *node);
/* "transport": 0 RCCL send/recv groups (default), 1 hipMemcpyPeerAsync (single-process nodes only);
 * "me_local_copy": 1 = the root's own motion-search stripes also go through stripe buffers (what a peer
 * receives) instead of being searched in place -- exercises the halo logic on one GPU (default 0). */
int  xHipNodeSetOption(x266hip_node *node, const char *key, int value);
/* Communication self-check: every rank sends a pattern to the next rank and receives from the previous
 * one inside one RCCL group (with one rank: to itself), then all ranks all-reduce a checksum. */
int  xHipNodeSelfTest(x266hip_node *node);
/* The RCCL this process uses: NCCL_VERSION_CODE-style version (0 if the library has no ncclGetVersion) and the path of
 * the shared object (dladdr).  X266HIP_ECOMM when no RCCL could be opened. */
int  xHipNodeRcclInfo(int *version, char *path, size_t path_cap);

/* A stream of frames, each a fixed set of "lanes" (one batch per lane).  op: 0 = DCT32 forward
 * (2048 B in / 2048 B out per unit), 1 = DCT32 inverse, 2 = 8x8 SATD (128 B in / 4 B out).
 * xNodeFrameStreamCreate: the two lanes of BASELINE configs[4] for a width x height luma frame:
 * (width/32)*(height/32) DCT32 blocks and (width/8)*(height/8) SATD blocks. */
int  xNodeStreamCreate(x266hip_node *node, int n_lanes, const int *ops, const size_t *max_units, x266hip_nstream **s);
int  xNodeFrameStreamCreate(x266hip_node *node, int width, int height, x266hip_nstream **s);
void xNodeStreamFree(x266hip_nstream *s);
/* Step t (the t-th Push/Flush step of this stream), pipelined and asynchronous: one RCCL group moves
 * frame t's shards root -> peers AND frame t-2's results peers -> root, then every rank's kernels for
 * frame t are enqueued; so while frame t is transformed, frame t+1's inputs and frame t-1's outputs
 * are on the links.  d_in[lane] / d_out[lane]: on the process that drives the root rank, device
 * pointers on the root's device of the lane's input units and of the caller-owned buffer that receives
 * its results (zero-copy: the root transforms its own shard in place, peers' shards are sent from /
 * received into these buffers directly); ignored on other processes (may be NULL).  units[lane]
 * (NULL: the stream's max_units) must be the same on every rank.  producer_stream: the root-device
 * stream the inputs were produced on (ordering by event; NULL = the default stream).
 * A step's ticket may be waited for once TWO later steps have been issued (Push or Flush); its inputs may be
 * overwritten once THREE later steps have been issued (or its ticket has been waited for): until then its buffers belong
 * to the stream -- the kernels of three consecutive frames run on three streams and may overlap (only a third frame in
 * flight covers the ramp and tail of ~30 us kernels), so frames t, t+1 and t+2 must not share buffers: input rings of four,
 * output rings of five or more (X266_STREAM_IN_RING / X266_STREAM_OUT_RING; they grew by one when the stream went from two to three
 * frames in flight, so a host sizes its rings by the constants, not by a number).  Input buffers may be SHARED between frames (they
 * are only read); an output buffer that overlaps the output of one of the previous X266_STREAM_OUT_RING - 1 frames of this stream
 * -- frames not yet flushed -- is refused with X266HIP_EINVAL: two frames in flight would write it.  *ticket (may be NULL) receives t. 
 * One process per GPU (xHipNodeInitRank): a push is a collective step.  A call that fails on ONE rank -- including an argument
 * error only the root can detect (its buffers) -- leaves the other ranks' step unmatched; the failing rank's node is marked
 * failed (every later call returns X266HIP_ECOMM) and the host must free the node on every rank, as after any ECOMM. */
#define X266_STREAM_IN_RING  4
#define X266_STREAM_OUT_RING 5
int  xNodeStreamPush(x266hip_nstream *s, const void *const *d_in, void *const *d_out, const size_t *units,
                     void *producer_stream, long *ticket);
/* The root-device stream on which the NEXT pushed frame's kernels will run (NULL on processes that do not drive the
 * root).  A producer that enqueues the frame's inputs on this stream and passes it as producer_stream needs no event:
 * stream order already puts the frame's kernels behind it (one event record and one stream wait less per frame). */
void *xNodeStreamNextSlotStream(x266hip_nstream *s);
/* Issues the two draining steps and blocks until every pushed frame's results are in place. */
int  xNodeStreamFlush(x266hip_nstream *s);
/* Blocks the host until the results of step `ticket` are complete in their d_out buffers (root), or
 * the step's shard has left (peers).  No communication; EINVAL when fewer than two later steps exist. */
int  xNodeStreamWait(x266hip_nstream *s, long ticket);
/* One batch through the same machinery (SURVEY 8e "end-to-end scatter -> compute -> gather"): the batch
 * is cut into chunks of chunk_units (0 = default: one launch with one rank, else 8 MiB of input per
 * rank and chunk but at least four chunks), pushed as frames, flushed.  Synchronous.  d_in / d_out on the root as above. */
int  xNodeBatchScatterGather(x266hip_node *node, int op, const void *d_in, void *d_out, size_t n_units,
                             size_t chunk_units);
/* Full-search motion estimation of one frame over the node (xSatd8x8SearchDev semantics and argument
 * meaning; d_cur / d_ref / d_best on the root's device, NULL elsewhere): the frame is cut into
 * n_stripes horizontal stripes (0: one per rank) dealt to the ranks in contiguous runs; the root sends
 * each stripe of `cur` and the stripe's rows +- range of the padded reference, every rank searches its
 * stripes, the (mv, cost) records return to d_best.  Results are identical to the single-device call
 * for any n_stripes.  Synchronous. */
int  xNodeSatd8x8Search(x266hip_node *node, const uint8_t *d_cur, intptr_t cur_stride, const uint8_t *d_ref,
                        intptr_t ref_stride, int width, int height, int range, int n_stripes,
                        x266_me_result_t *d_best);

/* ------------------------------------------------------------------------ */
/* host-only utilities (no device needed)                                     */
/* ------------------------------------------------------------------------ */
/* BDPI packing of two input rows, Vector#(2,Vector#(32,Bit#(16))):
 * res[w] = (x[2w+1] << 16) + x[2w], row `first_row` then `first_row+1`
 * (src_tb/dct32.c:205-220). */
void     xDct32PackDiffRows(const int16_t *mat, int first_row, unsigned int res[32]);
/* BDPI packing of 4 vertically adjacent coefficients, column-major walk:
 * idx -> col = idx>>5, row = idx&31 (src_tb/dct32.c:223-246). */
uint64_t xDct32PackDctWord(const int16_t *dct, int idx);
/* The N x N matrix the transform-set kernels use for a 1-D transform: type 0 = DCT-II (rows 0, 32/N, ... of g_t32,
 * first N columns: the taps of src/mkDct32.bsv:132-141), 1 = DST-VII; N in {4, 8, 16} (and 32 for DCT-II).
 * m[k*N + n], row k = frequency.  Lets a host (and the CPU tests) check the product's own tables. */
int      xTransformMatrix(int type, int size, int16_t *m);
const char *xHipVersion(void);

/* ------------------------------------------------------------------------ */
/* BDPI drop-in surface (stateful, non-reentrant, single-threaded -- exactly  */
/* like upstream; device from env X266HIP_DEVICE, default 0)                  */
/* ------------------------------------------------------------------------ */
/* const int16_t g_t32[32][32]        -- src_tb/dct32.c:30 (exported global)  */
#ifndef X266HIP_DEFINING_TABLE   /* the defining TU generates it at compile time */
extern const int16_t g_t32[32][32];
#endif
/* draws a new 32x32 stimulus block with rand() and transforms it on the GPU  */
void               dct32_genNew(void);                    /* src_tb/dct32.c:178 */
/* two input rows per call, 16 calls per block                                */
void               dct32_getDiff(unsigned int res[]);     /* src_tb/dct32.c:205 */
/* four coefficients per call, column-major, 256 calls per block              */
unsigned long long dct32_getDct(void);                    /* src_tb/dct32.c:223 */
void               satd8x8_genNew(void);                  /* src_tb/satd.c:124  */
/* one row (8 x int16 = 4 words) per call                                     */
void               satd8x8_getDiff(unsigned int res[]);   /* src_tb/satd.c:143  */
unsigned int       satd8x8_getSatd(void);                 /* src_tb/satd.c:149  */
/* Per-call twin of the RISC-V benchmark's sad() (riscv/programs/benchmarks/sad/sad.c:28-39): same arguments and
 * result, on the GPU, through the same lazily created context; n in {4, 8, 16, 32, 64}, -1 otherwise. */
int                x266_sad(const unsigned char *input_data1, const unsigned char *input_data2, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* X266HIP_H */
