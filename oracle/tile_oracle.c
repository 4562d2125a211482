/*
 * tile_oracle.c -- CPU restatement of the codec skeleton's frame container conversion and of
 * this repository's residual formation.  TEST INFRASTRUCTURE ONLY (see x266_oracle.h).
 *
 * Reference: ref_block_t, src/x266.cpp:56-63 (512-byte tile: m_Y[16*16], m_C[2*8*8] as 8 rows of
 * interleaved U,V pairs, m_I[128]); xConvInputFmt, src/x266.cpp:415-453; xConvOutput420,
 * src/x266.cpp:455-492.  Parity: restated from the source, NOT pinned by execution --
 * src/x266.cpp does not build on this toolchain (`_aligned_malloc`, `_stricmp`, MSVC
 * packing pragmas) and holds no vectors (its only self-test is
 * compiled out, src/x266.cpp:614-643).  The checks available are the round trip
 * unpack(pack(x)) == x and a numpy statement of the layout (tests/test_oracle_props.py).
 * Residual formation is UNPINNED (no upstream counterpart).
 */
#include "x266_oracle.h"

#include <string.h>

enum { TILE = 16, TILE_BYTES = 512, Y_BYTES = 256 };

void orc_conv_input_fmt(uint8_t *tiles, const uint8_t *y, const uint8_t *u, const uint8_t *v,
                        ptrdiff_t strd_y, int width, int height)
{
    const ptrdiff_t strd_c = strd_y >> 1;                 /* x266.cpp:426 */
    uint8_t *t = tiles;
    for (int ty = 0; ty < height; ty += TILE)
        for (int tx = 0; tx < width; tx += TILE, t += TILE_BYTES) {
            for (int r = 0; r < TILE; r++) memcpy(t + r * TILE, y + (ty + r) * strd_y + tx, TILE);
            for (int r = 0; r < TILE / 2; r++)
                for (int c = 0; c < TILE / 2; c++) {
                    t[Y_BYTES + r * TILE + 2 * c]     = u[((ty >> 1) + r) * strd_c + (tx >> 1) + c];
                    t[Y_BYTES + r * TILE + 2 * c + 1] = v[((ty >> 1) + r) * strd_c + (tx >> 1) + c];
                }
        }
}

void orc_conv_output_420(const uint8_t *tiles, uint8_t *y, ptrdiff_t strd_y, uint8_t *u, uint8_t *v,
                         ptrdiff_t strd_c, int width, int height)
{
    const uint8_t *t = tiles;
    for (int ty = 0; ty < height; ty += TILE)
        for (int tx = 0; tx < width; tx += TILE, t += TILE_BYTES) {
            for (int r = 0; r < TILE; r++) memcpy(y + (ty + r) * strd_y + tx, t + r * TILE, TILE);
            for (int r = 0; r < TILE / 2; r++)
                for (int c = 0; c < TILE / 2; c++) {
                    u[((ty >> 1) + r) * strd_c + (tx >> 1) + c] = t[Y_BYTES + r * TILE + 2 * c];
                    v[((ty >> 1) + r) * strd_c + (tx >> 1) + c] = t[Y_BYTES + r * TILE + 2 * c + 1];
                }
        }
}

/* residual blocks (edge 8 or 32) of cur - pred on the tiles' luma, raster order of blocks */
void orc_residual_luma(const uint8_t *cur_tiles, const uint8_t *pred_tiles, int width, int height, int edge,
                       int16_t *res)
{
    const int tiles_x = width / TILE, bx_n = width / edge;
    for (int py = 0; py < height; py++)
        for (int px = 0; px < width; px++) {
            const size_t tile = (size_t)(py / TILE) * tiles_x + px / TILE;
            const size_t o = tile * TILE_BYTES + (py % TILE) * TILE + px % TILE;
            const size_t blk = (size_t)(py / edge) * bx_n + px / edge;
            res[blk * edge * edge + (py % edge) * edge + px % edge] = (int16_t)((int)cur_tiles[o] - (int)pred_tiles[o]);
        }
}

/* residual blocks (edge 8 or 32) of cur - pred on the tiles' chroma: m_C holds row r of the tile's 8x8 U and V samples as
 * the interleaved pairs t[256 + r*16 + 2*c] (U) and t[256 + r*16 + 2*c + 1] (V), x266.cpp:441-449.  Plane U's blocks go to
 * res_u + blk * block_pitch * edge^2, V's to res_v likewise; blocks in raster order of the (width/2) x (height/2) plane. */
void orc_residual_chroma(const uint8_t *cur_tiles, const uint8_t *pred_tiles, int width, int height, int edge,
                         int16_t *res_u, int16_t *res_v, size_t block_pitch)
{
    const int tiles_x = width / TILE, cw = width / 2, ch = height / 2, bx_n = cw / edge;
    for (int py = 0; py < ch; py++)
        for (int px = 0; px < cw; px++) {
            const size_t tile = (size_t)(py / 8) * tiles_x + px / 8;
            const size_t o = tile * TILE_BYTES + Y_BYTES + (py % 8) * TILE + 2 * (px % 8);
            const size_t blk = (size_t)(py / edge) * bx_n + px / edge;
            const size_t e = blk * block_pitch * edge * edge + (py % edge) * edge + px % edge;
            res_u[e] = (int16_t)((int)cur_tiles[o] - (int)pred_tiles[o]);
            res_v[e] = (int16_t)((int)cur_tiles[o + 1] - (int)pred_tiles[o + 1]);
        }
}
