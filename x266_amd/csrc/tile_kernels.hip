// tile_kernels.hip -- frame container conversion and residual formation for gfx950
// (the data formats either side of the transform / SATD kernels).
//
// Reference: ref_block_t (src/x266.cpp:56-63): a frame is a raster of 512-byte tiles,
//   m_Y[16*16] luma, m_C[2*8*8] chroma as 8 rows of interleaved U,V pairs, m_I[128] info;
// xConvInputFmt (src/x266.cpp:415-453) packs planar YUV 4:2:0 into tiles (chroma stride =
// luma stride / 2), xConvOutput420 (src/x266.cpp:455-492) unpacks.  m_I is never written by
// either and is left untouched here as well.  Residual formation (cur - pred on the luma of two
// tiled frames, emitted as the row-major int16 blocks the DCT32 / SATD kernels consume) has no
// upstream counterpart -- upstream stops before the residual stage (xEncodeFrame,
// src/x266.cpp:526-555) -- and is defined in include/x266hip.h.
//
// Pure data movement, HBM-bound.  One thread per 16-byte luma row of a tile; the first 8 rows of
// a tile also carry its chroma row (8 U + 8 V bytes <-> 16 interleaved bytes).  A wave takes eight
// horizontally adjacent tiles x eight rows (lane = 8 * row + tile): on the planar side eight lanes
// then cover one whole 128-byte line of a picture row, on the tile side the eight rows of a tile are
// one whole line, and the residual rows of a block pair are contiguous -- every load and store
// instruction consumes whole lines, which is what lets the streaming cache hints pay.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "x266_device.hpp"
#include "x266_mfma_blocks.hpp"

namespace x266 {
namespace {

// wave -> (tile row ty, group of 8 tiles, upper / lower half of the tiles); lane -> (row, tile)
struct TileRow {
    size_t tile;      // tile index in the frame raster
    size_t ty, tx;
    int i;            // luma row inside the tile, 0..15
    bool live;        // this lane's tile exists
    bool unit_live;   // the wave's (tile row, group, half) exists: false for the whole wave at once
};

// A wave takes K consecutive units (template parameter), all of their loads issued before the first store.

__device__ __forceinline__ TileRow tile_row_of_unit(size_t unit, int lane, int tiles_x, int groups_x, size_t n_units)
{
    TileRow r;
    r.unit_live = unit < n_units;
    r.live = r.unit_live;
    const size_t u = r.live ? unit : 0;
    const int half = (int)(u & 1);
    const size_t gq = u >> 1;
    r.ty = gq / groups_x;
    r.tx = (gq - r.ty * groups_x) * 8 + (lane & 7);
    r.i = 8 * half + (lane >> 3);
    r.live = r.live && r.tx < (size_t)tiles_x;
    if (r.tx >= (size_t)tiles_x) r.tx = tiles_x - 1;
    r.tile = r.ty * tiles_x + r.tx;
    return r;
}

// PACK: planar -> tiles, else tiles -> planar
template <bool PACK, int kUnitsPerWave>
__global__ __launch_bounds__(256) void tile_convert_kernel(x266_ref_block_t *tiles, uint8_t *y, uint8_t *u, uint8_t *v,
                                                           long long strd_y, long long strd_c, int tiles_x, int groups_x,
                                                           size_t n_units)
{
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    TileRow r[kUnitsPerWave];
    v4i luma[kUnitsPerWave], cc[kUnitsPerWave];
    uint2 ua[kUnitsPerWave], ub[kUnitsPerWave];
#pragma unroll
    for (int k = 0; k < kUnitsPerWave; ++k) {
        r[k] = tile_row_of_unit(wave * kUnitsPerWave + k, lane, tiles_x, groups_x, n_units);
        luma[k] = cc[k] = v4i{0, 0, 0, 0};
        ua[k] = ub[k] = make_uint2(0, 0);
        if (!r[k].live) continue;
        const int i = r[k].i;
        uint8_t *t = reinterpret_cast<uint8_t *>(tiles + r[k].tile);
        uint8_t *py = y + (long long)(r[k].ty * 16 + i) * strd_y + r[k].tx * 16;
        luma[k] = PACK ? load16<true>(py) : load16<true>(t + i * 16);
        if (i < 8) {
            if (PACK) {
                ua[k] = *reinterpret_cast<const uint2 *>(u + (long long)(r[k].ty * 8 + i) * strd_c + r[k].tx * 8);
                ub[k] = *reinterpret_cast<const uint2 *>(v + (long long)(r[k].ty * 8 + i) * strd_c + r[k].tx * 8);
            } else {
                cc[k] = load16<true>(t + 256 + i * 16);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kUnitsPerWave; ++k) {
        if (!r[k].live) continue;
        const int i = r[k].i;
        uint8_t *t = reinterpret_cast<uint8_t *>(tiles + r[k].tile);
        uint8_t *py = y + (long long)(r[k].ty * 16 + i) * strd_y + r[k].tx * 16;
        if (PACK) store16_sc1nt(t + i * 16, luma[k]);
        else      store16_sc1nt(py, luma[k]);
        if (i < 8) {
            uint8_t *pu = u + (long long)(r[k].ty * 8 + i) * strd_c + r[k].tx * 8;
            uint8_t *pv = v + (long long)(r[k].ty * 8 + i) * strd_c + r[k].tx * 8;
            uint32_t *c = reinterpret_cast<uint32_t *>(t + 256 + i * 16);
            if (PACK) {
                const uint2 a = ua[k], b = ub[k];
                const v4i o = {(int)bperm(b.x, a.x, 0x05010400u),      // u0 v0 u1 v1
                               (int)bperm(b.x, a.x, 0x07030602u),      // u2 v2 u3 v3
                               (int)bperm(b.y, a.y, 0x05010400u), (int)bperm(b.y, a.y, 0x07030602u)};
                store16_sc1nt(c, o);
            } else {
                const uint32_t c0 = (uint32_t)cc[k][0], c1 = (uint32_t)cc[k][1], c2 = (uint32_t)cc[k][2], c3 = (uint32_t)cc[k][3];
                *reinterpret_cast<uint2 *>(pu) = make_uint2(bperm(c1, c0, 0x06040200u), bperm(c3, c2, 0x06040200u));
                *reinterpret_cast<uint2 *>(pv) = make_uint2(bperm(c1, c0, 0x07050301u), bperm(c3, c2, 0x07050301u));
            }
        }
    }
}

// residual[block][r][c] = cur - pred on tile luma; one thread per 16-pixel row segment.
// LOGB = log2(block edge): 3 (8x8, the SATD blocks) or 5 (32x32, the DCT blocks)
template <int LOGB, int kUnitsPerWave>
__global__ __launch_bounds__(256) void residual_luma_kernel(const x266_ref_block_t *__restrict__ cur,
                                                            const x266_ref_block_t *__restrict__ pred,
                                                            int16_t *__restrict__ res, int tiles_x, int groups_x, size_t n_units)
{
    constexpr int B = 1 << LOGB;
    __shared__ __attribute__((aligned(16))) unsigned char stage[B == 32 ? 4 * 2048 : 16];
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    TileRow r[kUnitsPerWave];
    v4i a[kUnitsPerWave], b[kUnitsPerWave];
#pragma unroll
    for (int k = 0; k < kUnitsPerWave; ++k) {
        r[k] = tile_row_of_unit(wave * kUnitsPerWave + k, lane, tiles_x, groups_x, n_units);
        a[k] = b[k] = v4i{0, 0, 0, 0};
        // B = 32 stores cooperatively: lanes whose tile lies beyond the frame edge (clamped to the last tile, so
        // their loads are valid) stay in the unit; their runs are skipped at the store
        if (B == 32 ? !r[k].unit_live : !r[k].live) continue;
        a[k] = load16<true>(reinterpret_cast<const uint8_t *>(cur + r[k].tile) + r[k].i * 16);
        b[k] = load16<true>(reinterpret_cast<const uint8_t *>(pred + r[k].tile) + r[k].i * 16);
    }
    const size_t width = (size_t)tiles_x * 16;
#pragma unroll
    for (int k = 0; k < kUnitsPerWave; ++k) {
        if (B == 32 ? !r[k].unit_live : !r[k].live) continue;                // B = 32: wave-uniform
        uint32_t d[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t x = (uint32_t)a[k][q], z = (uint32_t)b[k][q];
            const int d0 = (int)(x & 255) - (int)(z & 255), d1 = (int)((x >> 8) & 255) - (int)((z >> 8) & 255);
            const int d2 = (int)((x >> 16) & 255) - (int)((z >> 16) & 255), d3 = (int)(x >> 24) - (int)(z >> 24);
            d[2 * q] = ((uint32_t)d0 & 0xFFFFu) | ((uint32_t)d1 << 16);
            d[2 * q + 1] = ((uint32_t)d2 & 0xFFFFu) | ((uint32_t)d3 << 16);
        }
        const size_t py = r[k].ty * 16 + r[k].i, px = r[k].tx * 16;         // pixel coordinates of this segment
        if (B == 32) {
            // A unit's 2 KiB of residual are four 512-byte runs (8 rows x 64 bytes of four 32x32 blocks), but a
            // lane's 32 bytes are only a quarter line: written straight, every store instruction would touch half
            // of each line.  Through a wave-private 2 KiB LDS slot the wave stores its runs with two 1 KiB-linear
            // instructions instead.
            unsigned char *slot = stage + (threadIdx.x >> 6) * 2048;
            const int t = lane & 7, rr = lane >> 3;
            unsigned char *mine = slot + (t >> 1) * 512 + rr * 64 + (t & 1) * 32;
            *reinterpret_cast<v4i *>(mine) = v4i{(int)d[0], (int)d[1], (int)d[2], (int)d[3]};
            *reinterpret_cast<v4i *>(mine + 16) = v4i{(int)d[4], (int)d[5], (int)d[6], (int)d[7]};
            __builtin_amdgcn_wave_barrier();
            const size_t unit = wave * kUnitsPerWave + k;
            const size_t tx0 = (unit >> 1) % (size_t)groups_x * 8;          // first tile of the unit's group
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = lane + 64 * j, run = c >> 5;                  // 16-byte chunk of the 2 KiB, its 512-byte run
                const size_t bx = (tx0 >> 1) + run;                         // block column of that run
                if (bx * 2 < (size_t)tiles_x) {
                    const size_t blk = (py >> 5) * (width >> 5) + bx;
                    char *dst = reinterpret_cast<char *>(res + blk * 1024 + ((py & 31) & ~7) * 32) + (c & 31) * 16;
                    store16_sc1nt(dst, *reinterpret_cast<const v4i *>(slot + c * 16));
                }
            }
            __builtin_amdgcn_wave_barrier();                                // the slot is reused by the next unit
        } else {                                                           // two 8x8 blocks side by side
            const size_t blk = (py >> 3) * (width >> 3) + (px >> 3);
            int16_t *dst = res + blk * 64 + (py & 7) * 8;
            store16_sc1nt(dst, v4i{(int)d[0], (int)d[1], (int)d[2], (int)d[3]});
            store16_sc1nt(dst + 64, v4i{(int)d[4], (int)d[5], (int)d[6], (int)d[7]});
        }
    }
}


// ---- chroma residual --------------------------------------------------------------------------------------------------
// A tile's chroma is ONE 128-byte line (m_C, src/x266.cpp:60: 8 rows of 8 interleaved U,V pairs, packed at :441-449).  One lane
// takes one 16-byte row of it from both frames and emits 8 U and 8 V residuals (16 bytes each) -- the de-interleave is two
// masks and two packed 16-bit subtractions per dword.  A wave takes eight tiles x eight rows, lane = 8 * tile + row, so that a
// tile's eight lanes read its whole line:
//   LOGB = 3 (one 8x8 U and one 8x8 V block per tile): eight horizontally adjacent tiles; a tile's lanes write its two blocks'
//            128 bytes each -- with block_pitch 1 the wave's U (and V) store is 1 KiB linear;
//   LOGB = 5 (one 32x32 U and V block per 64x64 CTU = 4 x 4 tiles): 4 tiles x 2 tile rows = 16 rows of ONE block, re-ordered
//            through a wave-private LDS slot into that block's row order -- again one 1 KiB-linear store per plane.
// Both forms read whole lines and write whole lines, 1 KiB per instruction.
typedef short v2s __attribute__((ext_vector_type(2)));

template <int LOGB>
__global__ __launch_bounds__(256) void residual_chroma_kernel(const x266_ref_block_t *__restrict__ cur,
                                                              const x266_ref_block_t *__restrict__ pred,
                                                              int16_t *__restrict__ res_u, int16_t *__restrict__ res_v,
                                                              size_t block_pitch, int tiles_x, int groups_x, size_t n_units)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];  // LOGB = 5: 2 KiB per wave; the rest of the charge caps the resident waves
    const size_t unit = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (unit >= n_units) return;
    const int lane = threadIdx.x & 63;
    const size_t uy = unit / (size_t)groups_x, g = unit - uy * (size_t)groups_x;
    const int t = lane >> 3, row = lane & 7;                              // lane = 8 * tile + row: a tile's eight lanes read its whole line
    // LOGB = 3: the unit is 8 tiles of one tile row.  LOGB = 5: 4 tiles of two tile rows = a quarter-height slab of ONE block
    // (tiles_x and the tile row count are multiples of 4 there: no ragged units)
    const size_t ty = LOGB == 3 ? uy : uy * 2 + (size_t)(t >> 2);
    size_t tx = LOGB == 3 ? g * 8 + (size_t)t : g * 4 + (size_t)(t & 3);
    const bool live = tx < (size_t)tiles_x;
    if (!live) tx = (size_t)tiles_x - 1;                                 // lanes past the frame edge re-read its last tile, store nothing
    const size_t tile = ty * (size_t)tiles_x + tx;
    const v4i a = load16<true>(reinterpret_cast<const uint8_t *>(cur + tile) + 256 + row * 16);
    const v4i b = load16<true>(reinterpret_cast<const uint8_t *>(pred + tile) + 256 + row * 16);
    v4i du, dv;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t x = (uint32_t)a[q], z = (uint32_t)b[q];           // u v u v
        const uint32_t xu = x & 0x00FF00FFu, zu = z & 0x00FF00FFu, xv = (x >> 8) & 0x00FF00FFu, zv = (z >> 8) & 0x00FF00FFu;
        const v2s eu = __builtin_bit_cast(v2s, xu) - __builtin_bit_cast(v2s, zu);      // v_pk_sub_i16: two residuals at once
        const v2s ev = __builtin_bit_cast(v2s, xv) - __builtin_bit_cast(v2s, zv);
        du[q] = __builtin_bit_cast(int, eu);
        dv[q] = __builtin_bit_cast(int, ev);
    }
    if (LOGB == 5) {
        // The loads want a tile's rows in adjacent lanes, the stores a block row's four tiles: the 2 x 1 KiB change hands through a wave-private
        // LDS slot.  16-byte chunk (tile row j, row, tile) sits at j*32 + row*4 + (tile ^ ((row >> 1) & 3)): both sides conflict-free, and the
        // slab leaves as ONE 1 KiB-linear store per plane (16 block rows of 64 bytes).
        unsigned char *slot = stage + (threadIdx.x >> 6) * 2048;
        const unsigned wr = (unsigned)((t >> 2) * 32 + row * 4 + ((t & 3) ^ ((row >> 1) & 3)));
        *reinterpret_cast<v4i *>(slot + wr * 16) = du;
        *reinterpret_cast<v4i *>(slot + 1024 + wr * 16) = dv;
        __builtin_amdgcn_wave_barrier();
        const unsigned rd = (unsigned)((lane & ~3) + ((lane & 3) ^ ((lane >> 3) & 3)));
        const v4i su = *reinterpret_cast<const v4i *>(slot + rd * 16), sv = *reinterpret_cast<const v4i *>(slot + 1024 + rd * 16);
        const size_t blk = (uy >> 1) * (size_t)(tiles_x >> 2) + g;
        const size_t off = blk * block_pitch * 1024 + (uy & 1) * 512 + (size_t)lane * 8;
        store16_sc1nt(res_u + off, su);
        store16_sc1nt(res_v + off, sv);
        return;
    }
    if (!live) return;
    const size_t off = tile * block_pitch * 64 + (size_t)row * 8;         // tiles and 8x8 chroma blocks share their raster
    store16_sc1nt(res_u + off, du);
    store16_sc1nt(res_v + off, dv);
}

}  // namespace

// Units per wave, measured (tools/probes/gpu_tilefmt_probe.py, 32768^2 frame): unpacking gains 13 % from two units per wave (its planar
// stores are half lines per instruction; more of them in flight per wave), packing and the residual kernels lose 3-10 %.
constexpr int kUnitsPack = 1, kUnitsUnpack = 2, kUnitsResidual = 1;

hipError_t launch_tile_convert(bool pack, x266_ref_block_t *d_tiles, uint8_t *d_y, uint8_t *d_u, uint8_t *d_v,
                               long long strd_y, long long strd_c, int width, int height, hipStream_t stream)
{
    const int tiles_x = width / 16, groups_x = (tiles_x + 7) / 8;
    const size_t n_units = (size_t)groups_x * (height / 16) * 2;           // unit = (tile row, 8 tiles, half)
    if (n_units == 0) return hipSuccess;
    const size_t K = pack ? kUnitsPack : kUnitsUnpack;
    const size_t n_waves = (n_units + K - 1) / K;
    if ((n_waves + 3) / 4 > 0x7FFFFFFFull) return hipErrorInvalidValue;
    dim3 grid((unsigned)((n_waves + 3) / 4)), block(256);
    if (pack) hipLaunchKernelGGL((tile_convert_kernel<true, kUnitsPack>), grid, block, 0, stream, d_tiles, d_y, d_u, d_v, strd_y, strd_c, tiles_x, groups_x, n_units);
    else      hipLaunchKernelGGL((tile_convert_kernel<false, kUnitsUnpack>), grid, block, 0, stream, d_tiles, d_y, d_u, d_v, strd_y, strd_c, tiles_x, groups_x, n_units);
    return hipGetLastError();
}

hipError_t launch_residual_luma(int block_edge, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int16_t *d_res,
                                int width, int height, hipStream_t stream)
{
    const int tiles_x = width / 16, groups_x = (tiles_x + 7) / 8;
    const size_t n_units = (size_t)groups_x * (height / 16) * 2;
    if (n_units == 0) return hipSuccess;
    const size_t n_waves = (n_units + kUnitsResidual - 1) / kUnitsResidual;
    // two-wave workgroups, 20 KiB of LDS charged per workgroup = 16 resident waves per CU: paired in one process against round 4's shape (four-wave
    // workgroups, no cap) 0.662 against 0.707 ms for the 32x32 order of a 32768^2 frame, 0.658 against 0.709 for the 8x8 order (-6.5 / -7 %:
    // 0.81 of 8 TB/s, 0.97-0.98 of the box's copy stream; tools/probes/gpu_residual_shapes.py, profiles/r05_from_tiles_dma.txt)
    constexpr unsigned kThreads = 128, kLdsPerWorkgroup = 20480;
    const size_t wgs = (n_waves + 1) / 2;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    dim3 grid((unsigned)wgs), block(kThreads);
    if (block_edge == 32) hipLaunchKernelGGL((residual_luma_kernel<5, kUnitsResidual>), grid, block, kLdsPerWorkgroup - 4 * 2048, stream, d_cur, d_pred, d_res, tiles_x, groups_x, n_units);
    else                  hipLaunchKernelGGL((residual_luma_kernel<3, kUnitsResidual>), grid, block, kLdsPerWorkgroup, stream, d_cur, d_pred, d_res, tiles_x, groups_x, n_units);
    return hipGetLastError();
}

hipError_t launch_residual_chroma(int block_edge, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int16_t *d_res_u, int16_t *d_res_v,
                                  size_t block_pitch, int width, int height, hipStream_t stream)
{
    const int tiles_x = width / 16, tiles_y = height / 16;
    const int groups_x = block_edge == 32 ? tiles_x / 4 : (tiles_x + 7) / 8;
    const size_t n_units = (size_t)groups_x * (size_t)(block_edge == 32 ? tiles_y / 2 : tiles_y);     // one wave each: 8 tiles' chroma lines
    if (n_units == 0) return hipSuccess;
    // one-wave workgroups, 12 KiB of LDS charged = 13 resident waves per CU: paired in one process on a 32768^2 frame (tools/probes/gpu_chroma_shapes.py,
    // profiles/r06_chroma_shapes.txt) the 8x8 order runs at 0.977 of the box's copy of the same bytes (two-wave workgroups with 20 KiB, the luma shape: 0.96)
    constexpr unsigned kThreads = 64, kLdsPerWorkgroup = 12288;
    if (n_units > 0x7FFFFFFFull) return hipErrorInvalidValue;
    dim3 grid((unsigned)n_units), block(kThreads);
    if (block_edge == 32) hipLaunchKernelGGL(residual_chroma_kernel<5>, grid, block, kLdsPerWorkgroup, stream, d_cur, d_pred, d_res_u, d_res_v, block_pitch, tiles_x, groups_x, n_units);
    else                  hipLaunchKernelGGL(residual_chroma_kernel<3>, grid, block, kLdsPerWorkgroup, stream, d_cur, d_pred, d_res_u, d_res_v, block_pitch, tiles_x, groups_x, n_units);
    return hipGetLastError();
}

}  // namespace x266
