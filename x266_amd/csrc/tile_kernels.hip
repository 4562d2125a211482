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
// a tile also carry its chroma row (8 U + 8 V bytes <-> 16 interleaved bytes).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "x266_device.hpp"

namespace x266 {
namespace {

__device__ __forceinline__ uint32_t bperm(uint32_t hi_src, uint32_t lo_src, uint32_t sel)
{
    return __builtin_amdgcn_perm(hi_src, lo_src, sel);
}

// PACK: planar -> tiles, else tiles -> planar
template <bool PACK>
__global__ __launch_bounds__(256) void tile_convert_kernel(x266_ref_block_t *tiles, uint8_t *y, uint8_t *u, uint8_t *v,
                                                           long long strd_y, long long strd_c, int tiles_x, size_t n_rows)
{
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // (tile, luma row)
    if (id >= n_rows) return;
    const size_t tile = id >> 4;
    const int i = (int)(id & 15);
    const size_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    uint8_t *t = reinterpret_cast<uint8_t *>(tiles + tile);
    uint8_t *py = y + (long long)(ty * 16 + i) * strd_y + tx * 16;
    if (PACK) *reinterpret_cast<v4i *>(t + i * 16) = *reinterpret_cast<const v4i *>(py);
    else      *reinterpret_cast<v4i *>(py) = *reinterpret_cast<const v4i *>(t + i * 16);
    if (i < 8) {
        uint8_t *pu = u + (long long)(ty * 8 + i) * strd_c + tx * 8;
        uint8_t *pv = v + (long long)(ty * 8 + i) * strd_c + tx * 8;
        uint32_t *c = reinterpret_cast<uint32_t *>(t + 256 + i * 16);
        if (PACK) {
            const uint2 a = *reinterpret_cast<const uint2 *>(pu), b = *reinterpret_cast<const uint2 *>(pv);
            c[0] = bperm(b.x, a.x, 0x05010400u);      // u0 v0 u1 v1
            c[1] = bperm(b.x, a.x, 0x07030602u);      // u2 v2 u3 v3
            c[2] = bperm(b.y, a.y, 0x05010400u);
            c[3] = bperm(b.y, a.y, 0x07030602u);
        } else {
            const uint32_t c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
            *reinterpret_cast<uint2 *>(pu) = make_uint2(bperm(c1, c0, 0x06040200u), bperm(c3, c2, 0x06040200u));
            *reinterpret_cast<uint2 *>(pv) = make_uint2(bperm(c1, c0, 0x07050301u), bperm(c3, c2, 0x07050301u));
        }
    }
}

// residual[block][r][c] = cur - pred on tile luma; one thread per 16-pixel row segment.
// LOGB = log2(block edge): 3 (8x8, the SATD blocks) or 5 (32x32, the DCT blocks)
template <int LOGB>
__global__ __launch_bounds__(256) void residual_luma_kernel(const x266_ref_block_t *__restrict__ cur,
                                                            const x266_ref_block_t *__restrict__ pred,
                                                            int16_t *__restrict__ res, int tiles_x, size_t n_rows)
{
    constexpr int B = 1 << LOGB;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // (tile, luma row)
    if (id >= n_rows) return;
    const size_t tile = id >> 4;
    const int i = (int)(id & 15);
    const size_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const v4i a = *reinterpret_cast<const v4i *>(reinterpret_cast<const uint8_t *>(cur + tile) + i * 16);
    const v4i b = *reinterpret_cast<const v4i *>(reinterpret_cast<const uint8_t *>(pred + tile) + i * 16);
    uint32_t d[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t x = (uint32_t)a[k], z = (uint32_t)b[k];
        const int d0 = (int)(x & 255) - (int)(z & 255), d1 = (int)((x >> 8) & 255) - (int)((z >> 8) & 255);
        const int d2 = (int)((x >> 16) & 255) - (int)((z >> 16) & 255), d3 = (int)(x >> 24) - (int)(z >> 24);
        d[2 * k] = ((uint32_t)d0 & 0xFFFFu) | ((uint32_t)d1 << 16);
        d[2 * k + 1] = ((uint32_t)d2 & 0xFFFFu) | ((uint32_t)d3 << 16);
    }
    const size_t py = ty * 16 + i, px = tx * 16;                           // pixel coordinates of this segment
    const size_t width = (size_t)tiles_x * 16;
    if (B == 32) {
        const size_t blk = (py >> 5) * (width >> 5) + (px >> 5);
        int16_t *dst = res + blk * 1024 + (py & 31) * 32 + (px & 31);
        *reinterpret_cast<v4i *>(dst) = v4i{(int)d[0], (int)d[1], (int)d[2], (int)d[3]};
        *reinterpret_cast<v4i *>(dst + 8) = v4i{(int)d[4], (int)d[5], (int)d[6], (int)d[7]};
    } else {                                                               // two 8x8 blocks side by side
        const size_t blk = (py >> 3) * (width >> 3) + (px >> 3);
        int16_t *dst = res + blk * 64 + (py & 7) * 8;
        *reinterpret_cast<v4i *>(dst) = v4i{(int)d[0], (int)d[1], (int)d[2], (int)d[3]};
        *reinterpret_cast<v4i *>(dst + 64) = v4i{(int)d[4], (int)d[5], (int)d[6], (int)d[7]};
    }
}

}  // namespace

hipError_t launch_tile_convert(bool pack, x266_ref_block_t *d_tiles, uint8_t *d_y, uint8_t *d_u, uint8_t *d_v,
                               long long strd_y, long long strd_c, int width, int height, hipStream_t stream)
{
    const int tiles_x = width / 16;
    const size_t n_rows = (size_t)tiles_x * (height / 16) * 16;
    if (n_rows == 0) return hipSuccess;
    dim3 grid((unsigned)((n_rows + 255) / 256)), block(256);
    if (pack) hipLaunchKernelGGL((tile_convert_kernel<true>), grid, block, 0, stream, d_tiles, d_y, d_u, d_v, strd_y, strd_c, tiles_x, n_rows);
    else      hipLaunchKernelGGL((tile_convert_kernel<false>), grid, block, 0, stream, d_tiles, d_y, d_u, d_v, strd_y, strd_c, tiles_x, n_rows);
    return hipGetLastError();
}

hipError_t launch_residual_luma(int block_edge, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int16_t *d_res,
                                int width, int height, hipStream_t stream)
{
    const int tiles_x = width / 16;
    const size_t n_rows = (size_t)tiles_x * (height / 16) * 16;
    if (n_rows == 0) return hipSuccess;
    dim3 grid((unsigned)((n_rows + 255) / 256)), block(256);
    if (block_edge == 32) hipLaunchKernelGGL((residual_luma_kernel<5>), grid, block, 0, stream, d_cur, d_pred, d_res, tiles_x, n_rows);
    else                  hipLaunchKernelGGL((residual_luma_kernel<3>), grid, block, 0, stream, d_cur, d_pred, d_res, tiles_x, n_rows);
    return hipGetLastError();
}

}  // namespace x266
