// dct32_butterfly.hip -- the forward DCT32 with the 1-D passes on the VECTOR ALU in the reference's
// own even/odd decomposition (partialButterfly32, src_tb/dct32.c:66-170: 344 multiplies and about
// 400 additions per 32-point row instead of 1024 multiply-adds), kept ONLY as the comparison the
// task statement asks for ("the 1-D pass is tiled onto MFMA where rocprof shows it beats the
// butterfly"): xHipSetOption(ctx, "dct32_variant", 2).  Same results, bit for bit.
//
// One lane owns one 32-sample line, a wave two blocks (2 x 32 lanes); the corner turn between the
// passes (mkTranspose.bsv's job in the RTL) goes through a 2 KiB LDS tile per block.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "x266_device.hpp"
#include "x266_tables.hpp"

namespace x266 {
namespace {

// 32-point forward transform of one line: out[k] = (sum_n g[k][n] x[n] + round) >> SHIFT, truncated to
// int16 on store by the caller.  Coefficients are compile-time constants (literal operands).
template <int SHIFT>
__device__ __forceinline__ void butterfly32(const int (&x)[32], int (&out)[32])
{
    constexpr int R = 1 << (SHIFT - 1);
    int E[16], O[16], EE[8], EO[8], EEE[4], EEO[4], EEEE[2], EEEO[2];
#pragma unroll
    for (int k = 0; k < 16; ++k) { E[k] = x[k] + x[31 - k]; O[k] = x[k] - x[31 - k]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) { EE[k] = E[k] + E[15 - k]; EO[k] = E[k] - E[15 - k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { EEE[k] = EE[k] + EE[7 - k]; EEO[k] = EE[k] - EE[7 - k]; }
    EEEE[0] = EEE[0] + EEE[3]; EEEO[0] = EEE[0] - EEE[3];
    EEEE[1] = EEE[1] + EEE[2]; EEEO[1] = EEE[1] - EEE[2];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        int acc = R;
        if ((k & 15) == 0) {                                 // rows 0, 16: two taps on EEEE
#pragma unroll
            for (int n = 0; n < 2; ++n) acc += coef32(k, n) * EEEE[n];
        } else if ((k & 7) == 0) {                           // rows 8, 24: two taps on EEEO
#pragma unroll
            for (int n = 0; n < 2; ++n) acc += coef32(k, n) * EEEO[n];
        } else if ((k & 3) == 0) {                           // rows 4, 12, 20, 28: four taps on EEO
#pragma unroll
            for (int n = 0; n < 4; ++n) acc += coef32(k, n) * EEO[n];
        } else if ((k & 1) == 0) {                           // rows 2, 6, ..., 30: eight taps on EO
#pragma unroll
            for (int n = 0; n < 8; ++n) acc += coef32(k, n) * EO[n];
        } else {                                             // odd rows: sixteen taps on O
#pragma unroll
            for (int n = 0; n < 16; ++n) acc += coef32(k, n) * O[n];
        }
        out[k] = acc >> SHIFT;
    }
}

__device__ __forceinline__ void unpack_line(const v4i (&v)[4], int (&x)[32])
{
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            x[8 * q + 2 * d] = (int)(int16_t)(v[q][d] & 0xFFFF);
            x[8 * q + 2 * d + 1] = v[q][d] >> 16;
        }
}

__global__ __launch_bounds__(256) void dct32_butterfly_kernel(const int16_t *__restrict__ in, int16_t *__restrict__ out,
                                                              size_t n_blocks)
{
    __shared__ __attribute__((aligned(16))) int16_t tiles[8][1024];      // 2 blocks per wave, 4 waves
    const int lane = threadIdx.x & 63, line = lane & 31;
    const size_t blk = (((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 2 + (lane >> 5);
    const bool live = blk < n_blocks;
    const size_t b = live ? blk : n_blocks - 1;
    int16_t *tile = tiles[(threadIdx.x >> 6) * 2 + (lane >> 5)];

    v4i v[4];
    const v4i *src = reinterpret_cast<const v4i *>(in + b * 1024 + line * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = src[q];
    int x[32], y[32];
    unpack_line(v, x);
    butterfly32<4>(x, y);                                    // rows: coef[k][j], transposed store
#pragma unroll
    for (int k = 0; k < 32; ++k) tile[k * 32 + line] = (int16_t)y[k];
    __builtin_amdgcn_wave_barrier();
    const v4i *t = reinterpret_cast<const v4i *>(tile + line * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = t[q];
    __builtin_amdgcn_wave_barrier();
    unpack_line(v, x);
    butterfly32<11>(x, y);                                   // columns: dct[k2][k], transposed store again
#pragma unroll
    for (int k = 0; k < 32; ++k) tile[k * 32 + line] = (int16_t)y[k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = t[q];
    if (live) {
        v4i *dst = reinterpret_cast<v4i *>(out + b * 1024 + line * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = v[q];
    }
}

}  // namespace

hipError_t launch_dct32_butterfly(const int16_t *d_in, int16_t *d_out, size_t n_blocks, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    const size_t wgs = (n_blocks + 7) / 8;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dct32_butterfly_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, d_in, d_out, n_blocks);
    return hipGetLastError();
}

}  // namespace x266
