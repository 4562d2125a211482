// satd_butterfly.hip -- the 8x8 Hadamard SATD as radix-2 butterflies on the vector ALU (packed int16,
// the stage structure of satd8x8, src_tb/satd.c:38-103), one lane per block: the comparison variant
// for the matrix-core kernel of satd_kernels.hip, xHipSetOption(ctx, "satd_variant", 2).  Bit-identical:
// add/sub chains are order-independent modulo 2^16, which is exactly satd.c's int16 wraparound.
//
// A wave takes 64 consecutive blocks (8 KiB): eight 1 KiB-linear nontemporal loads into a wave-private
// LDS slot, then every lane reads its own block (row j of block n lives at n*128 + ((j ^ (n & 7)) << 4)).
// Row pass: distances 4 and 2 are packed adds between a row's dwords, distance 1 swaps the halves of a
// dword; column pass: packed adds between rows.  |.| and the sum: v_sad_u16 against the bias.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "x266_device.hpp"

namespace x266 {
namespace {

typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s as_v2s(uint32_t v) { return __builtin_bit_cast(v2s, v); }
__device__ __forceinline__ uint32_t as_u32(v2s v) { return __builtin_bit_cast(uint32_t, v); }

// (p, q) -> (p + q, p - q)
__device__ __forceinline__ uint32_t pair_butterfly(uint32_t d)
{
    const uint32_t sw = __builtin_amdgcn_alignbit(d, d, 16u);                   // (q, p)
    const uint32_t sum = as_u32(as_v2s(d) + as_v2s(sw)), dif = as_u32(as_v2s(d) - as_v2s(sw));
    return __builtin_amdgcn_perm(dif, sum, 0x05040100u);                        // low half of sum, low half of dif
}

__global__ __launch_bounds__(256) void satd8x8_butterfly_kernel(const int16_t *__restrict__ diff, uint32_t *__restrict__ out,
                                                                size_t n_blocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];       // 8 KiB per wave
    const int lane = threadIdx.x & 63;
    unsigned char *slot = stage + (threadIdx.x >> 6) * 8192;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t first = wave * 64;
    if (first >= n_blocks) return;
    const size_t total_bytes = n_blocks * 128;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const unsigned c = (unsigned)lane + 64u * k, n = c >> 3, j = c & 7;
        size_t off = first * 128 + (size_t)c * 16;
        if (off + 16 > total_bytes) off = total_bytes - 16;                     // ragged tail: stay inside the buffer
        *reinterpret_cast<v4i *>(slot + n * 128 + ((j ^ (n & 7)) << 4)) = load16<true>(reinterpret_cast<const char *>(diff) + off);
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t r[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const v4i v = *reinterpret_cast<const v4i *>(slot + lane * 128 + ((j ^ (lane & 7)) << 4));
        // row pass: distance 4, distance 2, distance 1
        const uint32_t a0 = as_u32(as_v2s((uint32_t)v[0]) + as_v2s((uint32_t)v[2])), a1 = as_u32(as_v2s((uint32_t)v[1]) + as_v2s((uint32_t)v[3]));
        const uint32_t a2 = as_u32(as_v2s((uint32_t)v[0]) - as_v2s((uint32_t)v[2])), a3 = as_u32(as_v2s((uint32_t)v[1]) - as_v2s((uint32_t)v[3]));
        r[j][0] = pair_butterfly(as_u32(as_v2s(a0) + as_v2s(a1)));
        r[j][1] = pair_butterfly(as_u32(as_v2s(a0) - as_v2s(a1)));
        r[j][2] = pair_butterfly(as_u32(as_v2s(a2) + as_v2s(a3)));
        r[j][3] = pair_butterfly(as_u32(as_v2s(a2) - as_v2s(a3)));
    }
    // column pass: distance 4, 2, 1 between rows
#pragma unroll
    for (int d = 4; d >= 1; d >>= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (!(j & d))
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const v2s x = as_v2s(r[j][q]), y = as_v2s(r[j + d][q]);
                    r[j][q] = as_u32(x + y);
                    r[j + d][q] = as_u32(x - y);
                }
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) sum = __builtin_amdgcn_sad_u16(r[j][q] ^ 0x80008000u, 0x80008000u, sum);
    const size_t b = first + lane;
    if (b < n_blocks) out[b] = (sum + 2) >> 2;
}

}  // namespace

hipError_t launch_satd8x8_butterfly(const int16_t *d_diff, uint32_t *d_out, size_t n_blocks, const LaunchCfg &cfg, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    const unsigned tpb = cfg.wg_threads > 0 ? (unsigned)cfg.wg_threads : 128u;
    const size_t wpw = tpb / 64, waves = (n_blocks + 63) / 64, wgs = (waves + wpw - 1) / wpw;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const size_t per_wave = cfg.lds_bytes_per_wave < 8192 ? 8192 : (size_t)cfg.lds_bytes_per_wave;
    // the slots are addressed at 8 KiB per wave; extra bytes only cap the resident waves per CU
    hipLaunchKernelGGL(satd8x8_butterfly_kernel, dim3((unsigned)wgs), dim3(tpb), wpw * per_wave, stream, d_diff, d_out, n_blocks);
    return hipGetLastError();
}

}  // namespace x266
