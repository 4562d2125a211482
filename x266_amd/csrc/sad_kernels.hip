// sad_kernels.hip -- batched sum of absolute differences of square 8-bit blocks for gfx950
// (SURVEY.md section 8 f3).
//
// Reference: sad(), riscv/programs/benchmarks/sad/sad.c:28-39 -- sum over an n x n block of
// |a[i][j] - b[i][j]| on unsigned bytes, int accumulator; the benchmark ships one known answer
// (64 x 64, 344807, riscv/programs/benchmarks/sad/dataset1.h:423-426), which the tests replay.
//
// Mapping: HBM-bound (2*n*n bytes in, 4 out per block pair).  Blocks are n*n contiguous bytes; a block is cut into 16-byte chunks,
// one v_sad_u8 chain (4 bytes per instruction, 4 per chunk) per chunk, chunk = lane for 1 KiB-linear loads; the partial sums of a
// block's chunks meet in DPP steps inside the row of 16 lanes (and four v_readlane for blocks of 64 chunks and more; blocks never
// straddle a wave).  Launch shape = the read stream's (diag_kernels.hip): every wave issues its 2 x STEPS loads up front (4 KiB of each
// input at STEPS = 4), four-wave workgroups charged 32 KiB of LDS each = twenty resident waves per CU
// (profiles/r04_sad_shapes.txt: 8x8 0.0886 -> 0.0842 ms, 16x16 0.0878 -> 0.0789 on a 16384 x 16384 frame pair).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "x266_device.hpp"

namespace x266 {
namespace {

__device__ __forceinline__ uint32_t sad_chunk(const v4i &a, const v4i &b, uint32_t s)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) s = __builtin_amdgcn_sad_u8((uint32_t)a[k], (uint32_t)b[k], s);
    return s;
}

// sum over aligned groups of SPAN lanes (every lane of the group ends up with it; SPAN = 64: wave-uniform)
template <int SPAN>
__device__ __forceinline__ uint32_t span_sum(uint32_t x)
{
    if (SPAN >= 2) x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
    if (SPAN >= 4) x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    if (SPAN >= 8) x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true);    // row_half_mirror
    if (SPAN >= 16) x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x140, 0xF, 0xF, true);   // row_mirror
    if (SPAN >= 64) x = (uint32_t)(__builtin_amdgcn_readlane((int)x, 0) + __builtin_amdgcn_readlane((int)x, 16) + __builtin_amdgcn_readlane((int)x, 32) + __builtin_amdgcn_readlane((int)x, 48));
    return x;
}

// LOGC = log2(16-byte chunks per block): 0 (4x4), 2 (8x8), 4 (16x16), 6 (32x32), 8 (64x64); STEPS = 1 KiB-linear loads per input and wave
template <int LOGC, int STEPS>
__global__ __launch_bounds__(256) void sad_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                                  uint32_t *__restrict__ out, size_t n_blocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char occupancy_cap[];   // never touched: only its size matters
    constexpr int CPB = 1 << LOGC;                          // chunks per block
    constexpr int SPAN = CPB > 64 ? 64 : CPB;               // lanes that share a block
    static_assert(CPB <= 64 * STEPS, "a block must not straddle waves");
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t chunk0 = wave * (64 * STEPS);
    const size_t total = n_blocks * CPB;
    if (chunk0 >= total) return;
    v4i va[STEPS], vb[STEPS];
    if (chunk0 + 64 * STEPS <= total) {                     // wave-uniform: every wave but the batch's last, no per-lane conditions
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            va[i] = load16<true>(a + (chunk0 + 64 * i + lane) * 16);       // line-dense, read once: streaming hint
            vb[i] = load16<true>(b + (chunk0 + 64 * i + lane) * 16);
        }
    } else {                                                // ragged tail: chunks past the end read the last chunk of a twice (SAD 0, never stored)
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            size_t c = chunk0 + 64 * i + lane;
            const bool live = c < total;
            if (!live) c = total - 1;
            va[i] = load16<true>(a + c * 16);
            vb[i] = live ? load16<true>(b + c * 16) : va[i];
        }
    }
    if (CPB > 64) {                                          // one block spans CPB / 64 steps
        constexpr int PER = CPB / 64;
#pragma unroll
        for (int g = 0; g < STEPS / PER; ++g) {
            uint32_t s = 0;
#pragma unroll
            for (int i = 0; i < PER; ++i) s = sad_chunk(va[g * PER + i], vb[g * PER + i], s);
            s = span_sum<64>(s);
            const size_t blk = chunk0 / CPB + g;
            if (lane == 0 && blk < n_blocks) store_result4(out + blk, s);
        }
    } else {
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const uint32_t s = span_sum<SPAN>(sad_chunk(va[i], vb[i], 0));
            const size_t blk = (chunk0 + 64 * i + lane) / CPB;
            if ((lane & (SPAN - 1)) == 0 && blk < n_blocks) {
                // agent-scope result stores (x266_device.hpp) where they pay, paired A/B on four boxes (tools/gpu_ab_sad.py): 4x4 -1 to -3 %, 8x8 -3 to -4.5 %,
                // 64x64 (above) -1 to -2.5 %; 16x16 0 to +3 % and 32x32 +4 to +6.5 % keep the plain store
                if (LOGC <= 2) store_result4(out + blk, s);
                else           out[blk] = s;
            }
        }
    }
}

}  // namespace

// waves_per_wg / lds_per_wg: 0 = the defaults below; other values are the autotuner's candidates (x266hip_abi.hip)
hipError_t launch_sad(int edge, const uint8_t *d_a, const uint8_t *d_b, uint32_t *d_out, size_t n_blocks, int waves_per_wg, int lds_per_wg, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    if (edge != 4 && edge != 8 && edge != 16 && edge != 32 && edge != 64) return hipErrorInvalidValue;
    const size_t chunks = n_blocks * (size_t)(edge * edge / 16);
    // 4x4 (an eighth of its traffic is results): one-wave workgroups of 1 KiB per input, 8 KiB charged; the rest: the read stream's shape
    const int steps = edge == 4 ? 1 : 4;
    const int wpw = edge == 4 ? 1 : (waves_per_wg > 0 ? waves_per_wg : 4);
    const size_t lds = edge == 4 ? 8192 : (lds_per_wg > 0 ? (size_t)lds_per_wg : 32768);
    if (wpw > 4 || lds > 65536) return hipErrorInvalidValue;
    const size_t waves = (chunks + 64 * steps - 1) / (64 * steps);
    const size_t wgs = (waves + wpw - 1) / wpw;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    dim3 grid((unsigned)wgs), block(64 * wpw);
    switch (edge) {
    case 4:  hipLaunchKernelGGL((sad_kernel<0, 1>), grid, block, lds, stream, d_a, d_b, d_out, n_blocks); break;
    case 8:  hipLaunchKernelGGL((sad_kernel<2, 4>), grid, block, lds, stream, d_a, d_b, d_out, n_blocks); break;
    case 16: hipLaunchKernelGGL((sad_kernel<4, 4>), grid, block, lds, stream, d_a, d_b, d_out, n_blocks); break;
    case 32: hipLaunchKernelGGL((sad_kernel<6, 4>), grid, block, lds, stream, d_a, d_b, d_out, n_blocks); break;
    default: hipLaunchKernelGGL((sad_kernel<8, 4>), grid, block, lds, stream, d_a, d_b, d_out, n_blocks); break;
    }
    return hipGetLastError();
}

}  // namespace x266
