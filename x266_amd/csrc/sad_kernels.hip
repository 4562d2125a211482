// sad_kernels.hip -- batched sum of absolute differences of square 8-bit blocks for gfx950
// (SURVEY.md section 8 f3).
//
// Reference: sad(), riscv/programs/benchmarks/sad/sad.c:28-39 -- sum over an n x n block of
// |a[i][j] - b[i][j]| on unsigned bytes, int accumulator; the benchmark ships one known answer
// (64 x 64, 344807, riscv/programs/benchmarks/sad/dataset1.h:423-426), which the tests replay.
//
// Mapping: HBM-bound (2*n*n bytes in, 4 out per block pair).  Blocks are n*n contiguous bytes;
// a block is cut into 16-byte chunks, one v_sad_u8 chain (4 bytes per instruction, 4 per chunk)
// per chunk, chunk = lane for linear 1 KiB loads; the partial sums of a block's chunks meet in a
// butterfly of DPP-free __shfl_xor steps (blocks never straddle a wave).
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

// LOGC = log2(16-byte chunks per block): 0 (4x4), 2 (8x8), 4 (16x16), 6 (32x32), 8 (64x64)
template <int LOGC>
__global__ __launch_bounds__(256) void sad_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b,
                                                  uint32_t *__restrict__ out, size_t n_blocks)
{
    constexpr int CPB = 1 << LOGC;                          // chunks per block
    constexpr int ITER = CPB > 64 ? CPB / 64 : 1;           // wave-instructions per block (64x64: 4)
    constexpr int BPW = CPB > 64 ? 1 : 64 / CPB;            // blocks per wave step
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t first = wave * BPW;                        // first block of this wave
    if (first >= n_blocks) return;
    const size_t total = n_blocks * (size_t)CPB * 16;
    uint32_t s = 0;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        size_t off = (first * CPB + (size_t)it * 64 + lane) * 16;
        const bool live = off + 16 <= total;
        if (!live) off = total - 16;                        // ragged tail
        const v4i va = load16<true>(a + off), vb = load16<true>(b + off);      // line-dense, read once: streaming hint
        const uint32_t part = sad_chunk(va, vb, 0);
        s += live ? part : 0u;
    }
    constexpr int SPAN = CPB > 64 ? 64 : CPB;               // lanes that share a block
#pragma unroll
    for (int m = SPAN >> 1; m >= 1; m >>= 1) s += (uint32_t)__shfl_xor((int)s, m);
    const size_t blk = first + (SPAN == 64 ? 0 : lane / SPAN);
    if ((lane & (SPAN - 1)) == 0 && blk < n_blocks) out[blk] = s;
}

}  // namespace

hipError_t launch_sad(int edge, const uint8_t *d_a, const uint8_t *d_b, uint32_t *d_out, size_t n_blocks, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    const int cpb = edge * edge / 16;
    const size_t bpw = cpb > 64 ? 1 : (size_t)(64 / cpb);
    const size_t waves = (n_blocks + bpw - 1) / bpw;
    const size_t wgs = (waves + 3) / 4;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    dim3 grid((unsigned)wgs), block(256);
    switch (edge) {
    case 4:  hipLaunchKernelGGL((sad_kernel<0>), grid, block, 0, stream, d_a, d_b, d_out, n_blocks); break;
    case 8:  hipLaunchKernelGGL((sad_kernel<2>), grid, block, 0, stream, d_a, d_b, d_out, n_blocks); break;
    case 16: hipLaunchKernelGGL((sad_kernel<4>), grid, block, 0, stream, d_a, d_b, d_out, n_blocks); break;
    case 32: hipLaunchKernelGGL((sad_kernel<6>), grid, block, 0, stream, d_a, d_b, d_out, n_blocks); break;
    case 64: hipLaunchKernelGGL((sad_kernel<8>), grid, block, 0, stream, d_a, d_b, d_out, n_blocks); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace x266
