// x266_hadamard.hpp -- the 8x8 Hadamard transform of 32 pixel windows on the int8 matrix core,
// shared by the motion search (me_kernels.hip) and the intra mode decision (intra_kernels.hip).
#pragma once

#include "x266_mfma_blocks.hpp"

namespace x266 {

// +-1 operand images of H64[m][s] = (-1)^popcount(m & s), built from the lane index
// (same construction as satd_kernels.hip; m = 32*tile + (lane & 31),
//  s = 32*(lane >> 5) + 16*step + t).
struct HadamardOps { v4i t0s0, t0s1, t1s0, t1s1; };

__device__ __forceinline__ HadamardOps make_hadamard_ops(int lane)
{
    const uint32_t NEG = 0xFEFEFEFEu;
    const uint32_t m = (uint32_t)lane & 31u, half = (uint32_t)lane >> 5;
    const uint32_t inner = (m & 1) ? ((m & 2) ? 0x01FFFF01u : 0xFF01FF01u) : ((m & 2) ? 0xFFFF0101u : 0x01010101u);
    const uint32_t f2 = (m & 4) ? NEG : 0u, f3 = (m & 8) ? NEG : 0u, f4 = (m & 16) ? NEG : 0u, fh = half ? NEG : 0u;
    const uint32_t b0 = inner, b1 = inner ^ f2, b2 = inner ^ f3, b3 = inner ^ f2 ^ f3;
    HadamardOps o;
    o.t0s0 = v4i{(int)b0, (int)b1, (int)b2, (int)b3};
    o.t0s1 = v4i{(int)(b0 ^ f4), (int)(b1 ^ f4), (int)(b2 ^ f4), (int)(b3 ^ f4)};
    o.t1s0 = v4i{(int)(b0 ^ fh), (int)(b1 ^ fh), (int)(b2 ^ fh), (int)(b3 ^ fh)};
    o.t1s1 = o.t0s1 ^ v4i{(int)fh, (int)fh, (int)fh, (int)fh};
    return o;
}

// 64 coefficients of 32 windows -> per lane 16 dwords of biased uint16 pairs.
// b0 / b1: the lane's half of the window, rows (4h, 4h+1) and (4h+2, 4h+3), 8 signed pixels each.
__device__ __forceinline__ void hadamard_pack(const HadamardOps &H, const v4i &b0, const v4i &b1, uint32_t (&p)[16])
{
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v16i a0 = mfma(H.t0s0, b0, zero);
    a0 = mfma(H.t0s1, b1, a0);
    v16i a1 = mfma(H.t1s0, b0, zero);
    a1 = mfma(H.t1s1, b1, a1);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        p[k]     = bperm((uint32_t)a0[2 * k + 1], (uint32_t)a0[2 * k], 0x05040100u) ^ 0x80008000u;
        p[8 + k] = bperm((uint32_t)a1[2 * k + 1], (uint32_t)a1[2 * k], 0x05040100u) ^ 0x80008000u;
    }
}


}  // namespace x266
