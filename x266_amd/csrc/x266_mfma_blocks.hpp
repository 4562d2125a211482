// x266_mfma_blocks.hpp -- device-side building blocks shared by the transform kernels:
// byte-plane split / re-pack around v_mfma_i32_32x32x32_i8 and the two-pass forward
// transform of one 32x32 tile held in registers (see dct32_kernels.hip for the derivation).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "x266_device.hpp"
#include "x266_tables.hpp"

namespace x266 {

// ---- byte-plane helpers ----------------------------------------------------
// v_perm_b32: result byte i = byte sel[i] of the 8-byte value {hi_src, lo_src}
// (indices 0-3 = lo_src, 4-7 = hi_src).
__device__ __forceinline__ uint32_t bperm(uint32_t hi_src, uint32_t lo_src, uint32_t sel)
{
    return __builtin_amdgcn_perm(hi_src, lo_src, sel);
}

// 8 dwords of int16 pairs -> 4 dwords of low bytes (offset to signed) + 4 of high bytes
__device__ __forceinline__ void split_planes(const v4i &w0, const v4i &w1, v4i &lo, v4i &hi)
{
    const uint32_t w[8] = {(uint32_t)w0[0], (uint32_t)w0[1], (uint32_t)w0[2], (uint32_t)w0[3],
                           (uint32_t)w1[0], (uint32_t)w1[1], (uint32_t)w1[2], (uint32_t)w1[3]};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        lo[p] = (int)(bperm(w[2 * p + 1], w[2 * p], 0x06040200u) ^ 0x80808080u);
        hi[p] = (int)bperm(w[2 * p + 1], w[2 * p], 0x07050301u);
    }
}

// 16 int32 whose bytes 0/1 hold the wanted low/high byte -> byte planes
__device__ __forceinline__ void pack_planes(const v16i &s, v4i &lo, v4i &hi)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t t01 = bperm((uint32_t)s[4 * q + 1], (uint32_t)s[4 * q + 0], 0x05010400u);
        const uint32_t t23 = bperm((uint32_t)s[4 * q + 3], (uint32_t)s[4 * q + 2], 0x05010400u);
        lo[q] = (int)(bperm(t23, t01, 0x05040100u) ^ 0x80808080u);
        hi[q] = (int)bperm(t23, t01, 0x07060302u);
    }
}

// 16 int32 holding one signed byte value each (byte 0) -> one plane of 4 dwords
__device__ __forceinline__ v4i pack_bytes(const v16i &s)
{
    v4i r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t t01 = bperm((uint32_t)s[4 * q + 1], (uint32_t)s[4 * q + 0], 0x0c0c0400u);
        const uint32_t t23 = bperm((uint32_t)s[4 * q + 3], (uint32_t)s[4 * q + 2], 0x0c0c0400u);
        r[q] = (int)bperm(t23, t01, 0x05040100u);
    }
    return r;
}

__device__ __forceinline__ v16i mfma(const v4i &a, const v4i &b, const v16i &c)
{
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
}

struct LaneConsts {
    v4i p1, p2, tr;
    int c1, c2;
};

__device__ __forceinline__ LaneConsts load_consts(const DctOps *ops, int lane)
{
    LaneConsts k;
    const DctLaneOps *r = &ops->lane[lane];
    k.p1 = *reinterpret_cast<const v4i *>(r->p1);
    k.p2 = *reinterpret_cast<const v4i *>(r->p2);
    k.tr = *reinterpret_cast<const v4i *>(r->tr);
    k.c1 = r->c1;
    k.c2 = r->c2;
    return k;
}

// ---- forward: one block held as (w0, w1) -> (o0, o1) ------------------------
// Second half of the forward transform: `acc` holds the pass-1 sums INCLUDING the rounding term;
// shift, re-pack to byte planes, pass 2, final shift and int16 packing.
template <int S1, int S2>
__device__ __forceinline__ void fwd_finish(v16i acc, const LaneConsts &k, v4i &o0, v4i &o1)
{
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc[r] >> S1;        // bytes 0/1 = int16 result
    v4i ylo, yhi;
    pack_planes(acc, ylo, yhi);

    // pass 2 (columns): data = A (pass-1 accumulators re-packed), coefficients = B
    acc = mfma(yhi, k.p2, zero);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = (int)(((uint32_t)acc[r] << 8) + (uint32_t)k.c2);
    acc = mfma(ylo, k.p2, acc);

    // (acc >> S2) truncated to int16, pairs packed into dwords
    uint32_t z[8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
        z[m] = bperm((uint32_t)(acc[2 * m + 1] >> S2), (uint32_t)(acc[2 * m] >> S2), 0x05040100u);
    o0 = v4i{(int)z[0], (int)z[1], (int)z[2], (int)z[3]};
    o1 = v4i{(int)z[4], (int)z[5], (int)z[6], (int)z[7]};
}

template <int S1, int S2>
__device__ __forceinline__ void fwd_block(const v4i &w0, const v4i &w1, const LaneConsts &k,
                                          v4i &o0, v4i &o1)
{
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v4i lo, hi;
    split_planes(w0, w1, lo, hi);

    // pass 1 (rows): data = A, coefficients = B
    v16i acc = mfma(hi, k.p1, zero);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = (int)(((uint32_t)acc[r] << 8) + (uint32_t)k.c1);
    acc = mfma(lo, k.p1, acc);
    fwd_finish<S1, S2>(acc, k, o0, o1);
}

// ---- one 32x32 tile through a wave-private 2 KiB LDS slot (dct32_kernels.hip, "LDS-staged variant") ----------
// Chunk (row r, quarter q) lives at r*64 + ((q ^ ((r >> 2) & 3)) << 4): linear writes, row-per-lane fragment reads and
// the way back are all bank-conflict-free.
__device__ __forceinline__ unsigned lds_slot(unsigned row, unsigned quarter)
{
    return row * 64u + ((quarter ^ ((row >> 2) & 3u)) << 4);
}

// g0 / g1: the lane's two 16-byte pieces of the tile in LINEAR order (bytes lane*16 and 1024 + lane*16);
// returns the transformed tile's pieces in the same linear order.
template <int S1, int S2>
__device__ __forceinline__ void fwd_tile_staged(unsigned char *slot, int lane, const LaneConsts &k, const v4i &g0, const v4i &g1, v4i &s0, v4i &s1)
{
    const unsigned c = lane & 31, h = lane >> 5;
    const unsigned lin0 = lds_slot(lane >> 2, lane & 3), lin1 = lds_slot(16 + (lane >> 2), lane & 3);
    const unsigned frag0 = lds_slot(c, 2 * h), frag1 = lds_slot(c, 2 * h + 1);
    *reinterpret_cast<v4i *>(slot + lin0) = g0;
    *reinterpret_cast<v4i *>(slot + lin1) = g1;
    __builtin_amdgcn_wave_barrier();
    const v4i a0 = *reinterpret_cast<const v4i *>(slot + frag0);
    const v4i a1 = *reinterpret_cast<const v4i *>(slot + frag1);
    v4i o0, o1;
    fwd_block<S1, S2>(a0, a1, k, o0, o1);
    __builtin_amdgcn_wave_barrier();
    *reinterpret_cast<v4i *>(slot + frag0) = o0;
    *reinterpret_cast<v4i *>(slot + frag1) = o1;
    __builtin_amdgcn_wave_barrier();
    s0 = *reinterpret_cast<const v4i *>(slot + lin0);
    s1 = *reinterpret_cast<const v4i *>(slot + lin1);
    __builtin_amdgcn_wave_barrier();
}

// ---- inverse passes (see dct32_kernels.hip, section "inverse") --------------------------------

// {clip16(lo), clip16(hi)} packed into one dword: v_cvt_pk_i16_i32 (full rate, profiles/r01_alubench.txt)
__device__ __forceinline__ uint32_t sat_pack16(int lo, int hi)
{
    typedef short v2s __attribute__((ext_vector_type(2)));
    const v2s r = __builtin_amdgcn_cvt_pk_i16(lo, hi);
    return __builtin_bit_cast(uint32_t, r);
}

// passes A and B on column data: zlo / zhi = byte planes of 16 samples of ONE COLUMN per lane.
// c2r_group(g): the pass-B constants of accumulator registers 4g .. 4g+3 (fetched where they are used: four live, not sixteen)
template <class C2RGroup>
__device__ __forceinline__ void inv_passes_with(const v4i &zlo, const v4i &zhi, const LaneConsts &k,
                                                C2RGroup c2r_group, v4i &o0, v4i &o1)
{
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // pass A (columns): data = A, coefficients = B, per-lane constant
    v16i acc = mfma(zhi, k.p1, zero);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = (int)(((uint32_t)acc[r] << 8) + (uint32_t)k.c1);
    acc = mfma(zlo, k.p1, acc);
    // shift, clip to int16 and re-pack to byte planes: v_cvt_pk_i16_i32 saturates and packs two
    // values per instruction (instead of two v_med3 + byte shuffles on 32-bit values)
    v4i tlo2, thi2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t p01 = sat_pack16(acc[4 * q + 0] >> 7, acc[4 * q + 1] >> 7);
        const uint32_t p23 = sat_pack16(acc[4 * q + 2] >> 7, acc[4 * q + 3] >> 7);
        tlo2[q] = (int)(bperm(p23, p01, 0x06040200u) ^ 0x80808080u);
        thi2[q] = (int)bperm(p23, p01, 0x07050301u);
    }

    // pass B (rows): coefficients = A, data = B, per-register constant
    acc = mfma(k.p2, thi2, zero);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const v4i c = c2r_group(g);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * g + i] = (int)(((uint32_t)acc[4 * g + i] << 8) + (uint32_t)c[i]);
    }
    acc = mfma(k.p2, tlo2, acc);

    uint32_t z[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) z[m] = sat_pack16(acc[2 * m] >> 12, acc[2 * m + 1] >> 12);
    o0 = v4i{(int)z[0], (int)z[1], (int)z[2], (int)z[3]};
    o1 = v4i{(int)z[4], (int)z[5], (int)z[6], (int)z[7]};
}

__device__ __forceinline__ void inv_passes(const v4i &zlo, const v4i &zhi, const LaneConsts &k,
                                           const v16i &c2r, v4i &o0, v4i &o1)
{
    inv_passes_with(zlo, zhi, k, [&](int g) { return v4i{c2r[4 * g], c2r[4 * g + 1], c2r[4 * g + 2], c2r[4 * g + 3]}; }, o0, o1);
}

}  // namespace x266
