// dct32_kernels.hip -- batched 32x32 integer DCT-II (forward / inverse) for
// gfx950 (MI355X, CDNA4).  Hand-written for wave64 + v_mfma_i32_32x32x32_i8.
//
// Arithmetic contract (bit-exact): partialButterfly32 called twice with shifts
// 4 and 11 (src_tb/dct32.c:66-170,180-198):
//     Y[j][k] = (int16)((sum_n g[k][n] X[j][n] + 8)    >> 4)
//     Z[v][k] = (int16)((sum_j g[v][j] Y[j][k] + 1024) >> 11)
// The reference evaluates each sum with an even/odd butterfly; the dense
// contraction is the same integer, so each 1-D pass is a 32x32x32 integer GEMM
// against a constant matrix (the RTL already computes it as dense 16-tap MACs,
// src/mkDct32.bsv:107-129).
//
// Mapping:
//  * one 32x32 block per wavefront per iteration; lane l loads 32 contiguous
//    bytes: row (l & 31), columns 16*(l >> 5) .. +15.
//  * int16 data x int8 coefficients on the int8 matrix core: the data is split
//    into byte planes, x = 256*hi + (lo ^ 0x80) + 128 (hi signed, lo offset to
//    signed); two chained MFMAs per pass, acc = (mfma(hi) << 8) + c, then
//    acc = mfma(lo', acc).  The "+128" becomes 128*sum_n g[k][n], folded into
//    the per-lane constant c together with the rounding term.
//  * no transpose between the passes: a lane's 16 pass-1 accumulators are 16
//    rows of one column -- exactly the fragment shape of an MFMA *input*
//    operand.  They are re-packed to bytes in registers and fed to pass 2,
//    whose constant operand has its K-slots permuted to the accumulator row
//    order (the contraction does not care about order).  No shuffles, no
//    barriers between the passes; the RTL's BRAM corner-turn
//    (src/mkDct32.bsv:176-210,287-325, src/mkTranspose.bsv) has no counterpart.
//    (LDS is used by the default kernels further down, but only as a layout
//    converter so that global loads and stores are 1 KiB-linear.)
//  * the lane<->frequency assignment of the constant operands is chosen so that
//    each lane finishes with 16 consecutive coefficients of one output row:
//    two 16-byte stores per lane, same address pattern as the loads.
//
// Kernels in this file: dct32_lds_kernel (LDS-staged line-dense traffic, forward / inverse),
// dct32_fwdinv_lds_kernel (coefficients + reconstruction in one pass), dct32_from_tiles_kernel
// (residual formation fused in), dct32_pass_kernel (the 1-D pass by itself, for checking).  The direct
// fragment-load forms and the variants without cache-policy hints of rounds 1-3 are gone: every A/B they
// served is frozen in profiles/r01_*.txt (line-dense traffic +9 %, "nt" loads / "sc1 nt" stores +3-5 %).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "x266_device.hpp"
#include "x266_mfma_blocks.hpp"
#include "x266_tables.hpp"

namespace x266 {
namespace {

// ---- inverse ---------------------------------------------------------------
// T[u][y] = clip16((sum_v g[v][y] Z[v][u] + 64)   >> 7)      (columns first)
// R[y][x] = clip16((sum_u g[u][x] T[u][y] + 2048) >> 12)
// The first contraction runs over the ROW index of the loaded block, which a row-per-lane fragment cannot
// feed: with the tile staged in LDS each lane reads its COLUMN instead (see dct32_lds_kernel).

// ---- kernels ---------------------------------------------------------------
// Each wave transforms blocks_per_wave consecutive blocks, one wave per chunk, a grid as large as
// the batch: the hardware dispatcher then walks the batch in address order,
// which is what HBM likes best (a plain one-element-per-thread copy is ~15 % faster on this chip than
// any persistent grid-stride copy).  The next block's loads are issued before the current block's arithmetic.

// ---- LDS-staged variant -------------------------------------------------------
// Same arithmetic; the difference is what the memory system sees.  A row-per-lane
// fragment load touches only 16 bytes of every 32 per instruction; measured on this
// chip, load/store instructions that cover whole 128-byte lines run ~9 % faster
// (DCT-II 8x8 tiles, whose fragment loads are line-dense, reach 6.0 TB/s where the
// 32x32 fragment pattern reaches 5.4-5.5 on the same box).  So each wave moves its
// tile with fully linear 1 KiB instructions and turns it into fragment order through
// a private 2 KiB LDS slot: linear global load -> ds_write -> ds_read (row-per-lane)
// -> transform -> ds_write (row-per-lane) -> ds_read (linear) -> linear global store.
// Chunk (row r, quarter q) lives at r*64 + ((q ^ ((r >> 2) & 3)) << 4): every one of
// the four DS accesses is bank-conflict-free.  Waves never share a slot, so there is
// no barrier, only the in-order LDS queue of the wave itself.
// (lds_slot itself lives in x266_mfma_blocks.hpp: the fused frame kernel of satd_kernels.hip stages DCT32 tiles the same way)

template <bool INVERSE>
__global__ __launch_bounds__(256) void dct32_lds_kernel(const int16_t *__restrict__ in,
                                                        int16_t *__restrict__ out, size_t n_blocks,
                                                        const DctOps *__restrict__ ops,
                                                        unsigned blocks_per_wave)
{
    // dynamic LDS: one 2 KiB slot per wave, plus whatever the launcher adds to cap the number of
    // resident waves per CU (fewer, smaller workgroups stream better: profiles/r01_wg_occupancy.txt)
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];
    const int lane = threadIdx.x & 63;
    unsigned char *slot = stage + (threadIdx.x >> 6) * 2048;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    size_t b = wave * blocks_per_wave;
    const size_t end = b + blocks_per_wave < n_blocks ? b + blocks_per_wave : n_blocks;
    if (b >= end) return;

    const unsigned c = lane & 31, h = lane >> 5;
    // this lane's two linear chunks (16 B each) and its two fragment chunks
    const unsigned lin0 = lds_slot(lane >> 2, lane & 3), lin1 = lds_slot(16 + (lane >> 2), lane & 3);
    const unsigned frag0 = lds_slot(c, 2 * h), frag1 = lds_slot(c, 2 * h + 1);
    // inverse: byte offset of element (row 16h + t, column u) is col_base[(t >> 2) & 3] + 64 t, u = kappa(c)
    unsigned col_base[4];
    {
        const unsigned u = (unsigned)kappa((int)c);
#pragma unroll
        for (unsigned j = 0; j < 4; ++j) col_base[j] = 16u * h * 64u + ((((u >> 3) ^ j) & 3u) << 4) + (u & 7u) * 2u;
    }
    const char *src = reinterpret_cast<const char *>(in) + lane * 16;
    char *dst = reinterpret_cast<char *>(out) + lane * 16;

    v4i g0 = load16<true>(src + b * 2048), g1 = load16<true>(src + b * 2048 + 1024);
    const LaneConsts k = load_consts(ops, lane);
    v16i c2r;
    if (INVERSE) {
        // the pass-B constants depend on (half, register) only: two scalar loads (wave-uniform
        // addresses) and a per-lane select instead of 64 bytes of vector loads per lane and wave
        const int *__restrict__ s0 = ops->c2r[0], *__restrict__ s1 = ops->c2r[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) c2r[r] = h ? s1[r] : s0[r];
    }
    while (true) {
        const size_t nb = b + 1;
        *reinterpret_cast<v4i *>(slot + lin0) = g0;
        *reinterpret_cast<v4i *>(slot + lin1) = g1;
        if (nb < end) {                                        // next tile's loads fly under this tile's arithmetic
            g0 = load16<true>(src + nb * 2048);
            g1 = load16<true>(src + nb * 2048 + 1024);
        }
        __builtin_amdgcn_wave_barrier();
        v4i o0, o1;
        if (INVERSE) {
            // The inverse contracts over the block's ROW index first: with the tile in LDS the lane
            // simply reads its COLUMN (16 x ds_read_u16).  Lane (c, h): column kappa(c), rows 16h..16h+15.
            uint32_t w[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const uint32_t e0 = *reinterpret_cast<const uint16_t *>(slot + col_base[((2 * m) >> 2) & 3] + (2 * m) * 64);
                const uint32_t e1 = *reinterpret_cast<const uint16_t *>(slot + col_base[((2 * m + 1) >> 2) & 3] + (2 * m + 1) * 64);
                w[m] = e0 | (e1 << 16);
            }
            v4i lo, hi;
            split_planes(v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]}, v4i{(int)w[4], (int)w[5], (int)w[6], (int)w[7]}, lo, hi);
            inv_passes(lo, hi, k, c2r, o0, o1);
        } else {
            const v4i a0 = *reinterpret_cast<const v4i *>(slot + frag0);
            const v4i a1 = *reinterpret_cast<const v4i *>(slot + frag1);
            fwd_block<4, 11>(a0, a1, k, o0, o1);
        }
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<v4i *>(slot + frag0) = o0;
        *reinterpret_cast<v4i *>(slot + frag1) = o1;
        __builtin_amdgcn_wave_barrier();
        const v4i s0 = *reinterpret_cast<const v4i *>(slot + lin0);
        const v4i s1 = *reinterpret_cast<const v4i *>(slot + lin1);
        __builtin_amdgcn_wave_barrier();
        store16_sc1nt(dst + b * 2048, s0);
        store16_sc1nt(dst + b * 2048 + 1024, s1);
        if (nb >= end) break;
        b = nb;
    }
}

// ---- fused forward + inverse (coefficients AND reconstruction) ------------------------------
// recon = IDCT32(DCT32(x)) with both results written: 2 KiB in, 2 + 2 KiB out per block (6144
// algorithmic bytes) instead of 4 + 4 KiB for the two kernels back to back.  After the
// forward passes the coefficient tile sits in the wave's LDS slot for its line-dense store anyway;
// the inverse reads its columns from there, as the staged inverse does from a loaded tile.
__global__ __launch_bounds__(256) void dct32_fwdinv_lds_kernel(const int16_t *__restrict__ in,
                                                               int16_t *__restrict__ coef_out,
                                                               int16_t *__restrict__ recon_out, size_t n_blocks,
                                                               const DctOps *__restrict__ fwd_ops,
                                                               const DctOps *__restrict__ inv_ops,
                                                               unsigned blocks_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];
    const int lane = threadIdx.x & 63;
    unsigned char *slot = stage + (threadIdx.x >> 6) * 2048;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    size_t b = wave * blocks_per_wave;
    const size_t end = b + blocks_per_wave < n_blocks ? b + blocks_per_wave : n_blocks;
    if (b >= end) return;

    const unsigned c = lane & 31, h = lane >> 5;
    const unsigned lin0 = lds_slot(lane >> 2, lane & 3), lin1 = lds_slot(16 + (lane >> 2), lane & 3);
    const unsigned frag0 = lds_slot(c, 2 * h), frag1 = lds_slot(c, 2 * h + 1);
    unsigned col_base[4];
    {
        const unsigned u = (unsigned)kappa((int)c);
#pragma unroll
        for (unsigned j = 0; j < 4; ++j) col_base[j] = 16u * h * 64u + ((((u >> 3) ^ j) & 3u) << 4) + (u & 7u) * 2u;
    }
    const char *src = reinterpret_cast<const char *>(in) + lane * 16;
    const size_t lane_off = (size_t)lane * 16;

    v4i g0 = load16<true>(src + b * 2048), g1 = load16<true>(src + b * 2048 + 1024);
    const LaneConsts kf = load_consts(fwd_ops, lane);
    const LaneConsts ki = load_consts(inv_ops, lane);
    v16i c2r;
    {
        const int *__restrict__ s0 = inv_ops->c2r[0], *__restrict__ s1 = inv_ops->c2r[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) c2r[r] = h ? s1[r] : s0[r];
    }
    while (true) {
        const size_t nb = b + 1;
        *reinterpret_cast<v4i *>(slot + lin0) = g0;
        *reinterpret_cast<v4i *>(slot + lin1) = g1;
        if (nb < end) {
            g0 = load16<true>(src + nb * 2048);
            g1 = load16<true>(src + nb * 2048 + 1024);
        }
        __builtin_amdgcn_wave_barrier();
        v4i o0, o1;
        {
            const v4i a0 = *reinterpret_cast<const v4i *>(slot + frag0);
            const v4i a1 = *reinterpret_cast<const v4i *>(slot + frag1);
            fwd_block<4, 11>(a0, a1, kf, o0, o1);
        }
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<v4i *>(slot + frag0) = o0;               // coefficient tile, row-major (swizzled)
        *reinterpret_cast<v4i *>(slot + frag1) = o1;
        __builtin_amdgcn_wave_barrier();
        if (coef_out) {
            const v4i s0 = *reinterpret_cast<const v4i *>(slot + lin0);
            const v4i s1 = *reinterpret_cast<const v4i *>(slot + lin1);
            char *dst = reinterpret_cast<char *>(coef_out) + b * 2048 + lane_off;
            store16_sc1nt(dst, s0);
            store16_sc1nt(dst + 1024, s1);
        }
        {
            uint32_t w[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const uint32_t e0 = *reinterpret_cast<const uint16_t *>(slot + col_base[((2 * m) >> 2) & 3] + (2 * m) * 64);
                const uint32_t e1 = *reinterpret_cast<const uint16_t *>(slot + col_base[((2 * m + 1) >> 2) & 3] + (2 * m + 1) * 64);
                w[m] = e0 | (e1 << 16);
            }
            v4i lo, hi;
            split_planes(v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]}, v4i{(int)w[4], (int)w[5], (int)w[6], (int)w[7]}, lo, hi);
            inv_passes(lo, hi, ki, c2r, o0, o1);
        }
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<v4i *>(slot + frag0) = o0;
        *reinterpret_cast<v4i *>(slot + frag1) = o1;
        __builtin_amdgcn_wave_barrier();
        {
            const v4i s0 = *reinterpret_cast<const v4i *>(slot + lin0);
            const v4i s1 = *reinterpret_cast<const v4i *>(slot + lin1);
            __builtin_amdgcn_wave_barrier();
            char *dst = reinterpret_cast<char *>(recon_out) + b * 2048 + lane_off;
            store16_sc1nt(dst, s0);
            store16_sc1nt(dst + 1024, s1);
        }
        if (nb >= end) break;
        b = nb;
    }
}

// ---- fused residual + forward transform -----------------------------------------------------
// coef = DCT32(cur - pred) straight from two tiled frames (ref_block_t, src/x266.cpp:56-63),
// without materialising the residual: 2 KiB of pixels in, 2 KiB of coefficients out per block
// instead of 2 + 4 + 4 KiB for residual formation followed by the transform.  The transform is
// linear before its first rounding, so pass 1 is G*cur + (-G)*pred on the 8-bit pixels as they
// are: ONE byte plane per frame, no plane split, and the (x ^ 0x80) signed-offset trick needs no
// correction because the +128 of both frames cancels.  The rounding constant 8 is the MFMA's
// inline C operand.  A lane's fragment (row c, columns 16h..16h+15 of the 32x32 block) is exactly
// one 16-byte luma row of one tile, so fragment loads are line-dense as they are; only the stores
// go through the LDS slot (section "LDS-staged variant").
__global__ __launch_bounds__(256) void dct32_from_tiles_kernel(const x266_ref_block_t *__restrict__ cur,
                                                               const x266_ref_block_t *__restrict__ pred,
                                                               int16_t *__restrict__ out, int blocks_x, int tiles_x,
                                                               size_t n_blocks, const DctOps *__restrict__ ops)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];
    const int lane = threadIdx.x & 63;
    unsigned char *slot = stage + (threadIdx.x >> 6) * 2048;
    const size_t blk = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (blk >= n_blocks) return;
    const unsigned c = lane & 31, h = lane >> 5;
    const size_t by = blk / blocks_x, bx = blk - by * blocks_x;
    const size_t tile = (by * 2 + (c >> 4)) * (size_t)tiles_x + bx * 2 + h;
    // each instruction reads the whole 256-byte luma part of four tiles: line-dense, so streaming hints pay
    const v4i a = load16<true>(reinterpret_cast<const unsigned char *>(cur + tile) + (c & 15) * 16);
    const v4i b = load16<true>(reinterpret_cast<const unsigned char *>(pred + tile) + (c & 15) * 16);
    const LaneConsts k = load_consts(ops, lane);
    const v4i bias = {(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};
    const v16i round1 = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8};
    v16i acc = mfma(a ^ bias, k.p1, round1);
    acc = mfma(b ^ bias, k.tr, acc);                          // k.tr = -p1 in the forward tables
    v4i o0, o1;
    fwd_finish<4, 11>(acc, k, o0, o1);
    *reinterpret_cast<v4i *>(slot + lds_slot(c, 2 * h)) = o0;
    *reinterpret_cast<v4i *>(slot + lds_slot(c, 2 * h + 1)) = o1;
    __builtin_amdgcn_wave_barrier();
    const v4i s0 = *reinterpret_cast<const v4i *>(slot + lds_slot(lane >> 2, lane & 3));
    const v4i s1 = *reinterpret_cast<const v4i *>(slot + lds_slot(16 + (lane >> 2), lane & 3));
    char *dst = reinterpret_cast<char *>(out) + blk * 2048 + lane * 16;
    store16_sc1nt(dst, s0);
    store16_sc1nt(dst + 1024, s1);
}


// ---- the 1-D pass on its own (partialButterfly32, src_tb/dct32.c:66-170; RTL stage src/mkDct32.bsv:213-284) --------
// dst[k*32 + j] = (int16)((sum_n g[k][n] * src[j*32 + n] + (1 << (shift-1))) >> shift): one MFMA pass of the forward
// kernel with the accumulators stored TRANSPOSED, as the reference does.  Lane (c, h) holds frequency kappa(c) for the 16
// input rows j = acc_row(r, h): four runs of four consecutive j, i.e. four 8-byte stores per lane.  An entry point for
// checking a Bluesim DUT's (or this library's) intermediate against the reference -- not a throughput kernel.
__global__ __launch_bounds__(256) void dct32_pass_kernel(const int16_t *__restrict__ in, int16_t *__restrict__ out, size_t n_blocks,
                                                         const DctOps *__restrict__ ops, int shift)
{
    const int lane = threadIdx.x & 63;
    const size_t b = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (b >= n_blocks) return;
    const LaneConsts k = load_consts(ops, lane);
    const char *src = reinterpret_cast<const char *>(in) + b * 2048 + (size_t)(lane & 31) * 64 + (size_t)(lane >> 5) * 32;
    const v4i w0 = load16<false>(src), w1 = load16<false>(src + 16);
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v4i lo, hi;
    split_planes(w0, w1, lo, hi);
    const int c = k.c1 - (1 << 3) + (1 << (shift - 1));               // the table holds the constant of shift 4
    v16i acc = mfma(hi, k.p1, zero);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = (int)(((uint32_t)acc[r] << 8) + (uint32_t)c);
    acc = mfma(lo, k.p1, acc);
    const int freq = kappa(lane & 31), h = lane >> 5;
    char *dst = reinterpret_cast<char *>(out) + b * 2048 + (size_t)freq * 64 + (size_t)h * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t d0 = bperm((uint32_t)(acc[4 * q + 1] >> shift), (uint32_t)(acc[4 * q] >> shift), 0x05040100u);
        const uint32_t d1 = bperm((uint32_t)(acc[4 * q + 3] >> shift), (uint32_t)(acc[4 * q + 2] >> shift), 0x05040100u);
        *reinterpret_cast<uint2 *>(dst + q * 16) = make_uint2(d0, d1);
    }
}

}  // namespace

// ---- launchers ---------------------------------------------------------------
hipError_t launch_dct32(bool inverse, const int16_t *d_in, int16_t *d_out, size_t n_blocks,
                        const DctOps *d_ops, const LaunchCfg &cfg, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    const unsigned tpb = cfg.wg_threads;                       // 64 .. 256, multiple of 64
    const size_t waves_per_wg = tpb / 64;
    const unsigned bpw = units_per_wave_for(cfg, n_blocks);
    const size_t waves = (n_blocks + bpw - 1) / bpw;
    const size_t wgs = (waves + waves_per_wg - 1) / waves_per_wg;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const size_t lds = waves_per_wg * (size_t)cfg.lds_bytes_per_wave;   // 2 KiB used per wave; the rest caps the resident waves per CU
    if (inverse) hipLaunchKernelGGL(dct32_lds_kernel<true>, dim3((unsigned)wgs), dim3(tpb), lds, stream, d_in, d_out, n_blocks, d_ops, bpw);
    else         hipLaunchKernelGGL(dct32_lds_kernel<false>, dim3((unsigned)wgs), dim3(tpb), lds, stream, d_in, d_out, n_blocks, d_ops, bpw);
    return hipGetLastError();
}

hipError_t launch_dct32_pass(const int16_t *d_in, int16_t *d_out, size_t n_blocks, int shift, const DctOps *d_fwd_ops, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    const size_t wgs = (n_blocks + 3) / 4;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dct32_pass_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, d_in, d_out, n_blocks, d_fwd_ops, shift);
    return hipGetLastError();
}

hipError_t launch_dct32_fwdinv(const int16_t *d_in, int16_t *d_coef, int16_t *d_recon, size_t n_blocks,
                               const DctOps *d_fwd_ops, const DctOps *d_inv_lds_ops, const LaunchCfg &cfg, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    const unsigned tpb = (unsigned)cfg.wg_threads;
    const unsigned bpw = units_per_wave_for(cfg, n_blocks);
    const size_t wpw = tpb / 64, waves = (n_blocks + bpw - 1) / bpw, wgs = (waves + wpw - 1) / wpw;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const size_t lds = wpw * (size_t)cfg.lds_bytes_per_wave;
    hipLaunchKernelGGL(dct32_fwdinv_lds_kernel, dim3((unsigned)wgs), dim3(tpb), lds, stream, d_in, d_coef, d_recon, n_blocks, d_fwd_ops, d_inv_lds_ops, bpw);
    return hipGetLastError();
}

hipError_t launch_dct32_from_tiles(const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int16_t *d_out,
                                   int width, int height, const DctOps *d_fwd_ops, const LaunchCfg &cfg, hipStream_t stream)
{
    const int blocks_x = width / 32;
    const size_t n_blocks = (size_t)blocks_x * (size_t)(height / 32);
    if (n_blocks == 0) return hipSuccess;
    const unsigned tpb = (unsigned)cfg.wg_threads;
    const size_t wpw = tpb / 64, wgs = (n_blocks + wpw - 1) / wpw;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const size_t lds = wpw * (size_t)cfg.lds_bytes_per_wave;
    hipLaunchKernelGGL(dct32_from_tiles_kernel, dim3((unsigned)wgs), dim3(tpb), lds, stream, d_cur, d_pred, d_out, blocks_x, width / 16, n_blocks, d_fwd_ops);
    return hipGetLastError();
}

}  // namespace x266
