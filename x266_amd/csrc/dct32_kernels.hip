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
// dct32_fwdinv_kernel (coefficients + reconstruction in one pass, LDS-DMA fed), dct32_from_tiles_kernel
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
// recon = IDCT32(DCT32(x)) with both results written: 2 KiB in, 2 + 2 KiB out per block (6144 algorithmic bytes)
// instead of 4 + 4 KiB for the two kernels back to back.  Round 5's kernel (profiles/r05_fused_variants.txt):
//  * The inverse is fed from REGISTERS.  The forward's pass 2 in the OTHER operand orientation -- coefficients = A, pass-1
//    data = B, the orientation the inverse's pass B uses -- leaves in lane (i, h) the 16 sums Z[acc_row(r, h)][kappa(i)]:
//    sixteen rows v of one column u, i.e. the A fragment of the inverse's first (column) contraction when that pass pairs its
//    K-slots in accumulator-row order (build_inv_ops(o, false)).  Both orientations use the SAME constant image (k.p2) and the
//    same data planes: two more MFMAs (the matrix core is 10 % busy) instead of round 4's trip of the coefficient tile through
//    LDS (16 x ds_read_u16 per lane behind the tile's ds_write).  The byte-plane offset fix of the swapped pass belongs to
//    output row v = 0 = accumulator register 0 of the lower half-wave.  With coef_out == NULL only the swapped orientation runs.
//  * Nothing inside the loop is visible to the compiler's vmcnt bookkeeping.  Round 4's loops carried an `s_waitcnt vmcnt(0)`
//    in the middle of pass 1 (the compiler cannot see the inline-asm stores and merged the loop-entry state, constant loads
//    pending, into the loop): the "prefetched" next block and the previous block's stores were waited for on the spot.  Here
//    inputs arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write) into DEPTH 2 KiB slots, stores are
//    the sc1 nt instructions with a scalar base, and the one wait per block is counted by hand -- a wave's vector memory
//    operations retire in issue order, so "block i has landed" = at most {2 DMA per younger block in flight + the stores
//    issued since its DMA} outstanding.  The DMA writes lane l's 16 bytes at slot + 16 l: the bank swizzle of lds_slot() is
//    applied on the GLOBAL side (the same whole lines per instruction).  Slot DEPTH is the layout converter of the outputs.
//  * Launch shape: 4-wave workgroups, 2 blocks per wave (both fetched up front), 12 KiB of LDS charged per wave = 12 resident
//    waves per CU.  Deeper pipelines with long-lived waves (3 slots, 16 blocks per wave, 8 waves per CU) are 3-5 % faster on
//    some boxes and 5-8 % slower on others; this shape gave 0.73-0.76 of 8 TB/s on every one of eight boxes (round 4's
//    kernel: 0.66-0.77 on the same boxes).
// two 1 KiB-linear stores of one tile: scalar base + the lane's 32-bit offset, "sc1 nt" (x266_device.hpp, store16_sc1nt)
constexpr unsigned kStoresPerTile = 2;       // global_store instructions store_tile_sc1nt issues (counted by the kernel's waits)
constexpr unsigned kDmaPerBlock = 2;         // global_load_lds instructions one fetch of a 2 KiB block issues
__device__ __forceinline__ void store_tile_sc1nt(char *base, unsigned lane_off, const v4i &s0, const v4i &s1)
{
    asm volatile("global_store_dwordx4 %0, %1, %3 sc1 nt\n\tglobal_store_dwordx4 %0, %2, %3 offset:1024 sc1 nt\n\ts_nop 1"
                 : : "v"(lane_off), "v"(s0), "v"(s1), "s"(base) : "memory");
}

template <int DEPTH, bool COEF>
__global__ __launch_bounds__(256) void dct32_fwdinv_kernel(const int16_t *__restrict__ in,
                                                               int16_t *__restrict__ coef_out,
                                                               int16_t *__restrict__ recon_out, size_t n_blocks,
                                                               const DctOps *__restrict__ fwd_ops,
                                                               const DctOps *__restrict__ inv_ops,
                                                               unsigned blocks_per_wave, unsigned lds_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];
    const int lane = threadIdx.x & 63;
    const unsigned wave_in_wg = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
    const size_t first = wave * blocks_per_wave;
    if (first >= n_blocks) return;
    const unsigned cnt = n_blocks - first < blocks_per_wave ? (unsigned)(n_blocks - first) : blocks_per_wave;
    unsigned char *slots = stage + wave_in_wg * lds_per_wave;
    unsigned char *conv = slots + DEPTH * 2048;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)slots;

    const unsigned c = lane & 31, h = lane >> 5;
    const unsigned lin0 = lds_slot(lane >> 2, lane & 3), lin1 = lds_slot(16 + (lane >> 2), lane & 3);
    const unsigned frag0 = lds_slot(c, 2 * h), frag1 = lds_slot(c, 2 * h + 1);
    // the lane's LDS position 16 * lane (+ 1024) holds chunk (row, (lane & 3) ^ swizzle(row)) of the block
    const unsigned goff0 = (unsigned)(lane >> 2) * 64u + ((((unsigned)lane & 3u) ^ (((unsigned)lane >> 4) & 3u)) << 4), goff1 = goff0 + 1024u;
    const unsigned lane_off = (unsigned)lane * 16u;
    const char *src = reinterpret_cast<const char *>(in) + first * 2048;

    LaneConsts kf = load_consts(fwd_ops, lane);
    LaneConsts ki = load_consts(inv_ops, lane);
    v16i c2r;
    {
        const int *__restrict__ s0 = inv_ops->c2r[0], *__restrict__ s1 = inv_ops->c2r[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) c2r[r] = h ? s1[r] : s0[r];
    }
    auto fetch = [&](unsigned j, unsigned s) {                    // block j of this wave's run into input slot s
        const char *blk = src + (size_t)j * 2048;
        const unsigned lds = lds0 + s * 2048u;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\t"
                     "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 nt\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(goff0), "v"(goff1), "s"(blk), "s"(lds), "s"(lds + 1024u) : "memory");
    };
#pragma unroll
    for (unsigned j = 0; j < (unsigned)DEPTH; ++j)
        if (j < cnt) fetch(j, j);
    // every compiler-visible load has landed before the loop (the empty asm reads the constants, so the compiler's wait sits HERE): no
    // s_waitcnt vmcnt of the compiler's inside the loop
    asm volatile("" : "+v"(kf.p1), "+v"(kf.p2), "+v"(kf.c1), "+v"(kf.c2), "+v"(ki.p1), "+v"(ki.p2), "+v"(ki.c1), "+v"(c2r));
    const int c2s0 = (1 << 10) + (h ? 0 : 128 * 2048);            // swapped pass 2: row v = acc_row(0, 0) = 0 carries the offset fix
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    constexpr unsigned kStoresPerBlock = (COEF ? 2u : 1u) * kStoresPerTile;          // coefficient tile + reconstruction tile
    unsigned si = 0;                                               // i mod DEPTH
    for (unsigned i = 0; i < cnt; ++i) {
        const unsigned younger_loads = cnt - 1 - i < (unsigned)(DEPTH - 1) ? cnt - 1 - i : (unsigned)(DEPTH - 1);
        const unsigned stores_since = i < (unsigned)DEPTH ? i : (unsigned)DEPTH;
        if (i >= (unsigned)DEPTH && i + DEPTH <= cnt) {             // steady state: DEPTH - 1 younger blocks in flight, DEPTH blocks' stores since
            constexpr unsigned kSteady = kDmaPerBlock * (DEPTH - 1) + kStoresPerBlock * DEPTH;
            static_assert(kSteady <= kWaitVmcntMax, "steady-state wait beyond wait_vmcnt's range");
            wait_vmcnt(kSteady);
        } else {
            wait_vmcnt(kDmaPerBlock * younger_loads + kStoresPerBlock * stores_since);
        }
        unsigned char *slot = slots + si * 2048u;
        v4i ylo, yhi;
        {
            const v4i a0 = *reinterpret_cast<const v4i *>(slot + frag0);
            const v4i a1 = *reinterpret_cast<const v4i *>(slot + frag1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the fragments are in registers: the slot may be refilled
            if (i + DEPTH < cnt) fetch(i + DEPTH, si);
            v4i lo, hi;
            split_planes(a0, a1, lo, hi);
            v16i acc = mfma(hi, kf.p1, zero);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = (int)(((uint32_t)acc[r] << 8) + (uint32_t)kf.c1);
            acc = mfma(lo, kf.p1, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = acc[r] >> 4;
            pack_planes(acc, ylo, yhi);
        }
        // pass 2, swapped orientation: the inverse's input fragment
        v4i zlo, zhi;
        {
            v16i acc = mfma(kf.p2, yhi, zero);
            acc[0] = (int)(((uint32_t)acc[0] << 8) + (uint32_t)c2s0);
#pragma unroll
            for (int r = 1; r < 16; ++r) acc[r] = (int)(((uint32_t)acc[r] << 8) + (1u << 10));
            acc = mfma(kf.p2, ylo, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = acc[r] >> 11;       // bytes 0 / 1 = the int16 coefficient
            pack_planes(acc, zlo, zhi);
        }
        if (COEF) {
            // pass 2, natural orientation: the coefficient tile, 16 consecutive coefficients of one row per lane
            v16i acc = mfma(yhi, kf.p2, zero);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = (int)(((uint32_t)acc[r] << 8) + (uint32_t)kf.c2);
            acc = mfma(ylo, kf.p2, acc);
            uint32_t z[8];
#pragma unroll
            for (int m = 0; m < 8; ++m)
                z[m] = bperm((uint32_t)(acc[2 * m + 1] >> 11), (uint32_t)(acc[2 * m] >> 11), 0x05040100u);
            __builtin_amdgcn_wave_barrier();
            *reinterpret_cast<v4i *>(conv + frag0) = v4i{(int)z[0], (int)z[1], (int)z[2], (int)z[3]};
            *reinterpret_cast<v4i *>(conv + frag1) = v4i{(int)z[4], (int)z[5], (int)z[6], (int)z[7]};
            __builtin_amdgcn_wave_barrier();
            const v4i s0 = *reinterpret_cast<const v4i *>(conv + lin0);
            const v4i s1 = *reinterpret_cast<const v4i *>(conv + lin1);
            store_tile_sc1nt(reinterpret_cast<char *>(coef_out) + (first + i) * 2048, lane_off, s0, s1);
        }
        v4i o0, o1;
        inv_passes(zlo, zhi, ki, c2r, o0, o1);
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<v4i *>(conv + frag0) = o0;
        *reinterpret_cast<v4i *>(conv + frag1) = o1;
        __builtin_amdgcn_wave_barrier();
        {
            const v4i s0 = *reinterpret_cast<const v4i *>(conv + lin0);
            const v4i s1 = *reinterpret_cast<const v4i *>(conv + lin1);
            __builtin_amdgcn_wave_barrier();
            store_tile_sc1nt(reinterpret_cast<char *>(recon_out) + (first + i) * 2048, lane_off, s0, s1);
        }
        si = si + 1 == (unsigned)DEPTH ? 0u : si + 1;
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


// ---- fused chroma residual + forward transform -------------------------------------------------
// A 64x64 CTU in 4:2:0 carries one 32x32 U and one 32x32 V block -- the headline transform size -- spread over the m_C lines
// of its 4 x 4 tiles (src/x266.cpp:60, packed at :441-449: 8 rows of 8 interleaved U,V pairs per tile).  One wave takes one
// CTU and BOTH planes: a lane's fragment (row c, columns 16h .. 16h+15 of the block) is the chroma row (c & 7) of tile row
// (c >> 3), tiles 2h and 2h+1 -- two 16-byte loads per frame, whose even bytes are the U fragment and odd bytes the V
// fragment.  Eight lanes share a tile's line, so every load instruction consumes whole lines.  Arithmetic as for luma
// (dct32_from_tiles_kernel): one byte plane per frame, G*cur + (-G)*pred, the +128 of the offset trick cancels.
__global__ __launch_bounds__(256) void dct32_chroma_from_tiles_kernel(const x266_ref_block_t *__restrict__ cur,
                                                                      const x266_ref_block_t *__restrict__ pred,
                                                                      int16_t *__restrict__ out_u, int16_t *__restrict__ out_v,
                                                                      size_t block_pitch, int ctus_x, int tiles_x, size_t n_ctus,
                                                                      const DctOps *__restrict__ ops, unsigned lds_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];
    const int lane = threadIdx.x & 63;
    unsigned char *slot = stage + (threadIdx.x >> 6) * lds_per_wave;
    const size_t ctu = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (ctu >= n_ctus) return;
    const unsigned c = lane & 31, h = lane >> 5;
    const size_t cy = ctu / ctus_x, cx = ctu - cy * ctus_x;
    const size_t tile = (cy * 4 + (c >> 3)) * (size_t)tiles_x + cx * 4 + 2 * h;
    const unsigned char *pc = reinterpret_cast<const unsigned char *>(cur + tile) + 256 + (c & 7) * 16;
    const unsigned char *pp = reinterpret_cast<const unsigned char *>(pred + tile) + 256 + (c & 7) * 16;
    const v4i a0 = load16<true>(pc), a1 = load16<true>(pc + 512), b0 = load16<true>(pp), b1 = load16<true>(pp + 512);
    const LaneConsts k = load_consts(ops, lane);
    const uint32_t S = 0x80808080u;
    const v16i round1 = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8};
    const unsigned lin0 = lds_slot(lane >> 2, lane & 3), lin1 = lds_slot(16 + (lane >> 2), lane & 3);
    const unsigned frag0 = lds_slot(c, 2 * h), frag1 = lds_slot(c, 2 * h + 1);
    // U, then V through the same slot (both planes computed first and converted through two slots: 1-2 % slower, profiles/r06_chroma_shapes.txt)
#pragma unroll
    for (int plane = 0; plane < 2; ++plane) {
        const uint32_t sel = plane ? 0x07050301u : 0x06040200u;           // odd bytes = V, even bytes = U
        const v4i a = {(int)(bperm((uint32_t)a0[1], (uint32_t)a0[0], sel) ^ S), (int)(bperm((uint32_t)a0[3], (uint32_t)a0[2], sel) ^ S),
                       (int)(bperm((uint32_t)a1[1], (uint32_t)a1[0], sel) ^ S), (int)(bperm((uint32_t)a1[3], (uint32_t)a1[2], sel) ^ S)};
        const v4i b = {(int)(bperm((uint32_t)b0[1], (uint32_t)b0[0], sel) ^ S), (int)(bperm((uint32_t)b0[3], (uint32_t)b0[2], sel) ^ S),
                       (int)(bperm((uint32_t)b1[1], (uint32_t)b1[0], sel) ^ S), (int)(bperm((uint32_t)b1[3], (uint32_t)b1[2], sel) ^ S)};
        v16i acc = mfma(a, k.p1, round1);
        acc = mfma(b, k.tr, acc);                                         // k.tr = -p1 in the forward tables
        v4i o0, o1;
        fwd_finish<4, 11>(acc, k, o0, o1);
        if (plane) __builtin_amdgcn_wave_barrier();                       // the U tile has left the slot
        *reinterpret_cast<v4i *>(slot + frag0) = o0;
        *reinterpret_cast<v4i *>(slot + frag1) = o1;
        __builtin_amdgcn_wave_barrier();
        const v4i s0 = *reinterpret_cast<const v4i *>(slot + lin0);
        const v4i s1 = *reinterpret_cast<const v4i *>(slot + lin1);
        char *dst = reinterpret_cast<char *>((plane ? out_v : out_u) + ctu * block_pitch * 1024) + lane * 16;
        store16_sc1nt(dst, s0);
        store16_sc1nt(dst + 1024, s1);
    }
}

// ---- a whole 4:2:0 CTU in one launch ----------------------------------------------------------------------------------------
// coef[ctu][0..3] = DCT32 of the CTU's four 32x32 luma residual quadrants (raster order inside the CTU), coef[ctu][4] = U,
// coef[ctu][5] = V: 12 KiB per 64x64 CTU, CTUs in raster order -- the order a per-CTU encoder loop consumes, from ONE grid
// instead of xDct32FwdFromTilesDev + xDct32FwdChromaFromTilesDev (frame-raster luma, separate chroma streams).  Five waves per
// CTU: four take one luma quadrant each (dct32_from_tiles_kernel's body), the fifth both chroma planes
// (dct32_chroma_from_tiles_kernel's body: their fragments come from the same loads).  The part index is wave-uniform.
__global__ __launch_bounds__(256) void dct32_ctu_from_tiles_kernel(const x266_ref_block_t *__restrict__ cur,
                                                                   const x266_ref_block_t *__restrict__ pred,
                                                                   int16_t *__restrict__ out, int ctus_x, int tiles_x, size_t n_ctus,
                                                                   const DctOps *__restrict__ ops, unsigned lds_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];
    const int lane = threadIdx.x & 63;
    const unsigned wave_in_wg = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned char *slot = stage + wave_in_wg * lds_per_wave;
    const size_t unit = (size_t)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
    const size_t ctu = unit / 5;
    const unsigned part = (unsigned)(unit - ctu * 5);
    if (ctu >= n_ctus) return;
    const unsigned c = lane & 31, h = lane >> 5;
    const size_t cy = ctu / ctus_x, cx = ctu - cy * ctus_x;
    const LaneConsts k = load_consts(ops, lane);
    const uint32_t S = 0x80808080u;
    const v16i round1 = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8};
    const unsigned lin0 = lds_slot(lane >> 2, lane & 3), lin1 = lds_slot(16 + (lane >> 2), lane & 3);
    const unsigned frag0 = lds_slot(c, 2 * h), frag1 = lds_slot(c, 2 * h + 1);
    char *dst = reinterpret_cast<char *>(out + ctu * 6144) + lane * 16;
    if (part < 4) {
        // luma quadrant (part >> 1, part & 1): row c of the quadrant, columns 16h .. 16h+15 = one 16-byte luma row of one tile
        const size_t tile = (cy * 4 + (part >> 1) * 2 + (c >> 4)) * (size_t)tiles_x + cx * 4 + (part & 1) * 2 + h;
        const v4i a = load16<true>(reinterpret_cast<const unsigned char *>(cur + tile) + (c & 15) * 16);
        const v4i b = load16<true>(reinterpret_cast<const unsigned char *>(pred + tile) + (c & 15) * 16);
        const v4i bias = {(int)S, (int)S, (int)S, (int)S};
        v16i acc = mfma(a ^ bias, k.p1, round1);
        acc = mfma(b ^ bias, k.tr, acc);
        v4i o0, o1;
        fwd_finish<4, 11>(acc, k, o0, o1);
        *reinterpret_cast<v4i *>(slot + frag0) = o0;
        *reinterpret_cast<v4i *>(slot + frag1) = o1;
        __builtin_amdgcn_wave_barrier();
        const v4i s0 = *reinterpret_cast<const v4i *>(slot + lin0);
        const v4i s1 = *reinterpret_cast<const v4i *>(slot + lin1);
        store16_sc1nt(dst + part * 2048, s0);
        store16_sc1nt(dst + part * 2048 + 1024, s1);
        return;
    }
    const size_t tile = (cy * 4 + (c >> 3)) * (size_t)tiles_x + cx * 4 + 2 * h;
    const unsigned char *pc = reinterpret_cast<const unsigned char *>(cur + tile) + 256 + (c & 7) * 16;
    const unsigned char *pp = reinterpret_cast<const unsigned char *>(pred + tile) + 256 + (c & 7) * 16;
    const v4i a0 = load16<true>(pc), a1 = load16<true>(pc + 512), b0 = load16<true>(pp), b1 = load16<true>(pp + 512);
#pragma unroll
    for (int plane = 0; plane < 2; ++plane) {
        const uint32_t sel = plane ? 0x07050301u : 0x06040200u;
        const v4i a = {(int)(bperm((uint32_t)a0[1], (uint32_t)a0[0], sel) ^ S), (int)(bperm((uint32_t)a0[3], (uint32_t)a0[2], sel) ^ S),
                       (int)(bperm((uint32_t)a1[1], (uint32_t)a1[0], sel) ^ S), (int)(bperm((uint32_t)a1[3], (uint32_t)a1[2], sel) ^ S)};
        const v4i b = {(int)(bperm((uint32_t)b0[1], (uint32_t)b0[0], sel) ^ S), (int)(bperm((uint32_t)b0[3], (uint32_t)b0[2], sel) ^ S),
                       (int)(bperm((uint32_t)b1[1], (uint32_t)b1[0], sel) ^ S), (int)(bperm((uint32_t)b1[3], (uint32_t)b1[2], sel) ^ S)};
        v16i acc = mfma(a, k.p1, round1);
        acc = mfma(b, k.tr, acc);
        v4i o0, o1;
        fwd_finish<4, 11>(acc, k, o0, o1);
        if (plane) __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<v4i *>(slot + frag0) = o0;
        *reinterpret_cast<v4i *>(slot + frag1) = o1;
        __builtin_amdgcn_wave_barrier();
        const v4i s0 = *reinterpret_cast<const v4i *>(slot + lin0);
        const v4i s1 = *reinterpret_cast<const v4i *>(slot + lin1);
        store16_sc1nt(dst + (4 + plane) * 2048, s0);
        store16_sc1nt(dst + (4 + plane) * 2048 + 1024, s1);
    }
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

template <int DEPTH>
static hipError_t launch_dct32_fwdinv_depth(const int16_t *d_in, int16_t *d_coef, int16_t *d_recon, size_t n_blocks,
                                            const DctOps *d_fwd_ops, const DctOps *d_inv_acc_ops, const LaunchCfg &cfg, hipStream_t stream)
{
    const unsigned tpb = (unsigned)cfg.wg_threads;
    const unsigned bpw = units_per_wave_for(cfg, n_blocks);
    const size_t wpw = tpb / 64, waves = (n_blocks + bpw - 1) / bpw, wgs = (waves + wpw - 1) / wpw;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const unsigned need = (DEPTH + 1) * 2048u;                                 // input slots + the converter
    const unsigned per_wave = (unsigned)cfg.lds_bytes_per_wave < need ? need : (unsigned)cfg.lds_bytes_per_wave;
    const size_t lds = wpw * (size_t)per_wave;
    if (lds > 65536) return hipErrorInvalidValue;
    if (d_coef) hipLaunchKernelGGL((dct32_fwdinv_kernel<DEPTH, true>), dim3((unsigned)wgs), dim3(tpb), lds, stream, d_in, d_coef, d_recon, n_blocks, d_fwd_ops, d_inv_acc_ops, bpw, per_wave);
    else        hipLaunchKernelGGL((dct32_fwdinv_kernel<DEPTH, false>), dim3((unsigned)wgs), dim3(tpb), lds, stream, d_in, d_coef, d_recon, n_blocks, d_fwd_ops, d_inv_acc_ops, bpw, per_wave);
    return hipGetLastError();
}

// cfg.shape = input slots per wave (DMA depth): 0 or 2 = the default, 3 / 4 = the deeper pipelines of the long-lived shapes ("autotune", x266hip_abi.hip)
hipError_t launch_dct32_fwdinv(const int16_t *d_in, int16_t *d_coef, int16_t *d_recon, size_t n_blocks,
                               const DctOps *d_fwd_ops, const DctOps *d_inv_acc_ops, const LaunchCfg &cfg, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    if (cfg.shape == 3) return launch_dct32_fwdinv_depth<3>(d_in, d_coef, d_recon, n_blocks, d_fwd_ops, d_inv_acc_ops, cfg, stream);
    if (cfg.shape == 4) return launch_dct32_fwdinv_depth<4>(d_in, d_coef, d_recon, n_blocks, d_fwd_ops, d_inv_acc_ops, cfg, stream);
    return launch_dct32_fwdinv_depth<2>(d_in, d_coef, d_recon, n_blocks, d_fwd_ops, d_inv_acc_ops, cfg, stream);
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

hipError_t launch_dct32_chroma_from_tiles(const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int16_t *d_out_u, int16_t *d_out_v,
                                          size_t block_pitch, int width, int height, const DctOps *d_fwd_ops, const LaunchCfg &cfg, hipStream_t stream)
{
    const int ctus_x = width / 64;
    const size_t n_ctus = (size_t)ctus_x * (size_t)(height / 64);
    if (n_ctus == 0) return hipSuccess;
    const unsigned tpb = (unsigned)cfg.wg_threads;
    const size_t wpw = tpb / 64, wgs = (n_ctus + wpw - 1) / wpw;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    // a wave lives for two blocks here: 12 KiB charged per wave (13 resident per CU) instead of the forward kernel's 8 -- paired on a 32768^2 frame
    // 0.332 against 0.340 ms, 0.97 of the box's copy of the same bytes (tools/probes/gpu_chroma_shapes.py, profiles/r06_chroma_shapes.txt)
    const unsigned per_wave = (unsigned)cfg.lds_bytes_per_wave < 12288u ? 12288u : (unsigned)cfg.lds_bytes_per_wave;
    hipLaunchKernelGGL(dct32_chroma_from_tiles_kernel, dim3((unsigned)wgs), dim3(tpb), wpw * (size_t)per_wave, stream, d_cur, d_pred, d_out_u, d_out_v, block_pitch,
                       ctus_x, width / 16, n_ctus, d_fwd_ops, per_wave);
    return hipGetLastError();
}

hipError_t launch_dct32_ctu_from_tiles(const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int16_t *d_out,
                                       int width, int height, const DctOps *d_fwd_ops, const LaunchCfg &cfg, hipStream_t stream)
{
    const int ctus_x = width / 64;
    const size_t n_ctus = (size_t)ctus_x * (size_t)(height / 64);
    if (n_ctus == 0) return hipSuccess;
    const unsigned tpb = (unsigned)cfg.wg_threads;
    const size_t wpw = tpb / 64, units = n_ctus * 5, wgs = (units + wpw - 1) / wpw;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const unsigned per_wave = (unsigned)cfg.lds_bytes_per_wave;
    hipLaunchKernelGGL(dct32_ctu_from_tiles_kernel, dim3((unsigned)wgs), dim3(tpb), wpw * (size_t)per_wave, stream, d_cur, d_pred, d_out,
                       ctus_x, width / 16, n_ctus, d_fwd_ops, per_wave);
    return hipGetLastError();
}

}  // namespace x266
