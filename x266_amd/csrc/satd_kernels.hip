// satd_kernels.hip -- batched 8x8 Hadamard SATD for gfx950 (MI355X, CDNA4).
//
// Arithmetic contract (bit-exact): satd8x8(), src_tb/satd.c:31-118 -- 8-point
// Hadamard on rows then columns with every intermediate stored to int16
// (wraps), sum of |coefficient|, (sum + 2) >> 2.  Only add/sub are involved, so
// the result equals "exact integer Hadamard, keep the low 16 bits as signed,
// then abs" (SURVEY.md section 9.3); the coefficient order is irrelevant under
// the sum.  RTL twin: mkSatd8, src/mkSatd.bsv:83-176.
//
// Mapping: the 2-D Hadamard of a block is one 64x64 +-1
// matrix applied to the block's 64 samples, so 32 blocks at a time are one
// 64 x 64 x 32 integer GEMM on the int8 matrix core:
//      D[m][blk] = sum_s H64[m][s] * d[blk][s]
// with the int16 samples split into byte planes exactly as in the DCT kernel
// (d = 256*hi + (lo ^ 0x80) + 128; the +128 only reaches the DC coefficient:
// 128 * sum_s H[m][s] = 8192 for m = 0, 0 otherwise).  8 MFMAs per 32 blocks; the
// +-1 operand images are built in registers from the lane index (no table).
// Lane l owns block (l & 31) and loads the contiguous 64-byte half (l >> 5) of
// it; the two halves of a block meet inside the MFMA's K reduction.  The
// epilogue truncates to int16 by packing, takes |.| of two int16 at once with
// v_sad_u16 against a bias, and one cross-half add finishes the block.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "x266_device.hpp"
#include "x266_mfma_blocks.hpp"

namespace x266 {
namespace {

// sum over 16 accumulators of |(int16)x|, the accumulators holding x + 0x8000 in their low 16 bits (biased, see satd_group)
__device__ __forceinline__ uint32_t abs_sum16(const v16i &acc, uint32_t sum)
{
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        // low halves side by side: unsigned |biased - 0x8000| = |int16|, two per v_sad_u16
        const uint32_t pk = bperm((uint32_t)acc[2 * m + 1], (uint32_t)acc[2 * m], 0x05040100u);
        sum = __builtin_amdgcn_sad_u16(pk, 0x80008000u, sum);
    }
    return sum;
}

// One 32-block group (held as the lane's 64-byte half of its block) -> 32 costs.
struct SatdOperands { v4i h00, h01, h10, h11; int bias0; };

__device__ __forceinline__ SatdOperands make_satd_operands(int lane)
{
    // The four +-1 operand images (tile x K-step) of H64[m][s] = (-1)^popcount(m & s) are built
    // in registers (~25 VALU) instead of being fetched: with one 4 KiB group per wave a table would
    // double the wave's load instructions.  m = 32*tile + blk, s = 32*half + 16*step + t:
    //   popcount parity splits over disjoint bit ranges, so
    //   sign = [bits 0-1 of m vs t] ^ [bits 2-3 of m vs t>>2] ^ [bit 4 of m & step] ^ [tile & half]
    // and negating a +-1 byte is an XOR with 0xFE.
    const uint32_t NEG = 0xFEFEFEFEu;
    const uint32_t m = (uint32_t)lane & 31u, half = (uint32_t)lane >> 5;
    const uint32_t inner = (m & 1) ? ((m & 2) ? 0x01FFFF01u : 0xFF01FF01u)     // bytes j = 0..3: (-1)^popcount(m & 3 & j)
                                   : ((m & 2) ? 0xFFFF0101u : 0x01010101u);
    const uint32_t f2 = (m & 4) ? NEG : 0u, f3 = (m & 8) ? NEG : 0u;          // dword q flips on popcount((m >> 2) & q)
    const uint32_t f4 = (m & 16) ? NEG : 0u, fh = half ? NEG : 0u;
    const uint32_t b0 = inner, b1 = inner ^ f2, b2 = inner ^ f3, b3 = inner ^ f2 ^ f3;
    SatdOperands o;
    o.h00 = v4i{(int)b0, (int)b1, (int)b2, (int)b3};                             // tile 0, step 0
    o.h01 = v4i{(int)(b0 ^ f4), (int)(b1 ^ f4), (int)(b2 ^ f4), (int)(b3 ^ f4)}; // tile 0, step 1: m bit 4 & step
    o.h10 = v4i{(int)(b0 ^ fh), (int)(b1 ^ fh), (int)(b2 ^ fh), (int)(b3 ^ fh)}; // tile 1, step 0: m bit 5 & s bit 5
    o.h11 = o.h01 ^ v4i{(int)fh, (int)fh, (int)fh, (int)fh};
    // What the "<< 8" between the byte planes adds on its way (v_lshl_add, same cost as the bare shift): 0x8000 = the int16
    // bias that lets v_sad_u16 take |.| of the truncated coefficient directly, plus -- coefficient m = 0 only: tile 0,
    // register 0, half 0 -- the byte-plane offset fix 128 * 64 = 0x2000.
    o.bias0 = half == 0 ? 0x8000 + 0x2000 : 0x8000;
    return o;
}

__device__ __forceinline__ uint32_t satd_group(const SatdOperands &H, const v4i &w0, const v4i &w1, const v4i &w2, const v4i &w3)
{
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v4i lo0, hi0, lo1, hi1;
    split_planes(w0, w1, lo0, hi0);                     // K-step 0: samples 32*half + 0..15
    split_planes(w2, w3, lo1, hi1);                     // K-step 1: samples 32*half + 16..31
    uint32_t sum = 0;
    // x = 256 * (H * hi) + H * lo: the planes are chained through ONE accumulator -- high plane, "<< 8" with the bias (and, for
    // the DC coefficient, the offset fix) added by the same v_lshl_add, low plane on top.
    {   // coefficients 0..31
        v16i a = mfma(H.h00, hi0, zero);
        a = mfma(H.h01, hi1, a);
        a[0] = (int)(((uint32_t)a[0] << 8) + (uint32_t)H.bias0);
#pragma unroll
        for (int r = 1; r < 16; ++r) a[r] = (int)(((uint32_t)a[r] << 8) + 0x8000u);
        a = mfma(H.h00, lo0, a);
        a = mfma(H.h01, lo1, a);
        sum = abs_sum16(a, sum);
    }
    {   // coefficients 32..63
        v16i a = mfma(H.h10, hi0, zero);
        a = mfma(H.h11, hi1, a);
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = (int)(((uint32_t)a[r] << 8) + 0x8000u);
        a = mfma(H.h10, lo0, a);
        a = mfma(H.h11, lo1, a);
        sum = abs_sum16(a, sum);
    }
    // the other half of the coefficient rows sits in lane ^ 32
    sum += (uint32_t)__shfl_xor((int)sum, 32);
    return (sum + 2) >> 2;
}

// LDS-staged variant: the group's 4 KiB are fetched with four fully linear 1 KiB instructions
// (whole 128-byte lines per instruction, see dct32_kernels.hip) and turned into block-per-lane
// order through a wave-private LDS slot.  Chunk (block n, row j) lives at
// n*128 + ((j ^ ((n >> 1) & 7)) << 4): linear writes and fragment reads are conflict-free.
// the body, per wave: `wave` = index of the wave in the launch, `slot` = its 4 KiB of LDS
__device__ __forceinline__ void satd8x8_lds_wave(const int16_t *__restrict__ diff, uint32_t *__restrict__ out, size_t n_blocks,
                                                 unsigned groups_per_wave, size_t wave, unsigned char *slot, int lane)
{
    const size_t n_groups = (n_blocks + 31) >> 5;
    size_t g = wave * groups_per_wave;
    const size_t end = g + groups_per_wave < n_groups ? g + groups_per_wave : n_groups;
    if (g >= end) return;
    const int blk = lane & 31, half = lane >> 5;
    const SatdOperands H = make_satd_operands(lane);
    const size_t total_bytes = n_blocks * 128;

    unsigned lin[4], frag[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned chunk = lane + 64 * i, n = chunk >> 3, j = chunk & 7;
        lin[i] = n * 128 + ((j ^ ((n >> 1) & 7)) << 4);
        frag[i] = blk * 128 + ((((unsigned)(4 * half + i)) ^ (((unsigned)blk >> 1) & 7)) << 4);
    }
    for (; g < end; ++g) {
        const size_t base = g * 4096;
        v4i v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            size_t off = base + (size_t)lane * 16 + 1024 * (size_t)i;
            if (off + 16 > total_bytes) off = total_bytes - 16;       // ragged tail: stay inside the buffer
            v[i] = load16<true>(reinterpret_cast<const char *>(diff) + off);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<v4i *>(slot + lin[i]) = v[i];
        __builtin_amdgcn_wave_barrier();
        const v4i w0 = *reinterpret_cast<const v4i *>(slot + frag[0]), w1 = *reinterpret_cast<const v4i *>(slot + frag[1]);
        const v4i w2 = *reinterpret_cast<const v4i *>(slot + frag[2]), w3 = *reinterpret_cast<const v4i *>(slot + frag[3]);
        __builtin_amdgcn_wave_barrier();
        const uint32_t cost = satd_group(H, w0, w1, w2, w3);
        const size_t b = g * 32 + blk;
        if (b < n_blocks && half == 0) {
            // agent-scope result store (x266_device.hpp) from 2^18 blocks on: -1 % at the 8K frame's 518 400 blocks, -12 % at 2^21; launch-bound batches
            // (4096 .. 65536 blocks in 4.3 us) keep the plain store -- the write-through's acknowledgement lengthens their one round by 0.25 us
            if (n_blocks >= ((size_t)1 << 18)) store_result4(out + b, cost);
            else                               out[b] = cost;
        }
    }
}

// ---- large batches: the groups go straight from HBM into LDS -------------------------------------------------------------
// global_load_lds_dwordx4 (LDS-DMA: no staging registers), two 4 KiB slots per wave in ping-pong.  The DMA writes lane l's 16
// bytes at slot + 16 l, i.e. the LDS image is linear in the order of the lanes' global addresses; the bank swizzle of the fragment
// reads is therefore applied on the GLOBAL side -- lane l of instruction i fetches chunk (n, j ^ ((n >> 1) & 7)) of its 128-byte
// line, n = (l + 64 i) >> 3, j = l & 7: the same whole lines per instruction, permuted inside each line.
// `group` = the group's first byte, wave-uniform: the address is an SGPR pair + the lane's constant 32-bit offset, no vector
// address arithmetic per group.  CLAMP: the batch's last, partly filled group -- lanes past the end re-read its last 16 bytes.
constexpr unsigned kDmaPerGroup = 4;         // global_load_lds instructions one group fetch issues -- satd8x8_dma_issue and tiles_dma_issue alike (counted by the kernels' waits)
template <bool CLAMP>
__device__ __forceinline__ void satd8x8_dma_issue(const char *__restrict__ group, size_t bytes_left, const unsigned (&goff)[4], unsigned char *slot)
{
    if (!CLAMP) {
        // the instructions themselves, because the compiler only produces the VGPR-pair address form here (a 64-bit vector add
        // per load).  M0 (the LDS destination) is put back: the compiler owns it.
        const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)slot;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5 nt\n\t"
                     "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5 nt\n\t"
                     "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5 nt\n\t"
                     "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5 nt\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(goff[0]), "v"(goff[1]), "v"(goff[2]), "v"(goff[3]), "s"(group), "s"(lds), "s"(lds + 1024u),
                       "s"(lds + 2048u), "s"(lds + 3072u)
                     : "memory");
        return;
    }
    asm volatile("; ragged last group" ::: "memory");    // keeps this (once per launch) path a branch: if-converted into the common path it costs every group vector selects
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned off = goff[i];
        if ((size_t)off + 16 > bytes_left) off = (unsigned)(bytes_left - 16);   // those blocks are never stored
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(group + off),
                                         (__attribute__((address_space(3))) void *)(slot + 1024 * i), 16, 0, 2 /* nt */);
    }
}

// The wave scores groups first .. first + count - 1 (clipped to the batch).  Per group: wait for its DMA, read the fragments,
// REFILL THE SLOT AT ONCE -- before the group is scored -- so that two groups are in flight all the time, not one while the
// wave computes; score; park the 32 costs in LDS.  The costs of up to eight groups leave as one line-dense 1 KiB store: a store
// just before the loop's s_waitcnt vmcnt(4) would be counted by it, i.e. every iteration would wait for the previous group's
// store to be acknowledged.  Group counters are 32-bit (the launcher refuses batches of 2^37 blocks): wave-uniform compares stay
// on the scalar unit.  `slots`: 2 x 4 KiB of DMA slots + 1 KiB of costs.
typedef int v4i_unaligned __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ void satd8x8_dma_wave(const int16_t *__restrict__ diff, uint32_t *__restrict__ out, size_t n_blocks,
                                                 unsigned first, unsigned count, unsigned char *slots, int lane)
{
    const unsigned n_groups = (unsigned)((n_blocks + 31) >> 5), full_groups = (unsigned)(n_blocks >> 5);
    if (first >= n_groups) return;
    const unsigned end = n_groups - first > count ? first + count : n_groups;
    const int blk = lane & 31, half = lane >> 5;
    const char *src = reinterpret_cast<const char *>(diff);
    unsigned goff[4], frag[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned chunk = lane + 64 * i, n = chunk >> 3, j = chunk & 7;
        goff[i] = n * 128 + ((j ^ ((n >> 1) & 7)) << 4);
        frag[i] = blk * 128 + ((((unsigned)(4 * half + i)) ^ (((unsigned)blk >> 1) & 7)) << 4);
    }
    uint32_t *costs = reinterpret_cast<uint32_t *>(slots + 8192);        // 8 groups x 32 costs
    auto fetch = [&](unsigned q, unsigned char *slot) {
        const char *group = src + (size_t)q * 4096;
        if (q < full_groups) satd8x8_dma_issue<false>(group, 0, goff, slot);
        else                 satd8x8_dma_issue<true>(group, (n_blocks & 31) * 128, goff, slot);
    };
    unsigned g = first;
    fetch(g, slots);
    if (g + 1 < end) fetch(g + 1, slots + 4096);
    const SatdOperands H = make_satd_operands(lane);
    for (unsigned i = 0; g < end; ++g, ++i) {
        unsigned char *slot = slots + (i & 1) * 4096;
        // group g has landed when only group g+1's DMA may still be out (the cost stores issued in between make the count conservative, never short)
        wait_vmcnt(g + 1 < end ? kDmaPerGroup : 0u);
        __builtin_amdgcn_wave_barrier();
        const v4i w0 = *reinterpret_cast<const v4i *>(slot + frag[0]), w1 = *reinterpret_cast<const v4i *>(slot + frag[1]);
        const v4i w2 = *reinterpret_cast<const v4i *>(slot + frag[2]), w3 = *reinterpret_cast<const v4i *>(slot + frag[3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the fragments are in registers: the slot may be refilled
        __builtin_amdgcn_wave_barrier();
        if (g + 2 < end) fetch(g + 2, slot);
        const uint32_t cost = satd_group(H, w0, w1, w2, w3);
        if (half == 0) costs[(i & 7) * 32 + blk] = cost;
        if ((i & 7) == 7 || g + 1 == end) {                             // up to 8 x 32 costs: 16 bytes per lane, one 1 KiB-linear store
            const size_t b0 = (size_t)(g - (i & 7)) * 32;
            const size_t have = (size_t)((i & 7) + 1) * 32 < n_blocks - b0 ? (size_t)((i & 7) + 1) * 32 : n_blocks - b0;
            __builtin_amdgcn_wave_barrier();
            const v4i c = *reinterpret_cast<const v4i *>(costs + lane * 4);
            __builtin_amdgcn_wave_barrier();
            uint32_t *dst = out + b0 + lane * 4;
            if ((size_t)lane * 4 + 4 <= have) store_result16(dst, c);
            else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if ((size_t)lane * 4 + k < have) store_result4(dst + k, (uint32_t)c[k]);
            }
        }
    }
}

// small and medium batches (and the SATD lane of the frame kernel below): the staged body, one 4 KiB LDS slot per wave
__global__ __launch_bounds__(256) void satd8x8_lds_kernel(const int16_t *__restrict__ diff, uint32_t *__restrict__ out, size_t n_blocks,
                                                          unsigned groups_per_wave, unsigned lds_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];   // 4 KiB per wave + occupancy padding
    satd8x8_lds_wave(diff, out, n_blocks, groups_per_wave, ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6,
                     stage + (threadIdx.x >> 6) * lds_per_wave, (int)(threadIdx.x & 63));
}

// large batches: the LDS-DMA body, 9 KiB of LDS per wave + occupancy padding
__global__ __launch_bounds__(256) void satd8x8_dma_kernel(const int16_t *__restrict__ diff, uint32_t *__restrict__ out, size_t n_blocks,
                                                          unsigned groups_per_wave, unsigned lds_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];
    // wave-uniform by construction; readfirstlane tells the compiler so (scalar group addresses, scalar LDS slot for M0)
    const unsigned wave_in_wg = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
    if (wave > 0xFFFFFFFFu / groups_per_wave) return;
    satd8x8_dma_wave(diff, out, n_blocks, (unsigned)wave * groups_per_wave, groups_per_wave, stage + wave_in_wg * lds_per_wave, (int)(threadIdx.x & 63));
}

// ---- the two lanes of a frame in ONE launch (BASELINE configs[4]: a 7680x4320 frame = 32 400 DCT32 blocks + 518 400
// SATD blocks) ----------------------------------------------------------------------------------
// Per frame the kernels are ~15 us each: two submissions plus their events cost more than they run.  One grid: the first
// dct_wgs workgroups transform one DCT32 block per wave (the staged forward kernel's body, 8 KiB of LDS charged per
// wave as there), the rest score SATD groups (the staged SATD kernel's body).  Both halves are one-wave-per-unit
// already, so the fused grid is just the two grids back to back; the two lanes also fill each other's tails.
__global__ __launch_bounds__(128) void frame_lanes_kernel(const int16_t *__restrict__ dct_in, int16_t *__restrict__ dct_out, size_t n_dct,
                                                          const DctOps *__restrict__ ops, unsigned dct_wgs,
                                                          const int16_t *__restrict__ diff, uint32_t *__restrict__ satd_out, size_t n_satd,
                                                          unsigned groups_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];   // 8 KiB per wave
    const int lane = threadIdx.x & 63;
    const unsigned wave_in_wg = threadIdx.x >> 6;
    unsigned char *slot = stage + wave_in_wg * 8192;
    if (blockIdx.x < dct_wgs) {                                             // wave-uniform, in fact workgroup-uniform
        const size_t b = (size_t)blockIdx.x * 2 + wave_in_wg;
        if (b >= n_dct) return;
        const char *src = reinterpret_cast<const char *>(dct_in) + b * 2048 + lane * 16;
        const v4i g0 = load16<true>(src), g1 = load16<true>(src + 1024);
        const LaneConsts k = load_consts(ops, lane);
        v4i s0, s1;
        fwd_tile_staged<4, 11>(slot, lane, k, g0, g1, s0, s1);
        char *dst = reinterpret_cast<char *>(dct_out) + b * 2048 + lane * 16;
        store16m<2>(dst, s0);
        store16m<2>(dst + 1024, s1);
    } else {
        satd8x8_lds_wave(diff, satd_out, n_satd, groups_per_wave, (size_t)(blockIdx.x - dct_wgs) * 2 + wave_in_wg, slot, lane);
    }
}

// ---- fused residual + SATD ---------------------------------------------------------------------
// cost = satd8x8(cur - pred) for every 8x8 luma block of two tiled frames (ref_block_t,
// src/x266.cpp:56-63), residual never materialised.  The Hadamard transform is linear and the
// difference of two 8-bit pixels cannot wrap int16 (|coefficient| <= 64*255), so
// H*(cur - pred) = H*cur + (-H)*pred on the pixels as they are: one byte plane per frame, the +128
// of the signed-offset trick cancels, and negating a +-1 operand byte is an XOR with 0xFE.
// One wave takes 8 consecutive tiles = 32 blocks (lane n: tile n/4, block n%4 of the tile), so a
// tile's 256 luma bytes are consumed whole by one wave; costs are stored in raster order of blocks.
__global__ __launch_bounds__(256) void satd8x8_from_tiles_kernel(const x266_ref_block_t *__restrict__ cur,
                                                                 const x266_ref_block_t *__restrict__ pred,
                                                                 uint32_t *__restrict__ out, int tiles_x, size_t n_tiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];   // 4 KiB per wave
    const int lane = threadIdx.x & 63, n = lane & 31, half = lane >> 5;
    const size_t group = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (group * 8 >= n_tiles) return;
    size_t tile = group * 8 + (n >> 2);
    const bool live = tile < n_tiles;
    if (!live) tile = n_tiles - 1;
    const int sub_y = (n >> 1) & 1, sub_x = n & 1;                      // block inside the tile
    uint2 a[4], b[4];
    // Line-dense loads: one instruction reads the whole 256-byte luma part of four tiles (16 lanes
    // x 16 B each), nontemporal; a wave-private LDS slot turns that into block-per-lane order.
    // Row r of tile t (both 0-based inside the wave's group) lives at t*256 + ((r & 8) | ((r & 7) ^ t))*16:
    // the b128 writes and the b64 fragment reads are both bank-conflict-free.
    unsigned char *slot = stage + (threadIdx.x >> 6) * 4096;
    const int lt = lane >> 4, lr = lane & 15;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        size_t t = group * 8 + 4 * k + lt;
        if (t >= n_tiles) t = n_tiles - 1;                          // ragged tail: stay inside the frame
        const unsigned dst = (unsigned)((4 * k + lt) * 256 + ((lr & 8) | ((lr & 7) ^ (4 * k + lt))) * 16);
        *reinterpret_cast<v4i *>(slot + dst) = load16<true>(reinterpret_cast<const unsigned char *>(cur + t) + lr * 16);
        *reinterpret_cast<v4i *>(slot + 2048 + dst) = load16<true>(reinterpret_cast<const unsigned char *>(pred + t) + lr * 16);
    }
    __builtin_amdgcn_wave_barrier();
    const int lt8 = n >> 2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = sub_y * 8 + 4 * half + r;
        const unsigned src = (unsigned)(lt8 * 256 + ((row & 8) | ((row & 7) ^ lt8)) * 16 + sub_x * 8);
        a[r] = *reinterpret_cast<const uint2 *>(slot + src);
        b[r] = *reinterpret_cast<const uint2 *>(slot + 2048 + src);
    }
    const SatdOperands H = make_satd_operands(lane);
    const uint32_t S = 0x80808080u;                                     // pixels -> signed (offset cancels)
    const v4i a0 = {(int)(a[0].x ^ S), (int)(a[0].y ^ S), (int)(a[1].x ^ S), (int)(a[1].y ^ S)};
    const v4i a1 = {(int)(a[2].x ^ S), (int)(a[2].y ^ S), (int)(a[3].x ^ S), (int)(a[3].y ^ S)};
    const v4i b0 = {(int)(b[0].x ^ S), (int)(b[0].y ^ S), (int)(b[1].x ^ S), (int)(b[1].y ^ S)};
    const v4i b1 = {(int)(b[2].x ^ S), (int)(b[2].y ^ S), (int)(b[3].x ^ S), (int)(b[3].y ^ S)};
    const v4i NEG = {(int)0xFEFEFEFEu, (int)0xFEFEFEFEu, (int)0xFEFEFEFEu, (int)0xFEFEFEFEu};
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t sum = 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const v4i h0 = t ? H.h10 : H.h00, h1 = t ? H.h11 : H.h01;
        v16i acc = mfma(h0, a0, zero);
        acc = mfma(h1, a1, acc);
        acc = mfma(h0 ^ NEG, b0, acc);
        acc = mfma(h1 ^ NEG, b1, acc);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const uint32_t pk = bperm((uint32_t)acc[2 * m + 1], (uint32_t)acc[2 * m], 0x05040100u) ^ 0x80008000u;
            sum = __builtin_amdgcn_sad_u16(pk, 0x80008000u, sum);
        }
    }
    sum += (uint32_t)__shfl_xor((int)sum, 32);
    if (live && half == 0) {
        const size_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
        store_result4(out + ((ty * 2 + sub_y) * (size_t)(tiles_x * 2) + tx * 2 + sub_x), (sum + 2) >> 2);
    }
}

// ---- fused residual + SATD, large frames: the LDS-DMA body (round 5) -------------------------------------------------------
// The same arithmetic fed the way satd8x8_dma_kernel is: the luma parts of a group's eight tiles of both frames go straight
// from HBM into a 4 KiB LDS slot (global_load_lds_dwordx4, 4 instructions per group: 1 KiB = the 256 luma bytes of four
// tiles, whole lines; the row swizzle of the slot is applied on the GLOBAL side), two slots per wave, a slot refilled as soon
// as its fragments sit in registers, and the one wait per group counted by hand: a wave's vector memory operations retire in
// issue order, so group i has landed when at most {the 4 DMA of the younger group + the cost stores issued since} are
// outstanding.  The costs leave per group as before (two 64-byte runs of the raster): their store is counted, not avoided.
template <bool CLAMP>
__device__ __forceinline__ void tiles_dma_issue(const x266_ref_block_t *__restrict__ cur, const x266_ref_block_t *__restrict__ pred,
                                                size_t first_tile, size_t n_tiles, const unsigned (&goff)[2], unsigned char *slot)
{
    const unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)slot;
    if (!CLAMP) {
        const char *c = reinterpret_cast<const char *>(cur + first_tile), *p = reinterpret_cast<const char *>(pred + first_tile);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\t"
                     "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 nt\n\t"
                     "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4 nt\n\t"
                     "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4 nt\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(goff[0]), "v"(goff[1]), "s"(c), "s"(p), "s"(lds), "s"(lds + 1024u), "s"(lds + 2048u), "s"(lds + 3072u)
                     : "memory");
        return;
    }
    asm volatile("; ragged last group" ::: "memory");
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        size_t t = first_tile + goff[k] / 512u;                          // lanes past the frame's last tile re-read it (their costs are never stored)
        if (t >= n_tiles) t = n_tiles - 1;
        const unsigned in_tile = goff[k] & 511u;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(cur + t) + in_tile),
                                         (__attribute__((address_space(3))) void *)(slot + 1024 * k), 16, 0, 2 /* nt */);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(pred + t) + in_tile),
                                         (__attribute__((address_space(3))) void *)(slot + 2048 + 1024 * k), 16, 0, 2 /* nt */);
    }
}

__global__ __launch_bounds__(256) void satd8x8_from_tiles_dma_kernel(const x266_ref_block_t *__restrict__ cur,
                                                                     const x266_ref_block_t *__restrict__ pred,
                                                                     uint32_t *__restrict__ out, unsigned tiles_x, unsigned n_tiles,
                                                                     unsigned groups_per_wave, unsigned lds_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];   // 2 x 4 KiB per wave + occupancy padding
    const int lane = threadIdx.x & 63, n = lane & 31, half = lane >> 5;
    const unsigned wave_in_wg = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
    const unsigned n_groups = (n_tiles + 7) >> 3, full_groups = n_tiles >> 3;
    const unsigned first = wave * groups_per_wave;
    if (wave > 0xFFFFFFFFu / groups_per_wave || first >= n_groups) return;
    const unsigned cnt = n_groups - first < groups_per_wave ? n_groups - first : groups_per_wave;
    unsigned char *slots = stage + wave_in_wg * lds_per_wave;
    // LDS position 16 * lane of 1 KiB instruction k holds row ((sr & 8) | ((sr & 7) ^ t)) of tile t = 4 k + (lane >> 4), sr = lane & 15
    unsigned goff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const unsigned t = 4u * k + ((unsigned)lane >> 4), sr = (unsigned)lane & 15u;
        goff[k] = t * 512u + ((sr & 8u) | ((sr & 7u) ^ t)) * 16u;
    }
    auto fetch = [&](unsigned g, unsigned char *slot) {
        if (g < full_groups) tiles_dma_issue<false>(cur, pred, (size_t)g * 8, n_tiles, goff, slot);
        else                 tiles_dma_issue<true>(cur, pred, (size_t)g * 8, n_tiles, goff, slot);
    };
    fetch(first, slots);
    if (cnt > 1) fetch(first + 1, slots + 4096);
    const SatdOperands H = make_satd_operands(lane);
    const int sub_y = (n >> 1) & 1, sub_x = n & 1, lt8 = n >> 2;
    unsigned frag[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = sub_y * 8 + 4 * half + r;
        frag[r] = (unsigned)(lt8 * 256 + ((row & 8) | ((row & 7) ^ lt8)) * 16 + sub_x * 8);
    }
    const uint32_t S = 0x80808080u;                                     // pixels -> signed (offset cancels)
    const v4i NEG = {(int)0xFEFEFEFEu, (int)0xFEFEFEFEu, (int)0xFEFEFEFEu, (int)0xFEFEFEFEu};
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (unsigned i = 0; i < cnt; ++i) {
        unsigned char *slot = slots + (i & 1) * 4096;
        // younger than group i's DMA: the 4 DMA of group i + 1 (when there is one) and the cost store of group i - 1 (issued after them)
        constexpr unsigned kStoresPerGroup = 1;                          // the ONE global_store_dword at the end of the loop body
        wait_vmcnt((i + 1 < cnt ? kDmaPerGroup : 0u) + (i ? kStoresPerGroup : 0u));
        __builtin_amdgcn_wave_barrier();
        uint2 a[4], b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a[r] = *reinterpret_cast<const uint2 *>(slot + frag[r]);
            b[r] = *reinterpret_cast<const uint2 *>(slot + 2048 + frag[r]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the fragments are in registers: the slot may be refilled
        __builtin_amdgcn_wave_barrier();
        if (i + 2 < cnt) fetch(first + i + 2, slot);
        const v4i a0 = {(int)(a[0].x ^ S), (int)(a[0].y ^ S), (int)(a[1].x ^ S), (int)(a[1].y ^ S)};
        const v4i a1 = {(int)(a[2].x ^ S), (int)(a[2].y ^ S), (int)(a[3].x ^ S), (int)(a[3].y ^ S)};
        const v4i b0 = {(int)(b[0].x ^ S), (int)(b[0].y ^ S), (int)(b[1].x ^ S), (int)(b[1].y ^ S)};
        const v4i b1 = {(int)(b[2].x ^ S), (int)(b[2].y ^ S), (int)(b[3].x ^ S), (int)(b[3].y ^ S)};
        uint32_t sum = 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const v4i h0 = t ? H.h10 : H.h00, h1 = t ? H.h11 : H.h01;
            v16i acc = mfma(h0, a0, zero);
            acc = mfma(h1, a1, acc);
            acc = mfma(h0 ^ NEG, b0, acc);
            acc = mfma(h1 ^ NEG, b1, acc);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const uint32_t pk = bperm((uint32_t)acc[2 * m + 1], (uint32_t)acc[2 * m], 0x05040100u) ^ 0x80008000u;
                sum = __builtin_amdgcn_sad_u16(pk, 0x80008000u, sum);
            }
        }
        sum += (uint32_t)__shfl_xor((int)sum, 32);
        // raster position of the lane's block: the group's first tile by scalar division, the lane's own tile by a short walk
        const unsigned t0 = (first + i) * 8u, ty0 = t0 / tiles_x, tx0 = t0 - ty0 * tiles_x;
        unsigned tx = tx0 + (unsigned)lt8, ty = ty0;
        while (tx >= tiles_x) { tx -= tiles_x; ++ty; }
        const bool live = t0 + (unsigned)lt8 < n_tiles;
        uint32_t *dst = out + ((size_t)ty * 2 + sub_y) * ((size_t)tiles_x * 2) + (size_t)tx * 2 + sub_x;
        const uint32_t cost = (sum + 2) >> 2;
        // ONE store instruction per group (some lane is always live: a group exists only with its first tile): the hand-counted waits rely on it
        unsigned long long keep_exec;
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(live && half == 0);
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %3\n\tglobal_store_dword %1, %2, off sc1\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep_exec) : "v"(dst), "v"(cost), "s"(mask) : "memory");
    }
}

// ---- fused chroma residual + SATD ----------------------------------------------------------------------------------------
// cost = satd8x8(cur - pred) for the 8x8 U block and the 8x8 V block every tile carries in its m_C line (src/x266.cpp:60,
// :441-449: 8 rows of 8 interleaved U,V pairs).  One wave takes 16 tiles = 32 blocks (lane n: tile n / 2, plane n % 2): two
// line-dense load instructions per frame (lane = 8 * tile + row: a tile's eight lanes read its whole 128-byte line) into a
// wave-private 4 KiB LDS slot, row r of tile t at t*128 + ((r ^ t) & 7)*16; the fragment reads fetch a row's 16 interleaved
// bytes and keep the even (U) or odd (V) ones.  Arithmetic as for luma: H*cur + (-H)*pred on the pixels as they are.
__global__ __launch_bounds__(256) void satd8x8_chroma_from_tiles_kernel(const x266_ref_block_t *__restrict__ cur,
                                                                        const x266_ref_block_t *__restrict__ pred,
                                                                        uint32_t *__restrict__ out_u, uint32_t *__restrict__ out_v,
                                                                        size_t pitch, size_t n_tiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];   // 4 KiB per wave
    const int lane = threadIdx.x & 63, n = lane & 31, half = lane >> 5;
    const size_t group = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (group * 16 >= n_tiles) return;
    unsigned char *slot = stage + (threadIdx.x >> 6) * 4096;
    const int lt = lane >> 3, lr = lane & 7;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        size_t t = group * 16 + 8 * k + lt;
        if (t >= n_tiles) t = n_tiles - 1;                                  // ragged tail: stay inside the frame
        const unsigned dst = (unsigned)((8 * k + lt) * 128 + ((lr ^ (8 * k + lt)) & 7) * 16);
        *reinterpret_cast<v4i *>(slot + dst) = load16<true>(reinterpret_cast<const unsigned char *>(cur + t) + 256 + lr * 16);
        *reinterpret_cast<v4i *>(slot + 2048 + dst) = load16<true>(reinterpret_cast<const unsigned char *>(pred + t) + 256 + lr * 16);
    }
    __builtin_amdgcn_wave_barrier();
    const int tl = n >> 1;
    const uint32_t sel = (n & 1) ? 0x07050301u : 0x06040200u;
    const uint32_t S = 0x80808080u;                                         // pixels -> signed (offset cancels)
    uint2 a[4], b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * half + r;
        const unsigned src = (unsigned)(tl * 128 + ((row ^ tl) & 7) * 16);
        const v4i wa = *reinterpret_cast<const v4i *>(slot + src), wb = *reinterpret_cast<const v4i *>(slot + 2048 + src);
        a[r] = make_uint2(bperm((uint32_t)wa[1], (uint32_t)wa[0], sel) ^ S, bperm((uint32_t)wa[3], (uint32_t)wa[2], sel) ^ S);
        b[r] = make_uint2(bperm((uint32_t)wb[1], (uint32_t)wb[0], sel) ^ S, bperm((uint32_t)wb[3], (uint32_t)wb[2], sel) ^ S);
    }
    const SatdOperands H = make_satd_operands(lane);
    const v4i a0 = {(int)a[0].x, (int)a[0].y, (int)a[1].x, (int)a[1].y}, a1 = {(int)a[2].x, (int)a[2].y, (int)a[3].x, (int)a[3].y};
    const v4i b0 = {(int)b[0].x, (int)b[0].y, (int)b[1].x, (int)b[1].y}, b1 = {(int)b[2].x, (int)b[2].y, (int)b[3].x, (int)b[3].y};
    const v4i NEG = {(int)0xFEFEFEFEu, (int)0xFEFEFEFEu, (int)0xFEFEFEFEu, (int)0xFEFEFEFEu};
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t sum = 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const v4i h0 = t ? H.h10 : H.h00, h1 = t ? H.h11 : H.h01;
        v16i acc = mfma(h0, a0, zero);
        acc = mfma(h1, a1, acc);
        acc = mfma(h0 ^ NEG, b0, acc);
        acc = mfma(h1 ^ NEG, b1, acc);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const uint32_t pk = bperm((uint32_t)acc[2 * m + 1], (uint32_t)acc[2 * m], 0x05040100u) ^ 0x80008000u;
            sum = __builtin_amdgcn_sad_u16(pk, 0x80008000u, sum);
        }
    }
    sum += (uint32_t)__shfl_xor((int)sum, 32);
    const size_t tile = group * 16 + (size_t)tl;
    if (tile < n_tiles && half == 0) store_result4(((n & 1) ? out_v : out_u) + tile * pitch, (sum + 2) >> 2);
}

// ---- synthetic residual stream ---------------------------------------------
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void fill_residual_kernel(int16_t *__restrict__ dst, size_t n_samples,
                                                            uint64_t seed, uint64_t first_index)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n_vec = n_samples >> 3;                    // 8 samples = 16 bytes per thread step
    for (size_t v = tid; v < n_vec; v += stride) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint64_t r0 = splitmix64_at(seed, first_index + v * 8 + 2 * i);
            const uint64_t r1 = splitmix64_at(seed, first_index + v * 8 + 2 * i + 1);
            const int s0 = (int)(r0 & 0xFF) - (int)((r0 >> 8) & 0xFF);
            const int s1 = (int)(r1 & 0xFF) - (int)((r1 >> 8) & 0xFF);
            w[i] = ((uint32_t)s0 & 0xFFFFu) | ((uint32_t)s1 << 16);
        }
        reinterpret_cast<v4i *>(dst)[v] = v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    }
    // tail (n_samples not a multiple of 8)
    for (size_t i = (n_vec << 3) + tid; i < n_samples; i += stride) {
        const uint64_t r = splitmix64_at(seed, first_index + i);
        dst[i] = (int16_t)((int)(r & 0xFF) - (int)((r >> 8) & 0xFF));
    }
}

}  // namespace

// Batches from kSatdDmaMinBlocks on run the LDS-DMA kernel (four-wave workgroups, four groups per wave, 16 KiB of LDS charged per wave
// = eight resident waves per CU), smaller ones the staged kernel (two-wave workgroups, two groups per wave, 6 KiB charged): measured
// crossover, profiles/r04_satd_batch.txt -- +4-5 % at 2^24 blocks on every box, -7 % at the 8K frame's 518 400.
hipError_t launch_satd8x8(const int16_t *d_diff, uint32_t *d_out, size_t n_blocks, const LaunchCfg &cfg, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    const size_t groups = (n_blocks + 31) / 32;
    if (groups > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const bool dma = cfg.shape == 3 || (cfg.shape == 0 && n_blocks >= kSatdDmaMinBlocks);
    LaunchCfg c = cfg;
    if (c.units_per_wave <= 0) c.units_per_wave = dma ? 4 : 2;
    const unsigned tpb = cfg.wg_threads > 0 ? (unsigned)cfg.wg_threads : (dma ? 256u : 128u);
    const unsigned min_lds = dma ? 9216u : 4096u;
    unsigned per_wave = cfg.lds_bytes_per_wave > 0 ? (unsigned)cfg.lds_bytes_per_wave : (dma ? 16384u : 6144u);
    if (per_wave < min_lds) per_wave = min_lds;
    const size_t waves_per_wg = tpb / 64;
    const unsigned gpw = units_per_wave_for(c, groups);
    const size_t waves = (groups + gpw - 1) / gpw;
    const size_t wgs = (waves + waves_per_wg - 1) / waves_per_wg;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const size_t lds = waves_per_wg * (size_t)per_wave;
    if (lds > 65536) return hipErrorInvalidValue;
    if (dma) hipLaunchKernelGGL(satd8x8_dma_kernel, dim3((unsigned)wgs), dim3(tpb), lds, stream, d_diff, d_out, n_blocks, gpw, per_wave);
    else     hipLaunchKernelGGL(satd8x8_lds_kernel, dim3((unsigned)wgs), dim3(tpb), lds, stream, d_diff, d_out, n_blocks, gpw, per_wave);
    return hipGetLastError();
}

hipError_t launch_frame_lanes(const int16_t *d_dct_in, int16_t *d_dct_out, size_t n_dct, const DctOps *d_fwd_ops,
                              const int16_t *d_diff, uint32_t *d_satd_out, size_t n_satd, const LaunchCfg &satd_cfg, hipStream_t stream)
{
    if (n_dct == 0 && n_satd == 0) return hipSuccess;
    const size_t dct_wgs = (n_dct + 1) / 2;
    const size_t groups = (n_satd + 31) / 32;
    LaunchCfg c = satd_cfg;
    if (c.units_per_wave <= 0) c.units_per_wave = 2;                  // the staged body's default
    const unsigned gpw = groups ? units_per_wave_for(c, groups) : 1u;
    const size_t satd_wgs = ((groups + gpw - 1) / gpw + 1) / 2;
    if (dct_wgs + satd_wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(frame_lanes_kernel, dim3((unsigned)(dct_wgs + satd_wgs)), dim3(128), (size_t)(2 * 8192), stream,
                       d_dct_in, d_dct_out, n_dct, d_fwd_ops, (unsigned)dct_wgs, d_diff, d_satd_out, n_satd, gpw);
    return hipGetLastError();
}

hipError_t launch_satd8x8_from_tiles(const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, uint32_t *d_out,
                                     int width, int height, int shape, hipStream_t stream)
{
    const int tiles_x = width / 16;
    const size_t n_tiles = (size_t)tiles_x * (size_t)(height / 16);
    if (n_tiles == 0) return hipSuccess;
    const size_t groups = (n_tiles + 7) / 8;                            // one wave per 8 tiles, one-wave workgroups
    if (groups > 0x7FFFFFFFull) return hipErrorInvalidValue;
    // Large frames (the batch kernel's crossover): the LDS-DMA body, one group per one-wave workgroup, 8 KiB charged = 20 resident waves per CU --
    // paired in one process it is 2.5 % faster than the staged body (0.350 against 0.359 ms for 2^24 blocks; two groups per wave 1.8 %, four-wave
    // workgroups with deep pipelines 3-15 % SLOWER: the tile format reads 256 of every 512 bytes, the short-lived dispatch-ordered shape keeps
    // the half-dense stream together; profiles/r05_from_tiles_dma.txt).  shape 1 / 3 force a body (tests).
    const bool dma = shape == 3 || (shape == 0 && n_tiles * 4 >= kSatdDmaMinBlocks);
    if (dma && n_tiles <= 0xFFFFFFFFull) {
        constexpr unsigned kGroupsPerWave = 1, kLdsPerWave = 8192;
        const size_t waves = (groups + kGroupsPerWave - 1) / kGroupsPerWave;
        hipLaunchKernelGGL(satd8x8_from_tiles_dma_kernel, dim3((unsigned)waves), dim3(64), (size_t)kLdsPerWave, stream, d_cur, d_pred, d_out,
                           (unsigned)tiles_x, (unsigned)n_tiles, kGroupsPerWave, kLdsPerWave);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(satd8x8_from_tiles_kernel, dim3((unsigned)groups), dim3(64), (size_t)6144, stream, d_cur, d_pred, d_out, tiles_x, n_tiles);
    return hipGetLastError();
}

hipError_t launch_satd8x8_chroma_from_tiles(const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, uint32_t *d_out_u, uint32_t *d_out_v,
                                            size_t pitch, int width, int height, hipStream_t stream)
{
    const size_t n_tiles = (size_t)(width / 16) * (size_t)(height / 16);
    if (n_tiles == 0) return hipSuccess;
    const size_t groups = (n_tiles + 15) / 16;                          // one wave per 16 tiles, one-wave workgroups (the luma kernel's shape)
    if (groups > 0x7FFFFFFFull) return hipErrorInvalidValue;
    // 6 KiB charged per one-wave workgroup (4 KiB used): the best of the 3 x 6 shapes of tools/probes/gpu_chroma_shapes.py, 0.81 of the box's DENSE read
    // probe of the same bytes -- the luma kernel's rate per byte less 6 % (a quarter-dense stream: one line of every four)
    hipLaunchKernelGGL(satd8x8_chroma_from_tiles_kernel, dim3((unsigned)groups), dim3(64), (size_t)6144, stream, d_cur, d_pred, d_out_u, d_out_v, pitch, n_tiles);
    return hipGetLastError();
}

hipError_t launch_fill_residual(int16_t *d_dst, size_t n_samples, uint64_t seed,
                                uint64_t first_index, const LaunchCfg &cfg, hipStream_t stream)
{
    if (n_samples == 0) return hipSuccess;
    size_t wgs = (n_samples / 8 + 255) / 256;
    const size_t cap = (size_t)cfg.cu_count * 8;
    if (wgs > cap) wgs = cap;
    if (wgs == 0) wgs = 1;
    hipLaunchKernelGGL(fill_residual_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, d_dst, n_samples, seed, first_index);
    return hipGetLastError();
}

}  // namespace x266
