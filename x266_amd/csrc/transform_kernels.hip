// transform_kernels.hip -- forward DCT-II 4/8/16 and DST-VII 4/8/16 for gfx950
// (BASELINE configs[3]: the mixed transform set: DCT-II + closed-form DST-VII, or caller-supplied matrices).
//
// Reference anchor: the N-point DCT-II matrices are sub-matrices of g_t32
// (src_tb/dct32.c:30-64; the RTL re-uses the adder-tree taps for 4/8/16,
// src/mkDct32.bsv:132-141) and the pass structure / rounding / truncating store
// are partialButterfly32's (src_tb/dct32.c:66-170) with shifts log2N-1 and
// log2N+6.  Upstream has no C model for these sizes nor any DST-VII: parity is
// UNPINNED and rests on the CPU statement kept with the tests.
//
// Mapping: (32/N)^2 small blocks form one 32x32 tile; transforming the tile with
// the block-diagonal matrix diag(M_N,...) on both sides transforms every small
// block on its own.  The tile runs through exactly the two-pass int8-MFMA
// pipeline of dct32_kernels.hip (x266_mfma_blocks.hpp); only the lane -> address
// pattern and the operand images differ.  Blocks are N x N int16 row-major and
// either contiguous (block b at b*N*N samples) or placed by a per-block offset
// table (the mixed per-CTU batches of configs[3]).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "x266_device.hpp"
#include "x266_mfma_blocks.hpp"
#include "x266_tables.hpp"

namespace x266 {
namespace {

// One 32x32 tile of (32/N)^2 small blocks, block-major, sitting in a wave-private 2 KiB LDS slot: read the
// lane's fragment pieces, run the two MFMA passes with the class's operand images, write the results back
// in place.  Shared by the per-class kernels and the mixed-class tile kernel.
template <int LOGN>
__device__ __forceinline__ void fwd_tile_in_slot(unsigned char *slot, int lane, const LaneConsts &k)
{
    constexpr int N = 1 << LOGN;
    constexpr int PER = 32 / N, PIECES = N >= 16 ? 1 : 16 / N;
    constexpr int S1 = LOGN - 1, S2 = LOGN + 6;
    const int c = lane & 31, h = lane >> 5;
    const int row = c & (N - 1), tile_row = c >> LOGN;
    // byte offsets inside the 2 KiB tile of this lane's fragment pieces (block-major tile layout)
    unsigned frag[PIECES];
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const unsigned sb = N == 32 ? 0u : (unsigned)(tile_row * PER + h * PIECES + q);
        frag[q] = (sb * (unsigned)(N * N) + (unsigned)row * N + (N == 32 ? 16u * h : 0u)) * 2u;
    }
    uint32_t w[8];
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        if (N >= 16) {
            const v4i a = *reinterpret_cast<const v4i *>(slot + frag[q]), b = *reinterpret_cast<const v4i *>(slot + frag[q] + 16);
            w[0] = a[0]; w[1] = a[1]; w[2] = a[2]; w[3] = a[3]; w[4] = b[0]; w[5] = b[1]; w[6] = b[2]; w[7] = b[3];
        } else if (N == 8) {
            const v4i a = *reinterpret_cast<const v4i *>(slot + frag[q]);
            w[4 * q] = a[0]; w[4 * q + 1] = a[1]; w[4 * q + 2] = a[2]; w[4 * q + 3] = a[3];
        } else {
            const uint2 a = *reinterpret_cast<const uint2 *>(slot + frag[q]);
            w[2 * q] = a.x; w[2 * q + 1] = a.y;
        }
    }
    __builtin_amdgcn_wave_barrier();
    v4i r0, r1;
    fwd_block<S1, S2>(v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]}, v4i{(int)w[4], (int)w[5], (int)w[6], (int)w[7]}, k, r0, r1);
    const uint32_t z[8] = {(uint32_t)r0[0], (uint32_t)r0[1], (uint32_t)r0[2], (uint32_t)r0[3],
                           (uint32_t)r1[0], (uint32_t)r1[1], (uint32_t)r1[2], (uint32_t)r1[3]};
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        if (N >= 16) {
            *reinterpret_cast<v4i *>(slot + frag[q]) = r0;
            *reinterpret_cast<v4i *>(slot + frag[q] + 16) = r1;
        } else if (N == 8) {
            *reinterpret_cast<v4i *>(slot + frag[q]) = v4i{(int)z[4 * q], (int)z[4 * q + 1], (int)z[4 * q + 2], (int)z[4 * q + 3]};
        } else {
            *reinterpret_cast<uint2 *>(slot + frag[q]) = make_uint2(z[2 * q], z[2 * q + 1]);
        }
    }
}

// A tile's (32/N)^2 blocks are 2 KiB of consecutive memory (contiguous batches) or (32/N)^2 pieces placed by the
// offset table, moved with linear 1 KiB instructions (whole 128-byte lines per instruction, see
// dct32_kernels.hip section "LDS-staged variant") and re-read from a wave-private LDS slot in
// fragment order.  Streaming cache hints only for contiguous batches: scattered blocks may share lines across instructions.
template <int LOGN, bool INDEXED>
__device__ __forceinline__ void tr_fwd_small_body(unsigned char *stage, size_t wave, const int16_t *__restrict__ in, int16_t *__restrict__ out,
                                                  size_t n_blocks, const DctOps *__restrict__ ops,
                                                  const uint32_t *__restrict__ offsets, unsigned tiles_per_wave)
{
    constexpr int N = 1 << LOGN;
    constexpr int PER = 32 / N, NSB = PER * PER;
    unsigned char *slot = stage + (threadIdx.x >> 6) * 2048;

    const int lane = threadIdx.x & 63;
    const size_t n_tiles = (n_blocks + NSB - 1) / NSB;
    size_t t = wave * tiles_per_wave;
    const size_t t_end = t + tiles_per_wave < n_tiles ? t + tiles_per_wave : n_tiles;
    if (t >= t_end) return;
    const LaneConsts k = load_consts(ops, lane);
    const size_t total_bytes = n_blocks * (size_t)(N * N * 2);
    for (; t < t_end; ++t) {
        size_t o0, o1;
        bool live0, live1;
        if (INDEXED) {
            // chunk c (16 bytes) of the tile belongs to block c / CPB of the tile, placed by the offset table:
            // blocks that sit next to each other in memory still make line-dense instructions
            constexpr int CPB = N * N / 8;                               // 16-byte chunks per block
            size_t b0 = t * NSB + (size_t)(lane / CPB), b1 = t * NSB + (size_t)((lane + 64) / CPB);
            live0 = b0 < n_blocks;
            live1 = b1 < n_blocks;
            if (!live0) b0 = n_blocks - 1;                               // ragged tail: re-read the last block
            if (!live1) b1 = n_blocks - 1;
            o0 = ((size_t)offsets[b0] * 2) + (size_t)(lane % CPB) * 16;
            o1 = ((size_t)offsets[b1] * 2) + (size_t)((lane + 64) % CPB) * 16;
        } else {
            const size_t base = t * 2048;
            o0 = base + (size_t)lane * 16;
            o1 = o0 + 1024;
            live0 = o0 + 16 <= total_bytes;
            live1 = o1 + 16 <= total_bytes;
            if (!live0) o0 = total_bytes - 16;                           // ragged tail: stay inside the buffer
            if (!live1) o1 = total_bytes - 16;
        }
        const v4i g0 = load16<true>(reinterpret_cast<const char *>(in) + o0);   // streaming hints, offset tables included (round 5: -8 % on the per-class calls over a CTU-ordered buffer, profiles/r05_result_stores.txt)
        const v4i g1 = load16<true>(reinterpret_cast<const char *>(in) + o1);
        *reinterpret_cast<v4i *>(slot + lane * 16) = g0;
        *reinterpret_cast<v4i *>(slot + 1024 + lane * 16) = g1;
        __builtin_amdgcn_wave_barrier();
        fwd_tile_in_slot<LOGN>(slot, lane, k);
        __builtin_amdgcn_wave_barrier();
        const v4i s0 = *reinterpret_cast<const v4i *>(slot + lane * 16);
        const v4i s1 = *reinterpret_cast<const v4i *>(slot + 1024 + lane * 16);
        __builtin_amdgcn_wave_barrier();
        if (live0) store16m<2>(reinterpret_cast<char *>(out) + o0, s0);   // "sc1 nt" (x266_device.hpp)
        if (live1) store16m<2>(reinterpret_cast<char *>(out) + o1, s1);
    }
}

// One kernel per size: the offset-table form and the contiguous form are two complete bodies (each with its own addressing AND cache
// policy at compile time) behind one wave-uniform test of the kernel argument.  A run-time choice INSIDE the loop costs 1-3.6 %; this
// form measures +-0.3 % on the forward family, +0.5-1 % FASTER on the tile kernel below and, once the inverse family's column gather
// is pinned pair by pair (inv_tile_in_slot_with<.., PAIRWISE>: left alone, the merged function issues the sixteen reads as one burst,
// 0.5-1.9 % slower), +0.1-1 % on the contiguous inverse family -- profiles/r04_kernel_prune.txt.  (32x32 exists only in the offset-table form:
// contiguous 32x32 batches are dct32_kernels.hip's.)
template <int LOGN>
__global__ __launch_bounds__(256) void tr_fwd_small_lds_kernel(const int16_t *__restrict__ in, int16_t *__restrict__ out,
                                                               size_t n_blocks, const DctOps *__restrict__ ops,
                                                               const uint32_t *__restrict__ offsets, unsigned tiles_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];   // 2 KiB per wave (+ occupancy padding)
    // everything both bodies need from the dispatch packet is read BEFORE the branch (a scalar load issued behind it would be a second
    // round trip in every wave's prologue)
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (LOGN == 5 || offsets) tr_fwd_small_body<LOGN, true>(stage, wave, in, out, n_blocks, ops, offsets, tiles_per_wave);
    else                      tr_fwd_small_body<LOGN, LOGN == 5>(stage, wave, in, out, n_blocks, ops, offsets, tiles_per_wave);
}

// The inverse of fwd_tile_in_slot: the first contraction runs over the tile's ROW index, so each lane reads
// its COLUMN out of the staged tile (16 x ds_read_u16), exactly as the staged DCT32 inverse does.
template <int LOGN, class C2RGroup, bool PAIRWISE = false>
__device__ __forceinline__ void inv_tile_in_slot_with(unsigned char *slot, int lane, const LaneConsts &k, C2RGroup c2r_group)
{
    constexpr int N = 1 << LOGN;
    constexpr int PER = 32 / N, PIECES = N >= 16 ? 1 : 16 / N;
    const int c = lane & 31, h = lane >> 5;
    const int row = c & (N - 1), tile_row = c >> LOGN;
    unsigned frag[PIECES];                                              // output fragment pieces (as in the forward body)
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const unsigned sb = N == 32 ? 0u : (unsigned)(tile_row * PER + h * PIECES + q);
        frag[q] = (sb * (unsigned)(N * N) + (unsigned)row * N + (N == 32 ? 16u * h : 0u)) * 2u;
    }
    // input column u = kappa(c), rows v = 16h + t: byte offset = col_base + ((t / N) * 64 * N + (t % N) * 2 * N)
    const unsigned u = (unsigned)kappa(c);
    const unsigned col_base = ((u >> LOGN) * (unsigned)(N * N) + (u & (N - 1))) * 2u + (unsigned)h * 1024u;
    uint32_t w[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int t0 = 2 * m, t1 = 2 * m + 1;
        const unsigned c0 = (unsigned)((t0 >> LOGN) * 64 * N + (t0 & (N - 1)) * 2 * N);
        const unsigned c1 = (unsigned)((t1 >> LOGN) * 64 * N + (t1 & (N - 1)) * 2 * N);
        const uint32_t e0 = *reinterpret_cast<const uint16_t *>(slot + col_base + c0);
        const uint32_t e1 = *reinterpret_cast<const uint16_t *>(slot + col_base + c1);
        w[m] = e0 | (e1 << 16);
        if (PAIRWISE) __builtin_amdgcn_sched_barrier(0);              // two column reads, then their packing, pair after pair: the order the per-size kernels run fastest in
    }
    __builtin_amdgcn_wave_barrier();
    v4i lo, hi, r0, r1;
    split_planes(v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]}, v4i{(int)w[4], (int)w[5], (int)w[6], (int)w[7]}, lo, hi);
    inv_passes_with(lo, hi, k, c2r_group, r0, r1);
    const uint32_t z[8] = {(uint32_t)r0[0], (uint32_t)r0[1], (uint32_t)r0[2], (uint32_t)r0[3],
                           (uint32_t)r1[0], (uint32_t)r1[1], (uint32_t)r1[2], (uint32_t)r1[3]};
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        if (N >= 16) {
            *reinterpret_cast<v4i *>(slot + frag[q]) = r0;
            *reinterpret_cast<v4i *>(slot + frag[q] + 16) = r1;
        } else if (N == 8) {
            *reinterpret_cast<v4i *>(slot + frag[q]) = v4i{(int)z[4 * q], (int)z[4 * q + 1], (int)z[4 * q + 2], (int)z[4 * q + 3]};
        } else {
            *reinterpret_cast<uint2 *>(slot + frag[q]) = make_uint2(z[2 * q], z[2 * q + 1]);
        }
    }
}

template <int LOGN>
__device__ __forceinline__ void inv_tile_in_slot(unsigned char *slot, int lane, const LaneConsts &k, const v16i &c2r)
{
    auto group = [&](int g) { return v4i{c2r[4 * g], c2r[4 * g + 1], c2r[4 * g + 2], c2r[4 * g + 3]}; };
    inv_tile_in_slot_with<LOGN, decltype(group), true>(slot, lane, k, group);
}

__device__ __forceinline__ v16i load_c2r(const DctOps *__restrict__ ops, int h)
{
    const int *__restrict__ s0 = ops->c2r[0], *__restrict__ s1 = ops->c2r[32];
    v16i c2r;
#pragma unroll
    for (int r = 0; r < 16; ++r) c2r[r] = h ? s1[r] : s0[r];
    return c2r;
}

// Inverse transforms of the set (UNPINNED upstream; columns first, shifts 7 and 12, int16 clipping
// after each pass -- DESIGN.md section 10), contiguous batches, same block-diagonal tile idea.  The
// first contraction runs over the tile's ROW index, so each lane reads its COLUMN out of the staged
// tile (16 x ds_read_u16), exactly as the staged DCT32 inverse does.
template <int LOGN, bool INDEXED>
__device__ __forceinline__ void tr_inv_small_body(unsigned char *stage, size_t wave, const int16_t *__restrict__ in, int16_t *__restrict__ out,
                                                  size_t n_blocks, const DctOps *__restrict__ ops,
                                                  const uint32_t *__restrict__ offsets, unsigned tiles_per_wave)
{
    constexpr int N = 1 << LOGN;
    constexpr int PER = 32 / N, NSB = PER * PER;
    unsigned char *slot = stage + (threadIdx.x >> 6) * 2048;

    const int lane = threadIdx.x & 63;
    const size_t n_tiles = (n_blocks + NSB - 1) / NSB;
    size_t t = wave * tiles_per_wave;
    const size_t t_end = t + tiles_per_wave < n_tiles ? t + tiles_per_wave : n_tiles;
    if (t >= t_end) return;
    const LaneConsts k = load_consts(ops, lane);
    const v16i c2r = load_c2r(ops, lane >> 5);
    const size_t total_bytes = n_blocks * (size_t)(N * N * 2);
    for (; t < t_end; ++t) {
        size_t o0, o1;
        bool live0, live1;
        if (INDEXED) {                                                   // as in the forward kernel: chunk -> block of the offset table
            constexpr int CPB = N * N / 8;
            size_t b0 = t * NSB + (size_t)(lane / CPB), b1 = t * NSB + (size_t)((lane + 64) / CPB);
            live0 = b0 < n_blocks;
            live1 = b1 < n_blocks;
            if (!live0) b0 = n_blocks - 1;
            if (!live1) b1 = n_blocks - 1;
            o0 = ((size_t)offsets[b0] * 2) + (size_t)(lane % CPB) * 16;
            o1 = ((size_t)offsets[b1] * 2) + (size_t)((lane + 64) % CPB) * 16;
        } else {
            const size_t base = t * 2048;
            o0 = base + (size_t)lane * 16;
            o1 = o0 + 1024;
            live0 = o0 + 16 <= total_bytes;
            live1 = o1 + 16 <= total_bytes;
            if (!live0) o0 = total_bytes - 16;
            if (!live1) o1 = total_bytes - 16;
        }
        const v4i g0 = load16<true>(reinterpret_cast<const char *>(in) + o0);   // streaming hints, offset tables included (round 5: -8 % on the per-class calls over a CTU-ordered buffer, profiles/r05_result_stores.txt)
        const v4i g1 = load16<true>(reinterpret_cast<const char *>(in) + o1);
        *reinterpret_cast<v4i *>(slot + lane * 16) = g0;
        *reinterpret_cast<v4i *>(slot + 1024 + lane * 16) = g1;
        __builtin_amdgcn_wave_barrier();
        inv_tile_in_slot<LOGN>(slot, lane, k, c2r);
        __builtin_amdgcn_wave_barrier();
        const v4i s0 = *reinterpret_cast<const v4i *>(slot + lane * 16);
        const v4i s1 = *reinterpret_cast<const v4i *>(slot + 1024 + lane * 16);
        __builtin_amdgcn_wave_barrier();
        if (live0) store16m<2>(reinterpret_cast<char *>(out) + o0, s0);   // "sc1 nt" (x266_device.hpp)
        if (live1) store16m<2>(reinterpret_cast<char *>(out) + o1, s1);
    }
}

template <int LOGN>
__global__ __launch_bounds__(256) void tr_inv_small_lds_kernel(const int16_t *__restrict__ in, int16_t *__restrict__ out,
                                                               size_t n_blocks, const DctOps *__restrict__ ops,
                                                               const uint32_t *__restrict__ offsets, unsigned tiles_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;       // read before the branch, as in the forward kernel
    if (LOGN == 5 || offsets) tr_inv_small_body<LOGN, true>(stage, wave, in, out, n_blocks, ops, offsets, tiles_per_wave);
    else                      tr_inv_small_body<LOGN, LOGN == 5>(stage, wave, in, out, n_blocks, ops, offsets, tiles_per_wave);
}

// ---- mixed classes, one launch (BASELINE configs[3], "batched per CTU") ------------------------------------
// A CTU's residual is a sequence of 32x32 regions ("tiles", 1024 samples), each cut into (32/N)^2 blocks of ONE
// (type, size) class, block-major.  The per-class calls above walk the buffer once per class; this kernel walks it
// once: tile t carries its class in tile_class[t] (type * 4 + log2N - 2) and sits at sample offset tile_offsets[t]
// (NULL: t * 1024).  Consecutive tiles are consecutive memory, so the access stream is that of a contiguous batch.
//
// Round 3: the operand images are no longer fetched per class from memory (class byte -> images -> first pass: two
// dependent round trips where the per-class kernels have one, 6-7 % of the launch, profiles/r03_tiles_one_launch.txt).
// Every wave copies the set's 1-D matrices in compact form (TileTab, 2 KiB, L2-resident) into LDS with its first two
// loads -- they do not depend on the class, so they travel with the tile's data -- and builds the class's images from
// there once the class byte has arrived: the block-diagonal 32x32 operand of an N-point transform is the N x N matrix
// repeated along the diagonal, so a lane's 16 operand bytes are at most 16 bytes of ONE matrix row, masked to the
// K-slots that fall into the lane's diagonal block.
// class byte of tile t through the scalar cache: the aligned dword that holds it (t is wave-uniform, so this is an
// s_load_dword and the class lands in an SGPR without a vector-memory round trip), for any alignment of the table
__device__ __forceinline__ int tile_class_of(const uint8_t *__restrict__ tile_class, size_t t)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(tile_class) + t;
    const uint32_t w = *reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
    return (int)((w >> (8 * (unsigned)(a & 3))) & 15u);
}

// Operand bytes B[16h + t][idx], t = 0..15, of blockdiag(M) given as rows: M[idx % N][(16h + t) % N] where the K-slot
// 16h + t lies in block idx / N, else 0.  `m` = the compact N x N matrix in LDS whose ROW idx % N holds those bytes.
template <int LOGN>
__device__ __forceinline__ v4i image_row_segment(const unsigned char *m, unsigned idx, unsigned h)
{
    constexpr unsigned N = 1u << LOGN;
    const unsigned rr = idx & (N - 1), b = idx >> LOGN;
    if constexpr (LOGN == 5) {
        return *reinterpret_cast<const v4i *>(m + rr * 32 + 16 * h);
    } else if constexpr (LOGN == 4) {                                    // K-slots 16h .. 16h+15 = block h
        const v4i v = *reinterpret_cast<const v4i *>(m + rr * 16);
        const int keep = b == h ? -1 : 0;
        return v4i{v[0] & keep, v[1] & keep, v[2] & keep, v[3] & keep};
    } else if constexpr (LOGN == 3) {                                    // dwords 0,1 = block 2h, dwords 2,3 = block 2h + 1
        const uint2 w = *reinterpret_cast<const uint2 *>(m + rr * 8);
        const int k0 = b == 2 * h ? -1 : 0, k1 = b == 2 * h + 1 ? -1 : 0;
        return v4i{(int)w.x & k0, (int)w.y & k0, (int)w.x & k1, (int)w.y & k1};
    } else {                                                             // dword q = block 4h + q
        const int w = *reinterpret_cast<const int *>(m + rr * 4);
        return v4i{b == 4 * h ? w : 0, b == 4 * h + 1 ? w : 0, b == 4 * h + 2 ? w : 0, b == 4 * h + 3 ? w : 0};
    }
}

// Forward pass 2: B[acc_row(t, h)][c], i.e. dword q holds columns 4h + 8q .. +3 of row c of blockdiag(M).
template <int LOGN>
__device__ __forceinline__ v4i image_acc_rows(const unsigned char *m, unsigned c, unsigned h)
{
    constexpr unsigned N = 1u << LOGN;
    const unsigned rr = c & (N - 1), b = c >> LOGN;
    if constexpr (LOGN == 5) {
        const int *q = reinterpret_cast<const int *>(m + rr * 32 + 4 * h);
        return v4i{q[0], q[2], q[4], q[6]};
    } else if constexpr (LOGN == 4) {                                    // columns 4h + 8q: block q >> 1, dword h + 2 (q & 1) of the row
        const int *q = reinterpret_cast<const int *>(m + rr * 16 + 4 * h);
        const int w0 = q[0], w1 = q[2];
        return v4i{b == 0 ? w0 : 0, b == 0 ? w1 : 0, b == 1 ? w0 : 0, b == 1 ? w1 : 0};
    } else if constexpr (LOGN == 3) {                                    // block q, dword h of the row
        const int w = *reinterpret_cast<const int *>(m + rr * 8 + 4 * h);
        return v4i{b == 0 ? w : 0, b == 1 ? w : 0, b == 2 ? w : 0, b == 3 ? w : 0};
    } else {                                                             // block h + 2q, the row's one dword
        const int w = *reinterpret_cast<const int *>(m + rr * 4);
        return v4i{b == h ? w : 0, b == h + 2 ? w : 0, b == h + 4 ? w : 0, b == h + 6 ? w : 0};
    }
}

// tab: the wave's LDS copy of the TileTab; hs / vs: the slots (0 / 1) of the class's horizontal and vertical transform
template <int LOGN>
__device__ __forceinline__ void fwd_tile_of_class(unsigned char *slot, const unsigned char *tab, int lane, unsigned hs, unsigned vs)
{
    asm volatile("" : "+v"(lane));          // per-size lane arithmetic stays inside its branch: hoisted out of the four-way switch (x 2 tiles) it costs registers
    constexpr unsigned N = 1u << LOGN;
    constexpr int S1 = LOGN - 1, S2 = LOGN + 6;
    const unsigned c = lane & 31, h = lane >> 5, kc = (unsigned)kappa((int)c);
    const unsigned mh = tile_tab_mat(0, LOGN - 2) + (LOGN == 5 ? 0u : hs * 336u), mv = tile_tab_mat(0, LOGN - 2) + (LOGN == 5 ? 0u : vs * 336u);
    const unsigned sh = tile_tab_sum(0, LOGN - 2) + (LOGN == 5 ? 0u : hs * 112u), sv = tile_tab_sum(0, LOGN - 2) + (LOGN == 5 ? 0u : vs * 112u);
    LaneConsts k;
    k.p1 = image_row_segment<LOGN>(tab + mh, kc, h);                     // pass 1: M[kappa(c)][16h + t]
    k.p2 = image_acc_rows<LOGN>(tab + mv, c, h);                         // pass 2: M[c][acc_row(t, h)]
    k.tr = v4i{0, 0, 0, 0};
    k.c1 = (1 << (S1 - 1)) + reinterpret_cast<const int *>(tab + sh)[kc & (N - 1)];
    k.c2 = (1 << (S2 - 1)) + reinterpret_cast<const int *>(tab + sv)[c & (N - 1)];
    fwd_tile_in_slot<LOGN>(slot, lane, k);
}

// The inverse reads the table built from the TRANSPOSED matrices: pass A (vertical, columns first) M[16h + t][c] and
// pass B (horizontal) M[16h + t][kappa(c)] are both row segments there.
template <int LOGN>
__device__ __forceinline__ void inv_tile_of_class(unsigned char *slot, const unsigned char *tab, int lane, unsigned hs, unsigned vs)
{
    asm volatile("" : "+v"(lane));          // as in the forward body
    constexpr unsigned N = 1u << LOGN;
    const unsigned c = lane & 31, h = lane >> 5, kc = (unsigned)kappa((int)c);
    const unsigned mh = tile_tab_mat(0, LOGN - 2) + (LOGN == 5 ? 0u : hs * 336u), mv = tile_tab_mat(0, LOGN - 2) + (LOGN == 5 ? 0u : vs * 336u);
    const unsigned sh = tile_tab_sum(0, LOGN - 2) + (LOGN == 5 ? 0u : hs * 112u), sv = tile_tab_sum(0, LOGN - 2) + (LOGN == 5 ? 0u : vs * 112u);
    LaneConsts k;
    k.p1 = image_row_segment<LOGN>(tab + mv, c, h);
    k.p2 = image_row_segment<LOGN>(tab + mh, kc, h);
    k.tr = v4i{0, 0, 0, 0};
    k.c1 = (1 << 6) + reinterpret_cast<const int *>(tab + sv)[c & (N - 1)];
    k.c2 = 0;
    // pass-B constants per accumulator register: output column x = 16h + r, i.e. sum (16h + r) % N of the horizontal matrix
    const v4i *sums = reinterpret_cast<const v4i *>(tab + sh);
    inv_tile_in_slot_with<LOGN>(slot, lane, k, [&](int g) {
        constexpr int G = (int)N / 4;                                    // v4i groups holding the N sums
        const v4i v = LOGN == 5 ? sums[4 * h + g] : sums[g % (G < 4 ? G : 4)];
        return v4i{v[0] + (1 << 11), v[1] + (1 << 11), v[2] + (1 << 11), v[3] + (1 << 11)};
    });
}

template <bool INVERSE>
__device__ __forceinline__ void tile_of_class(unsigned char *slot, const unsigned char *tab, int lane, int cls)
{
    const unsigned type = (unsigned)cls >> 2;                           // 0 DCT-II, 1 DST-VII, 2 h = slot 1 / v = slot 0, 3 the other way (wave-uniform)
    const unsigned hs = (type == 1 || type == 2) ? 1u : 0u, vs = (type == 1 || type == 3) ? 1u : 0u;
    if (INVERSE) {
        switch (cls & 3) {
        case 0: inv_tile_of_class<2>(slot, tab, lane, hs, vs); break;
        case 1: inv_tile_of_class<3>(slot, tab, lane, hs, vs); break;
        case 2: inv_tile_of_class<4>(slot, tab, lane, hs, vs); break;
        default: inv_tile_of_class<5>(slot, tab, lane, hs, vs); break;
        }
    } else {
        switch (cls & 3) {
        case 0: fwd_tile_of_class<2>(slot, tab, lane, hs, vs); break;
        case 1: fwd_tile_of_class<3>(slot, tab, lane, hs, vs); break;
        case 2: fwd_tile_of_class<4>(slot, tab, lane, hs, vs); break;
        default: fwd_tile_of_class<5>(slot, tab, lane, hs, vs); break;
        }
    }
}

// A wave takes its tiles two at a time: both tiles' data, their class bytes and (once) the table are requested with the
// wave's first instructions and parked in LDS as they arrive -- nothing is held in registers across a tile's passes.
// LDS per wave: table 2 KiB, two tile slots of 2 KiB.
template <bool INVERSE>
__device__ __forceinline__ void tr_tiles_body(unsigned char *stage, size_t wave, const int16_t *__restrict__ in, int16_t *__restrict__ out, size_t n_tiles,
                                              const uint32_t *__restrict__ tile_offsets,
                                              const uint8_t *__restrict__ tile_class, const TileTab *__restrict__ T,
                                              unsigned tiles_per_wave, unsigned lds_per_wave)
{
    unsigned char *tab = stage + (threadIdx.x >> 6) * lds_per_wave;
    unsigned char *slot0 = tab + 2048, *slot1 = tab + 4096;
    const int lane = threadIdx.x & 63;
    size_t t = wave * tiles_per_wave;
    const size_t t_end = t + tiles_per_wave < n_tiles ? t + tiles_per_wave : n_tiles;
    if (t >= t_end) return;
    t = (size_t)__builtin_amdgcn_readfirstlane((int)(t & 0xFFFFFFFFu)) | (t & ~(size_t)0xFFFFFFFFu);   // provably wave-uniform
    bool first = true;
    for (; t < t_end; t += 2) {
        const bool two = t + 1 < t_end;                                 // wave-uniform
        const int cls0 = tile_class_of(tile_class, t), cls1 = tile_class_of(tile_class, two ? t + 1 : t);
        const size_t base0 = (tile_offsets ? (size_t)tile_offsets[t] : t * 1024) * 2;
        const size_t base1 = two ? (tile_offsets ? (size_t)tile_offsets[t + 1] : (t + 1) * 1024) * 2 : base0;
        const v4i a0 = load16<true>(reinterpret_cast<const char *>(in) + base0 + lane * 16);
        const v4i a1 = load16<true>(reinterpret_cast<const char *>(in) + base0 + 1024 + lane * 16);
        if (first) {
            const char *src = reinterpret_cast<const char *>(T->b) + lane * 16;
            const v4i q0 = *reinterpret_cast<const v4i *>(src), q1 = *reinterpret_cast<const v4i *>(src + 1024);
            *reinterpret_cast<v4i *>(tab + lane * 16) = q0;
            *reinterpret_cast<v4i *>(tab + 1024 + lane * 16) = q1;
            first = false;
        }
        if (two) {
            const v4i b0 = load16<true>(reinterpret_cast<const char *>(in) + base1 + lane * 16);
            const v4i b1 = load16<true>(reinterpret_cast<const char *>(in) + base1 + 1024 + lane * 16);
            *reinterpret_cast<v4i *>(slot1 + lane * 16) = b0;
            *reinterpret_cast<v4i *>(slot1 + 1024 + lane * 16) = b1;
        }
        *reinterpret_cast<v4i *>(slot0 + lane * 16) = a0;
        *reinterpret_cast<v4i *>(slot0 + 1024 + lane * 16) = a1;
        __builtin_amdgcn_wave_barrier();
        tile_of_class<INVERSE>(slot0, tab, lane, cls0);
        __builtin_amdgcn_wave_barrier();
        {
            const v4i s0 = *reinterpret_cast<const v4i *>(slot0 + lane * 16);
            const v4i s1 = *reinterpret_cast<const v4i *>(slot0 + 1024 + lane * 16);
            char *dst = reinterpret_cast<char *>(out) + base0 + lane * 16;
            store16m<2>(dst, s0);
            store16m<2>(dst + 1024, s1);
        }
        if (two) {
            tile_of_class<INVERSE>(slot1, tab, lane, cls1);
            __builtin_amdgcn_wave_barrier();
            const v4i s0 = *reinterpret_cast<const v4i *>(slot1 + lane * 16);
            const v4i s1 = *reinterpret_cast<const v4i *>(slot1 + 1024 + lane * 16);
            char *dst = reinterpret_cast<char *>(out) + base1 + lane * 16;
            store16m<2>(dst, s0);
            store16m<2>(dst + 1024, s1);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// streaming cache hints when the tiles are the buffer in order, none behind an offset table: two complete bodies, one wave-uniform test
template <bool INVERSE>
__global__ __launch_bounds__(256) void tr_tiles_kernel(const int16_t *__restrict__ in, int16_t *__restrict__ out, size_t n_tiles,
                                                       const uint32_t *__restrict__ tile_offsets,
                                                       const uint8_t *__restrict__ tile_class, const TileTab *__restrict__ T,
                                                       unsigned tiles_per_wave, unsigned lds_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char stage[];
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;       // read before the branch
    tr_tiles_body<INVERSE>(stage, wave, in, out, n_tiles, tile_offsets, tile_class, T, tiles_per_wave, lds_per_wave);
}

}  // namespace

hipError_t launch_transform_small(int log2n, const int16_t *d_in, int16_t *d_out, size_t n_blocks, const DctOps *d_ops,
                                  const uint32_t *d_offsets, const LaunchCfg &cfg, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    const size_t per_tile = (size_t)(32 >> log2n) * (size_t)(32 >> log2n);
    const size_t tiles = (n_blocks + per_tile - 1) / per_tile;
    const unsigned tpw = units_per_wave_for(cfg, tiles);
    const size_t waves = (tiles + tpw - 1) / tpw;
    const unsigned tpb = (unsigned)cfg.wg_threads;                // same launch shape as the staged DCT32 kernel
    const size_t wpw = tpb / 64, wgs = (waves + wpw - 1) / wpw;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const size_t lds = wpw * (size_t)cfg.lds_bytes_per_wave;
    dim3 grid((unsigned)wgs), block(tpb);
#define X266_TRL(L) hipLaunchKernelGGL((tr_fwd_small_lds_kernel<L>), grid, block, lds, stream, d_in, d_out, n_blocks, d_ops, d_offsets, tpw)
    if (log2n == 2) X266_TRL(2); else if (log2n == 3) X266_TRL(3); else if (log2n == 4) X266_TRL(4);
    else if (log2n == 5 && d_offsets) X266_TRL(5);                     // contiguous 32x32 batches are the DCT32 kernel's
    else return hipErrorInvalidValue;
#undef X266_TRL
    return hipGetLastError();
}

hipError_t launch_transform_small_inv(int log2n, const int16_t *d_in, int16_t *d_out, size_t n_blocks, const DctOps *d_ops,
                                      const uint32_t *d_offsets, const LaunchCfg &cfg, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    const size_t per_tile = (size_t)(32 >> log2n) * (size_t)(32 >> log2n);
    const size_t tiles = (n_blocks + per_tile - 1) / per_tile;
    const unsigned tpw = units_per_wave_for(cfg, tiles);
    const size_t waves = (tiles + tpw - 1) / tpw;
    const unsigned tpb = (unsigned)cfg.wg_threads;
    const size_t wpw = tpb / 64, wgs = (waves + wpw - 1) / wpw;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const size_t lds = wpw * (size_t)cfg.lds_bytes_per_wave;
    dim3 grid((unsigned)wgs), block(tpb);
#define X266_TRI(L) hipLaunchKernelGGL((tr_inv_small_lds_kernel<L>), grid, block, lds, stream, d_in, d_out, n_blocks, d_ops, d_offsets, tpw)
    if (log2n == 2) X266_TRI(2); else if (log2n == 3) X266_TRI(3); else if (log2n == 4) X266_TRI(4);
    else if (log2n == 5 && d_offsets) X266_TRI(5);
    else return hipErrorInvalidValue;
#undef X266_TRI
    return hipGetLastError();
}

hipError_t launch_transform_tiles(bool inverse, const int16_t *d_in, int16_t *d_out, size_t n_tiles, const uint32_t *d_tile_offsets,
                                  const uint8_t *d_tile_class, const TileTab *d_tab, const LaunchCfg &cfg, hipStream_t stream)
{
    if (n_tiles == 0) return hipSuccess;
    const unsigned tpw = units_per_wave_for(cfg, n_tiles);
    const size_t waves = (n_tiles + tpw - 1) / tpw;
    const unsigned tpb = (unsigned)cfg.wg_threads;
    const size_t wpw = tpb / 64, wgs = (waves + wpw - 1) / wpw;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const unsigned per_wave = (unsigned)(cfg.lds_bytes_per_wave < 6144 ? 6144 : (cfg.lds_bytes_per_wave + 15) & ~15);   // table + two tiles, then padding
    const size_t lds = wpw * (size_t)per_wave;
    dim3 grid((unsigned)wgs), block(tpb);
#define X266_TT(INV) hipLaunchKernelGGL((tr_tiles_kernel<INV>), grid, block, lds, stream, d_in, d_out, n_tiles, d_tile_offsets, d_tile_class, d_tab, tpw, per_wave)
    if (inverse) X266_TT(true); else X266_TT(false);
#undef X266_TT
    return hipGetLastError();
}

}  // namespace x266
