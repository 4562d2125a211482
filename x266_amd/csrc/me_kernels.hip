// me_kernels.hip -- full-search 8x8 SATD motion estimation for gfx950 (BASELINE
// configs[2]: 3840x2160 luma, 8x8 blocks, window +-64 => 2.157e9 SATDs / frame).
//
// Per-candidate cost is pinned by the reference: satd8x8(cur - ref) with
// satd8x8 = src_tb/satd.c:31-118 (9-bit differences, as the testbench feeds
// them, src/mkSatd.bsv:229).  The search harness around it -- candidate
// order, tie-break, padding -- has no upstream counterpart and is defined in
// include/x266hip.h (raster order dy-major, first minimum wins).
//
// Algorithm.  The Hadamard transform is linear and a 9-bit
// difference cannot wrap int16 (|coefficient| <= 64*255), so
//      satd(cur - ref) = (sum_m |Hc[m] - Hr[m]| + 2) >> 2,   Hc = H64*cur, Hr = H64*ref
// exactly.  Each workgroup owns a tile of 8 x TBY blocks:
//   1. the reference window of the tile (pixels ^ 0x80, i.e. signed) is staged in
//      LDS once (<= 30 KB of the CU's 160 KB);
//   2. Hc of the tile's blocks: one int8 MFMA group, kept in LDS in the lanes' own
//      fragment order (128 B per block, read back as broadcasts);
//   3. a wave takes 32 consecutive candidate POSITIONS of one window row, forms
//      Hr for all 32 with 4 x v_mfma_i32_32x32x32_i8 (pixels are one byte plane;
//      the -128 offset hits Hc and Hr alike and cancels), packs the 64
//      coefficients to 16 dwords per lane, and then scores those positions
//      against every block of the tile whose window contains them: 16 x
//      v_sad_u16 per block and lane (two |a-b| per instruction).  Blocks are
//      scored in pairs so that one v_permlane32_swap + add joins the two
//      coefficient halves of both;
//   4. running minima are kept as (cost << 16 | candidate index) keys, so one
//      v_min_u32 implements "lowest cost, then first in raster order".
// The bound is VALU issue (v_sad_u16), not HBM: the frame pair is ~18 MB.
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cstdint>

#include "x266_device.hpp"
#include "x266_hadamard.hpp"

namespace x266 {
namespace {

constexpr int kTileBlocksX = 8;           // blocks per tile row (64 pixels)

__device__ __forceinline__ uint32_t sad16(const uint32_t (&p)[16], const uint32_t *__restrict__ c, uint32_t init)
{
    // c: 16 dwords of this lane's half of a block's coefficients (LDS, broadcast reads)
    const v4i c0 = *reinterpret_cast<const v4i *>(c), c1 = *reinterpret_cast<const v4i *>(c + 4);
    const v4i c2 = *reinterpret_cast<const v4i *>(c + 8), c3 = *reinterpret_cast<const v4i *>(c + 12);
    uint32_t s = init;
#pragma unroll
    for (int k = 0; k < 4; ++k) s = __builtin_amdgcn_sad_u16(p[k], (uint32_t)c0[k], s);
#pragma unroll
    for (int k = 0; k < 4; ++k) s = __builtin_amdgcn_sad_u16(p[4 + k], (uint32_t)c1[k], s);
#pragma unroll
    for (int k = 0; k < 4; ++k) s = __builtin_amdgcn_sad_u16(p[8 + k], (uint32_t)c2[k], s);
#pragma unroll
    for (int k = 0; k < 4; ++k) s = __builtin_amdgcn_sad_u16(p[12 + k], (uint32_t)c3[k], s);
    return s;
}

__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

struct MeParams {
    const uint8_t *cur;
    const uint8_t *ref;           // pixel (0,0); valid for x,y in [-range, dim + range)
    long long cur_stride, ref_stride;
    int width, height, range;
    int blocks_x, blocks_y;       // width / 8, height / 8
    int tiles_x;
    int n_groups;                 // 32-position groups per window row
    int n_rows;                   // candidate rows per tile
    int pitch;                    // LDS bytes per window row
    x266_me_result_t *best;
    uint32_t *costs;              // optional [block][(2R+1)^2]
    uint32_t *keys;               // variant 3: per-block running minimum (cost << 16 | window position), merged with atomicMin
    int splits;                   // variant 3: workgroups per tile (each takes a band of the tile's candidate rows)
};

template <int TBY>
__global__ __launch_bounds__(256) void satd_search_kernel(const MeParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NBLK = kTileBlocksX * TBY;
    const int R = P.range, span = 2 * R + 1;
    const int win_rows = P.n_rows + 7;
    // LDS carve: [coefficients NBLK*128 B][best NBLK*4 B, padded to 128][window win_rows*pitch]
    uint32_t *c_lds = reinterpret_cast<uint32_t *>(smem);
    uint32_t *best_lds = reinterpret_cast<uint32_t *>(smem + NBLK * 128);
    unsigned char *win = smem + NBLK * 128 + 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
    const int n = lane & 31, half = lane >> 5;
    const int tx = blockIdx.x % P.tiles_x, ty = blockIdx.x / P.tiles_x;
    const int x0 = tx * (8 * kTileBlocksX), y0 = ty * (8 * TBY);       // tile origin in pixels

    // ---- 1. stage the reference window (signed pixels), 4 bytes per thread step -------------
    {
        const int dwords_per_row = P.pitch >> 2;
        const int total = win_rows * dwords_per_row;
        for (int i = tid; i < total; i += blockDim.x) {
            const int ry = i / dwords_per_row, cx = (i - ry * dwords_per_row) * 4;
            int gy = y0 - R + ry;
            gy = gy < -R ? -R : (gy > P.height + R - 1 ? P.height + R - 1 : gy);
            const uint8_t *row = P.ref + (long long)gy * P.ref_stride;
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                int gx = x0 - R + cx + b;
                gx = gx < -R ? -R : (gx > P.width + R - 1 ? P.width + R - 1 : gx);
                v |= (uint32_t)row[gx] << (8 * b);
            }
            reinterpret_cast<uint32_t *>(win)[i] = v ^ 0x80808080u;
        }
    }
    const HadamardOps H = make_hadamard_ops(lane);

    // ---- 2. transform the tile's current blocks (wave 0), initialise the minima ---------------
    if (tid < NBLK) best_lds[tid] = 0x7FFFFFFFu;
    if (wave == 0) {
        const int blk = n < NBLK ? n : NBLK - 1;
        int bx = tx * kTileBlocksX + (blk % kTileBlocksX), by = ty * TBY + (blk / kTileBlocksX);
        bx = bx < P.blocks_x ? bx : P.blocks_x - 1;                   // tiles hanging over the frame edge
        by = by < P.blocks_y ? by : P.blocks_y - 1;
        const uint8_t *src = P.cur + (long long)(by * 8 + 4 * half) * P.cur_stride + bx * 8;
        uint32_t w[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint8_t *q = src + (long long)r * P.cur_stride;
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) { lo |= (uint32_t)q[b] << (8 * b); hi |= (uint32_t)q[4 + b] << (8 * b); }
            w[2 * r] = lo ^ 0x80808080u;
            w[2 * r + 1] = hi ^ 0x80808080u;
        }
        uint32_t p[16];
        hadamard_pack(H, v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]}, v4i{(int)w[4], (int)w[5], (int)w[6], (int)w[7]}, p);
        if (n < NBLK) {
            uint32_t *dst = c_lds + (n * 2 + half) * 16;
#pragma unroll
            for (int k = 0; k < 16; ++k) dst[k] = p[k];
        }
    }
    __syncthreads();

    // ---- 3. candidate rows x position groups, round-robin over the waves ----------------------
    uint32_t best[TBY][4];
#pragma unroll
    for (int j = 0; j < TBY; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) best[j][q] = 0x7FFFFFFFu;

    const uint32_t round_init = half ? 0u : 2u;                        // the "+2" of (sum + 2) >> 2, once per candidate
    const int sh = (n & 3) * 8;                                        // byte alignment of this lane's window column
    const int n_items = P.n_rows * P.n_groups;
    for (int item = wave; item < n_items; item += n_waves) {
        const int r = item / P.n_groups, g = item - r * P.n_groups;    // wave-uniform
        // window rows r + 4*half .. +3, columns 32g + n .. +7  (aligned dwords + funnel shift)
        const unsigned char *base = win + (r + 4 * half) * P.pitch + ((32 * g + n) & ~3);
        uint32_t px[8];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(base + rr * P.pitch);
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
            px[2 * rr]     = __builtin_amdgcn_alignbit(d1, d0, sh);
            px[2 * rr + 1] = __builtin_amdgcn_alignbit(d2, d1, sh);
        }
        uint32_t p[16];
        hadamard_pack(H, v4i{(int)px[0], (int)px[1], (int)px[2], (int)px[3]}, v4i{(int)px[4], (int)px[5], (int)px[6], (int)px[7]}, p);

#pragma unroll
        for (int j = 0; j < TBY; ++j) {
            const int dyi = r - 8 * j;                                 // candidate row index of block row j
            if (dyi < 0 || dyi >= span) continue;                      // wave-uniform
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int lo0 = 32 * g - 16 * q;                       // dx index of lane 0 for block 2q; block 2q+1: lo0 - 8
                if (!((lo0 + 31 >= 0 && lo0 < span) || (lo0 + 23 >= 0 && lo0 - 8 < span))) continue;   // wave-uniform
                const uint32_t *c = c_lds + ((j * kTileBlocksX + 2 * q) * 2 + half) * 16;
                const uint32_t s0 = sad16(p, c, round_init);           // block 2q,   this lane's coefficient half
                const uint32_t s1 = sad16(p, c + 32, round_init);      // block 2q+1
                // lanes 0-31 end with block 2q, lanes 32-63 with block 2q+1, both halves summed
                const auto sw = __builtin_amdgcn_permlane32_swap(s0, s1, false, false);
                const uint32_t tot = (uint32_t)sw[0] + (uint32_t)sw[1];
                const int dxi = lo0 + n - 8 * half;
                const uint32_t idx = (uint32_t)(dyi * span + dxi);
                uint32_t key = ((tot >> 2) << 16) | idx;
                const bool ok = (unsigned)dxi < (unsigned)span;
                key = ok ? key : 0x7FFFFFFFu;
                best[j][q] = key < best[j][q] ? key : best[j][q];
                if (P.costs) {
                    const int bx = tx * kTileBlocksX + 2 * q + half, by = ty * TBY + j;
                    if (ok && bx < P.blocks_x && by < P.blocks_y)
                        P.costs[((size_t)by * P.blocks_x + bx) * (size_t)(span * span) + idx] = tot >> 2;
                }
            }
        }
    }

    // ---- 4. minima: across the 32 lanes of each half, then across waves (LDS) -----------------
#pragma unroll
    for (int j = 0; j < TBY; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t v = best[j][q];
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)v, m);
                v = o < v ? o : v;
            }
            if (n == 0) atomicMin(&best_lds[j * kTileBlocksX + 2 * q + half], v);
        }
    __syncthreads();
    if (tid < NBLK) {
        const int bx = tx * kTileBlocksX + (tid % kTileBlocksX), by = ty * TBY + (tid / kTileBlocksX);
        if (bx < P.blocks_x && by < P.blocks_y) {
            const uint32_t key = best_lds[tid];
            const int idx = (int)(key & 0xFFFFu);
            x266_me_result_t res;
            res.mvx = (int16_t)(idx % span - R);
            res.mvy = (int16_t)(idx / span - R);
            res.cost = key >> 16;
            P.best[(size_t)by * P.blocks_x + bx] = res;
        }
    }
}


// ============================================================================
// Variant 2 ("scalar coefficients").  The same 32 columns of two consecutive
// candidate rows are transformed back to back and their coefficient halves
// exchanged between the half-waves
// (16 x v_permlane32_swap), after which EVERY lane owns all 64 coefficients of
// one position.  All lanes then need the same block coefficients, so those come
// from SGPRs (scalar loads of a table written by a small pre-pass kernel) and
// the v_sad_u16 chain takes them as its scalar operand: no LDS traffic and no
// cross-half reduction in the scoring loop.
// ============================================================================

// Pre-pass: Hc of every 8x8 block of the current frame, 32 dwords per block in the
// order the search kernel's lanes hold a position: [coefficient set of half 0][half 1].
// Blocks are stored TILE-MAJOR (search tile, then block row, then block column inside the tile):
// the search kernel addresses all blocks of its tile from one scalar base with immediate offsets.
__global__ __launch_bounds__(256) void me_coef_kernel(const uint8_t *__restrict__ cur, long long cur_stride,
                                                      int blocks_x, int n_blocks, int tiles_x, int tby,
                                                      uint32_t *__restrict__ coef, uint32_t *__restrict__ keys)
{
    const int lane = threadIdx.x & 63, n = lane & 31, half = lane >> 5;
    const int group = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (group * 32 >= n_blocks) return;
    if (keys && half == 0 && group * 32 + n < n_blocks) keys[group * 32 + n] = 0xFFFFFFFFu;   // variant 3 merges into these
    int blk = group * 32 + n;
    const bool live = blk < n_blocks;
    if (!live) blk = n_blocks - 1;
    const int bx = blk % blocks_x, by = blk / blocks_x;
    const uint8_t *src = cur + (long long)(by * 8 + 4 * half) * cur_stride + bx * 8;
    uint32_t w[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint8_t *q = src + (long long)r * cur_stride;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) { lo |= (uint32_t)q[b] << (8 * b); hi |= (uint32_t)q[4 + b] << (8 * b); }
        w[2 * r] = lo ^ 0x80808080u;
        w[2 * r + 1] = hi ^ 0x80808080u;
    }
    const HadamardOps H = make_hadamard_ops(lane);
    uint32_t p[16];
    hadamard_pack(H, v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]}, v4i{(int)w[4], (int)w[5], (int)w[6], (int)w[7]}, p);
    if (live) {
        const size_t slot = ((size_t)(by / tby) * tiles_x + bx / kTileBlocksX) * (size_t)(kTileBlocksX * tby)
                            + (size_t)(by % tby) * kTileBlocksX + bx % kTileBlocksX;
        v4i *dst = reinterpret_cast<v4i *>(coef + slot * 32 + 16 * half);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = v4i{(int)p[4 * k], (int)p[4 * k + 1], (int)p[4 * k + 2], (int)p[4 * k + 3]};
    }
}

// window rows r + 4*half .. +3, 8 columns from byte column `col` (aligned dwords + funnel shift)
__device__ __forceinline__ void load_window(const unsigned char *win, int pitch, int row, int col, int sh, v4i &b0, v4i &b1)
{
    const unsigned char *base = win + row * pitch + (col & ~3);
    uint32_t px[8];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(base + rr * pitch);
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
        px[2 * rr]     = __builtin_amdgcn_alignbit(d1, d0, sh);
        px[2 * rr + 1] = __builtin_amdgcn_alignbit(d2, d1, sh);
    }
    b0 = v4i{(int)px[0], (int)px[1], (int)px[2], (int)px[3]};
    b1 = v4i{(int)px[4], (int)px[5], (int)px[6], (int)px[7]};
}

// U = row pairs scored per coefficient fetch (each lane then holds U positions): the scalar
// loads of a block's 32 coefficient dwords are amortised over 32*U v_sad_u16.
// COSTS: also write the full cost map (test / analysis path; keeps its per-block pointers out of
// the search-only kernel's scalar registers).
template <int TBY, int U, bool COSTS>
__global__ __launch_bounds__(256) void satd_search_kernel_v2(const MeParams P, const uint32_t *__restrict__ coef)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NBLK = kTileBlocksX * TBY;
    const int R = P.range, span = 2 * R + 1;
    const int win_rows = P.n_rows + 7 + 2 * U;
    uint32_t *best_lds = reinterpret_cast<uint32_t *>(smem);
    unsigned char *win = smem + 128;

    const int tid = threadIdx.x, lane = tid & 63, n_waves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // provably wave-uniform: scalar branches, scalar loads
    const int n = lane & 31, half = lane >> 5;
    const int tx = blockIdx.x % P.tiles_x, ty = blockIdx.x / P.tiles_x;
    const int x0 = tx * (8 * kTileBlocksX), y0 = ty * (8 * TBY);

    {   // reference window, signed pixels
        const int dwords_per_row = P.pitch >> 2;
        const int total = win_rows * dwords_per_row;
        for (int i = tid; i < total; i += blockDim.x) {
            const int ry = i / dwords_per_row, cx = (i - ry * dwords_per_row) * 4;
            int gy = y0 - R + ry;
            gy = gy < -R ? -R : (gy > P.height + R - 1 ? P.height + R - 1 : gy);
            const uint8_t *row = P.ref + (long long)gy * P.ref_stride;
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                int gx = x0 - R + cx + b;
                gx = gx < -R ? -R : (gx > P.width + R - 1 ? P.width + R - 1 : gx);
                v |= (uint32_t)row[gx] << (8 * b);
            }
            reinterpret_cast<uint32_t *>(win)[i] = v ^ 0x80808080u;
        }
    }
    if (tid < NBLK) best_lds[tid] = 0x7FFFFFFFu;
    const HadamardOps H = make_hadamard_ops(lane);
    __syncthreads();

    uint32_t best[TBY][kTileBlocksX];
#pragma unroll
    for (int j = 0; j < TBY; ++j)
#pragma unroll
        for (int i = 0; i < kTileBlocksX; ++i) best[j][i] = 0x7FFFFFFFu;

    const int sh = (n & 3) * 8;
    const int n_strips = (P.n_rows + 2 * U - 1) / (2 * U);             // unit = 2*U consecutive candidate rows x one group
    const int n_items = n_strips * P.n_groups;
    const int blocks_left_x = P.blocks_x - tx * kTileBlocksX, blocks_left_y = P.blocks_y - ty * TBY;
    // after the half exchange lanes 0-31 own the odd row of a pair, lanes 32-63 the even row
    const int lane_row = half ? 0 : 1;
    const uint32_t *__restrict__ tile_coef = coef + (size_t)blockIdx.x * (NBLK * 32);
    const uint32_t lane_idx = (uint32_t)(lane_row * span + n);
    for (int item = wave; item < n_items; item += n_waves) {
        const int strip = item / P.n_groups, g = item - strip * P.n_groups;   // wave-uniform
        const int r = 2 * U * strip;
        uint32_t a[U][16], b[U][16];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v4i w0, w1;
            load_window(win, P.pitch, r + 2 * u + 4 * half, 32 * g + n, sh, w0, w1);
            hadamard_pack(H, w0, w1, a[u]);
            load_window(win, P.pitch, r + 2 * u + 1 + 4 * half, 32 * g + n, sh, w0, w1);
            hadamard_pack(H, w0, w1, b[u]);
            // exchange halves: every lane now owns all 64 coefficients of one position,
            // b[u][0..15] = coefficient set of half 0, a[u][0..15] = set of half 1
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const auto sw = __builtin_amdgcn_permlane32_swap(b[u][k], a[u][k], false, false);
                b[u][k] = (uint32_t)sw[0];
                a[u][k] = (uint32_t)sw[1];
            }
        }

#pragma unroll
        for (int j = 0; j < TBY; ++j) {
            const int dy0 = r - 8 * j;                                 // candidate row index of row r for block row j
            if (dy0 + 2 * U - 1 < 0 || dy0 >= span || j >= blocks_left_y) continue;   // wave-uniform
#pragma unroll
            for (int i = 0; i < kTileBlocksX; ++i) {
                const int lo = 32 * g - 8 * i;                         // dx index of lane 0 for block i
                if (lo + 31 < 0 || lo >= span || i >= blocks_left_x) continue;       // wave-uniform
                const uint32_t *__restrict__ c = tile_coef + (j * kTileBlocksX + i) * 32;
                uint32_t s[U];
#pragma unroll
                for (int u = 0; u < U; ++u) s[u] = 2u;                 // the "+2" of (sum + 2) >> 2
#pragma unroll
                for (int k = 0; k < 16; ++k)
#pragma unroll
                    for (int u = 0; u < U; ++u) s[u] = __builtin_amdgcn_sad_u16(b[u][k], c[k], s[u]);
#pragma unroll
                for (int k = 0; k < 16; ++k)
#pragma unroll
                    for (int u = 0; u < U; ++u) s[u] = __builtin_amdgcn_sad_u16(a[u][k], c[16 + k], s[u]);
                const bool cols_ok = lo >= 0 && lo + 31 < span;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int dyu = dy0 + 2 * u;
                    if (dyu + 1 < 0 || dyu >= span) continue;          // wave-uniform: this row pair is outside the window
                    const uint32_t idx = lane_idx + (uint32_t)(dyu * span + lo);
                    uint32_t key = ((s[u] << 14) & 0xFFFF0000u) | idx; // (s >> 2) << 16 | candidate index
                    bool ok = true;
                    if (!(cols_ok && dyu >= 0 && dyu + 1 < span)) {    // wave-uniform: unit straddles the window edge
                        ok = (unsigned)(lo + n) < (unsigned)span && (unsigned)(dyu + lane_row) < (unsigned)span;
                        key = ok ? key : 0x7FFFFFFFu;
                    }
                    best[j][i] = key < best[j][i] ? key : best[j][i];
                    if (COSTS && ok) {
                        const size_t blk = (size_t)(ty * TBY + j) * P.blocks_x + (tx * kTileBlocksX + i);
                        P.costs[blk * (size_t)(span * span) + idx] = s[u] >> 2;
                    }
                }
            }
        }
    }

#pragma unroll
    for (int j = 0; j < TBY; ++j)
#pragma unroll
        for (int i = 0; i < kTileBlocksX; ++i) {
            uint32_t v = best[j][i];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)v, m);
                v = o < v ? o : v;
            }
            if (lane == 0) atomicMin(&best_lds[j * kTileBlocksX + i], v);
        }
    __syncthreads();
    if (tid < NBLK) {
        const int bx = tx * kTileBlocksX + (tid % kTileBlocksX), by = ty * TBY + (tid / kTileBlocksX);
        if (bx < P.blocks_x && by < P.blocks_y) {
            const uint32_t key = best_lds[tid];
            const int idx = (int)(key & 0xFFFFu);
            x266_me_result_t res;
            res.mvx = (int16_t)(idx % span - R);
            res.mvy = (int16_t)(idx / span - R);
            res.cost = key >> 16;
            P.best[(size_t)by * P.blocks_x + bx] = res;
        }
    }
}

// ============================================================================
// Variant 3 (default).  Same algorithm as variant 2 -- transform every candidate POSITION once,
// exchange the coefficient halves so that a lane owns all 64 coefficients of one position, score
// against block coefficients held in SGPRs -- with the instruction count cut where the SQ counters
// of variant 2 showed it going (profiles/r01_pmc_sq_counters.csv: VALU 90 % busy at 1.84x the
// v_sad_u16 floor's instruction count):
//   * unit geometry 16 columns x 4 rows instead of 32 x 2: a block's window (2R+1 wide, starting at
//     a multiple of 8) is covered by 16-column groups with at most 15 + 7 idle columns instead of
//     31 + 24, and by 4-row units with at most 3 idle rows: 87 % of the scored lanes are valid
//     candidates at R = 64 (81 % before);
//   * the running minimum is keyed by POSITION, (cost << 16) | (window row << 8 | window column):
//     the same for every block, so it is formed once per unit instead of once per (unit, block),
//     and raster order of positions is raster order of candidates for any block;
//     per (unit, block): v_and, v_lshl_or and half a v_min3 (was ~10 instructions);
//   * window-edge units take a separate path with per-lane validity; full units have none;
//   * 192-thread workgroups when that fills the chip in fewer, fuller rounds (the kernel needs
//     ~150 VGPRs = 3 waves per SIMD: 4080 tiles of a 4K frame over 768 four-wave slots are 5.3
//     rounds, over 1024 three-wave slots 3.98).
// ============================================================================
template <int TBY, int U, bool COSTS>
__global__ __launch_bounds__(256, (U == 1 ? 4 : 3)) void satd_search_kernel_v3(const MeParams P, const uint32_t *__restrict__ coef)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NBLK = kTileBlocksX * TBY;
    const int R = P.range, span = 2 * R + 1;
    const int n_units_y = (P.n_rows + 3) / 4;                          // 4-row units
    const int n_items_y = (n_units_y + U - 1) / U;
    // this workgroup's band of the tile's item rows (P.splits workgroups per tile: finer dispatch granularity,
    // so that the last round of resident workgroups is not mostly empty -- profiles/r02_me_tail.txt)
    const int tile = blockIdx.x / P.splits, split = blockIdx.x - tile * P.splits;
    const int iy_begin = (n_items_y * split) / P.splits, iy_end = (n_items_y * (split + 1)) / P.splits;
    const int row_begin = 4 * U * iy_begin;                            // first window row this band reads
    const int win_rows = 4 * U * (iy_end - iy_begin) + 7;
    uint32_t *best_lds = reinterpret_cast<uint32_t *>(smem);
    unsigned char *win = smem + 128;

    const int tid = threadIdx.x, lane = tid & 63, n_waves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, half = lane >> 5;
    const int tx = tile % P.tiles_x, ty = tile / P.tiles_x;
    const int x0 = tx * (8 * kTileBlocksX), y0 = ty * (8 * TBY);

    {   // reference window, signed pixels
        const int dwords_per_row = P.pitch >> 2;
        const int total = win_rows * dwords_per_row;
        for (int i = tid; i < total; i += blockDim.x) {
            const int ry = i / dwords_per_row, cx = (i - ry * dwords_per_row) * 4;
            int gy = y0 - R + row_begin + ry;
            gy = gy < -R ? -R : (gy > P.height + R - 1 ? P.height + R - 1 : gy);
            const uint8_t *row = P.ref + (long long)gy * P.ref_stride;
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                int gx = x0 - R + cx + b;
                gx = gx < -R ? -R : (gx > P.width + R - 1 ? P.width + R - 1 : gx);
                v |= (uint32_t)row[gx] << (8 * b);
            }
            reinterpret_cast<uint32_t *>(win)[i] = v ^ 0x80808080u;
        }
    }
    if (tid < NBLK) best_lds[tid] = 0xFFFFFFFFu;
    const HadamardOps H = make_hadamard_ops(lane);
    __syncthreads();

    uint32_t best[TBY][kTileBlocksX];
#pragma unroll
    for (int j = 0; j < TBY; ++j)
#pragma unroll
        for (int i = 0; i < kTileBlocksX; ++i) best[j][i] = 0xFFFFFFFFu;

    const int col_in = n & 15, row_in = n >> 4;                        // this lane's column / row inside a 16 x 2 transform
    const int sh = (col_in & 3) * 8;
    // after the half exchange lanes 0-31 own rows 2, 3 of the unit, lanes 32-63 rows 0, 1
    const int lane_row = (half ? 0 : 2) + row_in;
    const int n_items = (iy_end - iy_begin) * P.n_groups;
    const int blocks_left_x = P.blocks_x - tx * kTileBlocksX, blocks_left_y = P.blocks_y - ty * TBY;
    const uint32_t *__restrict__ tile_coef = coef + (size_t)tile * (NBLK * 32);
    for (int item = wave; item < n_items; item += n_waves) {
        const int iyl = item / P.n_groups, g = item - iyl * P.n_groups;   // wave-uniform
        const int r = 4 * U * (iy_begin + iyl);                        // first window row (tile coordinates) of the item's U units
        uint32_t a[U][16], b[U][16], pid[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v4i w0, w1;
            load_window(win, P.pitch, r - row_begin + 4 * u + row_in + 4 * half, 16 * g + col_in, sh, w0, w1);
            hadamard_pack(H, w0, w1, a[u]);
            load_window(win, P.pitch, r - row_begin + 4 * u + 2 + row_in + 4 * half, 16 * g + col_in, sh, w0, w1);
            hadamard_pack(H, w0, w1, b[u]);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const auto sw = __builtin_amdgcn_permlane32_swap(b[u][k], a[u][k], false, false);
                b[u][k] = (uint32_t)sw[0];
                a[u][k] = (uint32_t)sw[1];
            }
            pid[u] = (uint32_t)((r + 4 * u + lane_row) << 8 | (16 * g + col_in));
        }

#pragma unroll
        for (int j = 0; j < TBY; ++j) {
            const int dy0 = r - 8 * j;                                 // candidate row index of window row r for block row j
            if (dy0 + 4 * U - 1 < 0 || dy0 >= span || j >= blocks_left_y) continue;   // wave-uniform
            const bool rows_full = dy0 >= 0 && dy0 + 4 * U - 1 < span;
#pragma unroll
            for (int i = 0; i < kTileBlocksX; ++i) {
                const int lo = 16 * g - 8 * i;                         // dx index of column 0 of the group for block i
                if (lo + 15 < 0 || lo >= span || i >= blocks_left_x) continue;       // wave-uniform
                const uint32_t *__restrict__ c = tile_coef + (j * kTileBlocksX + i) * 32;
                uint32_t s[U];
#pragma unroll
                for (int u = 0; u < U; ++u) s[u] = 2u;                 // the "+2" of (sum + 2) >> 2
#pragma unroll
                for (int k = 0; k < 16; ++k)
#pragma unroll
                    for (int u = 0; u < U; ++u) s[u] = __builtin_amdgcn_sad_u16(b[u][k], c[k], s[u]);
#pragma unroll
                for (int k = 0; k < 16; ++k)
#pragma unroll
                    for (int u = 0; u < U; ++u) s[u] = __builtin_amdgcn_sad_u16(a[u][k], c[16 + k], s[u]);
                uint32_t key[U];
#pragma unroll
                for (int u = 0; u < U; ++u) key[u] = ((s[u] & ~3u) << 14) | pid[u];      // (s >> 2) << 16 | position
                if (!(rows_full && lo >= 0 && lo + 15 < span)) {       // wave-uniform: the item straddles the block's window edge
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const bool ok = (unsigned)(lo + col_in) < (unsigned)span && (unsigned)(dy0 + 4 * u + lane_row) < (unsigned)span;
                        key[u] = ok ? key[u] : 0xFFFFFFFFu;
                    }
                }
                if (U == 2) {
                    best[j][i] = min3u(best[j][i], key[0], key[1]);
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) best[j][i] = key[u] < best[j][i] ? key[u] : best[j][i];
                }
                if (COSTS) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int dxi = lo + col_in, dyi = dy0 + 4 * u + lane_row;
                        if ((unsigned)dxi < (unsigned)span && (unsigned)dyi < (unsigned)span) {
                            const size_t blk = (size_t)(ty * TBY + j) * P.blocks_x + (tx * kTileBlocksX + i);
                            P.costs[blk * (size_t)(span * span) + (size_t)(dyi * span + dxi)] = s[u] >> 2;
                        }
                    }
                }
            }
        }
    }

#pragma unroll
    for (int j = 0; j < TBY; ++j)
#pragma unroll
        for (int i = 0; i < kTileBlocksX; ++i) {
            uint32_t v = best[j][i];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)v, m);
                v = o < v ? o : v;
            }
            if (lane == 0) atomicMin(&best_lds[j * kTileBlocksX + i], v);
        }
    __syncthreads();
    if (tid < NBLK) {                                                  // merge with the tile's other bands (me_decode_kernel reads the result)
        const int bi = tid % kTileBlocksX, bj = tid / kTileBlocksX;
        const int bx = tx * kTileBlocksX + bi, by = ty * TBY + bj;
        if (bx < P.blocks_x && by < P.blocks_y && best_lds[tid] != 0xFFFFFFFFu) atomicMin(&P.keys[(size_t)by * P.blocks_x + bx], best_lds[tid]);
    }
}

// keys -> (mv, cost) records: window position of the winner minus the block's offset inside its tile
__global__ __launch_bounds__(256) void me_decode_kernel(const uint32_t *__restrict__ keys, x266_me_result_t *__restrict__ best,
                                                        int blocks_x, int n_blocks, int tby, int range)
{
    const int blk = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (blk >= n_blocks) return;
    const int bx = blk % blocks_x, by = blk / blocks_x;
    const uint32_t key = keys[blk];
    x266_me_result_t res;
    res.mvx = (int16_t)((int)(key & 0xFFu) - 8 * (bx % kTileBlocksX) - range);          // window column -> dx
    res.mvy = (int16_t)((int)((key >> 8) & 0xFFu) - 8 * (by % tby) - range);            // window row -> dy
    res.cost = key >> 16;
    best[blk] = res;
}

// ============================================================================
// Variant 4 (default).  Variant 3's machinery with NO idle lanes in the main loop.
//
// A block's window is (2R+1)^2 candidates starting at a multiple of 8 in both directions.  Write
// 2R+1 = 8F + rem (rem odd, 1 for R = 64).  Units are 8 columns x 8 rows of positions (two 8x4
// transforms + the half exchange), on the 8-pixel grid of the blocks, so for ANY block the aligned part
// of its window, [8i, 8i+8F) x [8j, 8j+8F), is exactly F x F whole units: a (unit, block) pair is either
// entirely valid or not needed, and the scoring loop has no per-lane validity, no edge path.  Items pair
// two vertically adjacent units (one coefficient fetch per 64 v_sad_u16); a block row whose window starts on
// the odd unit of a pair takes single-unit paths at its first and last unit row instead of idle lanes.
// The running minima live in LDS, one slot per (block, lane), updated with ds_min_u32 (issued beside the
// VALU stream, 6 % of the LDS pipe): no registers, so the block-row loop is a real loop and the
// 64-instruction scoring bodies exist once per block column and path.
// The remaining rem columns and rem rows of every window ("+1" at R = 64: 257 of 16641 candidates) are
// scored by narrow units -- 64 positions down one column, or along one row -- with per-lane validity.
// Instruction count per 4K frame: 1.67e9 (variant 3) -> see profiles/r02_me_variants.txt.
// ============================================================================
template <int TBY, bool COSTS, int WG, int WPS>
__global__ __launch_bounds__(WG, WPS) void satd_search_kernel_v4(const MeParams P, const uint32_t *__restrict__ coef)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NBLK = kTileBlocksX * TBY;
    const int R = P.range, span = 2 * R + 1;
    const int F = span >> 3, rem = span - 8 * F;                        // window = 8F aligned + rem (odd) more
    const int n_ucols = kTileBlocksX - 1 + F;                          // main unit columns
    const int n_item_rows = (TBY - 1 + F + 1) >> 1;                    // main items: pairs of unit rows
    const int n_rows = 8 * (TBY - 1) + span, n_cols = 8 * (kTileBlocksX - 1) + span;   // candidate positions of the tile
    const int main_rows = 16 * n_item_rows;
    const int win_rows = (main_rows > n_rows ? main_rows : n_rows) + 7;
    uint32_t *slots = reinterpret_cast<uint32_t *>(smem);             // running minima: [block][lane], shared by the workgroup's waves
    unsigned char *win = smem + NBLK * 256;

    const int tid = threadIdx.x, lane = tid & 63, n_waves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, half = lane >> 5;
    const int tx = blockIdx.x % P.tiles_x, ty = blockIdx.x / P.tiles_x;
    const int x0 = tx * (8 * kTileBlocksX), y0 = ty * (8 * TBY);

    {   // reference window, signed pixels
        const int dwords_per_row = P.pitch >> 2;
        const int total = win_rows * dwords_per_row;
        for (int i = tid; i < total; i += blockDim.x) {
            const int ry = i / dwords_per_row, cx = (i - ry * dwords_per_row) * 4;
            int gy = y0 - R + ry;
            gy = gy < -R ? -R : (gy > P.height + R - 1 ? P.height + R - 1 : gy);
            const uint8_t *row = P.ref + (long long)gy * P.ref_stride;
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                int gx = x0 - R + cx + b;
                gx = gx < -R ? -R : (gx > P.width + R - 1 ? P.width + R - 1 : gx);
                v |= (uint32_t)row[gx] << (8 * b);
            }
            reinterpret_cast<uint32_t *>(win)[i] = v ^ 0x80808080u;
        }
    }
    for (int i = tid; i < NBLK * 64; i += blockDim.x) slots[i] = 0xFFFFFFFFu;
    const HadamardOps H = make_hadamard_ops(lane);
    __syncthreads();
    uint32_t *my_slot = slots + lane;

    int blocks_left_x = P.blocks_x - tx * kTileBlocksX, blocks_left_y = P.blocks_y - ty * TBY;
    blocks_left_x = blocks_left_x > kTileBlocksX ? kTileBlocksX : blocks_left_x;
    blocks_left_y = blocks_left_y > TBY ? TBY : blocks_left_y;
    const uint32_t *__restrict__ tile_coef = coef + (size_t)blockIdx.x * (NBLK * 32);

    // ---- item list: main (pairs of 8x8 units), then narrow columns, then narrow rows ---------------------
    const int n_main = F > 0 ? n_item_rows * n_ucols : 0;
    const int n_chunk_r = (n_rows + 63) >> 6, n_chunk_c = (n_cols + 63) >> 6;
    const int n_ncol = kTileBlocksX * rem * n_chunk_r;                  // (block column, extra column, 64-row chunk)
    const int n_nrow = F > 0 ? TBY * rem * n_chunk_c : 0;               // (block row, extra row, 64-column chunk); F = 0: the columns cover it all
    const int n_items = n_main + n_ncol + n_nrow;

    for (int item = wave; item < n_items; item += n_waves) {
        if (item < n_main) {
            // ================= main item: units (2m, ux) and (2m+1, ux) ====================================
            const int m = item / n_ucols, ux = item - m * n_ucols;     // wave-uniform
            const int col_in = n & 7, row_in = n >> 3;                 // 8 columns x 4 rows per transform
            const int sh = (col_in & 3) * 8;
            const int lane_row = (half ? 0 : 4) + row_in;              // after the exchange lanes 0-31 own rows 4..7 of the unit
            uint32_t a[2][16], b[2][16], pid[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = 8 * (2 * m + u);
                v4i w0, w1;
                load_window(win, P.pitch, r + row_in + 4 * half, 8 * ux + col_in, sh, w0, w1);
                hadamard_pack(H, w0, w1, a[u]);
                load_window(win, P.pitch, r + 4 + row_in + 4 * half, 8 * ux + col_in, sh, w0, w1);
                hadamard_pack(H, w0, w1, b[u]);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(b[u][k], a[u][k], false, false);
                    b[u][k] = (uint32_t)sw[0];
                    a[u][k] = (uint32_t)sw[1];
                }
                pid[u] = (uint32_t)((r + lane_row) << 8 | (8 * ux + col_in));
            }
#pragma unroll 1
            for (int j = 0; j < blocks_left_y; ++j) {
                const bool va = (unsigned)(2 * m - j) < (unsigned)F, vb = (unsigned)(2 * m + 1 - j) < (unsigned)F;   // unit rows inside block row j's window
                if (!va && !vb) continue;
                const uint32_t *__restrict__ crow = tile_coef + j * (kTileBlocksX * 32);
                if (va && vb) {
#pragma unroll
                    for (int i = 0; i < kTileBlocksX; ++i) {
                        if ((unsigned)(ux - i) >= (unsigned)F || i >= blocks_left_x) continue;          // wave-uniform
                        const uint32_t *__restrict__ c = crow + i * 32;
                        uint32_t s0 = 2u, s1 = 2u;                     // the "+2" of (sum + 2) >> 2
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            s0 = __builtin_amdgcn_sad_u16(b[0][k], c[k], s0);
                            s1 = __builtin_amdgcn_sad_u16(b[1][k], c[k], s1);
                        }
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            s0 = __builtin_amdgcn_sad_u16(a[0][k], c[16 + k], s0);
                            s1 = __builtin_amdgcn_sad_u16(a[1][k], c[16 + k], s1);
                        }
                        const uint32_t k0 = ((s0 & ~3u) << 14) | pid[0], k1 = ((s1 & ~3u) << 14) | pid[1];   // (s >> 2) << 16 | position
                        atomicMin(my_slot + (j * kTileBlocksX + i) * 64, k0);           // ds_min_u32, no return value
                        atomicMin(my_slot + (j * kTileBlocksX + i) * 64, k1);
                        if (COSTS) {
                            const size_t blk = (size_t)(ty * TBY + j) * P.blocks_x + (tx * kTileBlocksX + i);
                            uint32_t *cm = P.costs + blk * (size_t)(span * span) + (size_t)(8 * ux + col_in - 8 * i);
                            cm[(size_t)(16 * m + lane_row - 8 * j) * span] = s0 >> 2;
                            cm[(size_t)(16 * m + 8 + lane_row - 8 * j) * span] = s1 >> 2;
                        }
                    }
                } else {
                    const int u = va ? 0 : 1;                          // only one unit row of the pair is inside the window
#pragma unroll
                    for (int i = 0; i < kTileBlocksX; ++i) {
                        if ((unsigned)(ux - i) >= (unsigned)F || i >= blocks_left_x) continue;
                        const uint32_t *__restrict__ c = crow + i * 32;
                        uint32_t s0 = 2u;
                        if (u == 0) {
#pragma unroll
                            for (int k = 0; k < 16; ++k) s0 = __builtin_amdgcn_sad_u16(b[0][k], c[k], s0);
#pragma unroll
                            for (int k = 0; k < 16; ++k) s0 = __builtin_amdgcn_sad_u16(a[0][k], c[16 + k], s0);
                        } else {
#pragma unroll
                            for (int k = 0; k < 16; ++k) s0 = __builtin_amdgcn_sad_u16(b[1][k], c[k], s0);
#pragma unroll
                            for (int k = 0; k < 16; ++k) s0 = __builtin_amdgcn_sad_u16(a[1][k], c[16 + k], s0);
                        }
                        const uint32_t k0 = ((s0 & ~3u) << 14) | (u == 0 ? pid[0] : pid[1]);
                        atomicMin(my_slot + (j * kTileBlocksX + i) * 64, k0);
                        if (COSTS) {
                            const size_t blk = (size_t)(ty * TBY + j) * P.blocks_x + (tx * kTileBlocksX + i);
                            P.costs[blk * (size_t)(span * span) + (size_t)(16 * m + 8 * u + lane_row - 8 * j) * span + (size_t)(8 * ux + col_in - 8 * i)] = s0 >> 2;
                        }
                    }
                }
            }
        } else {
            // ================= narrow item: 64 positions down one column or along one row =====================
            int it = item - n_main;
            const bool is_col = it < n_ncol;
            int bi, bj, prow, pcol;                                    // the block column / row served; this lane's position
            const int q = (half ? 0 : 32) + n;                         // after the exchange lanes 0-31 own positions 32..63 of the unit
            if (is_col) {
                const int chunk = it % n_chunk_r, rest = it / n_chunk_r;
                const int c = rest % rem;
                bi = rest / rem; bj = -1;
                pcol = 8 * bi + 8 * F + c;
                prow = 64 * chunk + q;
            } else {
                it -= n_ncol;
                const int chunk = it % n_chunk_c, rest = it / n_chunk_c;
                const int rr = rest % rem;
                bj = rest / rem; bi = -1;
                prow = 8 * bj + 8 * F + rr;
                pcol = 64 * chunk + q;
            }
            // position of the lane BEFORE the exchange (transform 1: positions 0..31, transform 2: 32..63), clamped for the loads
            uint32_t a[16], b[16];
            {
                int r1 = is_col ? prow - q + n : prow, c1 = is_col ? pcol : pcol - q + n;
                int r2 = is_col ? r1 + 32 : r1, c2 = is_col ? c1 : c1 + 32;
                r1 = r1 > n_rows - 1 ? n_rows - 1 : r1; r2 = r2 > n_rows - 1 ? n_rows - 1 : r2;
                c1 = c1 > n_cols - 1 ? n_cols - 1 : c1; c2 = c2 > n_cols - 1 ? n_cols - 1 : c2;
                v4i w0, w1;
                load_window(win, P.pitch, r1 + 4 * half, c1, (c1 & 3) * 8, w0, w1);
                hadamard_pack(H, w0, w1, a);
                load_window(win, P.pitch, r2 + 4 * half, c2, (c2 & 3) * 8, w0, w1);
                hadamard_pack(H, w0, w1, b);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(b[k], a[k], false, false);
                    b[k] = (uint32_t)sw[0];
                    a[k] = (uint32_t)sw[1];
                }
            }
            const uint32_t pid = (uint32_t)(prow << 8 | pcol);
            const int j_lo = is_col ? 0 : bj, j_hi = is_col ? blocks_left_y : (bj < blocks_left_y ? bj + 1 : bj);
            const int i_lo = is_col ? bi : 0, i_hi = is_col ? (bi < blocks_left_x ? bi + 1 : bi) : blocks_left_x;
            const int unit_r0 = __builtin_amdgcn_readfirstlane(is_col ? prow - q : prow);
            const int unit_c0 = __builtin_amdgcn_readfirstlane(is_col ? pcol : pcol - q);
#pragma unroll 1
            for (int j = j_lo; j < j_hi; ++j) {
                if (is_col && (unit_r0 + 63 < 8 * j || unit_r0 >= 8 * j + span)) continue;            // no row of the unit in the window
#pragma unroll 1
                for (int i = i_lo; i < i_hi; ++i) {
                    if (!is_col && (unit_c0 + 63 < 8 * i || unit_c0 >= 8 * i + span)) continue;
                    const uint32_t *__restrict__ c = tile_coef + (j * kTileBlocksX + i) * 32;
                    uint32_t s0 = 2u;
#pragma unroll
                    for (int k = 0; k < 16; ++k) s0 = __builtin_amdgcn_sad_u16(b[k], c[k], s0);
#pragma unroll
                    for (int k = 0; k < 16; ++k) s0 = __builtin_amdgcn_sad_u16(a[k], c[16 + k], s0);
                    const int dyi = prow - 8 * j, dxi = pcol - 8 * i;
                    const bool ok = (unsigned)dyi < (unsigned)span && (unsigned)dxi < (unsigned)span;
                    const uint32_t key = ((s0 & ~3u) << 14) | pid;
                    if (ok) atomicMin(my_slot + (j * kTileBlocksX + i) * 64, key);
                    if (COSTS && ok) {
                        const size_t blk = (size_t)(ty * TBY + j) * P.blocks_x + (tx * kTileBlocksX + i);
                        P.costs[blk * (size_t)(span * span) + (size_t)dyi * span + (size_t)dxi] = s0 >> 2;
                    }
                }
            }
        }
    }

    __syncthreads();
    for (int blk = wave; blk < NBLK; blk += n_waves) {                 // minimum over the 64 lane slots of a block
        uint32_t v = slots[blk * 64 + lane];
#pragma unroll
        for (int mm = 32; mm >= 1; mm >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)v, mm);
            v = o < v ? o : v;
        }
        const int bi = blk % kTileBlocksX, bj = blk / kTileBlocksX;
        const int bx = tx * kTileBlocksX + bi, by = ty * TBY + bj;
        if (lane == 0 && bx < P.blocks_x && by < P.blocks_y) {
            x266_me_result_t res;
            res.mvx = (int16_t)((int)(v & 0xFFu) - 8 * bi - R);            // window column -> dx
            res.mvy = (int16_t)((int)((v >> 8) & 0xFFu) - 8 * bj - R);     // window row -> dy
            res.cost = v >> 16;
            P.best[(size_t)by * P.blocks_x + bx] = res;
        }
    }
}

// ============================================================================
// Full search with the cheaper metric: cost = sum |cur - ref| over the 8x8 block, i.e.
// sad() of riscv/programs/benchmarks/sad/sad.c:28-39 at n = 8, same harness (raster order, first
// minimum wins) as the SATD search above.
//
// Lane = candidate column.  A reference row's 8 pixels at that column (two dwords, aligned once with
// v_alignbit) are shared by the eight candidates (dy) whose blocks contain the row, one per block row
// p = 0..7: eight rotating accumulators per block, the slot that has just received p = 7 is a finished
// candidate.  Four horizontally adjacent blocks per pass share the row too; their 4 x 8 current rows
// sit in 64 SGPRs (scalar loads), so the inner loop is v_sad_u8 against scalar operands: 64 per
// reference row and lane, plus 5 instructions to fetch the row and 4 x 4 to score the finished
// candidates.  Work item = (block row of the tile, left / right half of its blocks, 64 columns).
// ============================================================================
template <int TBY, bool COSTS>
__global__ __launch_bounds__(256) void sad_search_kernel(const MeParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NBLK = kTileBlocksX * TBY;
    const int R = P.range, span = 2 * R + 1;
    const int win_rows = P.n_rows + 7;
    uint32_t *best_lds = reinterpret_cast<uint32_t *>(smem);
    unsigned char *win = smem + 128;

    const int tid = threadIdx.x, lane = tid & 63, n_waves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = blockIdx.x % P.tiles_x, ty = blockIdx.x / P.tiles_x;
    const int x0 = tx * (8 * kTileBlocksX), y0 = ty * (8 * TBY);

    {   // reference window, raw pixels (edge-clamped like the SATD search)
        const int dwords_per_row = P.pitch >> 2;
        const int total = win_rows * dwords_per_row;
        for (int i = tid; i < total; i += blockDim.x) {
            const int ry = i / dwords_per_row, cx = (i - ry * dwords_per_row) * 4;
            int gy = y0 - R + ry;
            gy = gy < -R ? -R : (gy > P.height + R - 1 ? P.height + R - 1 : gy);
            const uint8_t *row = P.ref + (long long)gy * P.ref_stride;
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                int gx = x0 - R + cx + b;
                gx = gx < -R ? -R : (gx > P.width + R - 1 ? P.width + R - 1 : gx);
                v |= (uint32_t)row[gx] << (8 * b);
            }
            reinterpret_cast<uint32_t *>(win)[i] = v;
        }
    }
    if (tid < NBLK) best_lds[tid] = 0xFFFFFFFFu;
    __syncthreads();

    const int n_items = TBY * 2 * P.n_groups;                        // here n_groups counts 64-column groups
    const int sh = (lane & 3) * 8;
    for (int item = wave; item < n_items; item += n_waves) {
        const int g = item % P.n_groups, jh = item / P.n_groups, ih = jh & 1, j = jh >> 1;   // wave-uniform
        const int by = ty * TBY + j;
        if (by >= P.blocks_y) continue;
        // current rows of the four blocks: scalar loads (addresses are wave-uniform)
        uint32_t c[4][8][2];
        bool have[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int bx = tx * kTileBlocksX + 4 * ih + i;
            have[i] = bx < P.blocks_x;
            const int bxc = have[i] ? bx : P.blocks_x - 1;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(P.cur + (long long)(by * 8 + p) * P.cur_stride + bxc * 8);
                c[i][p][0] = q[0];
                c[i][p][1] = q[1];
            }
        }
        // per-lane candidate column of each block, and whether it lies inside the block's window
        int dxl[4];
        bool okx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dxl[i] = 64 * g + lane - 8 * (4 * ih + i);
            okx[i] = have[i] && (unsigned)dxl[i] < (unsigned)span;
        }
        uint32_t best[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        uint32_t acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[i][k] = 0;
        const unsigned char *colbase = win + (8 * j) * P.pitch + ((64 * g + lane) & ~3);
        const int n_ry = span + 7;
        for (int ry8 = 0; ry8 < n_ry; ry8 += 8) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int ry = ry8 + m;
                if (ry >= n_ry) break;                               // wave-uniform
                const uint32_t *q = reinterpret_cast<const uint32_t *>(colbase + ry * P.pitch);
                const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
                const uint32_t a0 = __builtin_amdgcn_alignbit(d1, d0, sh), a1 = __builtin_amdgcn_alignbit(d2, d1, sh);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        const int slot = (m - p) & 7;
                        const uint32_t init = p == 0 ? 0u : acc[i][slot];
                        acc[i][slot] = __builtin_amdgcn_sad_u8(a1, c[i][p][1], __builtin_amdgcn_sad_u8(a0, c[i][p][0], init));
                    }
                const int dy = ry - 7;                               // the candidate row that has now seen all 8 block rows
                if (dy >= 0) {                                       // wave-uniform
                    const int fin = (m + 1) & 7;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t cost = acc[i][fin];
                        const uint32_t idx = (uint32_t)(dy * span + dxl[i]);
                        const uint32_t key = okx[i] ? ((cost << 16) | idx) : 0xFFFFFFFFu;
                        best[i] = key < best[i] ? key : best[i];
                        if (COSTS && okx[i]) {
                            const size_t blk = (size_t)by * P.blocks_x + (tx * kTileBlocksX + 4 * ih + i);
                            P.costs[blk * (size_t)(span * span) + idx] = cost;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t v = best[i];
#pragma unroll
            for (int mm = 32; mm >= 1; mm >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)v, mm);
                v = o < v ? o : v;
            }
            if (lane == 0 && have[i]) atomicMin(&best_lds[j * kTileBlocksX + 4 * ih + i], v);
        }
    }
    __syncthreads();
    if (tid < NBLK) {
        const int bx = tx * kTileBlocksX + (tid % kTileBlocksX), by = ty * TBY + (tid / kTileBlocksX);
        if (bx < P.blocks_x && by < P.blocks_y) {
            const uint32_t key = best_lds[tid];
            const int idx = (int)(key & 0xFFFFu);
            x266_me_result_t res;
            res.mvx = (int16_t)(idx % span - R);
            res.mvy = (int16_t)(idx / span - R);
            res.cost = key >> 16;
            P.best[(size_t)by * P.blocks_x + bx] = res;
        }
    }
}

// ============================================================================
// SAD search, variant 2 (default).  The same rotating-accumulator scheme (lane = candidate column, the 8 pixels of
// a reference row at that column serve the eight candidates whose block contains the row), reorganised so that
// the scored lanes are all valid and the per-candidate bookkeeping is two instructions:
//   * one pass = ONE block column x 64 candidate columns starting AT the column's own window, walked over all
//     window rows for the TBY vertically adjacent blocks of the tile -- they share the fetched row (their current
//     rows sit in 4 x 16 SGPRs) and, unlike horizontally adjacent blocks, the same aligned columns: a window of
//     2R+1 = 64 G + rem columns is G passes of 64 valid lanes (variant 1 scored 192 lanes for 129 columns);
//   * the rem extra columns (the "+1" at R = 64) go to narrow passes with lane = candidate ROW (rem <= 8), or to one
//     masked pass (rem > 8);
//   * keys hold the window POSITION, (cost << 16 | row << 8 | column) -- the finished row is the same position for
//     every block of the column, so it is one v_add per row; per (row, block) a v_lshl_or and a v_min.
// ============================================================================
template <int TBY, bool COSTS>
__global__ __launch_bounds__(256) void sad_search_kernel_v2(const MeParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NBLK = kTileBlocksX * TBY;
    const int R = P.range, span = 2 * R + 1;
    const int n_rows = 8 * (TBY - 1) + span;                           // candidate rows of the tile
    const int win_rows = n_rows + 7;
    uint32_t *best_lds = reinterpret_cast<uint32_t *>(smem);
    unsigned char *win = smem + 128;

    const int tid = threadIdx.x, lane = tid & 63, n_waves = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = blockIdx.x % P.tiles_x, ty = blockIdx.x / P.tiles_x;
    const int x0 = tx * (8 * kTileBlocksX), y0 = ty * (8 * TBY);

    {   // reference window, raw pixels (edge-clamped like the SATD search)
        const int dwords_per_row = P.pitch >> 2;
        const int total = win_rows * dwords_per_row;
        for (int i = tid; i < total; i += blockDim.x) {
            const int ry = i / dwords_per_row, cx = (i - ry * dwords_per_row) * 4;
            int gy = y0 - R + ry;
            gy = gy < -R ? -R : (gy > P.height + R - 1 ? P.height + R - 1 : gy);
            const uint8_t *row = P.ref + (long long)gy * P.ref_stride;
            uint32_t v = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                int gx = x0 - R + cx + b;
                gx = gx < -R ? -R : (gx > P.width + R - 1 ? P.width + R - 1 : gx);
                v |= (uint32_t)row[gx] << (8 * b);
            }
            reinterpret_cast<uint32_t *>(win)[i] = v;
        }
    }
    if (tid < NBLK) best_lds[tid] = 0xFFFFFFFFu;
    __syncthreads();

    int blocks_left_x = P.blocks_x - tx * kTileBlocksX, blocks_left_y = P.blocks_y - ty * TBY;
    blocks_left_x = blocks_left_x > kTileBlocksX ? kTileBlocksX : blocks_left_x;
    blocks_left_y = blocks_left_y > TBY ? TBY : blocks_left_y;
    const int G = span >> 6, rem = span - 64 * G;                       // full 64-column passes, extra columns
    const int n_wide = G + (rem > 8 ? 1 : 0);                           // + one masked pass
    const int n_narrow_cols = rem > 8 ? 0 : rem;
    const int n_chunks = (n_rows + 63) >> 6;
    const int n_main = kTileBlocksX * n_wide;
    const int n_items = n_main + kTileBlocksX * n_narrow_cols * n_chunks;

    for (int item = wave; item < n_items; item += n_waves) {
        const bool wide = item < n_main;
        int i, gq = 0, ncol = 0, chunk = 0;
        if (wide) { i = item / n_wide; gq = item - i * n_wide; }
        else { const int it = item - n_main; chunk = it % n_chunks; const int rest = it / n_chunks; ncol = rest % n_narrow_cols; i = rest / n_narrow_cols; }
        if (i >= blocks_left_x) continue;
        // Current rows of the column's TBY blocks, the same in every lane, held in VGPRs: 4 x 16 scalars next to the
        // loop state do not fit the SGPR file (98 spills when tried; variant 1 overflows it too), and the v_sad_u8
        // chain does not care which file its second operand comes from.  `lane0` is a zero the compiler cannot see
        // through, so these stay vector loads.
        int lane0;
        asm volatile("v_mov_b32 %0, 0" : "=v"(lane0));
        uint32_t c[TBY][8][2];
#pragma unroll
        for (int j = 0; j < TBY; ++j) {
            const int by = ty * TBY + (j < blocks_left_y ? j : blocks_left_y - 1);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(P.cur + (long long)(by * 8 + p) * P.cur_stride + (tx * kTileBlocksX + i) * 8) + lane0;
                c[j][p][0] = q[0];
                c[j][p][1] = q[1];
            }
        }
        uint32_t best[TBY];
#pragma unroll
        for (int j = 0; j < TBY; ++j) best[j] = 0xFFFFFFFFu;

        if (wide) {
            const int col = 8 * i + 64 * gq + lane;                    // window column of this lane's candidates
            const bool lane_ok = 64 * gq + lane < span;                 // false only in the masked pass
            const int sh = (col & 3) * 8;
            const unsigned char *colbase = win + (col & ~3);
            uint32_t acc[TBY][8];
#pragma unroll
            for (int j = 0; j < TBY; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[j][k] = 0;
            uint32_t pid = (uint32_t)col + ((uint32_t)(-7) << 8);       // position of the candidate finished by row ry: (ry - 7, col)
            const uint32_t dead = lane_ok ? 0u : 0xFFFFFFFFu;           // masked pass: lanes beyond the window get the all-ones key, once per row
            // Eight window rows.  STEADY: every block row of the tile is inside its band and finishes a candidate on every
            // one of the eight rows (all but the first and last few groups), so the body carries no wave-uniform tests at all.
            auto rows8 = [&](auto steady_tag, int ry8) {
                constexpr bool STEADY = decltype(steady_tag)::value;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int ry = ry8 + m;
                    if (!STEADY && ry >= win_rows) break;                // wave-uniform
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(colbase + ry * P.pitch);
                    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
                    const uint32_t a0 = __builtin_amdgcn_alignbit(d1, d0, sh), a1 = __builtin_amdgcn_alignbit(d2, d1, sh);
                    const uint32_t pidm = pid | dead;
#pragma unroll
                    for (int j = 0; j < TBY; ++j) {
                        const int rel = ry - 8 * j;                      // row inside block row j's band
                        if (!STEADY && (rel < 0 || rel >= span + 7 || j >= blocks_left_y)) continue;      // wave-uniform
#pragma unroll
                        for (int p = 0; p < 8; ++p) {
                            const int slot = (m - p) & 7;
                            const uint32_t init = p == 0 ? 0u : acc[j][slot];
                            acc[j][slot] = __builtin_amdgcn_sad_u8(a1, c[j][p][1], __builtin_amdgcn_sad_u8(a0, c[j][p][0], init));
                        }
                        const int dyi = rel - 7;                         // the candidate row that has now seen all 8 block rows
                        if (STEADY || (dyi >= 0 && dyi < span)) {        // wave-uniform
                            const uint32_t cost = acc[j][(m + 1) & 7];
                            const uint32_t key = (cost << 16) | pidm;
                            best[j] = key < best[j] ? key : best[j];
                            if (COSTS && lane_ok) {
                                const size_t blk = (size_t)(ty * TBY + j) * P.blocks_x + (tx * kTileBlocksX + i);
                                P.costs[blk * (size_t)(span * span) + (size_t)dyi * span + (size_t)(64 * gq + lane)] = cost;
                            }
                        }
                    }
                    pid += 256u;
                }
            };
            const bool full_tile = blocks_left_y == TBY;
            for (int ry8 = 0; ry8 < win_rows; ry8 += 8) {
                // steady: min rel = ry8 - 8 (TBY - 1) >= 7 and max rel = ry8 + 7 < span + 7
                if (full_tile && ry8 >= 8 * TBY && ry8 < span) rows8(std::true_type{}, ry8);
                else                                           rows8(std::false_type{}, ry8);
            }
        } else {
            // narrow pass: one extra column, lane = candidate row (window row 64 * chunk + lane)
            const int col = 8 * i + 64 * G + ncol;
            const int prow = 64 * chunk + lane;
            const int lrow = prow < n_rows - 1 ? prow : n_rows - 1;       // clamp the loads, not the position
            const int sh = (col & 3) * 8;
            const unsigned char *base = win + lrow * P.pitch + (col & ~3);
            uint32_t a[8][2];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(base + p * P.pitch);
                const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
                a[p][0] = __builtin_amdgcn_alignbit(d1, d0, sh);
                a[p][1] = __builtin_amdgcn_alignbit(d2, d1, sh);
            }
            const uint32_t pid = (uint32_t)(prow << 8 | col);
#pragma unroll
            for (int j = 0; j < TBY; ++j) {
                if (j >= blocks_left_y || 64 * chunk + 63 < 8 * j || 64 * chunk >= 8 * j + span) continue;   // wave-uniform
                uint32_t cost = 0;
#pragma unroll
                for (int p = 0; p < 8; ++p) cost = __builtin_amdgcn_sad_u8(a[p][1], c[j][p][1], __builtin_amdgcn_sad_u8(a[p][0], c[j][p][0], cost));
                const int dyi = prow - 8 * j;
                const bool ok = (unsigned)dyi < (unsigned)span;
                const uint32_t key = ok ? ((cost << 16) | pid) : 0xFFFFFFFFu;
                best[j] = key < best[j] ? key : best[j];
                if (COSTS && ok) {
                    const size_t blk = (size_t)(ty * TBY + j) * P.blocks_x + (tx * kTileBlocksX + i);
                    P.costs[blk * (size_t)(span * span) + (size_t)dyi * span + (size_t)(64 * G + ncol)] = cost;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TBY; ++j) {
            uint32_t v = best[j];
#pragma unroll
            for (int mm = 32; mm >= 1; mm >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)v, mm);
                v = o < v ? o : v;
            }
            if (lane == 0 && j < blocks_left_y) atomicMin(&best_lds[j * kTileBlocksX + i], v);
        }
    }
    __syncthreads();
    if (tid < NBLK) {
        const int bi = tid % kTileBlocksX, bj = tid / kTileBlocksX;
        const int bx = tx * kTileBlocksX + bi, by = ty * TBY + bj;
        if (bx < P.blocks_x && by < P.blocks_y) {
            const uint32_t key = best_lds[tid];
            x266_me_result_t res;
            res.mvx = (int16_t)((int)(key & 0xFFu) - 8 * bi - R);
            res.mvy = (int16_t)((int)((key >> 8) & 0xFFu) - 8 * bj - R);
            res.cost = key >> 16;
            P.best[(size_t)by * P.blocks_x + bx] = res;
        }
    }
}

}  // namespace

hipError_t launch_satd_search(const uint8_t *d_cur, long long cur_stride, const uint8_t *d_ref, long long ref_stride,
                              int width, int height, int range, x266_me_result_t *d_best, uint32_t *d_costs,
                              int tile_rows, int variant, int row_pairs, uint32_t *d_coef_scratch, int cu_count, int wg_threads,
                              int me_splits, hipStream_t stream)
{
    MeParams P;
    P.keys = nullptr; P.splits = 1;
    P.cur = d_cur; P.ref = d_ref; P.cur_stride = cur_stride; P.ref_stride = ref_stride;
    P.width = width; P.height = height; P.range = range;
    P.blocks_x = width / 8; P.blocks_y = height / 8;
    P.tiles_x = (P.blocks_x + kTileBlocksX - 1) / kTileBlocksX;
    if (tile_rows <= 0 && variant == 4) {
        // Tile height by frame size.  Measured (profiles/r02_me_sizes.txt, R = 64, 256 CUs): a launch of T tiles takes about
        // L + (ceil(T / CUs) - 1) * S with (L, S) = (0.40, 0.30) ms for 8-row tiles, (0.23, 0.167) for 4 and (0.146, 0.098) for 2
        // -- a lone tile costs L whatever the chip could do next to it, and tall tiles halve the position transforms per
        // block.  4K frames want 8 rows, a 544-row stripe 4, a 360p frame 2; only the ratios matter, they hold for any range.
        const int cus = cu_count > 0 ? cu_count : 256;
        const float L[3] = {4.0f, 2.3f, 1.46f}, S[3] = {3.0f, 1.67f, 0.98f};
        const int cand[3] = {8, 4, 2};
        float best_t = 0.f;
        for (int c = 0; c < 3; ++c) {
            const long long tiles = (long long)P.tiles_x * ((P.blocks_y + cand[c] - 1) / cand[c]);
            const float t = L[c] + (float)((tiles + cus - 1) / cus - 1) * S[c];
            if (c == 0 || t < best_t) { best_t = t; tile_rows = cand[c]; }
        }
    }
    if (tile_rows <= 0) tile_rows = 4;                               // the earlier variants: their measured best
    const int tby = (tile_rows == 8 && variant == 4) ? 8 : (tile_rows >= 4 ? 4 : (tile_rows == 1 ? 1 : 2));
    const int tiles_y = (P.blocks_y + tby - 1) / tby;
    const int span = 2 * range + 1;
    P.n_groups = (8 * (kTileBlocksX - 1) + span + 31) / 32;
    P.n_rows = 8 * (tby - 1) + span;
    P.pitch = 32 * P.n_groups + 12;
    P.best = d_best; P.costs = d_costs;
    dim3 grid((unsigned)(P.tiles_x * tiles_y)), block(256);
    if (variant == 2 || variant == 3 || variant == 4) {
        const int n_blocks = P.blocks_x * P.blocks_y;
        const int groups = (n_blocks + 31) / 32;
        uint32_t *d_keys = variant == 3 ? d_coef_scratch + (size_t)P.tiles_x * tiles_y * kTileBlocksX * tby * 32 : nullptr;   // behind the table
        hipLaunchKernelGGL(me_coef_kernel, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, stream, d_cur, cur_stride,
                           P.blocks_x, n_blocks, P.tiles_x, tby, d_coef_scratch, d_keys);
        {
            const hipError_t e0 = hipGetLastError();
            if (e0 != hipSuccess) return e0;
        }
        const int U = row_pairs == 1 ? 1 : (row_pairs == 3 ? 3 : 2);
        const uint32_t *cf = d_coef_scratch;
        if (variant == 4) {
            const int F = span >> 3;
            const int n_ucols = kTileBlocksX - 1 + F;
            const int n_item_rows = (tby - 1 + F + 1) >> 1;
            const int main_rows = 16 * n_item_rows;
            P.pitch = 8 * n_ucols + 20;                                  // last position column + 7 pixels + dword alignment
            const size_t lds4 = (size_t)kTileBlocksX * tby * 256 + (size_t)((main_rows > P.n_rows ? main_rows : P.n_rows) + 7) * P.pitch;
            const int wg4 = wg_threads ? wg_threads : 512;                // 8-wave workgroups, 4 waves per SIMD (profiles/r02_me_variants.txt)
            dim3 block4((unsigned)wg4);
#define X266_ME4W(T, WG, WPS) do { if (d_costs) hipLaunchKernelGGL((satd_search_kernel_v4<T, true, WG, WPS>), grid, block4, lds4, stream, P, cf); \
                                   else         hipLaunchKernelGGL((satd_search_kernel_v4<T, false, WG, WPS>), grid, block4, lds4, stream, P, cf); } while (0)
#define X266_ME4(T) do { if (wg4 == 512) X266_ME4W(T, 512, 4); else if (wg4 == 384) X266_ME4W(T, 384, 3); else X266_ME4W(T, 256, 3); } while (0)
            if (wg4 != 256 && wg4 != 384 && wg4 != 512) return hipErrorInvalidValue;
            if (tby == 8) X266_ME4(8); else if (tby == 4) X266_ME4(4); else if (tby == 1) X266_ME4(1); else X266_ME4(2);
#undef X266_ME4W
#undef X266_ME4
            return hipGetLastError();
        }
        if (variant == 3) {
            const int U3 = U == 1 ? 1 : 2;
            P.n_groups = (8 * (kTileBlocksX - 1) + span + 15) / 16;      // 16-column groups
            P.pitch = 16 * P.n_groups + 12;
            const int n_items_y = ((P.n_rows + 3) / 4 + U3 - 1) / U3;
            // Workgroups per tile ("me_splits" bands of candidate rows, merged with atomicMin): finer dispatch granularity.
            const long long tiles = (long long)P.tiles_x * tiles_y;
            (void)cu_count;
            int splits = 1;
            if (me_splits > 0) splits = me_splits;            // measured: no gain on MI355X (profiles/r02_me_variants.txt); kept for A/B
            if (splits > n_items_y) splits = n_items_y;
            P.splits = splits;
            P.keys = d_keys;
            const int band_items = (n_items_y + splits - 1) / splits;
            const size_t lds3 = 128 + (size_t)(4 * U3 * band_items + 7) * P.pitch;
            grid = dim3((unsigned)(tiles * splits));
            dim3 block3((unsigned)(wg_threads ? wg_threads : 256));
#define X266_ME3(T, UU) do { if (d_costs) hipLaunchKernelGGL((satd_search_kernel_v3<T, UU, true>), grid, block3, lds3, stream, P, cf); \
                             else         hipLaunchKernelGGL((satd_search_kernel_v3<T, UU, false>), grid, block3, lds3, stream, P, cf); } while (0)
            if (tby == 4)      { if (U3 == 1) X266_ME3(4, 1); else X266_ME3(4, 2); }
            else if (tby == 1) { if (U3 == 1) X266_ME3(1, 1); else X266_ME3(1, 2); }
            else               { if (U3 == 1) X266_ME3(2, 1); else X266_ME3(2, 2); }
#undef X266_ME3
            {
                const hipError_t e1 = hipGetLastError();
                if (e1 != hipSuccess) return e1;
            }
            hipLaunchKernelGGL(me_decode_kernel, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, stream, d_keys, d_best, P.blocks_x, n_blocks, tby, range);
            return hipGetLastError();
        }
        const size_t lds = 128 + (size_t)(P.n_rows + 7 + 2 * U) * P.pitch;   // the last strip may be partly empty
#define X266_ME(T, UU) do { if (d_costs) hipLaunchKernelGGL((satd_search_kernel_v2<T, UU, true>), grid, block, lds, stream, P, cf); \
                            else         hipLaunchKernelGGL((satd_search_kernel_v2<T, UU, false>), grid, block, lds, stream, P, cf); } while (0)
        if (tby == 4)      { if (U == 1) X266_ME(4, 1); else if (U == 2) X266_ME(4, 2); else X266_ME(4, 3); }
        else if (tby == 1) { if (U == 1) X266_ME(1, 1); else if (U == 2) X266_ME(1, 2); else X266_ME(1, 3); }
        else               { if (U == 1) X266_ME(2, 1); else if (U == 2) X266_ME(2, 2); else X266_ME(2, 3); }
#undef X266_ME
        return hipGetLastError();
    }
    const size_t lds = (size_t)kTileBlocksX * tby * 128 + 128 + (size_t)(P.n_rows + 7) * P.pitch;
    if (tby == 4)      hipLaunchKernelGGL((satd_search_kernel<4>), grid, block, lds, stream, P);
    else if (tby == 1) hipLaunchKernelGGL((satd_search_kernel<1>), grid, block, lds, stream, P);
    else               hipLaunchKernelGGL((satd_search_kernel<2>), grid, block, lds, stream, P);
    return hipGetLastError();
}

}  // namespace x266

namespace x266 {

hipError_t launch_sad_search(const uint8_t *d_cur, long long cur_stride, const uint8_t *d_ref, long long ref_stride,
                             int width, int height, int range, x266_me_result_t *d_best, uint32_t *d_costs,
                             int tile_rows, int variant, hipStream_t stream)
{
    MeParams P;
    P.keys = nullptr; P.splits = 1;
    P.cur = d_cur; P.ref = d_ref; P.cur_stride = cur_stride; P.ref_stride = ref_stride;
    P.width = width; P.height = height; P.range = range;
    P.blocks_x = width / 8; P.blocks_y = height / 8;
    P.tiles_x = (P.blocks_x + kTileBlocksX - 1) / kTileBlocksX;
    const int tby = tile_rows >= 4 ? 4 : (tile_rows == 1 ? 1 : 2);   // 0 (automatic) = 2: as fast as 4 on a 4K frame, finer-grained on small ones
    const int tiles_y = (P.blocks_y + tby - 1) / tby;
    const int span = 2 * range + 1;
    P.n_groups = (8 * (kTileBlocksX - 1) + span + 63) / 64;           // 64-column groups (lane = column)
    P.n_rows = 8 * (tby - 1) + span;
    P.pitch = 64 * P.n_groups + 12;
    P.best = d_best; P.costs = d_costs;
    dim3 grid((unsigned)(P.tiles_x * tiles_y)), block(256);
    const size_t lds = 128 + (size_t)(P.n_rows + 7) * P.pitch;
    if (variant == 2) {
#define X266_SADS2(T) do { if (d_costs) hipLaunchKernelGGL((sad_search_kernel_v2<T, true>), grid, block, lds, stream, P); \
                           else         hipLaunchKernelGGL((sad_search_kernel_v2<T, false>), grid, block, lds, stream, P); } while (0)
        if (tby == 4) X266_SADS2(4); else if (tby == 1) X266_SADS2(1); else X266_SADS2(2);
#undef X266_SADS2
        return hipGetLastError();
    }
#define X266_SADS(T) do { if (d_costs) hipLaunchKernelGGL((sad_search_kernel<T, true>), grid, block, lds, stream, P); \
                          else         hipLaunchKernelGGL((sad_search_kernel<T, false>), grid, block, lds, stream, P); } while (0)
    if (tby == 4) X266_SADS(4); else if (tby == 1) X266_SADS(1); else X266_SADS(2);
#undef X266_SADS
    return hipGetLastError();
}

}  // namespace x266
