// me_search.hip -- full-search 8x8 SATD motion estimation for gfx950 (BASELINE configs[2]:
// 3840x2160 luma, 8x8 blocks, window +-64 => 2.157e9 SATDs / frame), round-3 kernel.
//
// Per-candidate cost is pinned by the reference: satd8x8(cur - ref) with satd8x8 =
// src_tb/satd.c:31-118 (9-bit differences, as the testbench feeds them, src/mkSatd.bsv:229).
// The search harness around it -- candidate order, tie-break, padding -- has no upstream
// counterpart and is defined in include/x266hip.h (raster order dy-major, first minimum wins).
//
// Algorithm.  The Hadamard transform is linear and a 9-bit difference cannot
// wrap int16, so satd(cur - ref) = (sum_m |Hc[m] - Hr[m]| + 2) >> 2 with Hc = H64*cur, Hr = H64*ref,
// exactly.  The search runs in the transform domain: Hr is formed once per candidate POSITION and
// scored against every block whose window contains the position with v_sad_u16 (two |a-b| per
// instruction) against the block's coefficients held in SGPRs.  Bound: VALU issue of v_sad_u16.
//
// What round 3 changed against the round-2 kernel (profiles/r03_me_variants.txt):
//   * ONE LANE = ONE POSITION straight out of the matrix core.  The A operand of
//     v_mfma_i32_32x32x32_i8 is the 16x16 Hadamard matrix over (row parity, column) of the window,
//     zero outside the lane half that will receive the output row, so the K-slots a lane feeds
//     (16 of its own window's pixels per K-step) only ever meet the output rows the same lane
//     receives: four K-steps x four sign patterns (+A / -A per pixel-row pair) give the lane all 64
//     coefficients of its own window.  No half-wave exchange (16 v_permlane32_swap per 64 positions),
//     no v_perm packing (32) and no bias xor (32): two coefficient sets share one accumulator --
//     high set, << 16 (+ bias), low set on top -- and the int16 bias 0x8000 of the high field comes
//     from a constant MFMA.  16 v_alignbit + 32 v_lshl_add per 64 positions are all the VALU the
//     transform still costs.
//   * the coefficient rows of the blocks stream through TWO HALF-ROW SGPR sets: the half needed next
//     is requested right after the first v_sad_u16 of the current half, so every scalar load has 31
//     v_sad_u16 of its own wave to land in and the only wait sees exactly one load in flight
//     (s_waitcnt lgkmcnt(0) is all SMEM offers); the two ds_min_u32 of a block are issued one phase
//     late for the same reason.  32 SGPRs of operands instead of 64: no spills.
//   * the block loop is one run-time loop over the item's valid blocks (no per-column unrolling).
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cstdint>

#include "x266_device.hpp"
#include "x266_mfma_blocks.hpp"

namespace x266 {
namespace {

constexpr int kTileBlocksX = 8;           // blocks per tile row (64 pixels)
// LDS bytes per window row, the same for every range (R <= 64: 23 unit columns x 8 + 7 pixels + dword slack): a compile-time
// pitch turns the 24 row addresses of a position's window into immediate offsets.  51 dwords: odd, so the four rows a
// half-wave reads land in different banks.
constexpr int kPitch = 204;

typedef uint32_t u16v __attribute__((ext_vector_type(16)));

struct MeParams {
    const uint8_t *cur;
    const uint8_t *ref;           // pixel (0,0); valid for x,y in [-range, dim + range)
    long long cur_stride, ref_stride;
    int width, height, range;
    int blocks_x, blocks_y;       // width / 8, height / 8
    int tiles_x;
    x266_me_result_t *best;
    uint32_t *costs;              // optional [block][(2R+1)^2]
};

// ---- lane = position Hadamard transform on the matrix core ------------------------------------------
// Output row m of the MFMA lands in lane half (m >> 2) & 1, register r = (m & 3) | (m >> 3) << 2.
// A[m][k], k = 16 h + t: (-1)^popcount(r & t) for h == that half, else 0; t = 8 (y & 1) + x indexes the
// 16 pixels of one K-step (window rows 2s, 2s+1).  `neg` = -A.  `kc`: 32 slots x 32 x 32 = 0x8000.
struct LaneOps { v4i pos, neg, kc; };

__device__ __forceinline__ LaneOps make_lane_ops(int lane)
{
    const uint32_t NEG = 0xFEFEFEFEu;
    const uint32_t m = (uint32_t)lane & 31u, h = (uint32_t)lane >> 5;
    const uint32_t r = (m & 3u) | ((m >> 3) << 2);
    const bool active = ((m >> 2) & 1u) == h;
    const uint32_t inner = (r & 1) ? ((r & 2) ? 0x01FFFF01u : 0xFF01FF01u) : ((r & 2) ? 0xFFFF0101u : 0x01010101u);
    const uint32_t f0 = (r & 4) ? NEG : 0u, f1 = (r & 8) ? NEG : 0u;
    const uint32_t on = active ? 0xFFFFFFFFu : 0u;
    LaneOps o;
    o.pos = v4i{(int)(inner & on), (int)((inner ^ f0) & on), (int)((inner ^ f1) & on), (int)((inner ^ f0 ^ f1) & on)};
    o.neg = v4i{(int)((inner ^ NEG) & on), (int)((inner ^ f0 ^ NEG) & on), (int)((inner ^ f1 ^ NEG) & on), (int)((inner ^ f0 ^ f1 ^ NEG) & on)};
    o.kc = v4i{0x20202020, 0x20202020, 0x20202020, 0x20202020};
    return o;
}

// px[s] = window rows 2s, 2s+1 of this lane's 8x8 window (signed pixels, 16 bytes).  out[k], out[16+k]:
// biased uint16 pairs {coefficient 16 + k | coefficient k}, {48 + k | 32 + k} (natural H64 order
// c = 16 q + r over pixel index 8 y + x; the order is irrelevant under sum |.| as long as the block
// table uses the same one, and it does: me_coef_kernel calls this function too).
__device__ __forceinline__ void hadamard_lane(const LaneOps &O, const v4i (&px)[4], uint32_t (&out)[32])
{
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // high fields: q = 1 (signs + - + -) and q = 3 (+ - - +) share the bias and their first two K-steps
    v16i t = mfma(O.kc, O.kc, zero);
    t = mfma(O.pos, px[0], t); t = mfma(O.neg, px[1], t);
    v16i a = mfma(O.pos, px[2], t); a = mfma(O.neg, px[3], a);
    v16i b = mfma(O.neg, px[2], t); b = mfma(O.pos, px[3], b);
    // << 16 (a full-rate v_lshlrev_b32; the low field's bias is one more constant MFMA), then q = 0 (+ + + +) and q = 2 (+ + - -)
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = (int)((uint32_t)a[k] << 16);
    a = mfma(O.kc, O.kc, a);
    a = mfma(O.pos, px[0], a); a = mfma(O.pos, px[1], a); a = mfma(O.pos, px[2], a); a = mfma(O.pos, px[3], a);
#pragma unroll
    for (int k = 0; k < 16; ++k) b[k] = (int)((uint32_t)b[k] << 16);
    b = mfma(O.kc, O.kc, b);
    b = mfma(O.pos, px[0], b); b = mfma(O.pos, px[1], b); b = mfma(O.neg, px[2], b); b = mfma(O.neg, px[3], b);
#pragma unroll
    for (int k = 0; k < 16; ++k) { out[k] = (uint32_t)a[k]; out[16 + k] = (uint32_t)b[k]; }
}

// 8x8 bytes at (row, col) of the LDS window -> the four K-step operands: three aligned dwords per row and a funnel shift.
// (gfx950 does serve unaligned ds_read_b64 -- tools/probes/lds_unaligned_test.hip -- which would save the 16 v_alignbit_b32, but at a
// fraction of the aligned rate: 2.34 -> 2.49 ms per 4K frame, profiles/r03_me_variants.txt.)  The row offsets are immediates (kPitch).
__device__ __forceinline__ void load_window8(const unsigned char *lds0, int win_offset, int row, int col, v4i (&px)[4])
{
    const int sh = (col & 3) * 8;
    int off = win_offset + row * kPitch + (col & ~3);
    asm volatile("" : "+v"(off));         // keep the window's LDS offset in the register: y * kPitch then fits the DS offset fields
    const unsigned char *base = lds0 + off;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(base + y * kPitch);
        const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
        px[y >> 1][2 * (y & 1)]     = (int)__builtin_amdgcn_alignbit(d1, d0, sh);
        px[y >> 1][2 * (y & 1) + 1] = (int)__builtin_amdgcn_alignbit(d2, d1, sh);
    }
}

// Pre-pass: Hc of every 8x8 block of the current frame, 32 dwords per block in hadamard_lane's order.
// Blocks are stored TILE-MAJOR (search tile, then block row, then block column inside the tile): the
// search kernel addresses all blocks of its tile from one scalar base.
__global__ __launch_bounds__(256) void me_coef_kernel(const uint8_t *__restrict__ cur, long long cur_stride,
                                                      int blocks_x, int n_blocks, int tiles_x, int tby,
                                                      uint32_t *__restrict__ coef)
{
    const int lane = threadIdx.x & 63;
    const int wave = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (wave * 64 >= n_blocks) return;
    int blk = wave * 64 + lane;
    const bool live = blk < n_blocks;
    if (!live) blk = n_blocks - 1;
    const int bx = blk % blocks_x, by = blk / blocks_x;
    const uint8_t *src = cur + (long long)(by * 8) * cur_stride + bx * 8;
    v4i px[4];
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const uint8_t *q = src + (long long)y * cur_stride;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) { lo |= (uint32_t)q[b] << (8 * b); hi |= (uint32_t)q[4 + b] << (8 * b); }
        px[y >> 1][2 * (y & 1)]     = (int)(lo ^ 0x80808080u);
        px[y >> 1][2 * (y & 1) + 1] = (int)(hi ^ 0x80808080u);
    }
    const LaneOps O = make_lane_ops(lane);
    uint32_t p[32];
    hadamard_lane(O, px, p);
    if (live) {
        const size_t slot = ((size_t)(by / tby) * tiles_x + bx / kTileBlocksX) * (size_t)(kTileBlocksX * tby)
                            + (size_t)(by % tby) * kTileBlocksX + bx % kTileBlocksX;
        v4i *dst = reinterpret_cast<v4i *>(coef + slot * 32);
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = v4i{(int)p[4 * k], (int)p[4 * k + 1], (int)p[4 * k + 2], (int)p[4 * k + 3]};
    }
}

// x / d for 0 <= x < 64, d in {1, 2, 3} (the chunk counts of the narrow items) without a divider sequence
__device__ __forceinline__ int div_small(int x, int d) { return d == 1 ? x : (d == 2 ? x >> 1 : (x * 21846) >> 16); }

__device__ __forceinline__ uint32_t sad16(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_sad_u16(a, b, c); }
#define X266_FENCE() __builtin_amdgcn_sched_barrier(0)

// Scores NU (1 or 2) units against the blocks (j, i), j in [j_lo, j_hi], i in [i_lo, i_hi] of the tile
// (both ranges non-empty, wave-uniform).  pa / pb: the units' 32 coefficient dwords per lane; pid_a / pid_b
// their window positions (row << 8 | column); row_a / row_b the units' first candidate row (cost map only).
// MASKED (narrow units, NU = 1): a lane's position counts for block (j, i) only if it lies inside that block's window.
template <int NU, bool COSTS, bool MASKED = false>
__device__ __forceinline__ void score_blocks(const uint32_t (&pa)[32], const uint32_t (&pb)[32], uint32_t pid_a, uint32_t pid_b,
                                             int j_lo, int j_hi, int i_lo, int i_hi,
                                             const uint32_t *__restrict__ tile_coef, unsigned long long *my_slot,
                                             const MeParams &P, int span, int tile_bx, int tile_by, int prow_a, int prow_b, int pcol)
{
    int i = i_lo, j = j_lo;
    const uint32_t *cp = tile_coef + (j * kTileBlocksX + i) * 32;
    u16v LO = *reinterpret_cast<const u16v *>(cp);
    // keys are 64 bits, {cost : window position}: one v_lshrrev_b32 per candidate, the position half is the unit's own register
    // (a register pair per unit whose low half, the position, is written once: only the high half changes per block)
    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
    u2v key_a = {pid_a, 0xFFFFFFFFu}, key_b = {pid_b, 0xFFFFFFFFu};    // the previous block's keys, not yet in LDS
    unsigned long long *slot = my_slot + (j * kTileBlocksX + i) * 64;
    for (;;) {
        uint32_t s0 = sad16(pa[0], LO[0], 2u), s1 = 2u;               // the "+2" of (sum + 2) >> 2
        if (NU == 2) s1 = sad16(pb[0], LO[0], 2u);
        X266_FENCE();
        const u16v HI = *reinterpret_cast<const u16v *>(cp + 16);     // second half of this block's row: 31 v_sad_u16 to land in
        atomicMin(slot, __builtin_bit_cast(unsigned long long, key_a));   // ds_min_u64, no return value
        if (NU == 2) atomicMin(slot, __builtin_bit_cast(unsigned long long, key_b));
        X266_FENCE();
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            s0 = sad16(pa[k], LO[k], s0);
            if (NU == 2) s1 = sad16(pb[k], LO[k], s1);
        }
        slot = my_slot + (j * kTileBlocksX + i) * 64;
        const int bj = j, bi = i;
        const bool last = (i == i_hi) && (j == j_hi);
        if (i == i_hi) { i = i_lo; ++j; } else ++i;
        const uint32_t *cn = last ? cp : tile_coef + (j * kTileBlocksX + i) * 32;
        X266_FENCE();
        s0 = sad16(pa[16], HI[0], s0);
        if (NU == 2) s1 = sad16(pb[16], HI[0], s1);
        X266_FENCE();
        LO = *reinterpret_cast<const u16v *>(cn);                      // first half of the next block's row
        X266_FENCE();
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            s0 = sad16(pa[16 + k], HI[k], s0);
            if (NU == 2) s1 = sad16(pb[16 + k], HI[k], s1);
        }
        key_a.y = s0 >> 2;
        if (NU == 2) key_b.y = s1 >> 2;
        bool ok = true;
        if (MASKED) {
            ok = (unsigned)(prow_a - 8 * bj) < (unsigned)span && (unsigned)(pcol - 8 * bi) < (unsigned)span;
            key_a.y = ok ? key_a.y : 0xFFFFFFFFu;
        }
        if (COSTS && ok) {
            const size_t blk = (size_t)(tile_by + bj) * P.blocks_x + (tile_bx + bi);
            uint32_t *cm = P.costs + blk * (size_t)(span * span) + (size_t)(pcol - 8 * bi);
            cm[(size_t)(prow_a - 8 * bj) * span] = s0 >> 2;
            if (NU == 2) cm[(size_t)(prow_b - 8 * bj) * span] = s1 >> 2;
        }
        if (last) break;
        cp = cn;
    }
    atomicMin(slot, __builtin_bit_cast(unsigned long long, key_a));
    if (NU == 2) atomicMin(slot, __builtin_bit_cast(unsigned long long, key_b));
}

// A block's window is (2R+1)^2 candidates starting at a multiple of 8 in both directions.  Write
// 2R+1 = 8F + rem (rem odd, 1 for R = 64).  Units are 8 columns x 8 rows of positions on the 8-pixel grid
// of the blocks, one position per lane, so for ANY block the aligned part of its window,
// [8i, 8i+8F) x [8j, 8j+8F), is exactly F x F whole units: a (unit, block) pair is either entirely valid
// or not needed, and the scoring loop has no per-lane validity, no edge path.  Items pair two vertically
// adjacent units (one coefficient fetch per 64 v_sad_u16); the block row whose window starts on the odd unit
// of a pair, and the one whose window ends on the even unit, are scored with that unit alone.
// The running minima live in LDS, one slot per (block, lane), updated with ds_min_u32 (issued beside the
// VALU stream): no registers, so the block loop is a real loop.
// The remaining rem columns and rem rows of every window ("+1" at R = 64: 257 of 16641 candidates) are
// scored by narrow units -- 64 positions down one column, or along one row -- with per-lane validity.
template <int TBY, bool COSTS, int WG, int WPS>
__global__ __launch_bounds__(WG, WPS) void satd_search_kernel(const MeParams P, const uint32_t *__restrict__ coef)
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
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem);   // running minima: [block][lane], shared by the workgroup's waves
    unsigned char *win = smem + NBLK * 512;

    const int tid = threadIdx.x, lane = tid & 63, n_waves = WG >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = blockIdx.x % P.tiles_x, ty = blockIdx.x / P.tiles_x;
    const int x0 = tx * (8 * kTileBlocksX), y0 = ty * (8 * TBY);

#ifdef X266_ME_TIMING
    unsigned long long *tstamp = (!COSTS && P.costs) ? reinterpret_cast<unsigned long long *>(P.costs) + (size_t)blockIdx.x * 12 : nullptr;
    if (tstamp && tid == 0) tstamp[0] = __builtin_readcyclecounter();
#endif
    {   // reference window, signed pixels: a wave takes whole rows (one dword per lane), eight rows in flight
        typedef uint32_t u32_any_align __attribute__((aligned(1)));
        const int dwords_per_row = (8 * n_ucols + 20) >> 2;                // what this range reads of a row (51 at R = 64)
        const int gx0 = x0 - R + 4 * lane;                               // >= -R: only the right and bottom edges clamp
        const bool inside = gx0 + 3 <= P.width + R - 1;
        const int last_x = P.width + R - 1, last_y = P.height + R - 1;
        for (int ry0 = wave; ry0 < win_rows; ry0 += 8 * n_waves) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ry = ry0 + u * n_waves;
                int gy = y0 - R + (ry < win_rows ? ry : win_rows - 1);
                gy = gy > last_y ? last_y : gy;
                const uint8_t *row = P.ref + (long long)gy * P.ref_stride;
                if (inside) v[u] = *reinterpret_cast<const u32_any_align *>(row + gx0);
                else {
                    v[u] = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) { const int gx = gx0 + b > last_x ? last_x : gx0 + b; v[u] |= (uint32_t)row[gx] << (8 * b); }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ry = ry0 + u * n_waves;
                if (ry < win_rows && lane < dwords_per_row) reinterpret_cast<uint32_t *>(win + ry * kPitch)[lane] = v[u] ^ 0x80808080u;
            }
        }
    }
    for (int i = tid; i < NBLK * 64; i += WG) slots[i] = ~0ull;
    uint32_t *next_item = reinterpret_cast<uint32_t *>(win + win_rows * kPitch);   // items are handed out dynamically: the waves of a
    if (tid == 0) *next_item = (uint32_t)n_waves;                                    // SIMD do not advance at the same rate (issue priority by age)
    __syncthreads();
#ifdef X266_ME_TIMING
    if (tstamp && tid == 0) tstamp[1] = __builtin_readcyclecounter();
#endif
    unsigned long long *my_slot = slots + lane;

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
    const int py = lane >> 3, pxx = lane & 7;                           // this lane's position inside a main unit

    int fetched = 0;
    for (int item = wave; item < n_items; item = __builtin_amdgcn_readfirstlane(fetched)) {
        fetched = lane == 0 ? (int)atomicAdd(next_item, 1u) : 0;           // ds_add_rtn_u32, issued now, consumed after the item's work
        // The matrix-core operands are rebuilt per item (a dozen 2-cycle instructions) from a lane id the compiler cannot see
        // through: kept across the scoring loops they are what gets spilled.
        int opaque_lane = lane;
        asm volatile("" : "+v"(opaque_lane));
        const LaneOps O = make_lane_ops(opaque_lane);
        if (item < n_main) {
            // ================= main item: units (2m, ux) and (2m+1, ux) ====================================
            const int m = item / n_ucols, ux = item - m * n_ucols;     // wave-uniform
            int i_lo = ux - F + 1, i_hi = ux;
            i_lo = i_lo < 0 ? 0 : i_lo;
            i_hi = i_hi > blocks_left_x - 1 ? blocks_left_x - 1 : i_hi;
            if (i_lo > i_hi) continue;
            // block rows: pairs for j in [2m+2-F, 2m]; unit 2m alone for j = 2m+1-F; unit 2m+1 alone for j = 2m+1
            int jp_lo = 2 * m + 2 - F, jp_hi = 2 * m;
            jp_lo = jp_lo < 0 ? 0 : jp_lo;
            jp_hi = jp_hi > blocks_left_y - 1 ? blocks_left_y - 1 : jp_hi;
            const int j_a = 2 * m + 1 - F, j_b = 2 * m + 1;
            const bool only_a = j_a >= 0 && j_a < blocks_left_y, only_b = j_b < blocks_left_y;
            const bool pairs = jp_lo <= jp_hi;
            if (!pairs && !only_a && !only_b) continue;
            const int col = 8 * ux + pxx;
            const int row_a = 16 * m + py, row_b = row_a + 8;
            const uint32_t pid_a = (uint32_t)(row_a << 8 | col), pid_b = (uint32_t)(row_b << 8 | col);
            v4i px[4];
            uint32_t pa[32];
            load_window8(smem, NBLK * 512, row_a, col, px);
            hadamard_lane(O, px, pa);
            if (!pairs && !only_b) {                                   // the tile's last unit row: nothing pairs with it
                score_blocks<1, COSTS>(pa, pa, pid_a, pid_a, j_a, j_a, i_lo, i_hi, tile_coef, my_slot, P, span, tx * kTileBlocksX, ty * TBY, row_a, row_a, col);
                continue;
            }
            uint32_t pb[32];
            load_window8(smem, NBLK * 512, row_b, col, px);
            hadamard_lane(O, px, pb);
            if (pairs)  score_blocks<2, COSTS>(pa, pb, pid_a, pid_b, jp_lo, jp_hi, i_lo, i_hi, tile_coef, my_slot, P, span, tx * kTileBlocksX, ty * TBY, row_a, row_b, col);
            if (only_a) score_blocks<1, COSTS>(pa, pa, pid_a, pid_a, j_a, j_a, i_lo, i_hi, tile_coef, my_slot, P, span, tx * kTileBlocksX, ty * TBY, row_a, row_a, col);
            if (only_b) score_blocks<1, COSTS>(pb, pb, pid_b, pid_b, j_b, j_b, i_lo, i_hi, tile_coef, my_slot, P, span, tx * kTileBlocksX, ty * TBY, row_b, row_b, col);
        } else {
            // ================= narrow item: 64 positions down one column or along one row =====================
            // item order inside each kind: extra column / row, then 64-position chunk, then block column / row (fastest),
            // so that the only divisor is the chunk count, 1..3
            int it = item - n_main;
            const bool is_col = it < n_ncol;
            int bi, bj, prow, pcol;                                    // the block column / row served; this lane's position
            if (is_col) {
                bi = it & (kTileBlocksX - 1); bj = -1;
                const int rest = it >> 3;
                const int c = div_small(rest, n_chunk_r), chunk = rest - c * n_chunk_r;
                pcol = 8 * bi + 8 * F + c;
                prow = 64 * chunk + lane;
            } else {
                it -= n_ncol;
                bj = it & (TBY - 1); bi = -1;
                const int rest = it / TBY;
                const int rr = div_small(rest, n_chunk_c), chunk = rest - rr * n_chunk_c;
                prow = 8 * bj + 8 * F + rr;
                pcol = 64 * chunk + lane;
            }
            uint32_t pa[32];
            {
                const int r1 = prow > n_rows - 1 ? n_rows - 1 : prow, c1 = pcol > n_cols - 1 ? n_cols - 1 : pcol;   // clamped for the loads
                v4i px[4];
                load_window8(smem, NBLK * 512, r1, c1, px);
                hadamard_lane(O, px, pa);
            }
            const uint32_t pid = (uint32_t)(prow << 8 | pcol);
            // the blocks whose window the unit's 64 positions can touch (wave-uniform); the per-lane test is in score_blocks
            int j_lo, j_hi, i_lo, i_hi;
            if (is_col) {
                const int unit_r0 = prow - lane;                           // rows unit_r0 .. +63 against windows [8j, 8j + span)
                j_lo = (unit_r0 - span + 8) >> 3; j_hi = (unit_r0 + 63) >> 3;
                i_lo = i_hi = bi;
            } else {
                const int unit_c0 = pcol - lane;
                i_lo = (unit_c0 - span + 8) >> 3; i_hi = (unit_c0 + 63) >> 3;
                j_lo = j_hi = bj;
            }
            j_lo = __builtin_amdgcn_readfirstlane(j_lo < 0 ? 0 : j_lo); i_lo = __builtin_amdgcn_readfirstlane(i_lo < 0 ? 0 : i_lo);
            j_hi = __builtin_amdgcn_readfirstlane(j_hi > blocks_left_y - 1 ? blocks_left_y - 1 : j_hi);
            i_hi = __builtin_amdgcn_readfirstlane(i_hi > blocks_left_x - 1 ? blocks_left_x - 1 : i_hi);
            if (j_lo > j_hi || i_lo > i_hi) continue;
            score_blocks<1, COSTS, true>(pa, pa, pid, pid, j_lo, j_hi, i_lo, i_hi, tile_coef, my_slot, P, span, tx * kTileBlocksX, ty * TBY, prow, prow, pcol);
        }
    }

#ifdef X266_ME_TIMING
    if (tstamp && lane == 0) tstamp[2 + wave] = __builtin_readcyclecounter();
#endif
    __syncthreads();
    for (int blk = wave; blk < NBLK; blk += n_waves) {                 // minimum over the 64 lane slots of a block
        unsigned long long v = slots[blk * 64 + lane];
#pragma unroll
        for (int mm = 32; mm >= 1; mm >>= 1) {
            const unsigned long long o = (unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)v, mm)
                                         | ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mm) << 32);
            v = o < v ? o : v;
        }
        const int bi = blk % kTileBlocksX, bj = blk / kTileBlocksX;
        const int bx = tx * kTileBlocksX + bi, by = ty * TBY + bj;
        if (lane == 0 && bx < P.blocks_x && by < P.blocks_y) {
            x266_me_result_t res;
            res.mvx = (int16_t)((int)(v & 0xFFu) - 8 * bi - R);            // window column -> dx
            res.mvy = (int16_t)((int)((v >> 8) & 0xFFu) - 8 * bj - R);     // window row -> dy
            res.cost = (uint32_t)(v >> 32);
            P.best[(size_t)by * P.blocks_x + bx] = res;
        }
    }
#ifdef X266_ME_TIMING
    if (tstamp && tid == 0) { tstamp[10] = __builtin_readcyclecounter(); tstamp[11] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) << 32) /* XCC_ID */ | __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) /* HW_ID */; }
#endif
}


// ============================================================================
// The same search with the cheaper metric: cost = sum |cur - ref| over the 8x8 block, i.e.
// sad() of riscv/programs/benchmarks/sad/sad.c:28-39 at n = 8.  No transform, so the structure is its own:
// lane = candidate column; the 8 pixels of a reference row at that column (two dwords, aligned once with
// v_alignbit) belong to the eight candidates whose block contains the row, one per block row p = 0..7 -- eight
// rotating accumulators per block, the slot that has just received p = 7 is a finished candidate.
//   * one pass = ONE block column x 64 candidate columns starting AT the column's own window, walked over all
//     window rows for the TBY vertically adjacent blocks of the tile (they share the fetched row and the aligned
//     columns): a window of 2R+1 = 64 G + rem columns is G passes of 64 valid lanes;
//   * the rem extra columns (the "+1" at R = 64) go to narrow passes with lane = candidate ROW (rem <= 8), or to
//     one masked pass (rem > 8);
//   * round 3: the accumulators ARE the keys.  v_sad_hi_u8 adds its sum at bit 16, so a chain started from the
//     candidate's window position (row << 8 | column) ends as (cost << 16 | position): per (row, block) the
//     bookkeeping is one v_min_u32 (round 2: v_lshl_or + v_min after 16 v_sad_u8).  Passes are handed to the
//     waves dynamically (they are few and uneven).
// ============================================================================
__device__ __forceinline__ uint32_t sadhi8(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_sad_hi_u8(a, b, c); }

template <int TBY, bool COSTS>
__global__ __launch_bounds__(256) void sad_search_kernel(const MeParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NBLK = kTileBlocksX * TBY;
    constexpr uint32_t DEAD = 0xC0000000u;                             // above every (cost << 16 | position), no carry out when a cost is added
    const int R = P.range, span = 2 * R + 1;
    const int n_rows = 8 * (TBY - 1) + span;                           // candidate rows of the tile
    const int win_rows = n_rows + 7;
    uint32_t *best_lds = reinterpret_cast<uint32_t *>(smem);           // [NBLK <= 32] minima, then the pass counter: 256-byte header
    uint32_t *next_item = best_lds + NBLK;
    unsigned char *win = smem + 256;

    const int tid = threadIdx.x, lane = tid & 63, n_waves = 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx = blockIdx.x % P.tiles_x, ty = blockIdx.x / P.tiles_x;
    const int x0 = tx * (8 * kTileBlocksX), y0 = ty * (8 * TBY);

    {   // reference window, raw pixels (edge-clamped like the SATD search): whole rows per wave, eight in flight
        typedef uint32_t u32_any_align __attribute__((aligned(1)));
        const int dwords_per_row = (8 * (kTileBlocksX - 1) + span + 7 + 3 + 3) >> 2;
        const int gx0 = x0 - R + 4 * lane;
        const bool inside = gx0 + 3 <= P.width + R - 1;
        const int last_x = P.width + R - 1, last_y = P.height + R - 1;
        for (int ry0 = wave; ry0 < win_rows; ry0 += 8 * n_waves) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ry = ry0 + u * n_waves;
                int gy = y0 - R + (ry < win_rows ? ry : win_rows - 1);
                gy = gy > last_y ? last_y : gy;
                const uint8_t *row = P.ref + (long long)gy * P.ref_stride;
                if (inside) v[u] = *reinterpret_cast<const u32_any_align *>(row + gx0);
                else {
                    v[u] = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) { const int gx = gx0 + b > last_x ? last_x : gx0 + b; v[u] |= (uint32_t)row[gx] << (8 * b); }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ry = ry0 + u * n_waves;
                if (ry < win_rows && lane < dwords_per_row) reinterpret_cast<uint32_t *>(win + ry * kPitch)[lane] = v[u];
            }
        }
    }
    if (tid < NBLK) best_lds[tid] = 0xFFFFFFFFu;
    if (tid == 0) *next_item = (uint32_t)n_waves;
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

    int fetched = 0;
    for (int item = wave; item < n_items; item = __builtin_amdgcn_readfirstlane(fetched)) {
        fetched = lane == 0 ? (int)atomicAdd(next_item, 1u) : 0;
        const bool wide = item < n_main;
        int i, gq = 0, ncol = 0, chunk = 0;
        if (wide) { i = item & (kTileBlocksX - 1); gq = item >> 3; }            // block column fastest: the heavy passes come first
        else { const int it = item - n_main; i = it & (kTileBlocksX - 1); const int rest = it >> 3; ncol = rest / n_chunks; chunk = rest - ncol * n_chunks; }
        if (i >= blocks_left_x) continue;
        // Current rows of the column's TBY blocks, the same in every lane, held in VGPRs: TBY x 16 scalars next to the
        // loop state do not fit the SGPR file, and the v_sad chain does not care which file its operand comes from.
        // `lane0` is a zero the compiler cannot see through, so these stay vector loads.
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
            uint32_t pid = (uint32_t)col | (lane_ok ? 0u : DEAD);       // position of the candidate STARTED by row ry: (ry, col)
            // Eight window rows.  STEADY: every block row of the tile is inside its band and finishes a candidate on every
            // one of the eight rows (all but the first and last few groups), so the body carries no wave-uniform tests at all.
            auto rows8 = [&](auto steady_tag, int ry8) {
                constexpr bool STEADY = decltype(steady_tag)::value;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int ry = ry8 + m;
                    if (!STEADY && ry >= win_rows) break;                // wave-uniform
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(colbase + ry * kPitch);
                    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
                    const uint32_t a0 = __builtin_amdgcn_alignbit(d1, d0, sh), a1 = __builtin_amdgcn_alignbit(d2, d1, sh);
#pragma unroll
                    for (int j = 0; j < TBY; ++j) {
                        const int rel = ry - 8 * j;                      // row inside block row j's band
                        if (!STEADY && (rel < 0 || rel >= span + 7 || j >= blocks_left_y)) continue;      // wave-uniform
#pragma unroll
                        for (int p = 0; p < 8; ++p) {
                            const int slot = (m - p) & 7;
                            const uint32_t init = p == 0 ? pid : acc[j][slot];   // a candidate's chain starts from its position
                            acc[j][slot] = sadhi8(a1, c[j][p][1], sadhi8(a0, c[j][p][0], init));
                        }
                        const int dyi = rel - 7;                         // the candidate row that has now seen all 8 block rows
                        if (STEADY || (dyi >= 0 && dyi < span)) {        // wave-uniform
                            const uint32_t key = acc[j][(m + 1) & 7];    // (cost << 16 | row << 8 | column)
                            best[j] = key < best[j] ? key : best[j];
                            if (COSTS && lane_ok) {
                                const size_t blk = (size_t)(ty * TBY + j) * P.blocks_x + (tx * kTileBlocksX + i);
                                P.costs[blk * (size_t)(span * span) + (size_t)dyi * span + (size_t)(64 * gq + lane)] = key >> 16;
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
            const unsigned char *base = win + lrow * kPitch + (col & ~3);
            uint32_t a[8][2];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(base + p * kPitch);
                const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
                a[p][0] = __builtin_amdgcn_alignbit(d1, d0, sh);
                a[p][1] = __builtin_amdgcn_alignbit(d2, d1, sh);
            }
            const uint32_t pid = (uint32_t)(prow << 8 | col);
#pragma unroll
            for (int j = 0; j < TBY; ++j) {
                if (j >= blocks_left_y || 64 * chunk + 63 < 8 * j || 64 * chunk >= 8 * j + span) continue;   // wave-uniform
                const int dyi = prow - 8 * j;
                const bool ok = (unsigned)dyi < (unsigned)span;
                uint32_t key = ok ? pid : (pid | DEAD);
#pragma unroll
                for (int p = 0; p < 8; ++p) key = sadhi8(a[p][1], c[j][p][1], sadhi8(a[p][0], c[j][p][0], key));
                best[j] = key < best[j] ? key : best[j];
                if (COSTS && ok) {
                    const size_t blk = (size_t)(ty * TBY + j) * P.blocks_x + (tx * kTileBlocksX + i);
                    P.costs[blk * (size_t)(span * span) + (size_t)dyi * span + (size_t)(64 * G + ncol)] = key >> 16;
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
                               int tile_rows, uint32_t *d_coef_scratch, int cu_count, hipStream_t stream)
{
    MeParams P;
    P.cur = d_cur; P.ref = d_ref; P.cur_stride = cur_stride; P.ref_stride = ref_stride;
    P.width = width; P.height = height; P.range = range;
    P.blocks_x = width / 8; P.blocks_y = height / 8;
    P.tiles_x = (P.blocks_x + kTileBlocksX - 1) / kTileBlocksX;
    if (tile_rows <= 0) {
        // Tile height by frame size (profiles/r03_me_variants.txt, R = 64, 256 CUs; only the ratios matter, they hold for any
        // range).  Two 8-wave workgroups are resident per CU whatever the height (110 VGPRs), so a launch of T tiles is
        // floor(T / slots) full rounds of R_h each plus a last, partly filled one: up to one workgroup per CU it costs what a
        // lone tile costs (A_h: the workgroup has its CU to itself), from there to a full round it grows to R_h; behind full
        // rounds it overlaps their ragged end (x 0.9).  Tall tiles halve the position transforms per block (R_8 < 2 R_4),
        // short ones waste less of a thin launch: 4K wants 8 rows, a 544-row stripe 4, 360p 2.
        const int cus = cu_count > 0 ? cu_count : 256;
        const long long slots = 2LL * cus;
        const float R[3] = {0.565f, 0.312f, 0.182f}, A[3] = {0.35f, 0.20f, 0.13f};
        const int cand[3] = {8, 4, 2};
        float best_t = 0.f;
        for (int c = 0; c < 3; ++c) {
            const long long tiles = (long long)P.tiles_x * ((P.blocks_y + cand[c] - 1) / cand[c]);
            const long long full = tiles / slots, rest = tiles % slots;
            float t = (float)full * R[c];
            if (rest) {
                const float part = rest <= cus ? A[c] : A[c] + (R[c] - A[c]) * (float)(rest - cus) / (float)(slots - cus);
                t += full ? 0.9f * part : part;
            }
            if (c == 0 || t < best_t) { best_t = t; tile_rows = cand[c]; }
        }
    }
    const int tby = tile_rows >= 8 ? 8 : (tile_rows >= 4 ? 4 : 2);     // 1 is served by 2-row tiles (partial tiles are handled anyway)
    const int tiles_y = (P.blocks_y + tby - 1) / tby;
    const int span = 2 * range + 1;
    const int n_rows = 8 * (tby - 1) + span;
    P.best = d_best; P.costs = d_costs;
    const int n_blocks = P.blocks_x * P.blocks_y;
    hipLaunchKernelGGL(me_coef_kernel, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, stream, d_cur, cur_stride,
                       P.blocks_x, n_blocks, P.tiles_x, tby, d_coef_scratch);
    {
        const hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return e0;
    }
    const int F = span >> 3;
    const int n_item_rows = (tby - 1 + F + 1) >> 1;
    const int main_rows = 16 * n_item_rows;
    const size_t lds = (size_t)kTileBlocksX * tby * 512 + (size_t)((main_rows > n_rows ? main_rows : n_rows) + 7) * kPitch + 16;   // minima, window, item counter
    const dim3 grid((unsigned)(P.tiles_x * tiles_y)), block(512);        // 8-wave workgroups, two per CU: 4 waves per SIMD
    const uint32_t *cf = d_coef_scratch;
#ifdef X266_ME_TIMING
#define X266_ME5(T) hipLaunchKernelGGL((satd_search_kernel<T, false, 512, 4>), grid, block, lds, stream, P, cf)
#else
#define X266_ME5(T) do { if (d_costs) hipLaunchKernelGGL((satd_search_kernel<T, true, 512, 4>), grid, block, lds, stream, P, cf); \
                         else         hipLaunchKernelGGL((satd_search_kernel<T, false, 512, 4>), grid, block, lds, stream, P, cf); } while (0)
#endif
    if (tby == 8) X266_ME5(8); else if (tby == 4) X266_ME5(4); else X266_ME5(2);
#undef X266_ME5
    return hipGetLastError();
}

hipError_t launch_sad_search(const uint8_t *d_cur, long long cur_stride, const uint8_t *d_ref, long long ref_stride,
                              int width, int height, int range, x266_me_result_t *d_best, uint32_t *d_costs,
                              int tile_rows, hipStream_t stream)
{
    MeParams P;
    P.cur = d_cur; P.ref = d_ref; P.cur_stride = cur_stride; P.ref_stride = ref_stride;
    P.width = width; P.height = height; P.range = range;
    P.blocks_x = width / 8; P.blocks_y = height / 8;
    P.tiles_x = (P.blocks_x + kTileBlocksX - 1) / kTileBlocksX;
    (void)tile_rows;
    const int tby = 2;                                               // two block rows per tile whatever "me_tile_rows" says: faster than 4 on a 4K frame (1.21 against 1.27 ms) and finer-grained on small ones
    const int tiles_y = (P.blocks_y + tby - 1) / tby;
    const int span = 2 * range + 1;
    const int n_rows = 8 * (tby - 1) + span;
    P.best = d_best; P.costs = d_costs;
    dim3 grid((unsigned)(P.tiles_x * tiles_y)), block(256);
    const size_t lds = 256 + (size_t)(n_rows + 7) * kPitch;
#define X266_SADS(T) do { if (d_costs) hipLaunchKernelGGL((sad_search_kernel<T, true>), grid, block, lds, stream, P); \
                          else         hipLaunchKernelGGL((sad_search_kernel<T, false>), grid, block, lds, stream, P); } while (0)
    X266_SADS(2);
#undef X266_SADS
    return hipGetLastError();
}

}  // namespace x266
