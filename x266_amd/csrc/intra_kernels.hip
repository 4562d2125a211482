// intra_kernels.hip -- 32x32 intra prediction, gfx950.
//
// The HEVC 35-mode predictor the reference's WIP RTL sketch tabulates (src/mkIntra32-wip.bsv:
// IntraRef_t :36-39, iIdx / iFact tables :75-112, projected side references :150-316, two-tap
// interpolation :358-361, DC :380-384) -- H.265 8.4.4.2.4-6 at nTbS = 32: mode 0 planar, 1 DC,
// 2..34 angular, references used as given.  PARITY UNPINNED upstream (no C model); bit-exact with
// this repository's oracle.
//
// One wave per predicted block (reference set r, mode m): 144 bytes in, 1 KiB out -- a write-bound
// byte kernel.  The wave builds the extended reference array ref[-32 .. 65] in a private LDS slot
// (main side as it lies, negative positions projected from the other side with invAngle); lane
// (k = l >> 1, h = l & 1) then interpolates 16 consecutive samples of line k: its two taps are
// adjacent bytes of ref[], so one v_dot4_u32_u8 per sample against the weights (32 - f, f) with
// the rounding term as the accumulator.  The vertical family (modes 18..34) produces rows and is
// stored directly; the horizontal family (2..17) produces columns and is turned through a 1 KiB
// LDS tile with the transposing read ds_read_b64_tr_b8.  Stores are 1 KiB-linear, "sc1 nt" (x266_device.hpp).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "x266_device.hpp"
#include "x266_hadamard.hpp"
#include "x266_mfma_blocks.hpp"

namespace x266 {
namespace {

// intraPredAngle (H.265 Table 8-4) and |invAngle| (Table 8-5) from the mode, in scalar code: the
// mode is wave-uniform, so these are a handful of SALU selects instead of dependent memory reads.
__device__ __forceinline__ int angle_magnitude(int j)      // j = distance from the pure horizontal / vertical mode, 0..8
{
    const uint32_t lo = 0x09050200u, hi = 0x1A15110Du;      // 0,2,5,9 | 13,17,21,26
    return j >= 8 ? 32 : (int)(((j & 4) ? hi : lo) >> (8 * (j & 3))) & 0xFF;
}
__device__ __forceinline__ int intra_angle(int mode)
{
    const int pure = mode < 18 ? 10 : 26;
    const int j = mode - pure, m = angle_magnitude(j < 0 ? -j : j);
    return (mode < 18) == (j < 0) ? m : -m;                 // horizontal family: positive below 10; vertical: positive above 26
}
__device__ __forceinline__ int intra_inv_angle_magnitude(int j)   // j = 1..8
{
    const int t[8] = {4096, 1638, 910, 630, 482, 390, 315, 256};
    int v = t[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) v = (j == i + 1) ? t[i] : v;
    return v;
}

constexpr int kUnits = 7;           // predictions per wave: one round trip fetches all seven reference sets (7 x 9 pieces of 16 bytes = 63 lanes)
constexpr int kRawBytes = 144;      // x266_intra_ref_t
constexpr int kExtBytes = 128;      // ref[-32 .. 95]: negative-angle modes only
constexpr int kSlotBytes = 16 + 2 * kUnits * kRawBytes + kExtBytes + 1024;   // two raw areas (current round, next round)

// a * b + c on two 16-bit lanes, as ONE instruction (written as an expression, the compiler turns "two products plus a constant" into
// multiply, multiply-add, add)
__device__ __forceinline__ uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t d;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// 16 samples of one line: taps are the 17 bytes from `p` on (any alignment), weights (32 - f, f).
// Packed 16-bit arithmetic, two samples per instruction: even samples (32-f)*B[2i] + f*B[2i+1],
// odd samples (32-f)*B[2i+1] + f*B[2i+2].
__device__ __forceinline__ void interpolate16(const unsigned char *p, uint32_t f, uint32_t (&px)[4])
{
    const int o = (int)((uintptr_t)p & 3);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p - o);
    uint32_t d[6], a[5];
#pragma unroll
    for (int i = 0; i < 6; ++i) d[i] = q[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) a[i] = __builtin_amdgcn_alignbit(d[i + 1], d[i], (uint32_t)(8 * o));   // bytes p[4i .. 4i+3]
    // weights and rounding term times 8: the ">> 5" becomes ">> 8", i.e. the sample is the HIGH byte of its 16-bit lane and the byte
    // permute that interleaves even and odd samples picks it up for free (255 * 256 + 128 < 2^16: no overflow)
    const uint32_t w0 = (256u - 8u * f) * 0x00010001u, w1 = (8u * f) * 0x00010001u;
    const uint32_t R = 0x00800080u;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t t0 = __builtin_amdgcn_perm(0u, a[g], 0x0c020c00u);            // B0, B2
        const uint32_t t1 = __builtin_amdgcn_perm(0u, a[g], 0x0c030c01u);            // B1, B3
        const uint32_t t2 = __builtin_amdgcn_perm(a[g + 1], a[g], 0x0c040c02u);      // B2, B4
        const uint32_t e = pk_mad_u16(t0, w0, pk_mad_u16(t1, w1, R));     // samples 0, 2 (times 256)
        const uint32_t od = pk_mad_u16(t1, w0, pk_mad_u16(t2, w1, R));    // samples 1, 3
        px[g] = __builtin_amdgcn_perm(od, e, 0x07030501u);                // e0 o0 e1 o1, the high bytes
    }
}

// This lane's 16 samples (16h .. 16h+15) of line k of the prediction.  Lines are ROWS for
// planar, DC and the vertical family, COLUMNS for the horizontal family (return value true).
// left / top: the reference set in LDS (top[0] = corner); ext: 128 bytes of wave-private LDS scratch.
// FRAGMENT_LANES false: (k, h) = (lane >> 1, lane & 1), consecutive lanes = consecutive bytes of the 1 KiB prediction (the stores);
// true: (lane & 31, lane >> 5), the lane IS the A-operand fragment of the 32x32 matrix core (row k, K-slots 16h ..).
template <bool FRAGMENT_LANES = false>
__device__ __forceinline__ bool predict_line16(int mode, const unsigned char *left, const unsigned char *top,
                                               unsigned char *ext, int lane, uint32_t (&px)[4])
{
    const int k = FRAGMENT_LANES ? lane & 31 : lane >> 1, h = FRAGMENT_LANES ? lane >> 5 : lane & 1;
    if (mode >= 2) {
        const int angle = intra_angle(mode);
        const bool vertical = mode >= 18;
        const int t = (k + 1) * angle;
        const int idx = t >> 5;
        const uint32_t f = (uint32_t)(t & 31);
        const unsigned char *line;                          // line[x] = ref[x]
        if (angle >= 0) {
            // ref[x] = p[-1+x][-1] is the top array as it lies; ref[1+i] = p[-1][i] is the left array
            // (ref[0] is never a tap when the angle is not negative)
            line = vertical ? top : left - 1;
        } else {
            // negative angles: the other side's samples, projected with invAngle, sit in front of
            // ref[0].  Build ref[-32 .. 64] in ext (two positions per lane).
            const int inv = intra_inv_angle_magnitude(vertical ? 26 - mode : mode - 10);
            const int last = angle;                         // (32 * angle) >> 5
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                const int e = lane + 64 * rep, x = e - 32;
                unsigned v = 0;
                if (x >= 0) {
                    if (x <= 64) v = vertical ? top[x] : (x == 0 ? top[0] : left[x - 1]);
                } else if (last < -1 && x >= last) {
                    const int s = -1 + ((-x * inv + 128) >> 8);              // x * invAngle, invAngle = -inv
                    v = vertical ? left[s] : top[1 + s];
                }
                ext[e] = (unsigned char)v;
            }
            __builtin_amdgcn_wave_barrier();
            line = ext + 32;
        }
        interpolate16(line + 16 * h + idx + 1, f, px);
        return !vertical;
    }
    if (mode == 1) {                                        // DC: 32 top + 32 left samples
        uint32_t s = lane < 32 ? (uint32_t)top[1 + lane] + (uint32_t)left[lane] : 0u;
        s = sum_over_row16(s);
        s = (uint32_t)(__builtin_amdgcn_readlane((int)s, 0) + __builtin_amdgcn_readlane((int)s, 16));   // lanes 32.. hold zeros
        const uint32_t dc = (s + 32u) >> 6;
        px[0] = px[1] = px[2] = px[3] = dc * 0x01010101u;
        return false;
    }
    // planar, row y = k, columns 16h..: ((31-x) L + (x+1) TR + (31-y) T[x] + (y+1) BL + 32) >> 6
    //   = (C + x (TR - L) + (31-y) T[x]) >> 6,  C = 31 L + TR + (y+1) BL + 32: a per-lane ramp plus one
    // multiply per sample, in packed 16-bit lanes (the ramps wrap modulo 2^16 on the way; every finished sum is below 2^16).
    typedef unsigned short v2u __attribute__((ext_vector_type(2)));
    const int tr = top[33], bl = left[32], y = k, lv = left[y];
    const int D = tr - lv, x0 = 16 * h;
    // everything times 4: the ">> 6" becomes ">> 8" and the interleaving byte permute takes the high bytes (4 * (255 * 64 + 32) < 2^16)
    const int c0 = 4 * (31 * lv + tr + (y + 1) * bl + 32 + x0 * D), D4 = 4 * D;
    v2u re = {(unsigned short)c0, (unsigned short)(c0 + 2 * D4)}, ro = {(unsigned short)(c0 + D4), (unsigned short)(c0 + 3 * D4)};
    const v2u inc = {(unsigned short)(4 * D4), (unsigned short)(4 * D4)};
    const unsigned short wy = (unsigned short)(4 * (31 - y));
    const v2u W = {wy, wy};
    const uint32_t *q = reinterpret_cast<const uint32_t *>(top + x0);               // top[1 + x0 ..]: one byte past a dword boundary
    uint32_t d[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) d[i] = q[i];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint32_t a = __builtin_amdgcn_alignbit(d[g + 1], d[g], 8u);
        const v2u t0 = __builtin_bit_cast(v2u, __builtin_amdgcn_perm(0u, a, 0x0c020c00u));
        const v2u t1 = __builtin_bit_cast(v2u, __builtin_amdgcn_perm(0u, a, 0x0c030c01u));
        const v2u e = t0 * W + re, od = t1 * W + ro;
        px[g] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, od), __builtin_bit_cast(uint32_t, e), 0x07030501u);
        re += inc;
        ro += inc;
    }
    return false;
}

// lane (k, h) holds samples 16h..16h+15 of COLUMN k in px[]; afterwards it holds 16 consecutive samples of a ROW and the return value
// is their byte offset in the row-major 32x32 prediction.  The columns go to LDS as they are (a column-major tile, one b128 write per
// lane) and come back through gfx950's transposing read, which turns 8x8 byte blocks: ds_read_b64_tr_b8 hands lane 16q + 8p + e, as
// byte j, element e of the 8 bytes lane 16q + 2j + p addressed (tools/probes/lds_tr8_read_test.hip).  Source lane 16q + 2j + p addresses
// column 16H + j (second read: + 8), rows R..R+7 of block (R, H) = (8 ((2q + p) & 3), (2q + p) >> 2); the receiving lane holds row
// R + e, columns 16H..16H+15.  (Round 3 scattered the column with sixteen ds_write_b8 per lane.)
__device__ __forceinline__ unsigned turn_columns(unsigned char *cm, int lane, uint32_t (&px)[4])
{
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<v4i *>(cm + lane * 16) = v4i{(int)px[0], (int)px[1], (int)px[2], (int)px[3]};      // cm[32 k + 16 h ..]
    const int q = lane >> 4, s = lane & 15;
    const int src_blk = 2 * q + (s & 1);
    const unsigned src = (unsigned)(uintptr_t)cm + (unsigned)((16 * (src_blk >> 2) + (s >> 1)) * 32 + 8 * (src_blk & 3));   // low 32 bits of the generic pointer = LDS offset
    u2 a, b;
    asm volatile("ds_read_b64_tr_b8 %0, %2\n\tds_read_b64_tr_b8 %1, %2 offset:256\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(src) : "memory");
    px[0] = a.x; px[1] = a.y; px[2] = b.x; px[3] = b.y;
    const int blk = 2 * q + ((lane >> 3) & 1);
    return (unsigned)((8 * (blk & 3) + (lane & 7)) * 32 + 16 * (blk >> 2));
}

// A wave takes `rounds` x kUnits consecutive predictions.  Per round ONE round trip fetches the modes, the set indices and
// then all seven reference sets (7 x 9 pieces of 16 bytes, nontemporal); the NEXT round's indices and sets are fetched
// while the current round is computed (two raw areas in the wave's LDS slot), so only the first round trip of a wave is
// exposed.  (Four units per wave and no prefetch, round 1: 0.51 of the HBM peak written; seven: 0.62.)
__global__ __launch_bounds__(256) void intra32_predict_kernel(const x266_intra_ref_t *__restrict__ refs,
                                                              const uint8_t *__restrict__ modes,
                                                              const uint32_t *__restrict__ ref_index,
                                                              uint8_t *__restrict__ pred, size_t n, int rounds)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kSlotBytes];
    const int lane = threadIdx.x & 63;
    const int wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t unit0 = ((size_t)blockIdx.x * (blockDim.x >> 6) + wave_in_wg) * (size_t)(kUnits * rounds);
    if (unit0 >= n) return;
    unsigned char *slot = lds + wave_in_wg * kSlotBytes;
    unsigned char *raw2 = slot + 16;                        // two raw areas; set j of area a at raw2 + a*kUnits*144 + 144 j: left[64] | top[65]
    unsigned char *ext = raw2 + 2 * kUnits * kRawBytes;     // ext[e] = ref[e - 32]
    unsigned char *tile = ext + kExtBytes;

    const int ju = lane < kUnits ? lane : kUnits - 1;
    const int set = lane / 9, piece = lane - 9 * set;       // lanes 0..62 fetch 7 x 9 pieces of 16 bytes
    auto fetch_index = [&](size_t first, int &mode_out, uint32_t &ref_out) {
        size_t u = first + ju;
        if (u >= n) u = n - 1;
        mode_out = modes[u];
        ref_out = ref_index ? ref_index[u] : (uint32_t)u;
    };
    auto fetch_sets = [&](uint32_t my_ref) -> v4i {
        const uint32_t r = (uint32_t)__shfl((int)my_ref, set < kUnits ? set : 0);
        v4i v = {0, 0, 0, 0};
        if (lane < 9 * kUnits) v = load16<true>(reinterpret_cast<const unsigned char *>(refs + r) + piece * 16);
        return v;
    };
    int my_mode, next_mode = 0;
    uint32_t my_ref, next_ref = 0;
    fetch_index(unit0, my_mode, my_ref);
    v4i sets = fetch_sets(my_ref);
    if (rounds > 1) fetch_index(unit0 + kUnits, next_mode, next_ref);
    if (lane < 9 * kUnits) *reinterpret_cast<v4i *>(raw2 + lane * 16) = sets;
    __builtin_amdgcn_wave_barrier();

#pragma unroll 1
    for (int rd = 0; rd < rounds; ++rd) {
        const size_t base = unit0 + (size_t)rd * kUnits;
        if (base >= n) break;
        const bool more = rd + 1 < rounds && base + kUnits < n;
        int mode_after = 0;
        uint32_t ref_after = 0;
        if (more) {                                          // next round's sets (and the round after's indices) in flight during this round
            sets = fetch_sets(next_ref);
            if (rd + 2 < rounds) fetch_index(base + 2 * kUnits, mode_after, ref_after);
        }
        const unsigned char *raw_all = raw2 + (rd & 1) * (kUnits * kRawBytes);
#pragma unroll 1
        for (int j = 0; j < kUnits; ++j) {
            const size_t unit = base + j;
            if (unit >= n) break;
            const int mode = __builtin_amdgcn_readlane(my_mode, j);
            const unsigned char *left = raw_all + j * kRawBytes, *top = left + 64;   // top[0] = corner
            uint32_t px[4];
            unsigned at = (unsigned)lane * 16;                       // rows: lane (k, h) holds samples 16h.. of row k
            if (predict_line16(mode, left, top, ext, lane, px)) at = turn_columns(tile, lane, px);   // columns: turned through the tile
            store16_sc1(pred + unit * 1024 + at, v4i{(int)px[0], (int)px[1], (int)px[2], (int)px[3]});
        }
        if (more) {
            if (lane < 9 * kUnits) *reinterpret_cast<v4i *>(raw2 + ((rd + 1) & 1) * (kUnits * kRawBytes) + lane * 16) = sets;
            __builtin_amdgcn_wave_barrier();
            my_mode = next_mode;
            next_mode = mode_after;
            next_ref = ref_after;
        }
    }
}

// ---- prediction -> residual -> forward DCT32 in one kernel (round 5) ------------------------------------------------------
// coef[i] = DCT32(src[i] - prediction(refs[ref_index[i]], modes[i])): the encoder loop's form of intra coding -- the chosen mode's
// prediction is consumed where it is made and never reaches HBM (1 KiB of source + 144 B of references in, 2 KiB of coefficients
// out, against 144 B + 1 KiB out for the predictor, 2 + 2 KiB + 2 KiB for residual formation and 2 + 2 KiB for the transform).
// predict_line16<true> makes the prediction directly as the A-operand fragment of the 32x32 matrix core (lane = row, 16 samples of
// half h); the horizontal family's columns are turned by the transposing LDS read addressed so that the RECEIVING lane is the
// fragment lane.  The transform is linear before its first rounding, so pass 1 is G * src + (-G) * pred on the 8-bit samples as
// they are -- one byte plane per operand, the +128 of the signed-offset trick cancels (dct32_from_tiles_kernel's scheme).
// A wave takes kFusedUnits consecutive blocks and fetches everything they need UP FRONT (modes, set indices, the sets, the source
// fragments): no loop-carried prefetch, so the compiler's own wait counts stay exact (every load is older than every store).
constexpr int fused_slot_bytes(int units) { return 16 + units * kRawBytes + kExtBytes + 1024 + 2048; }   // raw sets | ext | column tile | output converter

// lane (k, h) = (lane & 31, lane >> 5) holds samples 16h.. of COLUMN k; afterwards rows: the same lane holds row k, columns 16h..
// (ds_read_b64_tr_b8, see turn_columns: receiving lane 16q + 8p + e gets row R + e of the 8x8 byte block the source lanes 16q + 2j + p
// address; here R = 16 (q & 1) + 8 p and the blocks' columns are 16 (q >> 1) + j [+ 8 for the second read])
__device__ __forceinline__ void turn_columns_to_fragment(unsigned char *cm, int lane, uint32_t (&px)[4])
{
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    *reinterpret_cast<v4i *>(cm + (lane & 31) * 32 + (lane >> 5) * 16) = v4i{(int)px[0], (int)px[1], (int)px[2], (int)px[3]};   // column-major tile
    const int q = lane >> 4, s = lane & 15;
    const unsigned src = (unsigned)(uintptr_t)cm + (unsigned)((16 * (q >> 1) + (s >> 1)) * 32 + 16 * (q & 1) + 8 * (s & 1));
    u2 a, b;
    asm volatile("ds_read_b64_tr_b8 %0, %2\n\tds_read_b64_tr_b8 %1, %2 offset:256\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(src) : "memory");
    px[0] = a.x; px[1] = a.y; px[2] = b.x; px[3] = b.y;
}

template <int kFusedUnits>
__global__ __launch_bounds__(256) void intra32_residual_dct32_kernel(const x266_intra_ref_t *__restrict__ refs,
                                                                     const uint8_t *__restrict__ modes,
                                                                     const uint32_t *__restrict__ ref_index,
                                                                     const uint8_t *__restrict__ src, int16_t *__restrict__ coef,
                                                                     size_t n, const DctOps *__restrict__ ops, unsigned lds_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t unit0 = ((size_t)blockIdx.x * (blockDim.x >> 6) + wave_in_wg) * (size_t)kFusedUnits;
    if (unit0 >= n) return;
    unsigned char *slot = lds + wave_in_wg * lds_per_wave;
    unsigned char *raw = slot + 16, *ext = raw + kFusedUnits * kRawBytes, *tile = ext + kExtBytes, *conv = tile + 1024;
    const int count = n - unit0 < (size_t)kFusedUnits ? (int)(n - unit0) : kFusedUnits;

    // everything the wave's blocks need, issued at once: modes + set indices (lanes 0..3), then the sets (4 x 9 pieces of 16 bytes)
    // and every block's source fragment (row lane & 31, bytes 16 (lane >> 5) ..: the instruction covers the block's 1 KiB whole)
    const int ju = lane < count ? lane : count - 1;
    const int my_mode = modes[unit0 + ju];
    const uint32_t my_ref = ref_index ? ref_index[unit0 + ju] : (uint32_t)(unit0 + ju);
    v4i sv[kFusedUnits];
    const unsigned char *sp = src + unit0 * 1024 + (size_t)(lane & 31) * 32 + (size_t)(lane >> 5) * 16;
#pragma unroll
    for (int j = 0; j < kFusedUnits; ++j) sv[j] = load16<true>(sp + (size_t)(j < count ? j : count - 1) * 1024);
    {
        const int set = lane / 9, piece = lane - 9 * set;
        const uint32_t r = (uint32_t)__shfl((int)my_ref, set < count ? set : 0);
        if (lane < 9 * count) *reinterpret_cast<v4i *>(raw + lane * 16) = load16<true>(reinterpret_cast<const unsigned char *>(refs + r) + piece * 16);
    }
    const LaneConsts k = load_consts(ops, lane);
    __builtin_amdgcn_wave_barrier();
    const v4i bias = {(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};
    const v16i round1 = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8};
    const unsigned c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < kFusedUnits; ++j) {
        if (j >= count) break;
        const int mode = __builtin_amdgcn_readlane(my_mode, j);
        const unsigned char *left = raw + j * kRawBytes, *top = left + 64;
        uint32_t px[4];
        if (predict_line16<true>(mode, left, top, ext, lane, px)) turn_columns_to_fragment(tile, lane, px);
        const v4i pv = {(int)px[0], (int)px[1], (int)px[2], (int)px[3]};
        v16i acc = mfma(sv[j] ^ bias, k.p1, round1);
        acc = mfma(pv ^ bias, k.tr, acc);                         // k.tr = -p1 in the forward tables
        v4i o0, o1;
        fwd_finish<4, 11>(acc, k, o0, o1);
        __builtin_amdgcn_wave_barrier();
        *reinterpret_cast<v4i *>(conv + lds_slot(c, 2 * h)) = o0;
        *reinterpret_cast<v4i *>(conv + lds_slot(c, 2 * h + 1)) = o1;
        __builtin_amdgcn_wave_barrier();
        const v4i s0 = *reinterpret_cast<const v4i *>(conv + lds_slot(lane >> 2, lane & 3));
        const v4i s1 = *reinterpret_cast<const v4i *>(conv + lds_slot(16 + (lane >> 2), lane & 3));
        char *dst = reinterpret_cast<char *>(coef) + (unit0 + j) * 2048 + lane * 16;
        store16_sc1nt(dst, s0);
        store16_sc1nt(dst + 1024, s1);
    }
}

// ---- mode decision ("Decide" channel of the RTL sketch, IntraChannel_t :41-44) ------------------
// costs[b][m] = sum over the sixteen 8x8 sub-blocks of satd8x8(src - prediction m), m = 0..34, without
// the predictions ever leaving the CU.  The Hadamard transform is linear and a 9-bit difference cannot
// wrap int16, so satd(src - pred) = (sum |H src - H pred| + 2) >> 2 per sub-block (as in me_kernels.hip):
// H src is formed once per block, H pred once per mode -- two modes per matrix-core pass, their
// 16 + 16 sub-blocks being the 32 columns of the 64x64x32 Hadamard GEMM -- and scored with v_sad_u16.
// Tiles of the decision kernel have a 40-byte row pitch and the second prediction tile starts 32 bytes past a
// 256-byte boundary: the 8-byte window reads of the 16 + 16 sub-blocks then fall into 32 distinct bank pairs
// (with a 32-byte pitch all four sub-block rows and both tiles share banks: 8-way conflicts).
constexpr int kPitch = 40, kTile = 32 * kPitch, kTileB = kTile + 32;
constexpr int kCostSlot = 16 + kRawBytes + kExtBytes + kTile + kTileB + kTile;   // raw | ext | src tile | two prediction tiles

__global__ __launch_bounds__(256) void intra32_costs_kernel(const x266_intra_ref_t *__restrict__ refs,
                                                            const uint8_t *__restrict__ src,
                                                            uint32_t *__restrict__ costs, uint8_t *__restrict__ best_mode,
                                                            size_t n)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kCostSlot];
    const int lane = threadIdx.x & 63;
    const int wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t b = (size_t)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
    if (b >= n) return;
    unsigned char *raw = lds + wave_in_wg * kCostSlot + 16;
    unsigned char *ext = raw + kRawBytes, *stile = ext + kExtBytes, *ptile = stile + kTile;
    const v4i S = {(int)0x80808080u, (int)0x80808080u, (int)0x80808080u, (int)0x80808080u};   // pixels -> signed; the offset cancels

    if (lane < 9) *reinterpret_cast<v4i *>(raw + lane * 16) = load16<true>(reinterpret_cast<const unsigned char *>(refs + b) + lane * 16);
    {
        const v4i sv = load16<true>(src + b * 1024 + lane * 16) ^ S;
        unsigned char *d = stile + (lane >> 1) * kPitch + (lane & 1) * 16;                      // 8-byte aligned rows
        *reinterpret_cast<uint2 *>(d) = make_uint2((uint32_t)sv[0], (uint32_t)sv[1]);
        *reinterpret_cast<uint2 *>(d + 8) = make_uint2((uint32_t)sv[2], (uint32_t)sv[3]);
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned char *left = raw, *top = raw + 64;

    // lane (n, half): rows 4*half .. +3 of sub-block n & 15 (sy = bits 3:2, sx = bits 1:0); columns 16..31 repeat 0..15
    const int nn = lane & 31, half = lane >> 5, sb = nn & 15;
    const unsigned frag = (unsigned)((8 * (sb >> 2) + 4 * half) * kPitch + 8 * (sb & 3));
    const HadamardOps H = make_hadamard_ops(lane);
    auto window = [&](const unsigned char *tile, v4i &w0, v4i &w1) {
        const uint2 r0 = *reinterpret_cast<const uint2 *>(tile + frag), r1 = *reinterpret_cast<const uint2 *>(tile + frag + kPitch);
        const uint2 r2 = *reinterpret_cast<const uint2 *>(tile + frag + 2 * kPitch), r3 = *reinterpret_cast<const uint2 *>(tile + frag + 3 * kPitch);
        w0 = v4i{(int)r0.x, (int)r0.y, (int)r1.x, (int)r1.y};
        w1 = v4i{(int)r2.x, (int)r2.y, (int)r3.x, (int)r3.y};
    };
    uint32_t cs[16];
    {
        v4i w0, w1;
        window(stile, w0, w1);
        hadamard_pack(H, w0, w1, cs);
    }
    uint32_t best_key = 0xFFFFFFFFu;
#pragma unroll 1
    for (int pair = 0; pair < 18; ++pair) {
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {
            const int mode = 2 * pair + t;
            if (mode > 34) break;
            uint32_t px[4];
            const bool columns = predict_line16(mode, left, top, ext, lane, px);
#pragma unroll
            for (int g = 0; g < 4; ++g) px[g] ^= 0x80808080u;
            unsigned char *tile = ptile + t * kTileB;
            unsigned at = (unsigned)((lane >> 1) * kPitch + (lane & 1) * 16);
            if (columns) {                                   // turned inside the tile's own first KiB (one wave: its LDS operations execute in order)
                const unsigned o = turn_columns(tile, lane, px);
                at = (o >> 5) * kPitch + (o & 31);
            }
            *reinterpret_cast<uint2 *>(tile + at) = make_uint2(px[0], px[1]);
            *reinterpret_cast<uint2 *>(tile + at + 8) = make_uint2(px[2], px[3]);
        }
        __builtin_amdgcn_wave_barrier();
        v4i w0, w1;
        window(ptile + (nn >> 4) * kTileB, w0, w1);
        uint32_t p[16];
        hadamard_pack(H, w0, w1, p);
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) s = __builtin_amdgcn_sad_u16(p[k], cs[k], s);
        s = sum_with_other_half(s);                          // the other half of the coefficient rows
        const uint32_t c = sum_over_row16((s + 2u) >> 2);    // satd8x8 of this sub-block, summed over the sixteen sub-blocks (one row of lanes per tile)
        __builtin_amdgcn_wave_barrier();
        const uint32_t c_a = (uint32_t)__builtin_amdgcn_readlane((int)c, 0), c_b = (uint32_t)__builtin_amdgcn_readlane((int)c, 16);
        const int m_a = 2 * pair, m_b = 2 * pair + 1;
        if (lane == 0) {
            costs[b * 35 + m_a] = c_a;
            if (m_b < 35) costs[b * 35 + m_b] = c_b;
        }
        const uint32_t k_a = (c_a << 6) | (uint32_t)m_a, k_b = m_b < 35 ? ((c_b << 6) | (uint32_t)m_b) : 0xFFFFFFFFu;
        best_key = k_a < best_key ? k_a : best_key;
        best_key = k_b < best_key ? k_b : best_key;
    }
    if (best_mode && lane == 0) best_mode[b] = (uint8_t)(best_key & 63u);
}

}  // namespace

hipError_t launch_intra32_predict(const x266_intra_ref_t *d_refs, const uint8_t *d_modes, const uint32_t *d_ref_index,
                                  uint8_t *d_pred, size_t n, int rounds, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    if (rounds < 1) rounds = 1;
    while (rounds > 1 && n / (size_t)(kUnits * rounds) < 8192) --rounds;      // small batches: keep the grid large enough to fill the chip
    const size_t per_wave = (size_t)kUnits * rounds;
    const size_t waves = (n + per_wave - 1) / per_wave, wgs = (waves + 3) / 4;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    // (not write-bound: capping the resident waves the way the write-only stream likes it -- 10 per CU, 7.4 TB/s -- slows this
    //  kernel from 5.4 to 3.2-5.1 TB/s written; it is paced by its own VALU + LDS work, profiles/r04_intra_occupancy.txt)
    hipLaunchKernelGGL(intra32_predict_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, d_refs, d_modes, d_ref_index, d_pred, n, rounds);
    return hipGetLastError();
}

hipError_t launch_intra32_residual_dct32(const x266_intra_ref_t *d_refs, const uint8_t *d_modes, const uint32_t *d_ref_index, const uint8_t *d_src,
                                         int16_t *d_coef, size_t n, const DctOps *d_fwd_ops, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    // launch shape (tools/probes/gpu_intra_fused.py over units 1 / 2 / 4 / 7 x workgroup 64 / 128 / 256 x LDS charge, profiles/r05_intra_fused.txt): everything
    // from 2 units per wave up lies within 3 %, occupancy caps only cost -- the kernel runs at what its 1 : 2 read : write mix allows
    constexpr int kUnitsPerWave = 4;
    const size_t waves = (n + kUnitsPerWave - 1) / kUnitsPerWave, wgs = (waves + 3) / 4;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    constexpr unsigned per_wave = (unsigned)((fused_slot_bytes(kUnitsPerWave) + 15) & ~15);
    hipLaunchKernelGGL(intra32_residual_dct32_kernel<kUnitsPerWave>, dim3((unsigned)wgs), dim3(256), 4 * per_wave, stream, d_refs, d_modes, d_ref_index, d_src, d_coef, n, d_fwd_ops, per_wave);
    return hipGetLastError();
}

}  // namespace x266

namespace x266 {

hipError_t launch_intra32_costs(const x266_intra_ref_t *d_refs, const uint8_t *d_src, uint32_t *d_costs, uint8_t *d_best_mode,
                                size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const size_t wgs = (n + 3) / 4;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(intra32_costs_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, d_refs, d_src, d_costs, d_best_mode, n);
    return hipGetLastError();
}

}  // namespace x266
