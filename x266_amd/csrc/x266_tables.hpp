// x266_tables.hpp -- coefficient matrix and the per-lane MFMA operand images
// derived from it (host side, built once per context).
//
// Reference: const int16_t g_t32[32][32], src_tb/dct32.c:30-64 (row k =
// frequency, column n = sample; half-table twin src/mkDct32.bsv:39-73).
// The matrix is generated at compile time from the 32 magnitudes of its first
// column: entry (k,n) is +/- magnitude[fold((2n+1)k mod 128)], the integer
// cosine cos((2n+1)k*pi/64) folded into the first quadrant.
#pragma once

#include <cstdint>
#include <cstring>

namespace x266 {

struct Table32 { int16_t v[32][32]; };

constexpr int kMag[33] = {64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                          61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4,  0};

constexpr int coef32(int k, int n)
{
    if (k == 0) return 64;
    int a = ((2 * n + 1) * k) % 128;          // angle, units of pi/64, period 128
    if (a > 64) a = 128 - a;                  // cos(2pi - t) = cos(t)
    return a > 32 ? -kMag[64 - a] : kMag[a];  // cos(pi - t) = -cos(t)
}

constexpr Table32 make_table32()
{
    Table32 t{};
    for (int k = 0; k < 32; ++k)
        for (int n = 0; n < 32; ++n) t.v[k][n] = static_cast<int16_t>(coef32(k, n));
    return t;
}

// ---------------------------------------------------------------------------
// MFMA fragment geometry for v_mfma_i32_32x32x32_i8 (wave64):
//   A (32 x 32, M x K):  lane l holds row  M = l & 31, 16 K-slots of half l >> 5
//   B (32 x 32, K x N):  lane l holds col  N = l & 31, the SAME 16 K-slots
//   D (32 x 32, M x N):  lane l holds col  N = l & 31, reg r -> row
//                        M = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
// A and B pair K-slots positionally (byte t of half h with byte t of half h),
// so any meaning may be given to slot (h, t) as long as both operands agree.
// ---------------------------------------------------------------------------
constexpr int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Output-order permutation: lane index i of a data operand is given the index
// kappa(i) such that a D fragment's reg r of half h lands on 16*h + r, i.e.
// every lane ends up with 16 CONSECUTIVE outputs (32 contiguous bytes).
constexpr int kappa(int i) { return ((i >> 2) & 1) * 16 + (i >> 3) * 4 + (i & 3); }

// One 64-byte record per lane; device code loads it once per wave.
struct DctLaneOps {
    uint32_t p1[4];     // constant operand of pass 1 (16 int8)
    uint32_t p2[4];     // constant operand of pass 2 (16 int8)
    uint32_t tr[4];     // inverse: permuted identity for the MFMA transpose; forward: the NEGATED pass-1 operand
    int32_t  c1;        // pass-1 accumulator constant: rounding + byte-plane offset fix
    int32_t  c2;        // pass-2 accumulator constant (forward: per lane)
    int32_t  pad[2];
};

struct DctOps {
    DctLaneOps lane[64];
    int32_t    c2r[64][16];   // inverse pass-2 constants, per lane and accumulator reg
};

// The 1-D matrices of the whole transform set in compact form, for the mixed-class tile kernel (xTransformTilesDev):
// 2 KiB that every wave copies into LDS with its first two instructions, so that the operand images of a tile's class
// are LDS reads behind the class byte instead of a second trip to memory (transform_kernels.hip).
//   [0, 1024)      the 32-point matrix (DCT-II, g_t32), 32 x 32 int8, row k = basis function
//   [1024, 1696)   slot 0 then slot 1, each: N = 4 (16 B), N = 8 (64 B), N = 16 (256 B), N x N int8 row-major
//   [1696, 2048)   int32 128 * (row sum), same order: 32 + 2 x (4 + 8 + 16) values -- the byte-plane offset fix
// The inverse kernels get the same table built from the TRANSPOSED matrices (their contractions run over the rows).
struct TileTab { alignas(16) uint8_t b[2048]; };
constexpr unsigned tile_tab_mat(unsigned slot, unsigned l) { return l == 3 ? 0u : 1024u + slot * 336u + (l == 0 ? 0u : (l == 1 ? 16u : 80u)); }
constexpr unsigned tile_tab_sum(unsigned slot, unsigned l) { return l == 3 ? 1696u : 1824u + slot * 112u + (l == 0 ? 0u : (l == 1 ? 16u : 48u)); }

inline uint32_t pack4(const int8_t *b)
{
    uint32_t w;
    std::memcpy(&w, b, 4);
    return w;
}

// Forward transform operand images.
//   pass 1:  D1[j][c]  = sum_n X[j][n] * g[kappa(c)][n]        data = A, const = B
//            slot (h,t) <-> n = 16h + t
//   pass 2:  D2[i][c]  = sum_j Y[j][kappa(i)] * g[c][j]        data = A, const = B
//            slot (h,t) <-> j = acc_row(t, h)   (the rows a pass-1 lane holds)
// The low byte plane is fed as (byte ^ 0x80) = byte - 128, so every sum misses
// 128 * sum_n g[k][n], which is 128*2048 for k = 0 and 0 otherwise.
inline void build_fwd_ops(DctOps &o)
{
    constexpr Table32 g = make_table32();
    for (int l = 0; l < 64; ++l) {
        const int c = l & 31, h = l >> 5;
        int8_t b1[16], b2[16], b1n[16];
        for (int t = 0; t < 16; ++t) {
            b1[t] = static_cast<int8_t>(g.v[kappa(c)][16 * h + t]);
            b2[t] = static_cast<int8_t>(g.v[c][acc_row(t, h)]);
            b1n[t] = static_cast<int8_t>(-b1[t]);                 // negated pass-1 operand: G*(a - b) = G*a + (-G)*b
        }
        for (int q = 0; q < 4; ++q) {
            o.lane[l].p1[q] = pack4(b1 + 4 * q);
            o.lane[l].p2[q] = pack4(b2 + 4 * q);
            o.lane[l].tr[q] = pack4(b1n + 4 * q);
        }
        o.lane[l].c1 = (1 << 3)  + (kappa(c) == 0 ? 128 * 2048 : 0);
        o.lane[l].c2 = (1 << 10) + (c == 0 ? 128 * 2048 : 0);
        o.lane[l].pad[0] = o.lane[l].pad[1] = 0;
        for (int r = 0; r < 16; ++r) o.c2r[l][r] = o.lane[l].c2;
    }
}

// Inverse transform operand images.
//   transpose: Dt[v][c] = sum_u Z[v][u] * [u == kappa(c)]       data = A, const = B
//              slot (h,t) <-> u = 16h + t
//   pass A:    Da[i][y] = sum_v Zt[v][kappa(i)] * g[v][y]       data = A, const = B
//              slot (h,t) <-> v = acc_row(t, h); per-lane constant (depends on y)
//   pass B:    Db[i][y] = sum_u g[u][kappa(i)] * T[u][y]        const = A, data = B
//              slot (h,t) <-> u = 16h + t; constant depends on x = 16h + r
// Offset fix: 128 * sum_k g[k][n] (a COLUMN sum here, non-zero for every n).
// natural_rows: the data operand of pass A holds rows v = 16h + t (a column read from LDS)
// instead of the accumulator order acc_row(t, h) left by the matrix-core transpose.
inline void build_inv_ops(DctOps &o, bool natural_rows = false)
{
    constexpr Table32 g = make_table32();
    int colsum[32] = {};
    for (int n = 0; n < 32; ++n)
        for (int k = 0; k < 32; ++k) colsum[n] += g.v[k][n];
    for (int l = 0; l < 64; ++l) {
        const int c = l & 31, h = l >> 5;
        int8_t ba[16], ab[16], id[16];
        for (int t = 0; t < 16; ++t) {
            ba[t] = static_cast<int8_t>(g.v[natural_rows ? 16 * h + t : acc_row(t, h)][c]);   // pass A: g[v][y=c]
            ab[t] = static_cast<int8_t>(g.v[16 * h + t][kappa(c)]);    // pass B: g[u][x=kappa(c)]
            id[t] = static_cast<int8_t>((16 * h + t) == kappa(c) ? 1 : 0);
        }
        for (int q = 0; q < 4; ++q) {
            o.lane[l].p1[q] = pack4(ba + 4 * q);
            o.lane[l].p2[q] = pack4(ab + 4 * q);
            o.lane[l].tr[q] = pack4(id + 4 * q);
        }
        o.lane[l].c1 = (1 << 6) + 128 * colsum[c];
        o.lane[l].c2 = 0;
        o.lane[l].pad[0] = o.lane[l].pad[1] = 0;
        for (int r = 0; r < 16; ++r) o.c2r[l][r] = (1 << 11) + 128 * colsum[16 * h + r];
    }
}

// ---------------------------------------------------------------------------
// The mixed transform set beyond DCT-II 32 (BASELINE configs[3]; parity UNPINNED
// upstream -- DESIGN.md section 10).
//  * DCT-II N = 4, 8, 16: rows 0, 32/N, 2*32/N, ... of g_t32 restricted to the
//    first N columns (the taps mkDct32Core re-uses, src/mkDct32.bsv:132-141).
//  * DST-VII N = 4, 8, 16: round(64*sqrt(N)*sqrt(4/(2N+1))*sin(pi(2k+1)(n+1)/(2N+1)));
//    for N = 4 this is the VVC table {29,55,74,84,...}.  Stored as literals so
//    that no floating point runs in the product.
// A group of (32/N)^2 small blocks is transformed as ONE 32x32 tile with the
// block-diagonal matrix diag(M_N, ..., M_N) on both sides, which yields every
// small block's own 2-D transform -- so the 32x32 MFMA pipeline is reused as is.
// ---------------------------------------------------------------------------
constexpr int8_t kDst7_4[16] = {
     29,  55,  74,  84,
     74,  74,   0, -74,
     84, -29, -74,  55,
     55, -84,  74, -29,
};
constexpr int8_t kDst7_8[64] = {
     16,  32,  46,  59,  70,  79,  84,  87,
     46,  79,  87,  70,  32, -16, -59, -84,
     70,  84,  32, -46, -87, -59,  16,  79,
     84,  46, -59, -79,  16,  87,  32, -70,
     87, -16, -84,  32,  79, -46, -70,  59,
     79, -70, -16,  84, -59, -32,  87, -46,
     59, -87,  70, -16, -46,  84, -79,  32,
     32, -59,  79, -87,  84, -70,  46, -16,
};
constexpr int8_t kDst7_16[256] = {
      8,  17,  25,  33,  41,  48,  55,  62,  67,  73,  77,  81,  84,  87,  88,  89,
     25,  48,  67,  81,  88,  88,  81,  67,  48,  25,   0, -25, -48, -67, -81, -88,
     41,  73,  88,  84,  62,  25, -17, -55, -81, -89, -77, -48,  -8,  33,  67,  87,
     55,  87,  81,  41, -17, -67, -89, -73, -25,  33,  77,  88,  62,   8, -48, -84,
     67,  88,  48, -25, -81, -81, -25,  48,  88,  67,   0, -67, -88, -48,  25,  81,
     77,  77,   0, -77, -77,   0,  77,  77,   0, -77, -77,   0,  77,  77,   0, -77,
     84,  55, -48, -87,  -8,  81,  62, -41, -88, -17,  77,  67, -33, -89, -25,  73,
     88,  25, -81, -48,  67,  67, -48, -81,  25,  88,   0, -88, -25,  81,  48, -67,
     89,  -8, -88,  17,  87, -25, -84,  33,  81, -41, -77,  48,  73, -55, -67,  62,
     87, -41, -67,  73,  33, -88,   8,  84, -48, -62,  77,  25, -89,  17,  81, -55,
     81, -67, -25,  88, -48, -48,  88, -25, -67,  81,   0, -81,  67,  25, -88,  48,
     73, -84,  25,  55, -89,  48,  33, -87,  67,   8, -77,  81, -17, -62,  88, -41,
     62, -89,  67,  -8, -55,  88, -73,  17,  48, -87,  77, -25, -41,  84, -81,  33,
     48, -81,  88, -67,  25,  25, -67,  88, -81,  48,   0, -48,  81, -88,  67, -25,
     33, -62,  81, -89,  84, -67,  41,  -8, -25,  55, -77,  88, -87,  73, -48,  17,
     17, -33,  48, -62,  73, -81,  87, -89,  88, -84,  77, -67,  55, -41,  25,  -8,
};

// Presets of slot 1 (xTransformUsePreset).  The H.266 DST-VII integers for N = 8 / 16 AS RECALLED from the VTM sources
// (DEFINE_DST7_P8_MATRIX / DEFINE_DST7_P16_MATRIX) -- UNVERIFIED OFFLINE: no copy of the standard or of VTM exists in the build
// environment.  They have the closed form's sign / index pattern with hand-tuned magnitudes, so they are stated as the
// magnitude substitution closed form -> recalled; N = 4 needs none (the closed form IS the standard's {29, 55, 74, 84} table).
constexpr int8_t kVtmDst7Mag8[2][8] = {{16, 32, 46, 59, 70, 79, 84, 87}, {17, 32, 46, 60, 71, 78, 85, 86}};
constexpr int8_t kVtmDst7Mag16[2][16] = {{8, 17, 25, 33, 41, 48, 55, 62, 67, 73, 77, 81, 84, 87, 88, 89},
                                         {8, 17, 25, 33, 40, 48, 55, 62, 68, 73, 77, 81, 85, 87, 88, 88}};

// m[k*n + c] = the recalled VTM DST-VII (dct8 = false) or the DCT-VIII derived from it, T8[k][c] = (-1)^k T7[k][n-1-c]
// (H.266's third MTS kernel is the flipped, sign-alternated DST-VII); n in {4, 8, 16}
inline void vtm_slot1_matrix(int n, bool dct8, int8_t *m)
{
    const int8_t *closed = n == 4 ? kDst7_4 : (n == 8 ? kDst7_8 : kDst7_16);
    int8_t t7[256];
    for (int i = 0; i < n * n; ++i) {
        int v = closed[i];
        const int mag = v < 0 ? -v : v;
        if (n == 8) { for (int j = 0; j < 8; ++j) if (kVtmDst7Mag8[0][j] == mag) { v = v < 0 ? -kVtmDst7Mag8[1][j] : kVtmDst7Mag8[1][j]; break; } }
        if (n == 16) { for (int j = 0; j < 16; ++j) if (kVtmDst7Mag16[0][j] == mag) { v = v < 0 ? -kVtmDst7Mag16[1][j] : kVtmDst7Mag16[1][j]; break; } }
        t7[i] = (int8_t)v;
    }
    for (int k = 0; k < n; ++k)
        for (int c = 0; c < n; ++c) m[k * n + c] = dct8 ? (int8_t)((k & 1) ? -t7[k * n + (n - 1 - c)] : t7[k * n + (n - 1 - c)]) : t7[k * n + c];
}

enum TransformType { kTrDct2 = 0, kTrDst7 = 1 };

struct Matrix32 { int8_t v[32][32]; };

// block-diagonal 32x32 matrix of the N-point transform `type`; N = 32 is g_t32 itself
inline Matrix32 make_transform_matrix(int type, int n)
{
    constexpr Table32 g = make_table32();
    Matrix32 m{};
    for (int k = 0; k < 32; ++k)
        for (int c = 0; c < 32; ++c) {
            int v = 0;
            if (k / n == c / n) {
                const int kk = k % n, cc = c % n;
                if (type == kTrDct2) v = g.v[kk * (32 / n)][cc];
                else v = n == 4 ? kDst7_4[kk * 4 + cc] : (n == 8 ? kDst7_8[kk * 8 + cc] : kDst7_16[kk * 16 + cc]);
            }
            m.v[k][c] = static_cast<int8_t>(v);
        }
    return m;
}

// Forward operand images for an arbitrary 32x32 int8 matrix and shift pair
// (same construction as build_fwd_ops; the byte-plane offset fix uses the row sums).
// m1 = matrix of pass 1 (along rows: the horizontal transform), m2 = matrix of pass 2 (vertical).
inline void build_fwd_ops_general(DctOps &o, const Matrix32 &m1, const Matrix32 &m2, int shift1, int shift2)
{
    int rowsum1[32] = {}, rowsum2[32] = {};
    for (int k = 0; k < 32; ++k)
        for (int c = 0; c < 32; ++c) { rowsum1[k] += m1.v[k][c]; rowsum2[k] += m2.v[k][c]; }
    for (int l = 0; l < 64; ++l) {
        const int c = l & 31, h = l >> 5;
        int8_t b1[16], b2[16];
        for (int t = 0; t < 16; ++t) {
            b1[t] = m1.v[kappa(c)][16 * h + t];
            b2[t] = m2.v[c][acc_row(t, h)];
        }
        for (int q = 0; q < 4; ++q) {
            o.lane[l].p1[q] = pack4(b1 + 4 * q);
            o.lane[l].p2[q] = pack4(b2 + 4 * q);
            o.lane[l].tr[q] = 0;
        }
        o.lane[l].c1 = (1 << (shift1 - 1)) + 128 * rowsum1[kappa(c)];
        o.lane[l].c2 = (1 << (shift2 - 1)) + 128 * rowsum2[c];
        o.lane[l].pad[0] = o.lane[l].pad[1] = 0;
        for (int r = 0; r < 16; ++r) o.c2r[l][r] = o.lane[l].c2;
    }
}

// Inverse operand images for an arbitrary 32x32 int8 matrix, data operand of pass A in natural row
// order (the staged kernel reads columns out of LDS); same construction as build_inv_ops(o, true).
// ma = matrix of pass A (the vertical inverse, columns first), mb = matrix of pass B (horizontal).
inline void build_inv_ops_general(DctOps &o, const Matrix32 &ma, const Matrix32 &mb)
{
    int colsum_a[32] = {}, colsum_b[32] = {};
    for (int c = 0; c < 32; ++c)
        for (int k = 0; k < 32; ++k) { colsum_a[c] += ma.v[k][c]; colsum_b[c] += mb.v[k][c]; }
    for (int l = 0; l < 64; ++l) {
        const int c = l & 31, h = l >> 5;
        int8_t ba[16], ab[16];
        for (int t = 0; t < 16; ++t) {
            ba[t] = ma.v[16 * h + t][c];             // pass A: M[v][y = c]
            ab[t] = mb.v[16 * h + t][kappa(c)];      // pass B: M[u][x = kappa(c)]
        }
        for (int q = 0; q < 4; ++q) {
            o.lane[l].p1[q] = pack4(ba + 4 * q);
            o.lane[l].p2[q] = pack4(ab + 4 * q);
            o.lane[l].tr[q] = 0;
        }
        o.lane[l].c1 = (1 << 6) + 128 * colsum_a[c];
        o.lane[l].c2 = 0;
        o.lane[l].pad[0] = o.lane[l].pad[1] = 0;
        for (int r = 0; r < 16; ++r) o.c2r[l][r] = (1 << 11) + 128 * colsum_b[16 * h + r];
    }
}

// slot_mat[slot][l]: the N x N matrices of the two 1-D transform slots, N = 4 << l (x266hip_abi.hip)
inline void build_tile_tab(TileTab &t, const int8_t (*slot_mat)[3][256], bool transposed)
{
    constexpr Table32 g = make_table32();
    std::memset(t.b, 0, sizeof t.b);
    auto put = [&](unsigned mat_off, unsigned sum_off, int n, auto at) {
        for (int k = 0; k < n; ++k) {
            int sum = 0;
            for (int c = 0; c < n; ++c) {
                const int v = transposed ? at(c, k) : at(k, c);
                t.b[mat_off + (unsigned)(k * n + c)] = static_cast<uint8_t>(static_cast<int8_t>(v));
                sum += v;
            }
            const int32_t fix = 128 * sum;
            std::memcpy(&t.b[sum_off + 4u * (unsigned)k], &fix, 4);
        }
    };
    put(tile_tab_mat(0, 3), tile_tab_sum(0, 3), 32, [&](int k, int c) { return (int)g.v[k][c]; });
    for (int slot = 0; slot < 2; ++slot)
        for (int l = 0; l < 3; ++l) {
            const int n = 4 << l;
            const int8_t *m = slot_mat[slot][l];
            put(tile_tab_mat((unsigned)slot, (unsigned)l), tile_tab_sum((unsigned)slot, (unsigned)l), n, [&](int k, int c) { return (int)m[k * n + c]; });
        }
}

// transform type codes of the API: 0 DCT-II both ways, 1 DST-VII both ways, 2 horizontal DST-VII + vertical DCT-II,
// 3 horizontal DCT-II + vertical DST-VII (the implicit-MTS style combinations)
constexpr int transform_htype(int type) { return (type == 1 || type == 2) ? kTrDst7 : kTrDct2; }
constexpr int transform_vtype(int type) { return (type == 1 || type == 3) ? kTrDst7 : kTrDct2; }
constexpr int transform_shift1(int n) { return (n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5) - 1; }   // log2N - 1  (8-bit video)
constexpr int transform_shift2(int n) { return (n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5) + 6; }   // log2N + 6

}  // namespace x266
