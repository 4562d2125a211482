// CPU check of the mixed-class tile kernel's host-side table (x266_tables.hpp: TileTab, build_tile_tab) against the
// per-class operand images (build_fwd_ops_general / build_inv_ops_general): the lane-level image construction the kernel
// performs from the compact table in LDS (transform_kernels.hip: image_row_segment, image_acc_rows, the sums) is restated
// here in plain C++ and must reproduce every lane's p1 / p2 / c1 / c2 / c2r of every class, for the built-in matrices and for
// random int8 ones.  Test infrastructure; built and run by tests/test_tile_table.py (g++, no GPU, no HIP).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "x266_tables.hpp"

using namespace x266;

static void row_segment(const uint8_t *m, int logn, unsigned idx, unsigned h, uint8_t out[16])
{
    const unsigned n = 1u << logn, rr = idx & (n - 1), b = idx >> logn;
    std::memset(out, 0, 16);
    if (logn == 5) std::memcpy(out, m + rr * 32 + 16 * h, 16);
    else if (logn == 4) { if (b == h) std::memcpy(out, m + rr * 16, 16); }
    else if (logn == 3) { for (unsigned q = 0; q < 4; ++q) if (b == 2 * h + (q >> 1)) std::memcpy(out + 4 * q, m + rr * 8 + 4 * (q & 1), 4); }
    else { for (unsigned q = 0; q < 4; ++q) if (b == 4 * h + q) std::memcpy(out + 4 * q, m + rr * 4, 4); }
}

static void acc_rows(const uint8_t *m, int logn, unsigned c, unsigned h, uint8_t out[16])
{
    const unsigned n = 1u << logn, rr = c & (n - 1), b = c >> logn;
    std::memset(out, 0, 16);
    for (unsigned q = 0; q < 4; ++q) {
        if (logn == 5) std::memcpy(out + 4 * q, m + rr * 32 + 4 * h + 8 * q, 4);
        else if (logn == 4) { if ((q >> 1) == b) std::memcpy(out + 4 * q, m + rr * 16 + 4 * h + 8 * (q & 1), 4); }
        else if (logn == 3) { if (q == b) std::memcpy(out + 4 * q, m + rr * 8 + 4 * h, 4); }
        else { if (b == h + 2 * q) std::memcpy(out + 4 * q, m + rr * 4, 4); }
    }
}

static int32_t sum_at(const TileTab &t, unsigned off, unsigned i)
{
    int32_t v;
    std::memcpy(&v, &t.b[off + 4 * i], 4);
    return v;
}

static Matrix32 block_diag(const int8_t *m, int n)
{
    Matrix32 r{};
    for (int k = 0; k < 32; ++k)
        for (int c = 0; c < 32; ++c) r.v[k][c] = k / n == c / n ? m[(k % n) * n + (c % n)] : (int8_t)0;
    return r;
}

static int check(const int8_t (*slot_mat)[3][256], const char *what)
{
    TileTab tf, ti;
    build_tile_tab(tf, slot_mat, false);
    build_tile_tab(ti, slot_mat, true);
    constexpr Table32 g = make_table32();
    int8_t m32[1024];
    for (int k = 0; k < 32; ++k)
        for (int c = 0; c < 32; ++c) m32[k * 32 + c] = (int8_t)g.v[k][c];
    int bad = 0;
    DctOps *ops = new DctOps;
    for (int type = 0; type < 4; ++type)
        for (int l = 0; l < 4; ++l) {
            const int n = 4 << l, logn = l + 2;
            const unsigned hs = transform_htype(type) == kTrDst7, vs = transform_vtype(type) == kTrDst7;
            const Matrix32 mh = block_diag(l == 3 ? m32 : slot_mat[hs][l], n), mv = block_diag(l == 3 ? m32 : slot_mat[vs][l], n);
            const unsigned moh = tile_tab_mat(hs, l), mov = tile_tab_mat(vs, l), soh = tile_tab_sum(hs, l), sov = tile_tab_sum(vs, l);
            // forward
            build_fwd_ops_general(*ops, mh, mv, transform_shift1(n), transform_shift2(n));
            for (unsigned lane = 0; lane < 64; ++lane) {
                const unsigned c = lane & 31, h = lane >> 5, kc = (unsigned)kappa((int)c);
                uint8_t p1[16], p2[16];
                row_segment(tf.b + moh, logn, kc, h, p1);
                acc_rows(tf.b + mov, logn, c, h, p2);
                const int32_t c1 = (1 << (transform_shift1(n) - 1)) + sum_at(tf, soh, kc & (n - 1));
                const int32_t c2 = (1 << (transform_shift2(n) - 1)) + sum_at(tf, sov, c & (n - 1));
                if (std::memcmp(p1, ops->lane[lane].p1, 16) || std::memcmp(p2, ops->lane[lane].p2, 16) || c1 != ops->lane[lane].c1 || c2 != ops->lane[lane].c2) {
                    if (bad++ < 5) std::fprintf(stderr, "%s: forward class (type %d, N %d) lane %u differs\n", what, type, n, lane);
                }
            }
            // inverse: pass A = vertical matrix, pass B = horizontal (build_inv_ops_general(ma = mv, mb = mh)), table = transposes
            build_inv_ops_general(*ops, mv, mh);
            for (unsigned lane = 0; lane < 64; ++lane) {
                const unsigned c = lane & 31, h = lane >> 5, kc = (unsigned)kappa((int)c);
                uint8_t p1[16], p2[16];
                row_segment(ti.b + mov, logn, c, h, p1);
                row_segment(ti.b + moh, logn, kc, h, p2);
                const int32_t c1 = (1 << 6) + sum_at(ti, sov, c & (n - 1));
                bool c2r_ok = true;
                for (unsigned r = 0; r < 16; ++r) c2r_ok = c2r_ok && ops->c2r[lane][r] == (1 << 11) + sum_at(ti, soh, (16 * h + r) & (n - 1));
                if (std::memcmp(p1, ops->lane[lane].p1, 16) || std::memcmp(p2, ops->lane[lane].p2, 16) || c1 != ops->lane[lane].c1 || !c2r_ok) {
                    if (bad++ < 5) std::fprintf(stderr, "%s: inverse class (type %d, N %d) lane %u differs\n", what, type, n, lane);
                }
            }
        }
    delete ops;
    return bad;
}

int main()
{
    static int8_t slot_mat[2][3][256];
    for (int slot = 0; slot < 2; ++slot)
        for (int l = 0; l < 3; ++l) {
            const int n = 4 << l;
            const Matrix32 d = make_transform_matrix(slot == 0 ? kTrDct2 : kTrDst7, n);
            for (int k = 0; k < n; ++k)
                for (int c = 0; c < n; ++c) slot_mat[slot][l][k * n + c] = d.v[k][c];
        }
    int bad = check(slot_mat, "built-in matrices");
    unsigned long long x = 0x266;
    for (int round = 0; round < 20; ++round) {
        for (int slot = 0; slot < 2; ++slot)
            for (int l = 0; l < 3; ++l)
                for (int i = 0; i < 256; ++i) {
                    x = x * 6364136223846793005ull + 1442695040888963407ull;
                    slot_mat[slot][l][i] = (int8_t)(x >> 56);
                }
        bad += check(slot_mat, "random int8 matrices");
    }
    if (bad) { std::fprintf(stderr, "%d lane images differ\n", bad); return 1; }
    std::printf("tile table: 16 classes x 64 lanes x (forward, inverse) x 21 matrix sets reproduce the per-class images\n");
    return 0;
}
