// x266hip_bdpi.cpp -- the BDPI drop-in surface of libx266hip.so plus the
// host-only packing utilities and the exported coefficient table.
//
// Upstream defines these six C symbols in src_tb/dct32.c:178-246 and
// src_tb/satd.c:124-152; the Bluespec testbenches import them
// (src/mkDct32.bsv:409-411, src/mkSatd.bsv:204-206) and call them in the order
// genNew -> N x getDiff -> M x getDct/getSatd (mkDct32.bsv:430-470,
// mkSatd.bsv:215-252).  The shims keep upstream's contract exactly -- file-
// static block state, non-reentrant, stimulus from libc rand() drawn in the
// same order, same word packing -- but the golden values served to the DUT are
// computed by the HIP kernels through the host-pointer batch API.
// There is no CPU fallback: if no gfx950 device can be opened the shims print
// a diagnostic and abort(), as SURVEY.md section 8(b) requires.
#define X266HIP_DEFINING_TABLE
#include "../../include/x266hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "x266_tables.hpp"

extern "C" {

// const int16_t g_t32[32][32] (src_tb/dct32.c:30): same name, size and layout,
// generated at compile time (x266_tables.hpp) instead of stored as a literal.
extern const x266::Table32 g_t32;
const x266::Table32 g_t32 = x266::make_table32();

void xDct32PackDiffRows(const int16_t *mat, int first_row, unsigned int res[32])
{
    const int16_t *row = mat + first_row * 32;
    for (int w = 0; w < 32; ++w) {                        // 16 words of row r, then 16 of row r+1
        const unsigned lo = (uint16_t)row[2 * w], hi = (uint16_t)row[2 * w + 1];
        res[w] = (hi << 16) + lo;
    }
}

int xTransformMatrix(int type, int size, int16_t *m)
{
    if (!m || (type != 0 && type != 1)) return X266HIP_EINVAL;
    if (!(size == 4 || size == 8 || size == 16 || (size == 32 && type == 0))) return X266HIP_EINVAL;
    const x266::Matrix32 t = x266::make_transform_matrix(type, size);       // block-diagonal 32x32: the first block is the matrix
    for (int k = 0; k < size; ++k)
        for (int n = 0; n < size; ++n) m[k * size + n] = t.v[k][n];
    return X266HIP_OK;
}

uint64_t xDct32PackDctWord(const int16_t *dct, int idx)
{
    const int col = idx >> 5, row = idx & 31;             // column-major walk, 4 rows per word
    uint64_t word = 0;
    for (int i = 3; i >= 0; --i) word = (word << 16) | (uint16_t)dct[(row + i) * 32 + col];
    return word;
}

}  // extern "C"

namespace {

x266hip_ctx *g_ctx = nullptr;

// The stimulus comes from libc rand(), whose state is process-global.  Anything
// the HIP runtime draws from it (it does, while initialising) would shift the
// stimulus stream away from upstream's.  Every trip into the runtime therefore
// runs on a private random state; the caller's state is put back untouched.
class RandStateGuard {
public:
    RandStateGuard() { prev_ = initstate(0x266u, scratch_, sizeof scratch_); }
    ~RandStateGuard() { if (prev_) setstate(prev_); }
    RandStateGuard(const RandStateGuard &) = delete;
    RandStateGuard &operator=(const RandStateGuard &) = delete;
private:
    char scratch_[256];
    char *prev_ = nullptr;
};

void *g_sad_buf = nullptr;                  // x266_sad's device scratch, released with the context

void release_ctx()
{
    if (g_sad_buf) (void)xHipFree(g_ctx, g_sad_buf);
    g_sad_buf = nullptr;
    xHipCodecFree(g_ctx);
    g_ctx = nullptr;
}

x266hip_ctx *bdpi_ctx()
{
    if (g_ctx) return g_ctx;
    int dev = 0;
    if (const char *e = std::getenv("X266HIP_DEVICE")) dev = std::atoi(e);
    const int rc = xHipCodecInit(&g_ctx, dev);
    if (rc != X266HIP_OK || !g_ctx) {
        std::fprintf(stderr, "x266hip BDPI: cannot open gfx950 device %d (rc=%d); no CPU fallback exists\n", dev, rc);
        std::abort();
    }
    std::atexit(release_ctx);
    return g_ctx;
}

[[noreturn]] void die(const char *what, int rc)
{
    std::fprintf(stderr, "x266hip BDPI: %s failed (rc=%d): %s\n", what, rc, xHipLastError(g_ctx));
    std::abort();
}

// block state, as upstream keeps it (src_tb/dct32.c:173-176, satd.c:120-122)
int16_t s_dct_in[32 * 32], s_dct_out[32 * 32];
int s_next_diff_row = 0, s_next_dct_idx = 0;
int16_t s_satd_in[8 * 8];
int s_satd_row = 0;
unsigned int s_satd_val = 0;

// one stimulus sample, drawn like upstream: a = rand()&0xFF, then b = rand()&0xFF
inline int16_t draw_residual()
{
    const int a = std::rand() & 0xFF;
    const int b = std::rand() & 0xFF;
    return (int16_t)(a - b);
}

}  // namespace

extern "C" {

void dct32_genNew(void)
{
    for (int i = 0; i < 32 * 32; ++i) s_dct_in[i] = draw_residual();
    {
        RandStateGuard guard;
        const int rc = xDct32FwdBatch(bdpi_ctx(), s_dct_in, s_dct_out, 1);
        if (rc != X266HIP_OK) die("xDct32FwdBatch", rc);
    }
    s_next_diff_row = 0;
    s_next_dct_idx = 0;
}

void dct32_getDiff(unsigned int res[])
{
    xDct32PackDiffRows(s_dct_in, s_next_diff_row, res);
    s_next_diff_row += 2;
}

unsigned long long dct32_getDct(void)
{
    const uint64_t w = xDct32PackDctWord(s_dct_out, s_next_dct_idx);
    s_next_dct_idx += 4;
    return w;
}

void satd8x8_genNew(void)
{
    for (int i = 0; i < 8 * 8; ++i) s_satd_in[i] = draw_residual();
    uint32_t v = 0;
    {
        RandStateGuard guard;
        const int rc = xSatd8x8Batch(bdpi_ctx(), s_satd_in, &v, 1);
        if (rc != X266HIP_OK) die("xSatd8x8Batch", rc);
    }
    s_satd_val = v;
    s_satd_row = 0;
}

void satd8x8_getDiff(unsigned int res[])
{
    std::memcpy(res, s_satd_in + 8 * s_satd_row, 8 * sizeof(int16_t));
    ++s_satd_row;
}

unsigned int satd8x8_getSatd(void) { return s_satd_val; }

// Per-call twin of sad() (riscv/programs/benchmarks/sad/sad.c:28-39: same arguments, same result), computed on
// the GPU through the lazily created context of the shims above.  n must be 4, 8, 16, 32 or 64.
int x266_sad(const unsigned char *input_data1, const unsigned char *input_data2, size_t n)
{
    if (!input_data1 || !input_data2 || !(n == 4 || n == 8 || n == 16 || n == 32 || n == 64)) return -1;
    // Like the BDPI shims: stateful, single-threaded by contract.  The guard comes first -- opening the
    // context initialises the HIP runtime, which draws from rand().
    RandStateGuard keep_rand_sequence;
    x266hip_ctx *ctx = bdpi_ctx();
    if (!g_sad_buf) {                                              // two 64x64 blocks + the result; freed by release_ctx at exit
        const int rc = xHipMalloc(ctx, &g_sad_buf, 2 * 4096 + 16);
        if (rc != X266HIP_OK) die("xHipMalloc", rc);
    }
    unsigned char *a = static_cast<unsigned char *>(g_sad_buf), *b = a + 4096;
    uint32_t *d_out = reinterpret_cast<uint32_t *>(a + 8192);
    uint32_t out = 0;
    int rc = xHipMemcpyH2D(ctx, a, input_data1, n * n);
    if (rc == X266HIP_OK) rc = xHipMemcpyH2D(ctx, b, input_data2, n * n);
    if (rc == X266HIP_OK) rc = xSadBatchDev(ctx, (int)n, a, b, d_out, 1, nullptr);
    if (rc == X266HIP_OK) rc = xHipMemcpyD2H(ctx, &out, d_out, sizeof(out));
    if (rc != X266HIP_OK) die("x266_sad", rc);
    return (int)out;
}

}  // extern "C"
