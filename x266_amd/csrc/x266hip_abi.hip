// x266hip_abi.hip -- host side of libx266hip.so: context, batch entry points,
// device-memory helpers and kernel timing behind the C ABI of include/x266hip.h.
//
// Conventions follow src/x266.cpp (xCodecInit/xCodecFree :494-524: context
// struct first, caller-owned buffers, int 0 / negative returns, no exceptions
// across the boundary).  There is deliberately NO CPU fallback: every compute
// entry point needs a gfx950 device and fails loudly without one.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cctype>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/x266hip.h"
#include "x266_device.hpp"
#include "x266_tables.hpp"

using namespace x266;

struct x266hip_ctx {
    int device = 0;
    hipDeviceProp_t prop{};
    DctOps *d_fwd = nullptr;
    DctOps *d_inv_lds = nullptr;                    // inverse operand images for the LDS-staged kernel (column reads)
    DctOps *d_inv_acc = nullptr;                    // the same with pass A's K-slots in accumulator-row order (fused kernel: the inverse is fed from the forward's registers)
    static constexpr int kTypes = 4;                // DCT-II, DST-VII, and the two mixed horizontal / vertical pairs
    DctOps *d_tr[kTypes][3] = {};                   // [type][log2N - 2], N = 4, 8, 16
    DctOps *d_tr_inv[kTypes][3] = {};
    TileTab *d_tile_fwd = nullptr, *d_tile_inv = nullptr;      // the set's 1-D matrices in compact form, as is / transposed (xTransformTilesDev)
    // the two 1-D transform slots of the set, N = 4, 8, 16: slot 0 = DCT-II sub-matrices of g_t32, slot 1 = closed-form
    // DST-VII unless the caller installed its own (xTransformSetMatrix); row k = basis function, N x N, row-major
    int8_t slot_mat[2][3][256] = {};
    int slot1_preset = X266_PRESET_CLOSED_FORM;     // -1 once the caller installed a matrix of its own in slot 1
    bool tr_tables_valid = true;                    // false after a matrix update whose device tables could be neither installed nor rolled back
    // Launch options (xHipSetOption): A/B knobs, results never depend on them.  Defaults = the measured optimum.
    int dct_variant = 0;                            // 0 = matrix-core kernel, 2 = VALU butterfly (the comparison variant north_star asks for)
    int satd_variant = 0;                           // 0 = by batch size (staged kernel below 3 Mi blocks, LDS-DMA kernel from there), 1 = staged, 2 = VALU butterfly, 3 = LDS-DMA
    int dct_blocks_per_wave = 1, dct_inv_blocks_per_wave = 2, dct_fwdinv_blocks_per_wave = 0;   // consecutive blocks one wave loops over (profiles/r01_launch_sweep.txt)
    int satd_groups_per_wave = 0, satd_wg_threads = 0, satd_lds_per_wave = 0;                   // 0 = the chosen SATD kernel's own default (satd_kernels.hip, launch_satd8x8)
    int adaptive_per_wave = 1;                      // shrink the per-wave run on small batches
    int dct_wg_threads = 0;                         // workgroup size of the DCT32 / transform-set kernels; 0 = the measured best: one-wave workgroups (profiles/r01_wg_occupancy.txt), four-wave ones for the fused forward + inverse kernel (profiles/r05_fused_variants.txt)
    int tile_tiles_per_wave = 0;                    // mixed-class tile kernel: consecutive tiles per wave (0 = 2: the wave's table copy serves two tiles)
    int autotune = 0;                               // 1: the first large batch of a kernel family times its box-dependent launch shapes on the caller's buffers and keeps the fastest (tune_family below)
    int me_tile_rows = 0;                           // block rows per ME tile: 0 = by frame size (SATD search: 8, 4 or 2; SAD search: 2), else 2, 4, 8 (8: SATD search only; 1 is served by 2)
    // fixed launch shapes (options in rounds 1-3; their sweeps are frozen in profiles/r01_*.txt, r03_tiles_one_launch.txt)
    static constexpr int kDctLdsPerWave = 8192;     // 2 KiB used: at most 20 resident waves per CU
    static constexpr int kDctInvLdsPerWave = 10240; // the DCT32 inverse: 16
    static constexpr int kFwdInvLdsPerWave = 12288; // fused forward + inverse: 6 KiB used (two DMA slots + the converter): 12 resident waves per CU
    static constexpr int kTileLdsPerWave = 8192;    // table + two tile slots = 6 KiB used
    static constexpr int kIntraRounds = 4;          // intra prediction: rounds of seven predictions per wave
    // Internal device scratch, ONE BUFFER PER STREAM AND KIND, so that calls enqueued on different streams never share it:
    // kind 0 = motion search (128 B per 8x8 block of the current frame).  Each table is bounded (kMeScratchMax streams, least recently used evicted after waiting for its
    // last use); a buffer that was handed out during a stream capture may be referenced by a recorded graph and is kept
    // until the context is freed (outgrown ones are retired, not released).
    struct MeScratch {
        hipStream_t stream;
        uint32_t *p;
        size_t bytes;
        hipEvent_t last_use;
        unsigned long long stamp;
        bool pinned;
        bool capturing_now;
    };
    // "autotune": per kernel family the chosen candidate (-1 = not tuned yet) and what every candidate measured
    enum { kTuneFwdInv, kTuneRecon, kTuneSatd, kTuneSad8, kTuneSad16, kTuneSad32, kTuneSad64, kTuneFamilies };
    static constexpr int kTuneMaxCands = 8;
    struct Tuned { int choice = -1; int n = 0; float ms[kTuneMaxCands] = {}; };
    Tuned tuned[kTuneFamilies];
    hipEvent_t tune_ev[2] = {};
    static constexpr size_t kMeScratchMax = 8;
    static constexpr int kScratchKinds = 1;
    std::vector<MeScratch> scratch[kScratchKinds];
    std::vector<void *> me_retired;
    unsigned long long me_stamp = 0;
    // host-pointer staging (lazily allocated)
    static constexpr int kSlots = 3;                // chunk i+1 uploading and chunk i-1 downloading while chunk i is transformed
    void *d_stage_in[kSlots] = {};
    void *d_stage_out[kSlots] = {};
    size_t stage_in_bytes = 0, stage_out_bytes = 0;
    // one stream per ENGINE, not per slot: every upload on [0], every kernel on [1], every download on [2], ordered by the
    // slots' events.  With a stream per slot (rounds 1-3) uploads and downloads of different chunks did not overlap on this
    // runtime: 28-30 GB/s each way from pinned memory against 43 this way (profiles/r04_hostpipe.txt; the link gives 48.5 both ways)
    hipStream_t stage_stream[3] = {};
    // small host-pointer calls (the BDPI shims: one block per call): two page-locked, device-visible 64 KiB buffers the kernel reads and writes in place
    static constexpr size_t kSmallCallBytes = (size_t)64 << 10;
    void *h_small_in = nullptr, *h_small_out = nullptr;
    hipStream_t small_stream = nullptr;
    hipEvent_t stage_up[kSlots] = {}, stage_done[kSlots] = {}, stage_down[kSlots] = {};
    std::string err;
};

namespace {

int fail(x266hip_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess)
{
    if (ctx) {
        ctx->err = what;
        if (e != hipSuccess) {
            ctx->err += ": ";
            ctx->err += hipGetErrorString(e);
        }
    }
    return code;
}

#define X_HIP(ctx, call)                                                     \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return fail((ctx), X266HIP_EDEVICE, #call, e_); \
    } while (0)

// Every entry point runs on the context's device and puts the caller's current device back on return
// (a host that also drives other GPUs, or torch with another current device, is not retargeted).
struct DeviceScope {
    int prev = -1;
    hipError_t status;
    explicit DeviceScope(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        status = prev == dev ? hipSuccess : hipSetDevice(dev);
        if (prev == dev) prev = -1;                                    // nothing to restore
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

#define X_DEV(ctx)                                                            \
    DeviceScope dev_scope_((ctx)->device);                                    \
    if (dev_scope_.status != hipSuccess) return fail((ctx), X266HIP_EDEVICE, "hipSetDevice", dev_scope_.status)

LaunchCfg cfg_for(const x266hip_ctx *ctx, int op)
{
    LaunchCfg c;
    c.cu_count = ctx->prop.multiProcessorCount;
    c.adaptive = ctx->adaptive_per_wave;
    if (op == 2) {
        c.units_per_wave = ctx->satd_groups_per_wave;
        c.wg_threads = ctx->satd_wg_threads;
        c.lds_bytes_per_wave = ctx->satd_lds_per_wave;
        c.shape = ctx->satd_variant == 1 || ctx->satd_variant == 3 ? ctx->satd_variant : 0;
    } else {
        c.units_per_wave = op == 1 ? ctx->dct_inv_blocks_per_wave : ctx->dct_blocks_per_wave;
        c.wg_threads = ctx->dct_wg_threads ? ctx->dct_wg_threads : 64;
        c.lds_bytes_per_wave = x266hip_ctx::kDctLdsPerWave;

        c.shape = 0;
    }
    return c;
}

int launch_op(x266hip_ctx *ctx, int op, const void *d_in, void *d_out, size_t n, hipStream_t s)
{
    hipError_t e;
    switch (op) {
    case 0:
        if (ctx->dct_variant == 2) e = launch_dct32_butterfly((const int16_t *)d_in, (int16_t *)d_out, n, s);   // VALU comparison variant
        else e = launch_dct32(false, (const int16_t *)d_in, (int16_t *)d_out, n, ctx->d_fwd, cfg_for(ctx, 0), s);
        break;
    case 1: {
        // the inverse does more arithmetic per tile than the forward: 10 KiB charged per wave = 16 resident waves per CU instead of 20 (paired sweep in one process: -1.8 %;
        // the forward, the small transforms and their inverses keep 8 KiB, tools/probes/gpu_dct_family_shapes.py)
        LaunchCfg cfg = cfg_for(ctx, 1);
        cfg.lds_bytes_per_wave = x266hip_ctx::kDctInvLdsPerWave;
        e = launch_dct32(true, (const int16_t *)d_in, (int16_t *)d_out, n, ctx->d_inv_lds, cfg, s);
        break;
    }
    case 2:
        if (ctx->satd_variant == 2) e = launch_satd8x8_butterfly((const int16_t *)d_in, (uint32_t *)d_out, n, cfg_for(ctx, 2), s);   // VALU comparison variant
        else e = launch_satd8x8((const int16_t *)d_in, (uint32_t *)d_out, n, cfg_for(ctx, 2), s);
        break;
    case 3: case 4: case 5: case 6: e = launch_mem_ceiling(op - 3, d_in, d_out, n * 2048, s); break;      // xHipTimeKernel only
    default: return fail(ctx, X266HIP_EINVAL, "unknown op");
    }
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "kernel launch", e);
    return X266HIP_OK;
}

// ---- opt-in launch-shape autotuning ("autotune" = 1; off by default; results never depend on it) ---------------------------
// Boxes of the pool differ in which launch shape a few kernels run fastest in (profiles/r05_fused_variants.txt: the fused forward +
// inverse kernel's best shape is a different one on four of six boxes, up to 5 % apart).  With the option on, the FIRST call of a
// family whose batch is large enough to time (the caller says so by `big`) runs every candidate on the caller's own buffers and
// stream -- one warm-up and three timed launches each, HIP events on that stream, so this one call is synchronous -- and the
// context keeps the fastest; the default shape (candidate 0) stays unless another one beats it by more than 2 %.  Every candidate
// writes the same bytes (tests/test_gpu_autotune.py compares option off / on per family; tests/test_gpu_waits.py every fused candidate against the
// wait-for-everything build), so the extra launches only rewrite the outputs.
// Not while the stream is being captured, and not for calls whose buffers overlap (the caller checks): those use the default.
template <class Launch>
int tune_family(x266hip_ctx *ctx, int family, int n_cands, bool big, hipStream_t stream, Launch &&launch)
{
    x266hip_ctx::Tuned &t = ctx->tuned[family];
    if (!ctx->autotune) return 0;
    if (t.choice >= 0) return t.choice;
    if (!big) return 0;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream) (void)hipStreamIsCapturing(stream, &cap);
    (void)hipGetLastError();
    if (cap != hipStreamCaptureStatusNone) return 0;
    for (hipEvent_t &e : ctx->tune_ev)
        if (!e && hipEventCreate(&e) != hipSuccess) { e = nullptr; (void)hipGetLastError(); return 0; }
    if (n_cands > x266hip_ctx::kTuneMaxCands) n_cands = x266hip_ctx::kTuneMaxCands;
    t.n = n_cands;
    int best = 0;
    for (int c = 0; c < n_cands; ++c) {
        t.ms[c] = -1.f;
        if (launch(c) != hipSuccess) { (void)hipGetLastError(); continue; }
        bool ok = hipEventRecord(ctx->tune_ev[0], stream) == hipSuccess;
        for (int r = 0; ok && r < 3; ++r) ok = launch(c) == hipSuccess;
        float ms = 0.f;
        ok = ok && hipEventRecord(ctx->tune_ev[1], stream) == hipSuccess && hipEventSynchronize(ctx->tune_ev[1]) == hipSuccess &&
             hipEventElapsedTime(&ms, ctx->tune_ev[0], ctx->tune_ev[1]) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); continue; }
        t.ms[c] = ms / 3.f;
        if (c && t.ms[c] > 0.f && (t.ms[best] <= 0.f || t.ms[c] < t.ms[best])) best = c;
    }
    if (best && t.ms[0] > 0.f && t.ms[best] > 0.98f * t.ms[0]) best = 0;   // within 2 % of the default (run-to-run noise is ~1 %): keep the default
    t.choice = best;
    return best;
}

struct ShapeCand { int units_per_wave, wg_threads, lds_bytes_per_wave, shape; };
// fused forward + inverse with both outputs: the default first, then the deep long-lived shapes of profiles/r05_fused_variants.txt
const ShapeCand kFwdInvCands[] = {{2, 256, 12288, 2}, {16, 256, 16384, 3}, {24, 256, 16384, 3}, {16, 128, 20480, 3}, {12, 256, 16384, 3}, {4, 128, 12288, 2},
                                  {16, 256, 16384, 4}, {4, 64, 16384, 3}};
const ShapeCand kReconCands[] = {{8, 256, 12288, 2}, {4, 256, 12288, 2}, {16, 256, 16384, 3}, {8, 128, 12288, 2}, {6, 256, 12288, 2}};
// SATD batch, LDS-DMA kernel (groups per wave, workgroup, LDS per wave)
const ShapeCand kSatdCands[] = {{4, 256, 16384, 3}, {2, 256, 16384, 3}, {8, 256, 16384, 3}, {4, 128, 12288, 3}, {6, 256, 12288, 3}, {3, 256, 12288, 3}};
// SAD batch: (-, waves per workgroup x 64, LDS per WORKGROUP)
const ShapeCand kSadCands[] = {{0, 256, 32768, 0}, {0, 128, 16384, 0}, {0, 64, 8192, 0}, {0, 256, 24576, 0}, {0, 256, 40960, 0}};

static bool ranges_overlap(const void *a, size_t na, const void *b, size_t nb)
{
    const uintptr_t x = (uintptr_t)a, y = (uintptr_t)b;
    return a && b && x < y + nb && y < x + na;
}

bool bad_ptrs(const void *a, const void *b, size_t n)
{
    if (n == 0) return false;
    if (!a || !b) return true;
    return (((uintptr_t)a | (uintptr_t)b) & 15u) != 0;
}

// Scratch of kind `kind` for `stream`, at least `need` bytes.  Allocation happens here, i.e. on the first call of a stream or
// size -- or ahead of time through xHipMeScratchReserve, which is what a host does before a stream
// capture or a real-time loop (hipMalloc is illegal under capture and synchronises).
int scratch_for(x266hip_ctx *ctx, int kind, hipStream_t stream, size_t need, x266hip_ctx::MeScratch **out)
{
    std::vector<x266hip_ctx::MeScratch> &table = ctx->scratch[kind];
    x266hip_ctx::MeScratch *slot = nullptr;
    for (x266hip_ctx::MeScratch &m : table)
        if (m.stream == stream) slot = &m;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream) (void)hipStreamIsCapturing(stream, &cap);
    (void)hipGetLastError();
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (!slot) {
        if (table.size() >= x266hip_ctx::kMeScratchMax && !capturing) {          // evict the least recently used stream's buffer
            size_t victim = table.size();
            for (size_t i = 0; i < table.size(); ++i)
                if (!table[i].pinned && (victim == table.size() || table[i].stamp < table[victim].stamp)) victim = i;
            if (victim < table.size()) {
                x266hip_ctx::MeScratch &v = table[victim];
                if (v.last_use) { (void)hipEventSynchronize(v.last_use); (void)hipEventDestroy(v.last_use); }
                if (v.p) (void)hipFree(v.p);
                table.erase(table.begin() + (long)victim);
            }
        }
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
        table.push_back({stream, nullptr, 0, ev, 0, false, false});
        slot = &table.back();
    }
    if (need > slot->bytes) {
        if (capturing) return fail(ctx, X266HIP_EINVAL, "internal scratch cannot be allocated under stream capture: reserve it (or run the call once) beforehand");
        void *fresh = nullptr;
        if (hipMalloc(&fresh, need) != hipSuccess) return fail(ctx, X266HIP_ENOMEM, "internal scratch");
        if (slot->p) {
            if (slot->pinned) ctx->me_retired.push_back(slot->p);        // a recorded graph may still use it
            else { if (slot->last_use) (void)hipEventSynchronize(slot->last_use); (void)hipFree(slot->p); }
        }
        slot->p = (uint32_t *)fresh;
        slot->bytes = need;
        slot->pinned = false;
    }
    if (capturing) slot->pinned = true;
    slot->capturing_now = capturing;
    slot->stamp = ++ctx->me_stamp;
    *out = slot;
    return X266HIP_OK;
}

// motion search: tile-major coefficient table of whole search tiles, partial edge tiles padded (tile heights 1, 2, 4, 8 all fit)
int me_scratch_for(x266hip_ctx *ctx, hipStream_t stream, int width, int height, x266hip_ctx::MeScratch **out)
{
    return scratch_for(ctx, 0, stream, (size_t)((width / 8 + 7) / 8) * 8 * (size_t)((height / 8 + 7) / 8) * 8 * 128, out);
}

// built-in matrix of a 1-D transform slot: 0 = DCT-II (rows 0, 32/N, ... of g_t32, first N columns), 1 = closed-form DST-VII
void default_slot_matrix(int slot, int n, int8_t *m)
{
    const Matrix32 d = make_transform_matrix(slot == 0 ? kTrDct2 : kTrDst7, n);
    for (int k = 0; k < n; ++k)
        for (int c = 0; c < n; ++c) m[k * n + c] = d.v[k][c];
}

Matrix32 block_diagonal(const int8_t *m, int n)
{
    Matrix32 r{};
    for (int k = 0; k < 32; ++k)
        for (int c = 0; c < 32; ++c) r.v[k][c] = k / n == c / n ? m[(k % n) * n + (c % n)] : (int8_t)0;
    return r;
}

typedef int8_t SlotMatrices[2][3][256];

// operand images of class (type, N = 4 << l) from a set of slot matrices: the two per-class tables (allocated on first use)
bool upload_class(x266hip_ctx *ctx, const SlotMatrices &mat, int type, int l, DctOps *h)
{
    const int n = 4 << l;
    const Matrix32 mh = block_diagonal(mat[transform_htype(type) == kTrDst7][l], n);
    const Matrix32 mv = block_diagonal(mat[transform_vtype(type) == kTrDst7][l], n);
    build_fwd_ops_general(*h, mh, mv, transform_shift1(n), transform_shift2(n));
    if (!ctx->d_tr[type][l] && hipMalloc((void **)&ctx->d_tr[type][l], sizeof(DctOps)) != hipSuccess) return false;
    if (hipMemcpy(ctx->d_tr[type][l], h, sizeof(DctOps), hipMemcpyHostToDevice) != hipSuccess) return false;
    build_inv_ops_general(*h, mv, mh);
    if (!ctx->d_tr_inv[type][l] && hipMalloc((void **)&ctx->d_tr_inv[type][l], sizeof(DctOps)) != hipSuccess) return false;
    return hipMemcpy(ctx->d_tr_inv[type][l], h, sizeof(DctOps), hipMemcpyHostToDevice) == hipSuccess;
}

// the mixed-class tile kernel's compact tables from a set of slot matrices
bool upload_tile_tabs(x266hip_ctx *ctx, const SlotMatrices &mat)
{
    TileTab t;
    build_tile_tab(t, mat, false);
    if (hipMemcpy(ctx->d_tile_fwd, &t, sizeof t, hipMemcpyHostToDevice) != hipSuccess) return false;
    build_tile_tab(t, mat, true);
    return hipMemcpy(ctx->d_tile_inv, &t, sizeof t, hipMemcpyHostToDevice) == hipSuccess;
}

// every device table that depends on (slot, N = 4 << l) of `mat`
bool upload_slot_tables(x266hip_ctx *ctx, const SlotMatrices &mat, int slot, int l, DctOps *h)
{
    for (int type = 0; type < x266hip_ctx::kTypes; ++type)
        if ((transform_htype(type) == kTrDst7) == (slot == 1) || (transform_vtype(type) == kTrDst7) == (slot == 1))
            if (!upload_class(ctx, mat, type, l, h)) return false;
    return upload_tile_tabs(ctx, mat);
}

}  // namespace

extern "C" {

const char *xHipVersion(void) { return "x266hip 0.1 (gfx950)"; }

int xHipDeviceCount(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0;
}

int xHipCodecInit(x266hip_ctx **out, int device_id)
{
    if (!out) return X266HIP_EINVAL;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        std::fprintf(stderr, "x266hip: no HIP device available (this library has no CPU path)\n");
        return X266HIP_EDEVICE;
    }
    if (device_id < 0 || device_id >= n_dev) return X266HIP_EINVAL;
    x266hip_ctx *ctx = new (std::nothrow) x266hip_ctx;
    if (!ctx) return X266HIP_ENOMEM;
    ctx->device = device_id;
    DeviceScope dev_scope_(device_id);
    if (dev_scope_.status != hipSuccess || hipGetDeviceProperties(&ctx->prop, device_id) != hipSuccess) {
        delete ctx;
        return X266HIP_EDEVICE;
    }
    if (std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
        std::fprintf(stderr, "x266hip: device %d is %s; the kernels are built for gfx950 only\n", device_id,
                     ctx->prop.gcnArchName);
        delete ctx;
        return X266HIP_EDEVICE;
    }
    DctOps *h = new (std::nothrow) DctOps;
    bool ok = h != nullptr;
    if (ok) ok = hipMalloc((void **)&ctx->d_fwd, sizeof(DctOps)) == hipSuccess;
    if (ok) {
        build_fwd_ops(*h);
        ok = hipMemcpy(ctx->d_fwd, h, sizeof(DctOps), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok) {
        build_inv_ops(*h, true);
        ok = hipMalloc((void **)&ctx->d_inv_lds, sizeof(DctOps)) == hipSuccess &&
             hipMemcpy(ctx->d_inv_lds, h, sizeof(DctOps), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok) {
        build_inv_ops(*h, false);
        ok = hipMalloc((void **)&ctx->d_inv_acc, sizeof(DctOps)) == hipSuccess &&
             hipMemcpy(ctx->d_inv_acc, h, sizeof(DctOps), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok) {
        for (int slot = 0; slot < 2; ++slot)
            for (int l = 0; l < 3; ++l) default_slot_matrix(slot, 4 << l, ctx->slot_mat[slot][l]);
        ok = hipMalloc((void **)&ctx->d_tile_fwd, sizeof(TileTab)) == hipSuccess &&
             hipMalloc((void **)&ctx->d_tile_inv, sizeof(TileTab)) == hipSuccess;
    }
    for (int type = 0; type < x266hip_ctx::kTypes && ok; ++type)
        for (int l = 0; l < 3 && ok; ++l) ok = upload_class(ctx, ctx->slot_mat, type, l, h);
    if (ok) ok = upload_tile_tabs(ctx, ctx->slot_mat);
    delete h;

    if (!ok) {
        xHipCodecFree(ctx);
        return X266HIP_ENOMEM;
    }
    *out = ctx;
    return X266HIP_OK;
}

void xHipCodecFree(x266hip_ctx *ctx)
{
    if (!ctx) return;
    DeviceScope dev_scope_(ctx->device);
    for (int i = 0; i < x266hip_ctx::kSlots; ++i) {
        if (ctx->d_stage_in[i]) (void)hipFree(ctx->d_stage_in[i]);
        if (ctx->d_stage_out[i]) (void)hipFree(ctx->d_stage_out[i]);
        if (ctx->stage_up[i]) (void)hipEventDestroy(ctx->stage_up[i]);
        if (ctx->stage_done[i]) (void)hipEventDestroy(ctx->stage_done[i]);
        if (ctx->stage_down[i]) (void)hipEventDestroy(ctx->stage_down[i]);
    }
    for (hipStream_t st : ctx->stage_stream)
        if (st) (void)hipStreamDestroy(st);
    if (ctx->h_small_in) (void)hipHostFree(ctx->h_small_in);
    if (ctx->h_small_out) (void)hipHostFree(ctx->h_small_out);
    if (ctx->small_stream) (void)hipStreamDestroy(ctx->small_stream);
    for (int type = 0; type < x266hip_ctx::kTypes; ++type)
        for (int l = 0; l < 3; ++l) {
            if (ctx->d_tr[type][l]) (void)hipFree(ctx->d_tr[type][l]);
            if (ctx->d_tr_inv[type][l]) (void)hipFree(ctx->d_tr_inv[type][l]);
        }
    for (auto &table : ctx->scratch)
        for (const x266hip_ctx::MeScratch &m : table) { if (m.p) (void)hipFree(m.p); if (m.last_use) (void)hipEventDestroy(m.last_use); }
    for (void *q : ctx->me_retired) (void)hipFree(q);
    for (hipEvent_t e : ctx->tune_ev) if (e) (void)hipEventDestroy(e);
    if (ctx->d_tile_fwd) (void)hipFree(ctx->d_tile_fwd);
    if (ctx->d_tile_inv) (void)hipFree(ctx->d_tile_inv);
    if (ctx->d_fwd) (void)hipFree(ctx->d_fwd);
    if (ctx->d_inv_lds) (void)hipFree(ctx->d_inv_lds);
    if (ctx->d_inv_acc) (void)hipFree(ctx->d_inv_acc);
    delete ctx;
}

const char *xHipLastError(const x266hip_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// The chip's name for humans.  hipDeviceProp_t::name is EMPTY on this pool's ROCm 7.2 boxes (BENCH_r05's "device" read
// "(gfx950:sramecc+:xnack-)"), so ask in turn: the property, hipDeviceGetName, the amdgpu driver's product_name in sysfs,
// and last a table keyed on what the property block does carry (arch, CU count, peak clock) -- marked as such.
static std::string device_marketing_name(const x266hip_ctx *ctx)
{
    auto trimmed = [](const char *s) {
        std::string t(s ? s : "");
        while (!t.empty() && (t.back() == ' ' || t.back() == '\n' || t.back() == '\r' || t.back() == '\t')) t.pop_back();
        size_t b = 0;
        while (b < t.size() && t[b] == ' ') ++b;
        return t.substr(b);
    };
    std::string n = trimmed(ctx->prop.name);
    if (!n.empty()) return n;
    char buf[256] = {0};
    if (hipDeviceGetName(buf, sizeof buf - 1, ctx->device) == hipSuccess && !(n = trimmed(buf)).empty()) return n;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus - 1, ctx->device) == hipSuccess) {
        for (char *c = bus; *c; ++c) *c = (char)std::tolower((unsigned char)*c);
        const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/product_name";
        if (FILE *f = std::fopen(path.c_str(), "r")) {
            const bool got = std::fgets(buf, sizeof buf, f) != nullptr;
            std::fclose(f);
            if (got && !(n = trimmed(buf)).empty()) return n;
        }
    }
    const std::string arch(ctx->prop.gcnArchName);
    if (arch.rfind("gfx950", 0) == 0 && ctx->prop.multiProcessorCount == 256)       // the two 256-CU gfx950 parts differ in peak clock
        return ctx->prop.clockRate >= 2300000 ? "AMD Instinct MI355X [by arch/CU/clock table]" : "AMD Instinct MI350X [by arch/CU/clock table]";
    return "unnamed AMD GPU";
}

int xHipDeviceInfo(const x266hip_ctx *ctx, char *name, size_t name_cap, int *cu_count, int *clock_mhz,
                   size_t *hbm_bytes)
{
    if (!ctx) return X266HIP_EINVAL;
    if (name && name_cap) {
        const std::string chip = device_marketing_name(ctx);
        std::snprintf(name, name_cap, "%s (%s)", chip.c_str(), ctx->prop.gcnArchName);
    }
    if (cu_count) *cu_count = ctx->prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = ctx->prop.clockRate / 1000;
    if (hbm_bytes) *hbm_bytes = ctx->prop.totalGlobalMem;
    return X266HIP_OK;
}

// One table for every launch option: key, the context member it sets, accepted range, and a
// required divisor (workgroup sizes are whole waves).
struct OptionDesc {
    const char *key;
    int x266hip_ctx::*member;
    int lo, hi, multiple_of;
};

static const OptionDesc kOptions[] = {
    {"adaptive_per_wave", &x266hip_ctx::adaptive_per_wave, 0, 1, 1},
    {"dct32_variant", &x266hip_ctx::dct_variant, 0, 2, 2},          // 0 or 2
    {"satd_variant", &x266hip_ctx::satd_variant, 0, 3, 1},
    {"dct32_blocks_per_wave", &x266hip_ctx::dct_blocks_per_wave, 1, 4096, 1},
    {"dct32_inv_blocks_per_wave", &x266hip_ctx::dct_inv_blocks_per_wave, 1, 4096, 1},
    {"dct32_fwdinv_blocks_per_wave", &x266hip_ctx::dct_fwdinv_blocks_per_wave, 0, 4096, 1},   // 0 = automatic: 2, and 8 when only the reconstruction is wanted
    {"dct32_wg_threads", &x266hip_ctx::dct_wg_threads, 0, 256, 64},
    {"satd_groups_per_wave", &x266hip_ctx::satd_groups_per_wave, 0, 4096, 1},
    {"satd_wg_threads", &x266hip_ctx::satd_wg_threads, 0, 256, 64},
    {"satd_lds_bytes_per_wave", &x266hip_ctx::satd_lds_per_wave, 0, 16384, 16},      // whole 16-byte rows; 16 KiB x the four waves of the largest workgroup = the 64 KiB a launch may ask for
    {"tile_tiles_per_wave", &x266hip_ctx::tile_tiles_per_wave, 0, 64, 1},
    {"me_tile_rows", &x266hip_ctx::me_tile_rows, 0, 8, 1},
    {"autotune", &x266hip_ctx::autotune, 0, 1, 1},
};

static const OptionDesc *find_option(const char *key)
{
    if (!key) return nullptr;
    for (const OptionDesc &o : kOptions)
        if (!std::strcmp(key, o.key)) return &o;
    return nullptr;
}

int xHipSetOption(x266hip_ctx *ctx, const char *key, int value)
{
    if (!ctx) return X266HIP_EINVAL;
    const OptionDesc *o = find_option(key);
    if (!o) return fail(ctx, X266HIP_EINVAL, "unknown option");
    if (value < o->lo || value > o->hi || value % o->multiple_of) return fail(ctx, X266HIP_EINVAL, "option value out of range");
    ctx->*(o->member) = value;
    return X266HIP_OK;
}

int xHipGetOption(const x266hip_ctx *ctx, const char *key, int *value)
{
    const OptionDesc *o = find_option(key);
    if (!ctx || !o || !value) return X266HIP_EINVAL;
    *value = ctx->*(o->member);
    return X266HIP_OK;
}

int xHipAutotuneReport(const x266hip_ctx *ctx, char *buf, size_t cap)
{
    if (!ctx || !buf || !cap) return X266HIP_EINVAL;
    static const char *const names[x266hip_ctx::kTuneFamilies] = {"dct32_fwd_inv", "dct32_recon_only", "satd8x8", "sad8", "sad16", "sad32", "sad64"};
    std::string out;
    char line[160];
    for (int f = 0; f < x266hip_ctx::kTuneFamilies; ++f) {
        const x266hip_ctx::Tuned &t = ctx->tuned[f];
        if (t.choice < 0) continue;
        std::snprintf(line, sizeof line, "%s choice %d ms", names[f], t.choice);
        out += line;
        for (int c = 0; c < t.n; ++c) { std::snprintf(line, sizeof line, " %.4f", t.ms[c]); out += line; }
        out += "\n";
    }
    std::snprintf(buf, cap, "%s", out.c_str());
    return X266HIP_OK;
}

// ---- device-pointer batch API ------------------------------------------------
int xDct32FwdBatchDev(x266hip_ctx *ctx, const int16_t *d_in, int16_t *d_out, size_t n, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (bad_ptrs(d_in, d_out, n)) return fail(ctx, X266HIP_EINVAL, "xDct32FwdBatchDev: NULL or unaligned buffer");
    X_DEV(ctx);
    return launch_op(ctx, 0, d_in, d_out, n, (hipStream_t)stream);
}

int xDct32InvBatchDev(x266hip_ctx *ctx, const int16_t *d_in, int16_t *d_out, size_t n, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (bad_ptrs(d_in, d_out, n)) return fail(ctx, X266HIP_EINVAL, "xDct32InvBatchDev: NULL or unaligned buffer");
    X_DEV(ctx);
    return launch_op(ctx, 1, d_in, d_out, n, (hipStream_t)stream);
}

int xDct32SatdFrameDev(x266hip_ctx *ctx, const int16_t *d_dct_in, int16_t *d_dct_out, size_t n_dct_blocks,
                       const int16_t *d_diff, uint32_t *d_satd_out, size_t n_satd_blocks, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (bad_ptrs(d_dct_in, d_dct_out, n_dct_blocks) || bad_ptrs(d_diff, d_satd_out, n_satd_blocks))
        return fail(ctx, X266HIP_EINVAL, "xDct32SatdFrameDev: NULL or unaligned buffer");
    X_DEV(ctx);
    const hipError_t e = launch_frame_lanes(d_dct_in, d_dct_out, n_dct_blocks, ctx->d_fwd, d_diff, d_satd_out, n_satd_blocks, cfg_for(ctx, 2), (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "frame lanes launch", e);
    return X266HIP_OK;
}

int xDct32PassDev(x266hip_ctx *ctx, const int16_t *d_in, int16_t *d_out, size_t n, int shift, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (bad_ptrs(d_in, d_out, n)) return fail(ctx, X266HIP_EINVAL, "xDct32PassDev: NULL or unaligned buffer");
    if (shift < 1 || shift > 15) return fail(ctx, X266HIP_EINVAL, "xDct32PassDev: shift must be 1..15");
    X_DEV(ctx);
    const hipError_t e = launch_dct32_pass(d_in, d_out, n, shift, ctx->d_fwd, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "1-D pass launch", e);
    return X266HIP_OK;
}

int xDct32FwdInvBatchDev(x266hip_ctx *ctx, const int16_t *d_in, int16_t *d_coef, int16_t *d_recon, size_t n, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (bad_ptrs(d_in, d_recon, n) || (n && d_coef && ((uintptr_t)d_coef & 15u)))
        return fail(ctx, X266HIP_EINVAL, "xDct32FwdInvBatchDev: NULL or unaligned buffer");
    X_DEV(ctx);
    LaunchCfg cfg = cfg_for(ctx, 1);
    // 2 blocks per wave with both outputs (the shape that held 0.73-0.76 of 8 TB/s on every box); without the coefficient output the wave's traffic is a
    // third less and its arithmetic the same: 8 blocks per wave (paired over four boxes x three allocations, profiles/r05_fused_variants.txt: 2 -> 4 -8 %, 4 -> 8 another -1 to -4 %)
    cfg.units_per_wave = ctx->dct_fwdinv_blocks_per_wave ? ctx->dct_fwdinv_blocks_per_wave : (d_coef ? 2 : 8);
    if (!ctx->dct_wg_threads) cfg.wg_threads = 256;
    cfg.lds_bytes_per_wave = x266hip_ctx::kFwdInvLdsPerWave;
    cfg.shape = 2;
    hipError_t e;
    const bool knobs_untouched = !ctx->dct_fwdinv_blocks_per_wave && !ctx->dct_wg_threads;
    const bool disjoint = !ranges_overlap(d_in, n * 2048, d_recon, n * 2048) && !ranges_overlap(d_in, n * 2048, d_coef, n * 2048);
    if (ctx->autotune && knobs_untouched && disjoint) {
        const ShapeCand *cands = d_coef ? kFwdInvCands : kReconCands;
        const int n_cands = d_coef ? (int)(sizeof kFwdInvCands / sizeof kFwdInvCands[0]) : (int)(sizeof kReconCands / sizeof kReconCands[0]);
        auto run = [&](int c) {
            LaunchCfg k = cfg;
            k.units_per_wave = cands[c].units_per_wave; k.wg_threads = cands[c].wg_threads; k.lds_bytes_per_wave = cands[c].lds_bytes_per_wave; k.shape = cands[c].shape;
            return launch_dct32_fwdinv(d_in, d_coef, d_recon, n, ctx->d_fwd, ctx->d_inv_acc, k, (hipStream_t)stream);
        };
        e = run(tune_family(ctx, d_coef ? x266hip_ctx::kTuneFwdInv : x266hip_ctx::kTuneRecon, n_cands, n >= ((size_t)1 << 18), (hipStream_t)stream, run));
    } else {
        e = launch_dct32_fwdinv(d_in, d_coef, d_recon, n, ctx->d_fwd, ctx->d_inv_acc, cfg, (hipStream_t)stream);
    }
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "fwd+inv launch", e);
    return X266HIP_OK;
}

int xSatd8x8BatchDev(x266hip_ctx *ctx, const int16_t *d_diff, uint32_t *d_out, size_t n, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (n && (!d_diff || !d_out || ((uintptr_t)d_diff & 15u) || ((uintptr_t)d_out & 3u)))
        return fail(ctx, X266HIP_EINVAL, "xSatd8x8BatchDev: NULL or unaligned buffer");
    X_DEV(ctx);
    const bool knobs_untouched = !ctx->satd_variant && !ctx->satd_groups_per_wave && !ctx->satd_wg_threads && !ctx->satd_lds_per_wave;
    if (ctx->autotune && knobs_untouched && n >= kSatdDmaMinBlocks && !ranges_overlap(d_diff, n * 128, d_out, n * 4)) {
        auto run = [&](int c) {
            LaunchCfg k = cfg_for(ctx, 2);
            k.units_per_wave = kSatdCands[c].units_per_wave; k.wg_threads = kSatdCands[c].wg_threads; k.lds_bytes_per_wave = kSatdCands[c].lds_bytes_per_wave; k.shape = kSatdCands[c].shape;
            return launch_satd8x8(d_diff, d_out, n, k, (hipStream_t)stream);
        };
        const hipError_t e = run(tune_family(ctx, x266hip_ctx::kTuneSatd, (int)(sizeof kSatdCands / sizeof kSatdCands[0]), n >= ((size_t)1 << 23), (hipStream_t)stream, run));
        if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "kernel launch", e);
        return X266HIP_OK;
    }
    return launch_op(ctx, 2, d_diff, d_out, n, (hipStream_t)stream);
}

int xHipMemCeilingDev(x266hip_ctx *ctx, int kind, const void *d_src, void *d_dst, size_t bytes, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (kind < X266_MEM_COPY || kind > X266_MEM_READ_PROBE) return fail(ctx, X266HIP_EINVAL, "xHipMemCeilingDev: kind must be X266_MEM_COPY, _READ, _WRITE or _READ_PROBE");
    if ((bytes & 15u) || bad_ptrs(kind == X266_MEM_WRITE ? d_dst : d_src, d_dst, bytes)) return fail(ctx, X266HIP_EINVAL, "xHipMemCeilingDev: NULL or unaligned buffer, or bytes not a multiple of 16");
    X_DEV(ctx);
    const hipError_t e = launch_mem_ceiling(kind, d_src, d_dst, bytes, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "memory ceiling launch", e);
    return X266HIP_OK;
}

int xIntra32PredictDev(x266hip_ctx *ctx, const x266_intra_ref_t *d_refs, const uint8_t *d_modes,
                       const uint32_t *d_ref_index, uint8_t *d_pred, size_t n, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (n && (!d_refs || !d_modes || !d_pred || (((uintptr_t)d_refs | (uintptr_t)d_pred) & 15u) || ((uintptr_t)d_ref_index & 3u)))
        return fail(ctx, X266HIP_EINVAL, "xIntra32PredictDev: NULL or unaligned buffer");
    X_DEV(ctx);
    hipError_t e = launch_intra32_predict(d_refs, d_modes, d_ref_index, d_pred, n, x266hip_ctx::kIntraRounds, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "intra launch", e);
    return X266HIP_OK;
}

int xIntra32ResidualDct32Dev(x266hip_ctx *ctx, const x266_intra_ref_t *d_refs, const uint8_t *d_modes, const uint32_t *d_ref_index,
                             const uint8_t *d_src, int16_t *d_coef, size_t n, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (n && (!d_refs || !d_modes || !d_src || !d_coef || (((uintptr_t)d_refs | (uintptr_t)d_src | (uintptr_t)d_coef) & 15u) || ((uintptr_t)d_ref_index & 3u)))
        return fail(ctx, X266HIP_EINVAL, "xIntra32ResidualDct32Dev: NULL or unaligned buffer");
    X_DEV(ctx);
    hipError_t e = launch_intra32_residual_dct32(d_refs, d_modes, d_ref_index, d_src, d_coef, n, ctx->d_fwd, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "intra residual + transform launch", e);
    return X266HIP_OK;
}

int xIntra32CostsDev(x266hip_ctx *ctx, const x266_intra_ref_t *d_refs, const uint8_t *d_src,
                     uint32_t *d_costs, uint8_t *d_best_mode, size_t n, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (n && (!d_refs || !d_src || !d_costs || (((uintptr_t)d_refs | (uintptr_t)d_src) & 15u) || ((uintptr_t)d_costs & 3u)))
        return fail(ctx, X266HIP_EINVAL, "xIntra32CostsDev: NULL or unaligned buffer");
    X_DEV(ctx);
    hipError_t e = launch_intra32_costs(d_refs, d_src, d_costs, d_best_mode, n, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "intra decision launch", e);
    return X266HIP_OK;
}

int xFillResidualDev(x266hip_ctx *ctx, int16_t *d_dst, size_t n_samples, uint64_t seed, uint64_t first_index,
                     void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (n_samples && (!d_dst || ((uintptr_t)d_dst & 15u)))
        return fail(ctx, X266HIP_EINVAL, "xFillResidualDev: NULL or unaligned buffer");
    X_DEV(ctx);
    hipError_t e = launch_fill_residual(d_dst, n_samples, seed, first_index, cfg_for(ctx, 0), (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "fill launch", e);
    return X266HIP_OK;
}

int xTransformFwdBatchDev(x266hip_ctx *ctx, int type, int size, const int16_t *d_in, int16_t *d_out, size_t n,
                          const uint32_t *d_offsets, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (type < 0 || type >= x266hip_ctx::kTypes) return fail(ctx, X266HIP_EINVAL, "xTransformFwdBatchDev: unknown transform type");
    if (size != 4 && size != 8 && size != 16 && !(size == 32 && type == X266_TR_DCT2))
        return fail(ctx, X266HIP_EINVAL, "xTransformFwdBatchDev: size must be 4, 8, 16 (or 32 for DCT-II)");
    if (bad_ptrs(d_in, d_out, n)) return fail(ctx, X266HIP_EINVAL, "xTransformFwdBatchDev: NULL or unaligned buffer");
    if (n && ((uintptr_t)d_offsets & 3u)) return fail(ctx, X266HIP_EINVAL, "xTransformFwdBatchDev: unaligned offset table");
    if (!ctx->tr_tables_valid) return fail(ctx, X266HIP_EDEVICE, "xTransformFwdBatchDev: the transform tables of this context are invalid (a failed xTransformSetMatrix)");
    X_DEV(ctx);
    if (size == 32 && !d_offsets) return launch_op(ctx, 0, d_in, d_out, n, (hipStream_t)stream);
    const int l = size == 4 ? 0 : (size == 8 ? 1 : 2);
    LaunchCfg cfg = cfg_for(ctx, 0);
    cfg.units_per_wave = 1;                                             // one 32x32 tile per wave (profiles/r01_launch_sweep.txt)
    hipError_t e = size == 32 ? launch_transform_small(5, d_in, d_out, n, ctx->d_fwd, d_offsets, cfg, (hipStream_t)stream)
                              : launch_transform_small(l + 2, d_in, d_out, n, ctx->d_tr[type][l], d_offsets, cfg, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "transform launch", e);
    return X266HIP_OK;
}

// All or nothing: the device tables are built from `next`, a COPY of the slot matrices with the wanted changes, and the context's
// own copy changes only once every upload has succeeded; after a failed upload the old tables are put back, and if even that fails
// the transform set of this context is marked unusable (its calls then fail) rather than left half old, half new.
static int commit_slot_matrices(x266hip_ctx *ctx, const SlotMatrices &next, const char *who)
{
    if (hipDeviceSynchronize() != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "transform matrices: device synchronisation");   // launches in flight read the tables
    DctOps *h = new (std::nothrow) DctOps;
    if (!h) return fail(ctx, X266HIP_ENOMEM, who);
    bool ok = true, touched = false;
    for (int slot = 0; slot < 2 && ok; ++slot)
        for (int l = 0; l < 3 && ok; ++l)
            if (std::memcmp(next[slot][l], ctx->slot_mat[slot][l], 256)) { touched = true; ok = upload_slot_tables(ctx, next, slot, l, h); }
    if (!ok) {
        bool back = true;
        for (int slot = 0; slot < 2 && back; ++slot)
            for (int l = 0; l < 3 && back; ++l)
                if (std::memcmp(next[slot][l], ctx->slot_mat[slot][l], 256)) back = upload_slot_tables(ctx, ctx->slot_mat, slot, l, h);
        ctx->tr_tables_valid = back;
    }
    delete h;
    if (!ok) return fail(ctx, X266HIP_EDEVICE, ctx->tr_tables_valid ? "transform matrices: table upload failed, previous matrices kept"
                                                                      : "transform matrices: table upload failed and the previous tables could not be restored: transform set unusable");
    if (touched) std::memcpy(ctx->slot_mat, next, sizeof(SlotMatrices));
    ctx->tr_tables_valid = true;
    return X266HIP_OK;
}

int xTransformSetMatrix(x266hip_ctx *ctx, int slot, int size, const int8_t *m)
{
    if (!ctx) return X266HIP_EINVAL;
    if (slot < 0 || slot > 1) return fail(ctx, X266HIP_EINVAL, "xTransformSetMatrix: slot must be 0 or 1");
    if (size != 4 && size != 8 && size != 16)
        return fail(ctx, X266HIP_EINVAL, "xTransformSetMatrix: size must be 4, 8 or 16 (the 32-point DCT-II is the reference's g_t32 and stays)");
    X_DEV(ctx);
    const int l = size == 4 ? 0 : (size == 8 ? 1 : 2);
    SlotMatrices next;
    std::memcpy(next, ctx->slot_mat, sizeof next);
    if (m) std::memcpy(next[slot][l], m, (size_t)size * size);
    else default_slot_matrix(slot, size, next[slot][l]);
    const int rc = commit_slot_matrices(ctx, next, "xTransformSetMatrix");
    if (rc == X266HIP_OK && slot == 1) ctx->slot1_preset = -1;
    return rc;
}

int xTransformUsePreset(x266hip_ctx *ctx, int preset)
{
    if (!ctx) return X266HIP_EINVAL;
    if (preset < X266_PRESET_CLOSED_FORM || preset > X266_PRESET_VTM_DCT8) return fail(ctx, X266HIP_EINVAL, "xTransformUsePreset: unknown preset");
    X_DEV(ctx);
    SlotMatrices next;
    std::memcpy(next, ctx->slot_mat, sizeof next);
    for (int l = 0; l < 3; ++l) {
        std::memset(next[1][l], 0, 256);
        if (preset == X266_PRESET_CLOSED_FORM) default_slot_matrix(1, 4 << l, next[1][l]);
        else vtm_slot1_matrix(4 << l, preset == X266_PRESET_VTM_DCT8, next[1][l]);
    }
    const int rc = commit_slot_matrices(ctx, next, "xTransformUsePreset");
    if (rc == X266HIP_OK) ctx->slot1_preset = preset;
    return rc;
}

int xTransformPreset(const x266hip_ctx *ctx) { return ctx ? ctx->slot1_preset : -1; }

int xTransformGetMatrix(const x266hip_ctx *ctx, int slot, int size, int8_t *m)
{
    if (!ctx || !m || slot < 0 || slot > 1 || (size != 4 && size != 8 && size != 16)) return X266HIP_EINVAL;
    std::memcpy(m, ctx->slot_mat[slot][size == 4 ? 0 : (size == 8 ? 1 : 2)], (size_t)size * size);
    return X266HIP_OK;
}

int xTransformTilesDev(x266hip_ctx *ctx, int inverse, const int16_t *d_in, int16_t *d_out, size_t n_tiles,
                       const uint32_t *d_tile_offsets, const uint8_t *d_tile_class, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (bad_ptrs(d_in, d_out, n_tiles) || (n_tiles && !d_tile_class)) return fail(ctx, X266HIP_EINVAL, "xTransformTilesDev: NULL or unaligned buffer");
    if (n_tiles && ((uintptr_t)d_tile_offsets & 3u)) return fail(ctx, X266HIP_EINVAL, "xTransformTilesDev: unaligned offset table");
    if (!ctx->tr_tables_valid) return fail(ctx, X266HIP_EDEVICE, "xTransformTilesDev: the transform tables of this context are invalid (a failed xTransformSetMatrix)");
    X_DEV(ctx);
    LaunchCfg cfg = cfg_for(ctx, inverse ? 1 : 0);
    cfg.lds_bytes_per_wave = x266hip_ctx::kTileLdsPerWave;
    cfg.units_per_wave = ctx->tile_tiles_per_wave ? ctx->tile_tiles_per_wave : 2;   // measured optimum (profiles/r03_tiles_one_launch.txt): the wave's table copy serves two tiles
    hipError_t e = launch_transform_tiles(inverse != 0, d_in, d_out, n_tiles, d_tile_offsets, d_tile_class,
                                          inverse ? ctx->d_tile_inv : ctx->d_tile_fwd, cfg, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "tile transform launch", e);
    return X266HIP_OK;
}

int xConvInputFmtDev(x266hip_ctx *ctx, x266_ref_block_t *d_tiles, const uint8_t *d_y, const uint8_t *d_u, const uint8_t *d_v,
                     intptr_t strdY, int width, int height, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (width <= 0 || height <= 0 || (width & 15) || (height & 15)) return fail(ctx, X266HIP_EINVAL, "xConvInputFmtDev: width/height must be multiples of 16");
    if (!d_tiles || !d_y || !d_u || !d_v) return fail(ctx, X266HIP_EINVAL, "xConvInputFmtDev: NULL buffer");
    if (strdY < width || (strdY & 15) || (((uintptr_t)d_y | (uintptr_t)d_tiles) & 15u) || (((uintptr_t)d_u | (uintptr_t)d_v) & 7u))
        return fail(ctx, X266HIP_EINVAL, "xConvInputFmtDev: stride / alignment");
    X_DEV(ctx);
    hipError_t e = launch_tile_convert(true, d_tiles, const_cast<uint8_t *>(d_y), const_cast<uint8_t *>(d_u), const_cast<uint8_t *>(d_v),
                                       (long long)strdY, (long long)(strdY >> 1), width, height, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "tile pack launch", e);
    return X266HIP_OK;
}

int xConvOutput420Dev(x266hip_ctx *ctx, const x266_ref_block_t *d_tiles, uint8_t *d_y, intptr_t strdY, uint8_t *d_u, uint8_t *d_v,
                      intptr_t strdC, int width, int height, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (width <= 0 || height <= 0 || (width & 15) || (height & 15)) return fail(ctx, X266HIP_EINVAL, "xConvOutput420Dev: width/height must be multiples of 16");
    if (!d_tiles || !d_y || !d_u || !d_v) return fail(ctx, X266HIP_EINVAL, "xConvOutput420Dev: NULL buffer");
    if (strdY < width || strdC < width / 2 || (strdY & 15) || (strdC & 7) || (((uintptr_t)d_y | (uintptr_t)d_tiles) & 15u) ||
        (((uintptr_t)d_u | (uintptr_t)d_v) & 7u))
        return fail(ctx, X266HIP_EINVAL, "xConvOutput420Dev: stride / alignment");
    X_DEV(ctx);
    hipError_t e = launch_tile_convert(false, const_cast<x266_ref_block_t *>(d_tiles), d_y, d_u, d_v, (long long)strdY, (long long)strdC,
                                       width, height, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "tile unpack launch", e);
    return X266HIP_OK;
}

int xResidualLumaDev(x266hip_ctx *ctx, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int width, int height,
                     int block_edge, int16_t *d_residual, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (block_edge != 8 && block_edge != 32) return fail(ctx, X266HIP_EINVAL, "xResidualLumaDev: block_edge must be 8 or 32");
    const int mask = block_edge == 32 ? 31 : 15;
    if (width <= 0 || height <= 0 || (width & mask) || (height & mask)) return fail(ctx, X266HIP_EINVAL, "xResidualLumaDev: frame size");
    if (!d_cur || !d_pred || !d_residual || ((((uintptr_t)d_cur | (uintptr_t)d_pred | (uintptr_t)d_residual)) & 15u))
        return fail(ctx, X266HIP_EINVAL, "xResidualLumaDev: NULL or unaligned buffer");
    X_DEV(ctx);
    hipError_t e = launch_residual_luma(block_edge, d_cur, d_pred, d_residual, width, height, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "residual launch", e);
    return X266HIP_OK;
}

int xDct32FwdFromTilesDev(x266hip_ctx *ctx, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int width, int height,
                          int16_t *d_coef, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (width <= 0 || height <= 0 || (width & 31) || (height & 31)) return fail(ctx, X266HIP_EINVAL, "xDct32FwdFromTilesDev: width/height must be multiples of 32");
    if (!d_cur || !d_pred || !d_coef || ((((uintptr_t)d_cur | (uintptr_t)d_pred | (uintptr_t)d_coef)) & 15u))
        return fail(ctx, X266HIP_EINVAL, "xDct32FwdFromTilesDev: NULL or unaligned buffer");
    X_DEV(ctx);
    hipError_t e = launch_dct32_from_tiles(d_cur, d_pred, d_coef, width, height, ctx->d_fwd, cfg_for(ctx, 0), (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "fused transform launch", e);
    return X266HIP_OK;
}

int xSatd8x8FromTilesDev(x266hip_ctx *ctx, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int width, int height,
                         uint32_t *d_out, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (width <= 0 || height <= 0 || (width & 15) || (height & 15)) return fail(ctx, X266HIP_EINVAL, "xSatd8x8FromTilesDev: width/height must be multiples of 16");
    if (!d_cur || !d_pred || !d_out || ((((uintptr_t)d_cur | (uintptr_t)d_pred)) & 15u) || ((uintptr_t)d_out & 3u))
        return fail(ctx, X266HIP_EINVAL, "xSatd8x8FromTilesDev: NULL or unaligned buffer");
    X_DEV(ctx);
    hipError_t e = launch_satd8x8_from_tiles(d_cur, d_pred, d_out, width, height, ctx->satd_variant == 1 || ctx->satd_variant == 3 ? ctx->satd_variant : 0, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "fused satd launch", e);
    return X266HIP_OK;
}

// the two output streams of a chroma call must not overlap: plane V starts at or after U's first block ends (interleaved
// form) or anywhere else outside [U, U + span)
static bool chroma_outputs_ok(const void *u, const void *v, size_t block_bytes, size_t n_blocks, size_t pitch)
{
    if (!u || !v || pitch < 1 || n_blocks == 0) return u && v && pitch >= 1;
    const uintptr_t a = (uintptr_t)u, b = (uintptr_t)v;
    const uintptr_t lo = a < b ? a : b, hi = a < b ? b : a;
    const size_t gap = (size_t)(hi - lo);
    if (gap >= ((n_blocks - 1) * pitch + 1) * block_bytes) return true;     // disjoint spans
    // interleaved: the other plane's blocks sit in the holes of this one's pitch
    return pitch >= 2 && gap % block_bytes == 0 && gap / block_bytes >= 1 && gap / block_bytes <= pitch - 1;
}

int xResidualChromaDev(x266hip_ctx *ctx, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int width, int height,
                       int block_edge, int16_t *d_res_u, int16_t *d_res_v, size_t block_pitch, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (block_edge != 8 && block_edge != 32) return fail(ctx, X266HIP_EINVAL, "xResidualChromaDev: block_edge must be 8 or 32");
    const int mask = block_edge == 32 ? 63 : 15;                             // luma dimensions: a 32x32 chroma block is a 64x64 CTU's
    if (width <= 0 || height <= 0 || (width & mask) || (height & mask)) return fail(ctx, X266HIP_EINVAL, "xResidualChromaDev: frame size");
    if (!d_cur || !d_pred || !d_res_u || !d_res_v || ((((uintptr_t)d_cur | (uintptr_t)d_pred | (uintptr_t)d_res_u | (uintptr_t)d_res_v)) & 15u))
        return fail(ctx, X266HIP_EINVAL, "xResidualChromaDev: NULL or unaligned buffer");
    const size_t n_blocks = (size_t)(width / 2 / block_edge) * (size_t)(height / 2 / block_edge);
    if (!chroma_outputs_ok(d_res_u, d_res_v, (size_t)block_edge * block_edge * 2, n_blocks, block_pitch))
        return fail(ctx, X266HIP_EINVAL, "xResidualChromaDev: block_pitch < 1 or overlapping U / V outputs");
    X_DEV(ctx);
    hipError_t e = launch_residual_chroma(block_edge, d_cur, d_pred, d_res_u, d_res_v, block_pitch, width, height, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "chroma residual launch", e);
    return X266HIP_OK;
}

int xDct32FwdChromaFromTilesDev(x266hip_ctx *ctx, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int width, int height,
                                int16_t *d_coef_u, int16_t *d_coef_v, size_t block_pitch, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (width <= 0 || height <= 0 || (width & 63) || (height & 63)) return fail(ctx, X266HIP_EINVAL, "xDct32FwdChromaFromTilesDev: width/height must be multiples of 64");
    if (!d_cur || !d_pred || !d_coef_u || !d_coef_v || ((((uintptr_t)d_cur | (uintptr_t)d_pred | (uintptr_t)d_coef_u | (uintptr_t)d_coef_v)) & 15u))
        return fail(ctx, X266HIP_EINVAL, "xDct32FwdChromaFromTilesDev: NULL or unaligned buffer");
    if (!chroma_outputs_ok(d_coef_u, d_coef_v, 2048, (size_t)(width / 64) * (size_t)(height / 64), block_pitch))
        return fail(ctx, X266HIP_EINVAL, "xDct32FwdChromaFromTilesDev: block_pitch < 1 or overlapping U / V outputs");
    X_DEV(ctx);
    hipError_t e = launch_dct32_chroma_from_tiles(d_cur, d_pred, d_coef_u, d_coef_v, block_pitch, width, height, ctx->d_fwd, cfg_for(ctx, 0), (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "fused chroma transform launch", e);
    return X266HIP_OK;
}

int xDct32FwdCtuFromTilesDev(x266hip_ctx *ctx, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int width, int height,
                             int16_t *d_coef, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (width <= 0 || height <= 0 || (width & 63) || (height & 63)) return fail(ctx, X266HIP_EINVAL, "xDct32FwdCtuFromTilesDev: width/height must be multiples of 64");
    if (!d_cur || !d_pred || !d_coef || ((((uintptr_t)d_cur | (uintptr_t)d_pred | (uintptr_t)d_coef)) & 15u))
        return fail(ctx, X266HIP_EINVAL, "xDct32FwdCtuFromTilesDev: NULL or unaligned buffer");
    X_DEV(ctx);
    hipError_t e = launch_dct32_ctu_from_tiles(d_cur, d_pred, d_coef, width, height, ctx->d_fwd, cfg_for(ctx, 0), (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "fused CTU transform launch", e);
    return X266HIP_OK;
}

int xSatd8x8ChromaFromTilesDev(x266hip_ctx *ctx, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int width, int height,
                               uint32_t *d_out_u, uint32_t *d_out_v, size_t pitch, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (width <= 0 || height <= 0 || (width & 15) || (height & 15)) return fail(ctx, X266HIP_EINVAL, "xSatd8x8ChromaFromTilesDev: width/height must be multiples of 16");
    if (!d_cur || !d_pred || !d_out_u || !d_out_v || ((((uintptr_t)d_cur | (uintptr_t)d_pred)) & 15u) || (((uintptr_t)d_out_u | (uintptr_t)d_out_v) & 3u))
        return fail(ctx, X266HIP_EINVAL, "xSatd8x8ChromaFromTilesDev: NULL or unaligned buffer");
    if (!chroma_outputs_ok(d_out_u, d_out_v, 4, (size_t)(width / 16) * (size_t)(height / 16), pitch))
        return fail(ctx, X266HIP_EINVAL, "xSatd8x8ChromaFromTilesDev: pitch < 1 or overlapping U / V outputs");
    X_DEV(ctx);
    hipError_t e = launch_satd8x8_chroma_from_tiles(d_cur, d_pred, d_out_u, d_out_v, pitch, width, height, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "fused chroma satd launch", e);
    return X266HIP_OK;
}

int xSadBatchDev(x266hip_ctx *ctx, int edge, const uint8_t *d_a, const uint8_t *d_b, uint32_t *d_out, size_t n, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (edge != 4 && edge != 8 && edge != 16 && edge != 32 && edge != 64) return fail(ctx, X266HIP_EINVAL, "xSadBatchDev: edge must be 4, 8, 16, 32 or 64");
    if (n && (!d_a || !d_b || !d_out || (((uintptr_t)d_a | (uintptr_t)d_b) & 15u) || ((uintptr_t)d_out & 3u)))
        return fail(ctx, X266HIP_EINVAL, "xSadBatchDev: NULL or unaligned buffer");
    X_DEV(ctx);
    hipError_t e;
    const size_t in_bytes = n * (size_t)(edge * edge);
    if (ctx->autotune && edge >= 8 && !ranges_overlap(d_a, in_bytes, d_out, n * 4) && !ranges_overlap(d_b, in_bytes, d_out, n * 4)) {
        auto run = [&](int c) { return launch_sad(edge, d_a, d_b, d_out, n, kSadCands[c].wg_threads / 64, kSadCands[c].lds_bytes_per_wave, (hipStream_t)stream); };
        const int family = edge == 8 ? x266hip_ctx::kTuneSad8 : edge == 16 ? x266hip_ctx::kTuneSad16 : edge == 32 ? x266hip_ctx::kTuneSad32 : x266hip_ctx::kTuneSad64;
        e = run(tune_family(ctx, family, (int)(sizeof kSadCands / sizeof kSadCands[0]), in_bytes >= ((size_t)128 << 20), (hipStream_t)stream, run));
    } else {
        e = launch_sad(edge, d_a, d_b, d_out, n, 0, 0, (hipStream_t)stream);
    }
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "sad launch", e);
    return X266HIP_OK;
}

int xTransformInvBatchDev(x266hip_ctx *ctx, int type, int size, const int16_t *d_in, int16_t *d_out, size_t n,
                          const uint32_t *d_offsets, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (type < 0 || type >= x266hip_ctx::kTypes) return fail(ctx, X266HIP_EINVAL, "xTransformInvBatchDev: unknown transform type");
    if (size != 4 && size != 8 && size != 16 && !(size == 32 && type == X266_TR_DCT2))
        return fail(ctx, X266HIP_EINVAL, "xTransformInvBatchDev: size must be 4, 8, 16 (or 32 for DCT-II)");
    if (bad_ptrs(d_in, d_out, n)) return fail(ctx, X266HIP_EINVAL, "xTransformInvBatchDev: NULL or unaligned buffer");
    if (n && ((uintptr_t)d_offsets & 3u)) return fail(ctx, X266HIP_EINVAL, "xTransformInvBatchDev: unaligned offset table");
    if (!ctx->tr_tables_valid) return fail(ctx, X266HIP_EDEVICE, "xTransformInvBatchDev: the transform tables of this context are invalid (a failed xTransformSetMatrix)");
    X_DEV(ctx);
    if (size == 32 && !d_offsets) return launch_op(ctx, 1, d_in, d_out, n, (hipStream_t)stream);
    const int l = size == 4 ? 0 : (size == 8 ? 1 : 2);
    const LaunchCfg cfg = cfg_for(ctx, 1);
    hipError_t e = size == 32 ? launch_transform_small_inv(5, d_in, d_out, n, ctx->d_inv_lds, d_offsets, cfg, (hipStream_t)stream)
                              : launch_transform_small_inv(l + 2, d_in, d_out, n, ctx->d_tr_inv[type][l], d_offsets, cfg, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "inverse transform launch", e);
    return X266HIP_OK;
}

int xHipMeScratchReserve(x266hip_ctx *ctx, void *stream, int width, int height)
{
    if (!ctx) return X266HIP_EINVAL;
    if (width < 8 || height < 8) return fail(ctx, X266HIP_EINVAL, "xHipMeScratchReserve: frame size");
    X_DEV(ctx);
    x266hip_ctx::MeScratch *slot = nullptr;
    return me_scratch_for(ctx, (hipStream_t)stream, width, height, &slot);
}

int xSatd8x8SearchDev(x266hip_ctx *ctx, const uint8_t *d_cur, intptr_t cur_stride, const uint8_t *d_ref,
                      intptr_t ref_stride, int width, int height, int range, x266_me_result_t *d_best,
                      uint32_t *d_costs, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (!d_cur || !d_ref || !d_best) return fail(ctx, X266HIP_EINVAL, "xSatd8x8SearchDev: NULL buffer");
    if (width < 8 || height < 8 || (width & 7) || (height & 7)) return fail(ctx, X266HIP_EINVAL, "xSatd8x8SearchDev: frame size must be a multiple of 8");
    if (range < 1 || range > 64) return fail(ctx, X266HIP_EINVAL, "xSatd8x8SearchDev: range must be 1..64");
    if (cur_stride < width || ref_stride < width + 2 * range) return fail(ctx, X266HIP_EINVAL, "xSatd8x8SearchDev: stride too small");
    if (((uintptr_t)d_best & 7u) || ((uintptr_t)d_costs & 3u)) return fail(ctx, X266HIP_EINVAL, "xSatd8x8SearchDev: unaligned output");
    X_DEV(ctx);
    uint32_t *d_me_coef = nullptr;
    x266hip_ctx::MeScratch *slot = nullptr;
    if (const int rc = me_scratch_for(ctx, (hipStream_t)stream, width, height, &slot)) return rc;
    d_me_coef = slot->p;
    (void)hipGetLastError();
    hipError_t e = launch_satd_search(d_cur, (long long)cur_stride, d_ref, (long long)ref_stride, width, height, range,
                                      d_best, d_costs, ctx->me_tile_rows, d_me_coef, ctx->prop.multiProcessorCount, (hipStream_t)stream);
    if (e == hipSuccess && slot->last_use && !slot->capturing_now) (void)hipEventRecord(slot->last_use, (hipStream_t)stream);   // what an eviction waits for
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "search launch", e);
    return X266HIP_OK;
}

int xSad8x8SearchDev(x266hip_ctx *ctx, const uint8_t *d_cur, intptr_t cur_stride, const uint8_t *d_ref,
                     intptr_t ref_stride, int width, int height, int range, x266_me_result_t *d_best,
                     uint32_t *d_costs, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (!d_cur || !d_ref || !d_best) return fail(ctx, X266HIP_EINVAL, "xSad8x8SearchDev: NULL buffer");
    if (width < 8 || height < 8 || (width & 7) || (height & 7)) return fail(ctx, X266HIP_EINVAL, "xSad8x8SearchDev: frame size must be a multiple of 8");
    if (range < 1 || range > 64) return fail(ctx, X266HIP_EINVAL, "xSad8x8SearchDev: range must be 1..64");
    if (cur_stride < width || ref_stride < width + 2 * range) return fail(ctx, X266HIP_EINVAL, "xSad8x8SearchDev: stride too small");
    if (((uintptr_t)d_cur & 3u) || (cur_stride & 3)) return fail(ctx, X266HIP_EINVAL, "xSad8x8SearchDev: current frame must be 4-byte aligned with a stride multiple of 4");
    if (((uintptr_t)d_best & 7u) || ((uintptr_t)d_costs & 3u)) return fail(ctx, X266HIP_EINVAL, "xSad8x8SearchDev: unaligned output");
    X_DEV(ctx);
    hipError_t e = launch_sad_search(d_cur, (long long)cur_stride, d_ref, (long long)ref_stride, width, height, range,
                                     d_best, d_costs, ctx->me_tile_rows, (hipStream_t)stream);
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "SAD search launch", e);
    return X266HIP_OK;
}

// ---- host-pointer batch API --------------------------------------------------
// Chunks of the batch rotate over three staging slots; uploads, kernels and downloads each have a stream of their own
// and are ordered by the slots' events: H2D(i+1) and D2H(i-1) overlap kernel(i).
static int ensure_staging(x266hip_ctx *ctx, size_t in_bytes, size_t out_bytes)
{
    for (hipStream_t &st : ctx->stage_stream)
        if (!st) X_HIP(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int i = 0; i < x266hip_ctx::kSlots; ++i) {
        if (!ctx->stage_up[i]) X_HIP(ctx, hipEventCreateWithFlags(&ctx->stage_up[i], hipEventDisableTiming));
        if (!ctx->stage_done[i]) X_HIP(ctx, hipEventCreateWithFlags(&ctx->stage_done[i], hipEventDisableTiming));
        if (!ctx->stage_down[i]) X_HIP(ctx, hipEventCreateWithFlags(&ctx->stage_down[i], hipEventDisableTiming));
    }
    if (in_bytes > ctx->stage_in_bytes) {
        for (int i = 0; i < x266hip_ctx::kSlots; ++i) {
            if (ctx->d_stage_in[i]) (void)hipFree(ctx->d_stage_in[i]);
            ctx->d_stage_in[i] = nullptr;
            if (hipMalloc(&ctx->d_stage_in[i], in_bytes) != hipSuccess) { ctx->stage_in_bytes = 0; return fail(ctx, X266HIP_ENOMEM, "staging alloc"); }
        }
        ctx->stage_in_bytes = in_bytes;
    }
    if (out_bytes > ctx->stage_out_bytes) {
        for (int i = 0; i < x266hip_ctx::kSlots; ++i) {
            if (ctx->d_stage_out[i]) (void)hipFree(ctx->d_stage_out[i]);
            ctx->d_stage_out[i] = nullptr;
            if (hipMalloc(&ctx->d_stage_out[i], out_bytes) != hipSuccess) { ctx->stage_out_bytes = 0; return fail(ctx, X266HIP_ENOMEM, "staging alloc"); }
        }
        ctx->stage_out_bytes = out_bytes;
    }
    return X266HIP_OK;
}

// Uploads + kernels are issued by the calling thread, downloads by a helper thread that lives for the call: a copy from / to
// PAGEABLE memory blocks the thread that issues it while the runtime stages it, so one thread keeps only one direction of the
// link busy (27 GB/s each way); two threads reach what pinned buffers reach, 43-44 GB/s each way of the 48.5 the link gives
// with both directions running (profiles/r04_hostpipe.txt).
static int host_batch(x266hip_ctx *ctx, int op, const void *in, void *out, size_t n, size_t in_unit, size_t out_unit)
{
    if (!ctx) return X266HIP_EINVAL;
    if (n == 0) return X266HIP_OK;
    if (!in || !out) return fail(ctx, X266HIP_EINVAL, "NULL host buffer");
    X_DEV(ctx);
    // A call of a block or a few (the BDPI shims hand over ONE: dct32_genNew, satd8x8_genNew) is all latency: upload, kernel and download
    // as three stream operations cost 60-80 us per call.  Up to 64 KiB each way the kernel instead reads its input from, and writes its result
    // to, page-locked host memory the device sees (hipHostMalloc: coherent, uncached on the device side): two memcpy on the host, ONE launch,
    // one synchronize -- profiles/r06_bdpi_latency.txt.  Same kernels, same bytes (tests/test_gpu_parity.py runs both paths against each other).
    if (n * in_unit <= x266hip_ctx::kSmallCallBytes && n * out_unit <= x266hip_ctx::kSmallCallBytes) {
        if (!ctx->h_small_in) X_HIP(ctx, hipHostMalloc(&ctx->h_small_in, x266hip_ctx::kSmallCallBytes, hipHostMallocDefault));
        if (!ctx->h_small_out) X_HIP(ctx, hipHostMalloc(&ctx->h_small_out, x266hip_ctx::kSmallCallBytes, hipHostMallocDefault));
        if (!ctx->small_stream) X_HIP(ctx, hipStreamCreateWithFlags(&ctx->small_stream, hipStreamNonBlocking));
        std::memcpy(ctx->h_small_in, in, n * in_unit);
        const int rc_small = launch_op(ctx, op, ctx->h_small_in, ctx->h_small_out, n, ctx->small_stream);
        const hipError_t es = hipStreamSynchronize(ctx->small_stream);   // also after a failed launch: nothing of this call stays in flight
        if (rc_small != X266HIP_OK) return rc_small;
        if (es != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "hipStreamSynchronize(small call)", es);
        std::memcpy(out, ctx->h_small_out, n * out_unit);
        return X266HIP_OK;
    }
    const size_t chunk_bytes = (size_t)16 << 20;                       // 16 MiB of input per chunk
    size_t chunk = chunk_bytes / in_unit;
    if (chunk > n) chunk = n;
    int rc = ensure_staging(ctx, chunk * in_unit, chunk * out_unit);
    if (rc) return rc;
    hipStream_t s_up = ctx->stage_stream[0], s_k = ctx->stage_stream[1], s_down = ctx->stage_stream[2];
    const size_t n_chunks = (n + chunk - 1) / chunk;
    constexpr long kSlots = x266hip_ctx::kSlots;

    struct Shared {
        std::mutex m;
        std::condition_variable cv;
        long launched = 0, drained = 0;                               // chunks whose kernel is enqueued / whose results have arrived
        bool stop = false;                                            // a failure on either side: both loops leave
        hipError_t down_error = hipSuccess;
    } sh;

    std::thread down;
    if (n_chunks > 1) try {
        down = std::thread([&]() {
            (void)hipSetDevice(ctx->device);
            for (size_t i = 0; i < n_chunks; ++i) {
                {
                    std::unique_lock<std::mutex> lk(sh.m);
                    sh.cv.wait(lk, [&] { return sh.stop || sh.launched > (long)i; });
                    if (sh.launched <= (long)i) return;              // stopped before this chunk was issued
                }
                const int slot = (int)(i % kSlots);
                const size_t done = i * chunk, cnt = n - done < chunk ? n - done : chunk;
                hipError_t e = hipStreamWaitEvent(s_down, ctx->stage_done[slot], 0);
                if (e == hipSuccess) e = hipMemcpyAsync((char *)out + done * out_unit, ctx->d_stage_out[slot], cnt * out_unit, hipMemcpyDeviceToHost, s_down);
                if (e == hipSuccess) e = hipStreamSynchronize(s_down);
                std::lock_guard<std::mutex> lk(sh.m);
                if (e != hipSuccess) { sh.down_error = e; sh.stop = true; }
                sh.drained = (long)i + 1;
                sh.cv.notify_all();
                if (e != hipSuccess) return;
            }
        });
    } catch (const std::system_error &) {                               // no thread to be had: nothing has been issued yet
        return fail(ctx, X266HIP_ENOMEM, "could not start the download thread of a host-pointer batch call");
    }
    hipError_t e = hipSuccess;
    const char *what = "";
#define STAGE(call) do { if (e == hipSuccess && rc == X266HIP_OK) { e = (call); what = #call; } } while (0)
    for (size_t i = 0; i < n_chunks && e == hipSuccess && rc == X266HIP_OK; ++i) {
        const int slot = (int)(i % kSlots);
        const size_t done = i * chunk, cnt = n - done < chunk ? n - done : chunk;
        if (n_chunks > 1) {                                             // the slot's previous chunk has left
            std::unique_lock<std::mutex> lk(sh.m);
            sh.cv.wait(lk, [&] { return sh.stop || (long)i - sh.drained < kSlots; });
            if (sh.stop) break;
        }
        STAGE(hipMemcpyAsync(ctx->d_stage_in[slot], (const char *)in + done * in_unit, cnt * in_unit, hipMemcpyHostToDevice, s_up));
        STAGE(hipEventRecord(ctx->stage_up[slot], s_up));
        STAGE(hipStreamWaitEvent(s_k, ctx->stage_up[slot], 0));
        if (e == hipSuccess) rc = launch_op(ctx, op, ctx->d_stage_in[slot], ctx->d_stage_out[slot], cnt, s_k);
        STAGE(hipEventRecord(ctx->stage_done[slot], s_k));
        if (n_chunks == 1) {                                            // a small call: no helper thread, the download follows in stream order
            STAGE(hipStreamWaitEvent(s_down, ctx->stage_done[slot], 0));
            STAGE(hipMemcpyAsync(out, ctx->d_stage_out[slot], cnt * out_unit, hipMemcpyDeviceToHost, s_down));
        } else if (e == hipSuccess && rc == X266HIP_OK) {
            std::lock_guard<std::mutex> lk(sh.m);
            sh.launched = (long)i + 1;
            sh.cv.notify_all();
        }
    }
#undef STAGE
    if (down.joinable()) {
        {
            std::lock_guard<std::mutex> lk(sh.m);
            if (e != hipSuccess || rc != X266HIP_OK) sh.stop = true;
            sh.cv.notify_all();
        }
        down.join();                                                    // it leaves once every issued chunk has been downloaded (or on a failure)
    }
    // Drain EVERY staging stream on every path: after a failure an already enqueued D2H copy must not still be
    // writing the caller's `out` once this function has returned (the caller may free it).
    for (hipStream_t st : ctx->stage_stream) {
        const hipError_t es = hipStreamSynchronize(st);
        if (e == hipSuccess && es != hipSuccess) { e = es; what = "hipStreamSynchronize(stage)"; }
    }
    if (rc != X266HIP_OK) return rc;
    if (e == hipSuccess && sh.down_error != hipSuccess) { e = sh.down_error; what = "download (helper thread)"; }
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, what, e);
    return X266HIP_OK;
}

int xDct32FwdBatch(x266hip_ctx *ctx, const int16_t *in, int16_t *out, size_t n) { return host_batch(ctx, 0, in, out, n, 2048, 2048); }
int xDct32InvBatch(x266hip_ctx *ctx, const int16_t *in, int16_t *out, size_t n) { return host_batch(ctx, 1, in, out, n, 2048, 2048); }
int xSatd8x8Batch(x266hip_ctx *ctx, const int16_t *diff, uint32_t *out, size_t n) { return host_batch(ctx, 2, diff, out, n, 128, 4); }

// ---- memory / stream helpers ---------------------------------------------------
int xHipMalloc(x266hip_ctx *ctx, void **d_ptr, size_t bytes)
{
    if (!ctx || !d_ptr) return X266HIP_EINVAL;
    X_DEV(ctx);
    *d_ptr = nullptr;
    if (bytes == 0) return X266HIP_OK;
    hipError_t e = hipMalloc(d_ptr, bytes);
    if (e != hipSuccess) return fail(ctx, X266HIP_ENOMEM, "hipMalloc", e);
    return X266HIP_OK;
}

int xHipHostAlloc(x266hip_ctx *ctx, void **h_ptr, size_t bytes)
{
    if (!ctx || !h_ptr) return X266HIP_EINVAL;
    X_DEV(ctx);
    *h_ptr = nullptr;
    if (bytes == 0) return X266HIP_OK;
    hipError_t e = hipHostMalloc(h_ptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, X266HIP_ENOMEM, "hipHostMalloc", e);
    return X266HIP_OK;
}

int xHipHostFree(x266hip_ctx *ctx, void *h_ptr)
{
    if (!ctx) return X266HIP_EINVAL;
    if (!h_ptr) return X266HIP_OK;
    X_DEV(ctx);
    X_HIP(ctx, hipHostFree(h_ptr));
    return X266HIP_OK;
}

int xHipFree(x266hip_ctx *ctx, void *d_ptr)
{
    if (!ctx) return X266HIP_EINVAL;
    if (!d_ptr) return X266HIP_OK;
    X_DEV(ctx);
    X_HIP(ctx, hipFree(d_ptr));
    return X266HIP_OK;
}

int xHipMemcpyH2D(x266hip_ctx *ctx, void *d_dst, const void *src, size_t bytes)
{
    if (!ctx || (bytes && (!d_dst || !src))) return X266HIP_EINVAL;
    X_DEV(ctx);
    X_HIP(ctx, hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return X266HIP_OK;
}

int xHipMemcpyD2H(x266hip_ctx *ctx, void *dst, const void *d_src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !d_src))) return X266HIP_EINVAL;
    X_DEV(ctx);
    X_HIP(ctx, hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return X266HIP_OK;
}

int xHipStreamSync(x266hip_ctx *ctx, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    X_DEV(ctx);
    X_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return X266HIP_OK;
}

int xHipStreamCreate(x266hip_ctx *ctx, void **stream)
{
    if (!ctx || !stream) return X266HIP_EINVAL;
    X_DEV(ctx);
    hipStream_t s = nullptr;
    X_HIP(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return X266HIP_OK;
}

int xHipStreamDestroy(x266hip_ctx *ctx, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (!stream) return X266HIP_OK;
    X_DEV(ctx);
    X_HIP(ctx, hipStreamDestroy((hipStream_t)stream));
    return X266HIP_OK;
}

struct x266hip_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

int xHipGraphBegin(x266hip_ctx *ctx, void *stream)
{
    if (!ctx) return X266HIP_EINVAL;
    if (!stream) return fail(ctx, X266HIP_EINVAL, "xHipGraphBegin: capture needs a stream of its own, not the NULL stream");
    X_DEV(ctx);
    X_HIP(ctx, hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return X266HIP_OK;
}

int xHipGraphEnd(x266hip_ctx *ctx, void *stream, x266hip_graph **graph)
{
    if (!ctx || !graph || !stream) return X266HIP_EINVAL;
    *graph = nullptr;
    X_DEV(ctx);
    hipGraph_t g = nullptr;
    X_HIP(ctx, hipStreamEndCapture((hipStream_t)stream, &g));
    if (!g) return fail(ctx, X266HIP_EDEVICE, "xHipGraphEnd: the capture was invalidated");
    hipGraphExec_t e = nullptr;
    const hipError_t rc = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    if (rc != hipSuccess) {
        (void)hipGraphDestroy(g);
        return fail(ctx, X266HIP_EDEVICE, "hipGraphInstantiate", rc);
    }
    x266hip_graph *out = new (std::nothrow) x266hip_graph;
    if (!out) {
        (void)hipGraphExecDestroy(e);
        (void)hipGraphDestroy(g);
        return X266HIP_ENOMEM;
    }
    out->graph = g;
    out->exec = e;
    *graph = out;
    return X266HIP_OK;
}

int xHipGraphLaunch(x266hip_ctx *ctx, x266hip_graph *graph, void *stream)
{
    if (!ctx || !graph || !graph->exec) return X266HIP_EINVAL;
    X_DEV(ctx);
    X_HIP(ctx, hipGraphLaunch(graph->exec, (hipStream_t)stream));
    return X266HIP_OK;
}

void xHipGraphFree(x266hip_ctx *ctx, x266hip_graph *graph)
{
    if (!graph) return;
    DeviceScope dev_scope_(ctx ? ctx->device : 0);
    if (graph->exec) (void)hipGraphExecDestroy(graph->exec);
    if (graph->graph) (void)hipGraphDestroy(graph->graph);
    delete graph;
}

int xHipEventCreate(x266hip_ctx *ctx, void **event)
{
    if (!ctx || !event) return X266HIP_EINVAL;
    X_DEV(ctx);
    hipEvent_t e = nullptr;
    X_HIP(ctx, hipEventCreate(&e));
    *event = (void *)e;
    return X266HIP_OK;
}

int xHipEventDestroy(x266hip_ctx *ctx, void *event)
{
    if (!ctx) return X266HIP_EINVAL;
    if (!event) return X266HIP_OK;
    X_DEV(ctx);
    X_HIP(ctx, hipEventDestroy((hipEvent_t)event));
    return X266HIP_OK;
}

int xHipEventRecord(x266hip_ctx *ctx, void *event, void *stream)
{
    if (!ctx || !event) return X266HIP_EINVAL;
    X_DEV(ctx);
    X_HIP(ctx, hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return X266HIP_OK;
}

int xHipEventElapsedMs(x266hip_ctx *ctx, void *start, void *stop, double *ms)
{
    if (!ctx || !start || !stop || !ms) return X266HIP_EINVAL;
    X_DEV(ctx);
    X_HIP(ctx, hipEventSynchronize((hipEvent_t)stop));
    float f = 0.f;
    X_HIP(ctx, hipEventElapsedTime(&f, (hipEvent_t)start, (hipEvent_t)stop));
    *ms = (double)f;
    return X266HIP_OK;
}

int xHipTimeKernel(x266hip_ctx *ctx, int op, const void *d_in, void *d_out, size_t n, int reps, void *stream,
                   double *ms_per_launch)
{
    if (!ctx || !ms_per_launch || reps < 1) return X266HIP_EINVAL;
    X_DEV(ctx);
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    X_HIP(ctx, hipEventCreate(&e0));
    X_HIP(ctx, hipEventCreate(&e1));
    int rc = X266HIP_OK;
    hipError_t e = hipEventRecord(e0, s);
    for (int i = 0; i < reps && rc == X266HIP_OK && e == hipSuccess; ++i) rc = launch_op(ctx, op, d_in, d_out, n, s);
    if (e == hipSuccess) e = hipEventRecord(e1, s);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, X266HIP_EDEVICE, "event timing", e);
    *ms_per_launch = (double)ms / reps;
    return X266HIP_OK;
}

}  // extern "C"
