// x266_device.hpp -- shared device-side types and helpers (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/x266hip.h"

namespace x266 {

typedef int v4i  __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// 16-byte global accesses; NT = streaming (non-temporal) cache policy.
template <bool NT>
__device__ __forceinline__ v4i load16(const void *p)
{
    const v4i *q = reinterpret_cast<const v4i *>(p);
    if (NT) return __builtin_nontemporal_load(q);
    return *q;
}

template <bool NT>
__device__ __forceinline__ void store16(void *p, const v4i &v)
{
    v4i *q = reinterpret_cast<v4i *>(p);
    if (NT) __builtin_nontemporal_store(v, q);
    else    *q = v;
}

// Streaming store with the cache-policy bits that measured best for write-once output on gfx950
// (profiles/r01_membench_cache_policy.txt: "sc1 nt" copy 6.65 TB/s, "nt" 6.56, default 6.06).  The
// compiler has no spelling for this combination, hence the instruction itself.  The s_nop covers
// the gfx9 hazard "VMEM store of more than 64 bits, then a VALU write of its data VGPRs" (one wait
// state), which the compiler's hazard recognizer cannot see through inline asm -- without it the
// next address computation may overwrite the first 8 bytes of the data still being read.
__device__ __forceinline__ void store16_sc1nt(void *p, const v4i &v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" : : "v"(p), "v"(v));
}

// the same store with agent scope only (no streaming hint): what the WRITE-ONLY intra predictor (long-lived waves) wants -- -1 to -6 % paired on the same
// buffers at 1.0e6 / 2.1e6 / 4.2e6 predictions (profiles/r05_result_stores.txt).  Every kernel with a read stream beside its writes (the transforms, the fused
// intra kernel) and the arithmetic-free copy / write streams themselves are fastest with "sc1 nt"
__device__ __forceinline__ void store16_sc1(void *p, const v4i &v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v));
}

// ---- hand-counted waits ---------------------------------------------------------------------------------------------------
// The LDS-DMA kernels (dct32_fwdinv_kernel, satd8x8_dma_kernel, satd8x8_from_tiles_dma_kernel) issue their loads and streaming stores as
// inline asm, invisible to the compiler's vmcnt bookkeeping, and wait by COUNT: a wave's vector memory operations retire in issue
// order, so "fetch f has landed" = at most {the operations issued after f} outstanding.  Every such count is written as a sum of
// the per-call instruction counts below (kDmaPer*, kStoresPer*), which the issuing helpers static_assert against what they emit.
// The check that the sums are right is by execution: `make -C x266_amd/csrc waits0` builds libx266hip_waits0.so with X266_WAIT_ALL,
// where every counted wait waits for EVERYTHING, and tests/test_gpu_waits.py compares the two libraries byte for byte on ragged,
// steady-state and single-block runs -- a count that is too high reads an LDS slot before its DMA landed and shows there.
__device__ __forceinline__ void wait_vmcnt(unsigned n)
{
#ifdef X266_WAIT_ALL
    (void)n;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    switch (n) {
#define X266_WAIT(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    X266_WAIT(1) X266_WAIT(2) X266_WAIT(3) X266_WAIT(4) X266_WAIT(5) X266_WAIT(6) X266_WAIT(7) X266_WAIT(8) X266_WAIT(9) X266_WAIT(10)
    X266_WAIT(11) X266_WAIT(12) X266_WAIT(13) X266_WAIT(14) X266_WAIT(15) X266_WAIT(16) X266_WAIT(17) X266_WAIT(18) X266_WAIT(19) X266_WAIT(20)
    X266_WAIT(21) X266_WAIT(22) X266_WAIT(23) X266_WAIT(24) X266_WAIT(25) X266_WAIT(26) X266_WAIT(27) X266_WAIT(28) X266_WAIT(29) X266_WAIT(30)
    X266_WAIT(31) X266_WAIT(32) X266_WAIT(33) X266_WAIT(34) X266_WAIT(35) X266_WAIT(36) X266_WAIT(37) X266_WAIT(38) X266_WAIT(39) X266_WAIT(40)
#undef X266_WAIT
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // 0, and anything beyond the counter's range
    }
#endif
}
constexpr unsigned kWaitVmcntMax = 40;

// ---- cross-lane sums without LDS traffic (a __shfl_xor is a ds_bpermute: address arithmetic, an LDS instruction and its latency) ----
// x + (lane ^ 32's x): gfx950's v_permlane32_swap exchanges the upper half of one register with the lower half of another
__device__ __forceinline__ uint32_t sum_with_other_half(uint32_t x)
{
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);   // r[0] = {lo, lo}, r[1] = {hi, hi}
    return r[0] + r[1];
}
// sum over the lane's row of 16 lanes (every lane of the row ends up with it): two quad permutes, then the two row mirrors
__device__ __forceinline__ uint32_t sum_over_row16(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true);   // row_half_mirror
    x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x140, 0xF, 0xF, true);   // row_mirror
    return x;
}

// Result streams -- a few bytes written per hundred read (SATD / SAD costs): agent-scope "sc1" stores without the streaming hint.  Measured on
// the SATD batch, paired over eight allocation sets (profiles/r05_result_stores.txt): plain or sc0 0.360 ms, nt 0.334-0.360, sc1 nt 0.329-0.339,
// sc1 or sc0 sc1 0.325-0.330 -- and the kernel no longer follows where its small output buffer landed.
__device__ __forceinline__ void store_result16(void *p, const v4i &v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_result4(uint32_t *p, uint32_t v)    // a relaxed agent-scope store IS "global_store_dword ... sc1", and the compiler still counts it
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// SM: 0 plain, 1 nontemporal, 2 "sc1 nt"
template <int SM>
__device__ __forceinline__ void store16m(void *p, const v4i &v)
{
    if (SM == 2) store16_sc1nt(p, v);
    else         store16<SM == 1>(p, v);
}

struct LaunchCfg {
    int cu_count;             // compute units of the device
    int adaptive;             // shrink units_per_wave on small batches so the grid still fills the chip
    int units_per_wave;       // streaming launch: consecutive units (DCT blocks / 32-block SATD groups / tiles) per wave; SATD batch: 0 = the kernel's own default
    int wg_threads;           // workgroup size, multiple of 64; SATD batch: 0 = the kernel's own default
    int lds_bytes_per_wave;   // LDS charged per wave (what the kernel uses + padding): 160 KiB / this = cap on resident waves per CU; SATD batch: 0 = default
    int shape;                // SATD batch: 0 kernel by batch size, 1 staged kernel, 3 LDS-DMA kernel; fused forward + inverse: input slots per wave (0 = 2, or 3)
};

// SATD batch: from this many blocks on the LDS-DMA kernel runs (satd_kernels.hip, launch_satd8x8)
constexpr size_t kSatdDmaMinBlocks = (size_t)3 << 20;

struct DctOps;
struct TileTab;

// Units (blocks, groups, tiles) one wave loops over: the configured count, reduced on small batches
// until the launch has at least two full rounds of resident waves (about 40 per CU).
inline unsigned units_per_wave_for(const LaunchCfg &cfg, size_t n_units)
{
    unsigned u = cfg.units_per_wave < 1 ? 1u : (unsigned)cfg.units_per_wave;
    if (cfg.adaptive) {
        const size_t fill = (size_t)cfg.cu_count * 40u;
        const size_t cap = n_units / fill;
        if (cap < u) u = cap < 1 ? 1u : (unsigned)cap;
    }
    return u;
}

hipError_t launch_intra32_predict(const x266_intra_ref_t *d_refs, const uint8_t *d_modes, const uint32_t *d_ref_index,
                                  uint8_t *d_pred, size_t n, int rounds, hipStream_t stream);
hipError_t launch_intra32_residual_dct32(const x266_intra_ref_t *d_refs, const uint8_t *d_modes, const uint32_t *d_ref_index, const uint8_t *d_src,
                                         int16_t *d_coef, size_t n, const DctOps *d_fwd_ops, hipStream_t stream);
hipError_t launch_intra32_costs(const x266_intra_ref_t *d_refs, const uint8_t *d_src, uint32_t *d_costs, uint8_t *d_best_mode,
                                size_t n, hipStream_t stream);
hipError_t launch_satd8x8_butterfly(const int16_t *d_diff, uint32_t *d_out, size_t n_blocks, const LaunchCfg &cfg, hipStream_t stream);
hipError_t launch_transform_tiles(bool inverse, const int16_t *d_in, int16_t *d_out, size_t n_tiles, const uint32_t *d_tile_offsets,
                                  const uint8_t *d_tile_class, const TileTab *d_tab, const LaunchCfg &cfg, hipStream_t stream);
hipError_t launch_dct32_butterfly(const int16_t *d_in, int16_t *d_out, size_t n_blocks, hipStream_t stream);
hipError_t launch_dct32_pass(const int16_t *d_in, int16_t *d_out, size_t n_blocks, int shift, const DctOps *d_fwd_ops, hipStream_t stream);
hipError_t launch_dct32_fwdinv(const int16_t *d_in, int16_t *d_coef, int16_t *d_recon, size_t n_blocks,
                               const DctOps *d_fwd_ops, const DctOps *d_inv_acc_ops, const LaunchCfg &cfg, hipStream_t stream);
hipError_t launch_dct32(bool inverse, const int16_t *d_in, int16_t *d_out, size_t n_blocks,
                        const DctOps *d_ops, const LaunchCfg &cfg, hipStream_t stream);
hipError_t launch_transform_small(int log2n, const int16_t *d_in, int16_t *d_out, size_t n_blocks, const DctOps *d_ops,
                                  const uint32_t *d_offsets, const LaunchCfg &cfg, hipStream_t stream);
hipError_t launch_dct32_from_tiles(const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int16_t *d_out,
                                   int width, int height, const DctOps *d_fwd_ops, const LaunchCfg &cfg, hipStream_t stream);
hipError_t launch_transform_small_inv(int log2n, const int16_t *d_in, int16_t *d_out, size_t n_blocks, const DctOps *d_ops,
                                      const uint32_t *d_offsets, const LaunchCfg &cfg, hipStream_t stream);
hipError_t launch_satd8x8_from_tiles(const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, uint32_t *d_out,
                                     int width, int height, int shape, hipStream_t stream);
hipError_t launch_frame_lanes(const int16_t *d_dct_in, int16_t *d_dct_out, size_t n_dct, const DctOps *d_fwd_ops,
                              const int16_t *d_diff, uint32_t *d_satd_out, size_t n_satd, const LaunchCfg &satd_cfg, hipStream_t stream);
hipError_t launch_satd8x8(const int16_t *d_diff, uint32_t *d_out, size_t n_blocks,
                          const LaunchCfg &cfg, hipStream_t stream);
hipError_t launch_satd_search(const uint8_t *d_cur, long long cur_stride, const uint8_t *d_ref, long long ref_stride,
                               int width, int height, int range, x266_me_result_t *d_best, uint32_t *d_costs,
                               int tile_rows, uint32_t *d_coef_scratch, int cu_count, hipStream_t stream);
hipError_t launch_sad_search(const uint8_t *d_cur, long long cur_stride, const uint8_t *d_ref, long long ref_stride,
                              int width, int height, int range, x266_me_result_t *d_best, uint32_t *d_costs,
                              int tile_rows, hipStream_t stream);
hipError_t launch_sad(int edge, const uint8_t *d_a, const uint8_t *d_b, uint32_t *d_out, size_t n_blocks, int waves_per_wg, int lds_per_wg, hipStream_t stream);
hipError_t launch_tile_convert(bool pack, x266_ref_block_t *d_tiles, uint8_t *d_y, uint8_t *d_u, uint8_t *d_v,
                               long long strd_y, long long strd_c, int width, int height, hipStream_t stream);
hipError_t launch_residual_luma(int block_edge, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int16_t *d_res,
                                int width, int height, hipStream_t stream);
hipError_t launch_residual_chroma(int block_edge, const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int16_t *d_res_u, int16_t *d_res_v,
                                  size_t block_pitch, int width, int height, hipStream_t stream);
hipError_t launch_dct32_chroma_from_tiles(const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int16_t *d_out_u, int16_t *d_out_v,
                                          size_t block_pitch, int width, int height, const DctOps *d_fwd_ops, const LaunchCfg &cfg, hipStream_t stream);
hipError_t launch_dct32_ctu_from_tiles(const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, int16_t *d_out,
                                       int width, int height, const DctOps *d_fwd_ops, const LaunchCfg &cfg, hipStream_t stream);
hipError_t launch_satd8x8_chroma_from_tiles(const x266_ref_block_t *d_cur, const x266_ref_block_t *d_pred, uint32_t *d_out_u, uint32_t *d_out_v,
                                            size_t pitch, int width, int height, hipStream_t stream);
hipError_t launch_mem_ceiling(int kind, const void *d_src, void *d_dst, size_t bytes, hipStream_t stream);
hipError_t launch_fill_residual(int16_t *d_dst, size_t n_samples, uint64_t seed,
                                uint64_t first_index, const LaunchCfg &cfg, hipStream_t stream);

}  // namespace x266
