// diag_kernels.hip -- what THIS box's memory system gives the launch shape the streaming kernels use (gfx950).
//
// Every HBM-bound kernel of the library is priced against the 8 TB/s spec, but boxes of the pool differ by 3-10 % in what a
// plain stream reaches (profiles/README.md), which is more than most effects worth chasing.  xHipMemCeilingDev runs the
// two arithmetic-free streams on the caller's own buffers so that a bench line (and the perf-floor tests) can put
// "fraction of this box's copy / read rate" next to "fraction of the spec":
//   kind 0  copy         dst[i] = src[i], 16 bytes per lane, 1 KiB-linear nontemporal loads, "sc1 nt" stores
//   kind 1  read stream  the same loads, nothing stored but one 32-bit XOR checksum per wave (4 B per 2 KiB read)
//   kind 2  write stream nothing loaded, dst = a counter pattern, the same stores
//   kind 3  read probe   the loads of kind 1 and NO store, except for pieces whose XOR equals X266_MEM_PROBE_MAGIC: what the
//                        memory system delivers when nothing at all flows back (a wave's checksum store costs the read stream
//                        2 % at twenty resident waves per CU and 17 % at eight, profiles/r04_membench_read_epilogue.txt)
// each in the launch shape that measured fastest for it (see launch_mem_ceiling).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "x266_device.hpp"

namespace x266 {
namespace {

// XOR over the wave without LDS traffic (four DPP steps inside each row of 16 lanes, then the four rows through SGPRs): the
// stream's waves live for one memory round trip, and six dependent ds_bpermute would stretch that by a fifth
__device__ __forceinline__ int wave_xor(int x)
{
    x ^= __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);           // quad_perm [1,0,3,2]
    x ^= __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);           // quad_perm [2,3,0,1]
    x ^= __builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true);          // row_half_mirror
    x ^= __builtin_amdgcn_mov_dpp(x, 0x140, 0xF, 0xF, true);          // row_mirror
    return __builtin_amdgcn_readlane(x, 0) ^ __builtin_amdgcn_readlane(x, 16) ^ __builtin_amdgcn_readlane(x, 32) ^ __builtin_amdgcn_readlane(x, 48);
}

// One wave moves KB KiB: KB 1 KiB-linear instructions of 16 bytes per lane.  Pieces of 2 KiB are the unit of the read
// checksum (one per wave at KB = 2, two at KB = 4).
template <int KIND, int KB>
__global__ __launch_bounds__(256) void mem_ceiling_kernel(const char *__restrict__ src, char *__restrict__ dst, size_t n_chunks, int probe)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char occupancy_cap[];   // never touched: only its size matters
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // provably wave-uniform
    const size_t c0 = wave * (64 * KB) + lane;                                     // the lane's chunks: c0 + 64 i
    v4i a[KB];
    const bool whole = (wave + 1) * (64 * KB) <= n_chunks;                         // wave-uniform: every wave but the batch's last
    if (whole) {
        // no per-lane conditions here: a select on a loaded value would make every load wait for the one before it
#pragma unroll
        for (int i = 0; i < KB; ++i) a[i] = KIND == 2 ? v4i{(int)(c0 + 64 * (size_t)i), 0, 0, 0} : load16<true>(src + (c0 + 64 * (size_t)i) * 16);
        if (KIND != 1) {
#pragma unroll
            for (int i = 0; i < KB; ++i) store16_sc1nt(dst + (c0 + 64 * (size_t)i) * 16, a[i]);
            return;
        }
    } else {
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const size_t c = c0 + 64 * (size_t)i;
            a[i] = KIND == 2 ? v4i{(int)c, 0, 0, 0} : v4i{0, 0, 0, 0};             // kind 2: dword 0 of chunk c = (uint32)c, the rest 0
            if (KIND != 2 && c < n_chunks) a[i] = load16<true>(src + c * 16);
            if (KIND != 1 && c < n_chunks) store16_sc1nt(dst + c * 16, a[i]);
        }
        if (KIND != 1) return;
    }
#pragma unroll
    for (int p = 0; p < KB / 2; ++p) {
        const int x = wave_xor(a[2 * p][0] ^ a[2 * p][1] ^ a[2 * p][2] ^ a[2 * p][3] ^ a[2 * p + 1][0] ^ a[2 * p + 1][1] ^ a[2 * p + 1][2] ^ a[2 * p + 1][3]);
        const size_t piece = wave * (KB / 2) + p;
        if (probe && x != (int)X266_MEM_PROBE_MAGIC) continue;                      // wave-uniform (read probe: practically never stores)
        if (lane == 0 && piece * 128 < n_chunks) reinterpret_cast<int *>(dst)[piece] = x;
    }
}

}  // namespace

// Launch shapes = the fastest of the sweeps in profiles/r04_membench_stream_shapes.txt: few resident waves, short-lived workgroups
// in dispatch (= address) order.  read: four-wave workgroups, 4 KiB per wave, 32 KiB of LDS charged per workgroup (twenty waves per
// CU: the checksum store needs the company), the probe 64 KiB (two workgroups = eight waves per CU); copy / write: one-wave workgroups, 2 KiB per wave, 8 KiB (twenty waves per CU) / 16 KiB (ten) charged.
hipError_t launch_mem_ceiling(int kind, const void *d_src, void *d_dst, size_t bytes, hipStream_t stream)
{
    const size_t n_chunks = bytes / 16;
    if (n_chunks == 0) return hipSuccess;
    if (kind == 1 || kind == 3) {
        const size_t wgs = (n_chunks + 1023) / 1024;                               // 4 waves x 4 KiB
        if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
        hipLaunchKernelGGL((mem_ceiling_kernel<1, 4>), dim3((unsigned)wgs), dim3(256), kind == 1 ? 32768 : 65536, stream, (const char *)d_src, (char *)d_dst, n_chunks, kind == 3 ? 1 : 0);
        return hipGetLastError();
    }
    const size_t wgs = (n_chunks + 127) / 128;
    if (wgs > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (kind == 0) hipLaunchKernelGGL((mem_ceiling_kernel<0, 2>), dim3((unsigned)wgs), dim3(64), 8192, stream, (const char *)d_src, (char *)d_dst, n_chunks, 0);
    else           hipLaunchKernelGGL((mem_ceiling_kernel<2, 2>), dim3((unsigned)wgs), dim3(64), 16384, stream, (const char *)d_src, (char *)d_dst, n_chunks, 0);
    return hipGetLastError();
}

}  // namespace x266
