// membench_fanout: what this box's memory system gives a stream that reads 1 byte for every 2 it writes -- the traffic mix of the
// fused forward + inverse DCT32 kernel (2 KiB in, 2 + 2 KiB out per block), so that its hbm_frac can be priced against the mix's own
// ceiling instead of the 1:1 copy's.  Arithmetic-free: dst0[i] = dst1[i] = src[i], nt loads, "sc1 nt" stores, 1 KiB-linear accesses.
// Per wave: `iters` iterations of KB KiB; WPW waves per workgroup; `lds` bytes of dynamic LDS per workgroup = cap on resident workgroups.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/membench_fanout tools/probes/membench_fanout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sc1nt(char *p, v4i v) { asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }

template <int KB, int OUTS>   // OUTS 1: copy, 2: fan-out
__global__ __launch_bounds__(256) void k(const char *__restrict__ in, char *__restrict__ out0, char *__restrict__ out1, size_t n_kib, unsigned iters)
{
    extern __shared__ __attribute__((aligned(16))) char cap[];
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    size_t p = wave * iters * KB;
    const size_t end = p + (size_t)iters * KB < n_kib ? p + (size_t)iters * KB : n_kib;
    if (p >= end) return;
    v4i a[KB];
    for (; p < end; p += KB) {
#pragma unroll
        for (int i = 0; i < KB; ++i) a[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(in + lane * 16 + (p + i) * 1024));
#pragma unroll
        for (int i = 0; i < KB; ++i) st_sc1nt(out0 + lane * 16 + (p + i) * 1024, a[i]);
        if (OUTS == 2) {
#pragma unroll
            for (int i = 0; i < KB; ++i) st_sc1nt(out1 + lane * 16 + (p + i) * 1024, a[i]);
        }
    }
}

template <int KB, int OUTS>
static void run(const char *in, char *out0, char *out1, size_t n_kib, unsigned iters, unsigned wpw, size_t lds)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t waves = (n_kib + (size_t)iters * KB - 1) / ((size_t)iters * KB);
    const unsigned grid = (unsigned)((waves + wpw - 1) / wpw);
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((k<KB, OUTS>), dim3(grid), dim3(64 * wpw), lds, 0, in, out0, out1, n_kib, iters);
    (void)hipDeviceSynchronize();
    float best = 1e9f, sum = 0;
    const int R = 30;
    for (int r = 0; r < R; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<KB, OUTS>), dim3(grid), dim3(64 * wpw), lds, 0, in, out0, out1, n_kib, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    const double bytes = (double)n_kib * 1024 * (1 + OUTS);
    printf("%s KB %d iters %3u waves/wg %u lds/wave %6zu : mean %.4f ms (min %.4f)  %.3f TB/s\n", OUTS == 1 ? "copy   " : "fan-out", KB, iters, wpw, lds / wpw, sum / R, best, bytes / (sum / R) / 1e9);
    fflush(stdout);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

int main()
{
    const size_t n_kib = (size_t)2 << 20;                            // 2 GiB in: the fused kernel's 2^20 blocks
    char *in, *out0, *out1;
    (void)hipMalloc(&in, n_kib * 1024); (void)hipMalloc(&out0, n_kib * 1024); (void)hipMalloc(&out1, n_kib * 1024);
    (void)hipMemset(in, 0x5a, n_kib * 1024);
    for (int rnd = 0; rnd < 2; ++rnd) {
        printf("# round %d\n", rnd);
        run<2, 1>(in, out0, out1, n_kib, 1, 1, 8192);                // the library's copy shape
        for (unsigned wpw : {1u, 4u})
            for (size_t lds_per_wave : {(size_t)4096, (size_t)8192, (size_t)12288, (size_t)16384, (size_t)32768})
                for (unsigned iters : {1u, 2u}) {
                    const size_t lds = lds_per_wave * wpw;
                    if (lds > 65536) continue;
                    run<2, 2>(in, out0, out1, n_kib, iters, wpw, lds);
                    run<4, 2>(in, out0, out1, n_kib, iters, wpw, lds);
                }
    }
    return 0;
}
