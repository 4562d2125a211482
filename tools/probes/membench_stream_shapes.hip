// membench_stream_shapes: which launch shape of an arithmetic-free nontemporal stream is fastest on this box -- copy (nt loads, "sc1 nt"
// stores) and write-only; the read-only twin is membench_read_shapes.  Per wave: `iters` iterations of KB KiB (KB 1 KiB-linear 16 B/lane
// accesses); WPW waves per workgroup; `lds` bytes of dynamic LDS per workgroup = cap on resident workgroups.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/membench_stream_shapes tools/probes/membench_stream_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sc1nt(char *p, v4i v) { asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }

template <int KB, int MODE>   // MODE 0 copy, 1 write
__global__ __launch_bounds__(256) void k(const char *__restrict__ in, char *__restrict__ out, size_t n_kib, unsigned iters)
{
    extern __shared__ __attribute__((aligned(16))) char cap[];
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    size_t p = wave * iters * KB;
    const size_t end = p + (size_t)iters * KB < n_kib ? p + (size_t)iters * KB : n_kib;
    if (p >= end) return;
    const char *src = in + lane * 16;
    char *dst = out + lane * 16;
    v4i a[KB];
    for (; p < end; p += KB) {
#pragma unroll
        for (int i = 0; i < KB; ++i) a[i] = MODE == 0 ? __builtin_nontemporal_load(reinterpret_cast<const v4i *>(src + (p + i) * 1024)) : v4i{(int)p, lane, i, 0};
#pragma unroll
        for (int i = 0; i < KB; ++i) st_sc1nt(dst + (p + i) * 1024, a[i]);
    }
}

template <int KB, int MODE>
static void run(const char *in, char *out, size_t n_kib, unsigned iters, unsigned wpw, size_t lds)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t waves = (n_kib + (size_t)iters * KB - 1) / ((size_t)iters * KB);
    const unsigned grid = (unsigned)((waves + wpw - 1) / wpw);
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((k<KB, MODE>), dim3(grid), dim3(64 * wpw), lds, 0, in, out, n_kib, iters);
    (void)hipDeviceSynchronize();
    float best = 1e9f, sum = 0;
    const int R = 30;
    for (int r = 0; r < R; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<KB, MODE>), dim3(grid), dim3(64 * wpw), lds, 0, in, out, n_kib, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    const double bytes = (double)n_kib * 1024 * (MODE == 0 ? 2 : 1);
    printf("%s KB %d iters %3u waves/wg %u lds/wave %6zu : mean %.4f ms (min %.4f)  %.3f TB/s\n", MODE == 0 ? "copy " : "write", KB, iters, wpw, lds / wpw, sum / R, best, bytes / (sum / R) / 1e9);
    fflush(stdout);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

int main()
{
    const size_t n_kib = (size_t)2 << 20;                            // 2 GiB
    char *in, *out;
    (void)hipMalloc(&in, n_kib * 1024); (void)hipMalloc(&out, n_kib * 1024);
    (void)hipMemset(in, 0x5a, n_kib * 1024);
    for (int rnd = 0; rnd < 2; ++rnd) {
        printf("# round %d\n", rnd);
        for (unsigned wpw : {1u, 2u, 4u})
            for (size_t lds_per_wave : {(size_t)4096, (size_t)8192, (size_t)12288, (size_t)16384})
                for (unsigned iters : {1u, 2u, 4u}) {
                    const size_t lds = lds_per_wave * wpw;
                    run<2, 0>(in, out, n_kib, iters, wpw, lds);
                    run<4, 0>(in, out, n_kib, iters, wpw, lds);
                    run<2, 1>(in, out, n_kib, iters, wpw, lds);
                    run<4, 1>(in, out, n_kib, iters, wpw, lds);
                }
    }
    return 0;
}
