// membench_write_order: does a write-only stream of LONG-LIVED waves (each wave stores `iters` pieces of 1 KiB, like the intra predictor's
// 28 predictions per wave) depend on the ORDER in which the waves walk the buffer?  blocked: wave g writes pieces g*iters .. (its own
// contiguous run; the chip's open write window = resident waves x iters KiB); interleaved: piece = t * n_waves + g (all resident waves
// write one compact window that moves through the buffer).  "sc1 nt" 16 B/lane stores, 2 GiB.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/membench_write_order tools/probes/membench_write_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sc1nt(char *p, v4i v) { asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }

template <int ORDER>   // 0 blocked, 1 interleaved over all waves of the launch, 2 interleaved over the workgroup's waves only
__global__ __launch_bounds__(256) void k(char *__restrict__ out, size_t n_kib, unsigned iters, unsigned spin)
{
    extern __shared__ __attribute__((aligned(16))) char cap[];
    const int lane = threadIdx.x & 63;
    const unsigned wpw = blockDim.x >> 6;
    const size_t wave = (size_t)blockIdx.x * wpw + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * wpw;
    v4i v = {(int)wave, lane, 0, 0};
    for (unsigned t = 0; t < iters; ++t) {
        size_t piece;
        if (ORDER == 0) piece = wave * iters + t;
        else if (ORDER == 1) piece = (size_t)t * n_waves + wave;
        else piece = ((size_t)blockIdx.x * iters + t) * wpw + (threadIdx.x >> 6);
        for (unsigned s = 0; s < spin; ++s) v[2] = v[2] * 3 + 1;          // stand-in for the work between two stores
        if (piece < n_kib) st_sc1nt(out + piece * 1024 + lane * 16, v);
    }
}

template <int ORDER>
static void run(char *out, size_t n_kib, unsigned iters, unsigned wpw, size_t lds, unsigned spin)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t waves = (n_kib + iters - 1) / iters;
    const unsigned grid = (unsigned)((waves + wpw - 1) / wpw);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<ORDER>), dim3(grid), dim3(64 * wpw), lds, 0, out, n_kib, iters, spin);
    (void)hipDeviceSynchronize();
    float sum = 0; const int R = 20;
    for (int r = 0; r < R; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<ORDER>), dim3(grid), dim3(64 * wpw), lds, 0, out, n_kib, iters, spin);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); sum += ms;
    }
    printf("%-22s iters %2u waves/wg %u lds/wg %6zu spin %3u : %.4f ms  %.3f TB/s\n", ORDER == 0 ? "blocked" : ORDER == 1 ? "interleaved (launch)" : "interleaved (workgroup)", iters, wpw, lds, spin, sum / R, (double)n_kib * 1024 / (sum / R) / 1e9);
    fflush(stdout);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

int main()
{
    const size_t n_kib = (size_t)2 << 20;
    char *out;
    (void)hipMalloc(&out, n_kib * 1024);
    for (unsigned spin : {0u, 40u})
        for (unsigned wpw : {1u, 4u})
            for (size_t lds_per_wave : {(size_t)0, (size_t)4096, (size_t)8192, (size_t)16384})
                for (unsigned iters : {2u, 7u, 28u}) {
                    run<0>(out, n_kib, iters, wpw, lds_per_wave * wpw, spin);
                    run<1>(out, n_kib, iters, wpw, lds_per_wave * wpw, spin);
                    if (wpw > 1) run<2>(out, n_kib, iters, wpw, lds_per_wave * wpw, spin);
                }
    return 0;
}
