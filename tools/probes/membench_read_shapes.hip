// membench_read_shapes: what shape of a read-only nontemporal stream is fastest on this box?  (round 4: the one-tile-per-wave stream that
// served as "read ceiling" in round 2 turned out SLOWER than the SATD kernel on some boxes -- wave-slot bound, not HBM bound.)
//   per wave: `iters` iterations of KB KiB (KB x 1 KiB-linear 16 B/lane nt loads), DEPTH iterations in flight (register ping-pong),
//   words XORed per lane; WPW waves per workgroup; `lds` bytes of dynamic LDS per workgroup = cap on resident workgroups.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/membench_read_shapes tools/probes/membench_read_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int KB, int DEPTH>
__global__ __launch_bounds__(256) void k(const char *__restrict__ in, int *__restrict__ out, size_t n_kib, unsigned iters)
{
    extern __shared__ __attribute__((aligned(16))) char cap[];
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    size_t p = wave * iters * KB;                                    // in KiB
    const size_t end = p + (size_t)iters * KB < n_kib ? p + (size_t)iters * KB : n_kib;
    if (p >= end) return;
    const char *src = in + lane * 16;
    v4i a[KB], b[KB];
    int acc = 0;
#pragma unroll
    for (int i = 0; i < KB; ++i) a[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(src + (p + i) * 1024));
    for (; p < end; p += KB) {
        if (DEPTH == 2 && p + KB < end) {
#pragma unroll
            for (int i = 0; i < KB; ++i) b[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(src + (p + KB + i) * 1024));
        }
#pragma unroll
        for (int i = 0; i < KB; ++i) acc ^= a[i][0] ^ a[i][1] ^ a[i][2] ^ a[i][3];
        if (DEPTH == 2) {
#pragma unroll
            for (int i = 0; i < KB; ++i) a[i] = b[i];
        } else if (p + KB < end) {
#pragma unroll
            for (int i = 0; i < KB; ++i) a[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(src + (p + KB + i) * 1024));
        }
    }
    if (acc == 0x12345678) out[wave] = acc;
}

template <int KB, int DEPTH>
static double run(const char *in, int *out, size_t n_kib, unsigned iters, unsigned wpw, size_t lds, bool print = true)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t waves = (n_kib + (size_t)iters * KB - 1) / ((size_t)iters * KB);
    const unsigned grid = (unsigned)((waves + wpw - 1) / wpw);
    for (int i = 0; i < 60; ++i) hipLaunchKernelGGL((k<KB, DEPTH>), dim3(grid), dim3(64 * wpw), lds, 0, in, out, n_kib, iters);
    hipDeviceSynchronize();
    float best = 1e9f, sum = 0;
    const int R = 30;
    for (int r = 0; r < R; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KB, DEPTH>), dim3(grid), dim3(64 * wpw), lds, 0, in, out, n_kib, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    const double bytes = (double)n_kib * 1024;
    if (print) { printf("KB %d depth %d iters %3u waves/wg %u lds/wg %6zu : mean %.4f ms (min %.4f)  %.3f TB/s\n", KB, DEPTH, iters, wpw, lds, sum / R, best, bytes / (sum / R) / 1e9); fflush(stdout); }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return sum / R;
}

int main()
{
    const size_t n_kib = (size_t)2 << 20;                            // 2 GiB
    char *in; int *out;
    hipMalloc(&in, n_kib * 1024); hipMalloc(&out, 64 << 20);
    hipMemset(in, 0x5a, n_kib * 1024);
    for (int rnd = 0; rnd < 2; ++rnd) {
        printf("# round %d\n", rnd);
        for (unsigned wpw : {1u, 4u})
            for (size_t lds_per_wave : {(size_t)4096, (size_t)8192, (size_t)16384})
                for (unsigned iters : {1u, 2u, 4u, 8u, 16u, 64u}) {
                    const size_t lds = lds_per_wave * wpw;
                    run<2, 1>(in, out, n_kib, iters, wpw, lds);
                    run<2, 2>(in, out, n_kib, iters, wpw, lds);
                    run<4, 1>(in, out, n_kib, iters, wpw, lds);
                    run<4, 2>(in, out, n_kib, iters, wpw, lds);
                }
    }
    return 0;
}
