// membench4.hip -- developer probe #4: from the classic copy towards the DCT kernel's structure, one step at a time
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(256) void k0_classic(const v4i* __restrict__ in, v4i* __restrict__ out, const v4i* tab, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v4i v = in[i]; v[0] ^= 1; out[i] = v; }
// one wave = 2 KiB, two loads; PAT 0 linear, 1 row-per-lane;  TAB: also read 48 B/lane of constants
template <int PAT, bool TAB> __global__ __launch_bounds__(256) void k_wave2k(const v4i* __restrict__ in, v4i* __restrict__ out, const v4i* __restrict__ tab, size_t n) {
    const int lane = threadIdx.x & 63; const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    size_t o0, o1; if (PAT == 0) { o0 = lane * 16; o1 = o0 + 1024; } else { o0 = (lane & 31) * 64 + (lane >> 5) * 32; o1 = o0 + 16; }
    const char* s = (const char*)in + wave * 2048; char* d = (char*)out + wave * 2048;
    v4i a = *(const v4i*)(s + o0), b = *(const v4i*)(s + o1);
    if (TAB) { v4i t0 = tab[lane * 4], t1 = tab[lane * 4 + 1], t2 = tab[lane * 4 + 2]; a ^= t0; b ^= t1 + t2; } else { a[0] ^= 1; }
    *(v4i*)(d + o0) = a; *(v4i*)(d + o1) = b; }
// BPW consecutive blocks per wave, with or without register prefetch
template <int BPW, bool PREFETCH> __global__ __launch_bounds__(256) void k_run(const v4i* __restrict__ in, v4i* __restrict__ out, const v4i* __restrict__ tab, size_t n) {
    const int lane = threadIdx.x & 63; const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t o0 = (lane & 31) * 64 + (lane >> 5) * 32, o1 = o0 + 16;
    const char* s = (const char*)in + wave * BPW * 2048; char* d = (char*)out + wave * BPW * 2048;
    v4i t0 = tab[lane * 4], t1 = tab[lane * 4 + 1];
    if (PREFETCH) {
        v4i a = *(const v4i*)(s + o0), b = *(const v4i*)(s + o1);
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            v4i na = a, nb = b; if (i + 1 < BPW) { na = *(const v4i*)(s + (i + 1) * 2048 + o0); nb = *(const v4i*)(s + (i + 1) * 2048 + o1); }
            a ^= t0; b ^= t1; *(v4i*)(d + i * 2048 + o0) = a; *(v4i*)(d + i * 2048 + o1) = b; a = na; b = nb; }
    } else {
#pragma unroll
        for (int i = 0; i < BPW; ++i) { v4i a = *(const v4i*)(s + i * 2048 + o0), b = *(const v4i*)(s + i * 2048 + o1); a ^= t0; b ^= t1; *(v4i*)(d + i * 2048 + o0) = a; *(v4i*)(d + i * 2048 + o1) = b; }
    } }
typedef void (*kern_t)(const v4i*, v4i*, const v4i*, size_t);
int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16, nblk = bytes / 2048;
    v4i *in, *out, *tab; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&tab, 8192)); CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes)); CK(hipMemset(tab, 3, 8192));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct C { const char* name; kern_t k; size_t threads; int tpb; };
    C cs[] = {
        {"classic 16B/thread tpb256", k0_classic, n, 256}, {"classic 16B/thread tpb64", k0_classic, n, 64},
        {"wave=2KiB linear", k_wave2k<0, false>, nblk * 64, 256}, {"wave=2KiB linear tpb64", k_wave2k<0, false>, nblk * 64, 64},
        {"wave=2KiB rowlane", k_wave2k<1, false>, nblk * 64, 256}, {"wave=2KiB rowlane tpb64", k_wave2k<1, false>, nblk * 64, 64},
        {"wave=2KiB rowlane+tab", k_wave2k<1, true>, nblk * 64, 256}, {"wave=2KiB rowlane+tab tpb64", k_wave2k<1, true>, nblk * 64, 64},
        {"run2 noprefetch", k_run<2, false>, nblk * 32, 256}, {"run2 prefetch", k_run<2, true>, nblk * 32, 256},
        {"run4 noprefetch", k_run<4, false>, nblk * 16, 256}, {"run4 prefetch", k_run<4, true>, nblk * 16, 256},
        {"run2 prefetch tpb64", k_run<2, true>, nblk * 32, 64},
    };
    for (int rnd = 0; rnd < 3; ++rnd)
        for (auto& c : cs) {
            dim3 grid(c.threads / c.tpb), block(c.tpb);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(c.k, grid, block, 0, 0, in, out, tab, n);
            CK(hipEventRecord(e0, 0)); const int reps = 20; for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(c.k, grid, block, 0, 0, in, out, tab, n);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            printf("r%d %-30s %.3f ms %5.2f TB/s\n", rnd, c.name, ms, 2.0 * bytes / ms * 1e3 / 1e12);
        }
    return 0;
}
