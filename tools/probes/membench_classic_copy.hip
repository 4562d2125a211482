// membench3.hip -- developer probe #3: classic one-element-per-thread float4 copy vs persistent copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <bool NT, int TPB> __global__ __launch_bounds__(TPB) void classic(const v4i* __restrict__ in, v4i* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i < n) { v4i v = NT ? __builtin_nontemporal_load(in + i) : in[i]; v[0] ^= 1; if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
}
template <int UNR, int TPB> __global__ __launch_bounds__(TPB) void classic_unr(const v4i* __restrict__ in, v4i* __restrict__ out, size_t n) {
    size_t base = ((size_t)blockIdx.x * UNR) * TPB + threadIdx.x;
    v4i v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) v[u] = in[base + (size_t)u * TPB];
#pragma unroll
    for (int u = 0; u < UNR; ++u) { v[u][0] ^= 1; out[base + (size_t)u * TPB] = v[u]; }
}
int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t mib : {256, 1024, 2048, 4096}) {
        size_t bytes = mib << 20, n = bytes / 16;
        v4i *in, *out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMemset(in, 1, bytes)); CK(hipMemset(out, 0, bytes));
        auto t = [&](const char* name, auto launch) {
            for (int i = 0; i < 3; ++i) launch();
            CK(hipEventRecord(e0, 0)); const int reps = 20; for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            printf("%5zu MiB %-24s %.3f ms %5.2f TB/s\n", mib, name, ms, 2.0 * bytes / ms * 1e3 / 1e12);
        };
        t("classic 256", [&] { hipLaunchKernelGGL((classic<false, 256>), dim3(n / 256), dim3(256), 0, 0, in, out, n); });
        t("classic 256 nt", [&] { hipLaunchKernelGGL((classic<true, 256>), dim3(n / 256), dim3(256), 0, 0, in, out, n); });
        t("classic 1024", [&] { hipLaunchKernelGGL((classic<false, 1024>), dim3(n / 1024), dim3(1024), 0, 0, in, out, n); });
        t("classic 64", [&] { hipLaunchKernelGGL((classic<false, 64>), dim3(n / 64), dim3(64), 0, 0, in, out, n); });
        t("unroll2 256", [&] { hipLaunchKernelGGL((classic_unr<2, 256>), dim3(n / 512), dim3(256), 0, 0, in, out, n); });
        t("unroll4 256", [&] { hipLaunchKernelGGL((classic_unr<4, 256>), dim3(n / 1024), dim3(256), 0, 0, in, out, n); });
        t("unroll8 256", [&] { hipLaunchKernelGGL((classic_unr<8, 256>), dim3(n / 2048), dim3(256), 0, 0, in, out, n); });
        t("hipMemcpyDtoD", [&] { CK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0)); });
        CK(hipFree(in)); CK(hipFree(out));
    }
    return 0;
}
