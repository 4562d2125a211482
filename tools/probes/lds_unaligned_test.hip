#include <hip/hip_runtime.h>
#include <cstdint>
typedef uint64_t u64u __attribute__((aligned(1)));
typedef uint32_t u32u __attribute__((aligned(1)));
__global__ void k(const int *idx, uint64_t *out, uint32_t *out2) {
    extern __shared__ unsigned char smem[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) smem[i] = (unsigned char)(i * 7 + (i >> 8));
    __syncthreads();
    const int o = idx[threadIdx.x];
    out[threadIdx.x] = *reinterpret_cast<const u64u *>(smem + o);
    out2[threadIdx.x] = *reinterpret_cast<const u32u *>(smem + o + 100);
}
int main() {
    int h_idx[64]; for (int i = 0; i < 64; ++i) h_idx[i] = i * 13 + (i & 7);
    int *d_idx; uint64_t *d_out; uint32_t *d_out2;
    hipMalloc(&d_idx, sizeof h_idx); hipMalloc(&d_out, 64 * 8); hipMalloc(&d_out2, 64 * 4);
    hipMemcpy(d_idx, h_idx, sizeof h_idx, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d_idx, d_out, d_out2);
    uint64_t h_out[64]; uint32_t h_out2[64];
    hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost); hipMemcpy(h_out2, d_out2, sizeof h_out2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        uint64_t e = 0; uint32_t e2 = 0;
        for (int b = 0; b < 8; ++b) { int a = h_idx[i] + b; e |= (uint64_t)(unsigned char)(a * 7 + (a >> 8)) << (8 * b); }
        for (int b = 0; b < 4; ++b) { int a = h_idx[i] + 100 + b; e2 |= (uint32_t)(unsigned char)(a * 7 + (a >> 8)) << (8 * b); }
        if (e != h_out[i] || e2 != h_out2[i]) ++bad;
    }
    printf("unaligned LDS reads: %d of 64 lanes wrong\n", bad);
    return bad != 0;
}
