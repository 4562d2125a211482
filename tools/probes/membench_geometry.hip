// membench5.hip -- developer probe #5: read-dominated streaming (SATD-like: 128 B in, 4 B out per unit)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
// LOADS 16-byte loads per lane per wave-iteration; PAT 0: linear (16*lane + 1024*i); 1: satd (128*(l&31) + 64*(l>>5) + 16*i); 2: sector (128*(l&31)+32*i+16*(l>>5))
// RUN consecutive iterations per wave (streaming launch).  Output: one dword per 128 input bytes, written by lanes < 32*LOADS/4... simplified: lane<8*LOADS writes 1 dword
template <int LOADS, int PAT, int RUN> __global__ __launch_bounds__(256) void rd(const char* __restrict__ in, int* __restrict__ out, size_t n_iter) {
    const int lane = threadIdx.x & 63; const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int r = 0; r < RUN; ++r) {
        const size_t it = wave * RUN + r; if (it >= n_iter) return;
        const char* s = in + it * (size_t)(LOADS * 1024);
        v4i acc = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            size_t o = PAT == 0 ? (size_t)lane * 16 + 1024 * i : PAT == 1 ? (size_t)(lane & 31) * (32 * LOADS) + (lane >> 5) * (16 * LOADS) + 16 * i
                                                                         : (size_t)(lane & 31) * (32 * LOADS) + 32 * i + 16 * (lane >> 5);
            acc += *(const v4i*)(s + o);
        }
        int v = acc[0] + acc[1] + acc[2] + acc[3]; v += __shfl_xor(v, 32);
        if (lane < 8 * LOADS) out[it * (8 * LOADS) + lane] = v;
    }
}
typedef void (*kern_t)(const char*, int*, size_t);
int main() {
    const size_t bytes = (size_t)2 << 30;
    char* in; int* out; CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes / 32)); CK(hipMemset(in, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct C { const char* name; kern_t k; int loads, run, tpb; };
    C cs[] = {
        {"1 load linear", rd<1, 0, 1>, 1, 1, 256}, {"1 load linear tpb64", rd<1, 0, 1>, 1, 1, 64},
        {"2 loads linear", rd<2, 0, 1>, 2, 1, 256}, {"2 loads rowlane", rd<2, 1, 1>, 2, 1, 256},
        {"4 loads linear", rd<4, 0, 1>, 4, 1, 256}, {"4 loads satd", rd<4, 1, 1>, 4, 1, 256}, {"4 loads sector", rd<4, 2, 1>, 4, 1, 256},
        {"4 loads satd tpb64", rd<4, 1, 1>, 4, 1, 64},
        {"4 loads satd run4", rd<4, 1, 4>, 4, 4, 256}, {"4 loads satd run8", rd<4, 1, 8>, 4, 8, 256}, {"4 loads linear run8", rd<4, 0, 8>, 4, 8, 256},
        {"1 load linear run8", rd<1, 0, 8>, 1, 8, 256}, {"2 loads linear run4", rd<2, 0, 4>, 2, 4, 256},
    };
    for (int rnd = 0; rnd < 2; ++rnd)
        for (auto& c : cs) {
            size_t n_iter = bytes / (c.loads * 1024), waves = (n_iter + c.run - 1) / c.run, threads = waves * 64;
            dim3 grid((threads + c.tpb - 1) / c.tpb), block(c.tpb);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(c.k, grid, block, 0, 0, in, out, n_iter);
            CK(hipEventRecord(e0, 0)); const int reps = 20; for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(c.k, grid, block, 0, 0, in, out, n_iter);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            printf("r%d %-24s %.3f ms %5.2f TB/s\n", rnd, c.name, ms, (bytes + bytes / 32.0) / ms * 1e3 / 1e12);
        }
    return 0;
}
