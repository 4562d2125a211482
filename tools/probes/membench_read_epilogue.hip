// membench_read_epilogue: what does a checksum store cost a short-lived read-stream wave?  The library's read-ceiling kernel (xHipMemCeilingDev
// kind 1) has to write something checkable; in its first form it ran at half the rate of the same loads with a never-taken store.
//   EPI 0  never-taken store (the membench baseline)      1  lane 0 stores its own XOR            2  DPP/readlane wave XOR, lane 0 stores
//   EPI 3  wave XOR, all 64 lanes store the same dword (no exec change)                           4  wave XOR, s_store-like scalar... (not available) -> atomic xor into one dword per workgroup
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/membench_read_epilogue tools/probes/membench_read_epilogue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int wave_xor(int x)
{
    x ^= __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true);
    x ^= __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true);
    x ^= __builtin_amdgcn_mov_dpp(x, 0x141, 0xF, 0xF, true);
    x ^= __builtin_amdgcn_mov_dpp(x, 0x140, 0xF, 0xF, true);
    return __builtin_amdgcn_readlane(x, 0) ^ __builtin_amdgcn_readlane(x, 16) ^ __builtin_amdgcn_readlane(x, 32) ^ __builtin_amdgcn_readlane(x, 48);
}

template <int KB, int EPI>
__global__ __launch_bounds__(256) void k(const char *__restrict__ in, int *__restrict__ out, size_t n_kib)
{
    extern __shared__ __attribute__((aligned(16))) char cap[];
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t p = wave * KB;
    if (p + KB > n_kib) return;
    const char *src = in + p * 1024 + lane * 16;
    v4i a[KB];
#pragma unroll
    for (int i = 0; i < KB; ++i) a[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(src + i * 1024));
    int acc = 0;
#pragma unroll
    for (int i = 0; i < KB; ++i) acc ^= a[i][0] ^ a[i][1] ^ a[i][2] ^ a[i][3];
    if (EPI == 0) { if (acc == 0x12345678) out[wave] = acc; }
    else if (EPI == 1) { if (lane == 0) out[wave] = acc; }
    else if (EPI == 2) { const int x = wave_xor(acc); if (lane == 0) out[wave] = x; }
    else if (EPI == 3) { const int x = wave_xor(acc); out[wave] = x; }
    else { const int x = wave_xor(acc); if (lane == 0) atomicXor(&out[blockIdx.x], x); }
}

template <int KB, int EPI>
static void run(const char *in, int *out, size_t n_kib, unsigned wpw, size_t lds)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t waves = n_kib / KB;
    const unsigned grid = (unsigned)((waves + wpw - 1) / wpw);
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((k<KB, EPI>), dim3(grid), dim3(64 * wpw), lds, 0, in, out, n_kib);
    (void)hipDeviceSynchronize();
    float sum = 0;
    const int R = 30;
    for (int r = 0; r < R; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<KB, EPI>), dim3(grid), dim3(64 * wpw), lds, 0, in, out, n_kib);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        sum += ms;
    }
    printf("KB %d epilogue %d waves/wg %u lds/wave %6zu : mean %.4f ms  %.3f TB/s\n", KB, EPI, wpw, lds / wpw, sum / R, (double)n_kib * 1024 / (sum / R) / 1e9);
    fflush(stdout);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

template <int KB>
static void all(const char *in, int *out, size_t n_kib, unsigned wpw, size_t lds_per_wave)
{
    run<KB, 0>(in, out, n_kib, wpw, lds_per_wave * wpw);
    run<KB, 1>(in, out, n_kib, wpw, lds_per_wave * wpw);
    run<KB, 2>(in, out, n_kib, wpw, lds_per_wave * wpw);
    run<KB, 3>(in, out, n_kib, wpw, lds_per_wave * wpw);
    run<KB, 4>(in, out, n_kib, wpw, lds_per_wave * wpw);
}

int main()
{
    const size_t n_kib = (size_t)2 << 20;
    char *in; int *out;
    (void)hipMalloc(&in, n_kib * 1024); (void)hipMalloc(&out, 64 << 20);
    (void)hipMemset(in, 0x5a, n_kib * 1024);
    for (int rnd = 0; rnd < 2; ++rnd) {
        printf("# round %d\n", rnd);
        all<4>(in, out, n_kib, 4, 16384);
        all<4>(in, out, n_kib, 4, 8192);
        all<4>(in, out, n_kib, 1, 16384);
        all<4>(in, out, n_kib, 1, 8192);
        all<2>(in, out, n_kib, 1, 8192);
        all<2>(in, out, n_kib, 4, 8192);
        all<2>(in, out, n_kib, 1, 4096);
    }
    return 0;
}
