// membench_sad_shapes: launch shapes of the batched SAD kernel (two 8-bit inputs, 16-byte chunks, v_sad_u8) -- bytes per wave (STEPS x 1 KiB per
// input), waves per workgroup, LDS charged per workgroup (cap on resident waves), reduction by DPP.  16384 x 16384 samples per input.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/membench_sad_shapes tools/probes/membench_sad_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t sad_chunk(const v4i &a, const v4i &b, uint32_t s)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) s = __builtin_amdgcn_sad_u8((uint32_t)a[k], (uint32_t)b[k], s);
    return s;
}
template <int SPAN> __device__ __forceinline__ uint32_t span_sum(uint32_t x)
{
    if (SPAN >= 2) x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);
    if (SPAN >= 4) x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);
    if (SPAN >= 8) x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x141, 0xF, 0xF, true);
    if (SPAN >= 16) x += (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x140, 0xF, 0xF, true);
    if (SPAN >= 64) x = (uint32_t)(__builtin_amdgcn_readlane((int)x, 0) + __builtin_amdgcn_readlane((int)x, 16) + __builtin_amdgcn_readlane((int)x, 32) + __builtin_amdgcn_readlane((int)x, 48));
    return x;
}
// LOGC = log2(chunks per block); STEPS = 1 KiB-linear loads per input and wave
template <int LOGC, int STEPS>
__global__ __launch_bounds__(256) void k(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint32_t *__restrict__ out, size_t n_blocks)
{
    extern __shared__ __attribute__((aligned(16))) char cap[];
    constexpr int CPB = 1 << LOGC;
    constexpr int SPAN = CPB > 64 ? 64 : CPB;
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t chunk0 = wave * (64 * STEPS);
    const size_t total = n_blocks * CPB;
    if (chunk0 >= total) return;
    v4i va[STEPS], vb[STEPS];
    if (chunk0 + 64 * STEPS <= total) {
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            va[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(a + (chunk0 + 64 * i + lane) * 16));
            vb[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(b + (chunk0 + 64 * i + lane) * 16));
        }
    } else {
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            size_t c = chunk0 + 64 * i + lane;
            const bool live = c < total;
            if (!live) c = total - 1;
            va[i] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(a + c * 16));
            vb[i] = live ? __builtin_nontemporal_load(reinterpret_cast<const v4i *>(b + c * 16)) : va[i];
        }
    }
    if (CPB > 64) {                                          // one block spans CPB / 64 steps
        constexpr int PER = CPB / 64;
#pragma unroll
        for (int g = 0; g < STEPS / PER; ++g) {
            uint32_t s = 0;
#pragma unroll
            for (int i = 0; i < PER; ++i) s = sad_chunk(va[g * PER + i], vb[g * PER + i], s);
            s = span_sum<64>(s);
            const size_t blk = chunk0 / CPB + g;
            if (lane == 0 && blk < n_blocks) out[blk] = s;
        }
    } else {
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const uint32_t s = span_sum<SPAN>(sad_chunk(va[i], vb[i], 0));
            const size_t blk = (chunk0 + 64 * i + lane) / CPB;
            if ((lane & (SPAN - 1)) == 0 && blk < n_blocks) out[blk] = s;
        }
    }
}
__global__ void ref_sad(const uint8_t *a, const uint8_t *b, uint32_t *out, size_t n_blocks, int cpb)
{
    size_t blk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= n_blocks) return;
    uint32_t s = 0;
    for (int i = 0; i < cpb * 16; ++i) { int d = (int)a[blk * cpb * 16 + i] - (int)b[blk * cpb * 16 + i]; s += d < 0 ? -d : d; }
    out[blk] = s;
}
static const uint8_t *A, *B; static uint32_t *O, *O2; static size_t NB;
template <int LOGC, int STEPS>
static void run(unsigned wpw, size_t lds_per_wg, bool check)
{
    constexpr int CPB = 1 << LOGC;
    const size_t n_blocks = NB / (CPB * 16) - 3;             // ragged
    const size_t waves = (n_blocks * CPB + 64 * STEPS - 1) / (64 * STEPS);
    const unsigned grid = (unsigned)((waves + wpw - 1) / wpw);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<LOGC, STEPS>), dim3(grid), dim3(64 * wpw), lds_per_wg, 0, A, B, O, n_blocks);
    (void)hipDeviceSynchronize();
    float sum = 0; const int R = 30;
    for (int r = 0; r < R; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<LOGC, STEPS>), dim3(grid), dim3(64 * wpw), lds_per_wg, 0, A, B, O, n_blocks);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); sum += ms;
    }
    const char *ok = "";
    if (check) {
        hipLaunchKernelGGL(ref_sad, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, 0, A, B, O2, n_blocks, CPB);
        uint32_t *h = (uint32_t *)malloc(n_blocks * 4), *h2 = (uint32_t *)malloc(n_blocks * 4);
        (void)hipMemcpy(h, O, n_blocks * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h2, O2, n_blocks * 4, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < n_blocks; ++i) bad += h[i] != h2[i];
        ok = bad ? "  MISMATCH" : "  ok"; free(h); free(h2);
    }
    const double bytes = (double)n_blocks * (CPB * 32 + 4);
    printf("edge %2d steps %d waves/wg %u lds/wg %6zu : %.4f ms  %.3f TB/s%s\n", LOGC == 0 ? 4 : LOGC == 2 ? 8 : LOGC == 4 ? 16 : LOGC == 6 ? 32 : 64, STEPS, wpw, lds_per_wg, sum / R, bytes / (sum / R) / 1e9, ok);
    fflush(stdout);
}
template <int LOGC> static void sweep()
{
    for (unsigned wpw : {1u, 4u})
        for (size_t lds : {(size_t)0, (size_t)8192, (size_t)16384, (size_t)32768}) {
            const size_t l = lds * wpw / 4 * (wpw == 1 ? 4 : 1);      // lds is per 4 waves
            (void)l;
            if (LOGC <= 6) run<LOGC, 1>(wpw, wpw == 4 ? lds : lds / 4, false);
            if (LOGC <= 6) run<LOGC, 2>(wpw, wpw == 4 ? lds : lds / 4, false);
            run<LOGC, 4>(wpw, wpw == 4 ? lds : lds / 4, lds == 0 && wpw == 4);
            run<LOGC, 8>(wpw, wpw == 4 ? lds : lds / 4, false);
        }
}
int main()
{
    NB = (size_t)16384 * 16384;
    uint8_t *a, *b;
    (void)hipMalloc(&a, NB); (void)hipMalloc(&b, NB); (void)hipMalloc(&O, NB / 16 * 4); (void)hipMalloc(&O2, NB / 16 * 4);
    uint8_t *h = (uint8_t *)malloc(NB);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < NB; i += 8) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; *(uint64_t *)(h + i) = x; }
    (void)hipMemcpy(a, h, NB, hipMemcpyHostToDevice);
    for (size_t i = 0; i < NB; i += 8) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; *(uint64_t *)(h + i) = x; }
    (void)hipMemcpy(b, h, NB, hipMemcpyHostToDevice);
    A = a; B = b;
    sweep<2>(); sweep<4>(); sweep<8>(); sweep<0>(); sweep<6>();
    return 0;
}
