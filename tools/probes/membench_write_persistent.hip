// membench_write_persistent (round 5): can LONG-LIVED waves write at the short-lived write stream's rate if they keep the chip's open write
// window compact?  A persistent grid (exactly the resident waves: 256 CUs x `per_cu` one-wave workgroups, held there by the LDS charge) walks
// the buffer in lockstep: chunk = t * n_waves + wave, `kib` KiB per chunk (kib consecutive 1 KiB "sc1 nt" stores), so the window is
// n_waves x kib KiB and moves through the buffer.  Next to it: the short-lived dispatch-ordered stream (xHipMemCeilingDev's write shape:
// one-wave workgroups, 2 KiB per wave, 16 KiB of LDS charged) and the blocked long-lived walk (the intra predictor's: 28 KiB runs per wave).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/membench_write_persistent tools/probes/membench_write_persistent.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sc1nt(char *p, v4i v) { asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }

// MODE 0: persistent lockstep walk; 1: blocked runs of `iters` chunks per wave (grid covers the buffer); 
template <int MODE>
__global__ __launch_bounds__(64) void k(char *__restrict__ out, size_t n_kib, unsigned kib, unsigned iters, unsigned spin)
{
    extern __shared__ __attribute__((aligned(16))) char cap[];
    const int lane = threadIdx.x;
    const size_t wave = blockIdx.x, n_waves = gridDim.x;
    v4i v = {(int)wave, lane, 0, 0};
    const size_t n_chunks = n_kib / kib;
    if (MODE == 0) {
        for (size_t c = wave; c < n_chunks; c += n_waves) {
            for (unsigned s = 0; s < spin; ++s) v[2] = v[2] * 3 + 1;
            char *p = out + c * kib * 1024 + lane * 16;
            for (unsigned i = 0; i < kib; ++i) st_sc1nt(p + i * 1024, v);
        }
    } else {
        for (unsigned t = 0; t < iters; ++t) {
            const size_t c = wave * iters + t;
            if (c >= n_chunks) break;
            for (unsigned s = 0; s < spin; ++s) v[2] = v[2] * 3 + 1;
            char *p = out + c * kib * 1024 + lane * 16;
            for (unsigned i = 0; i < kib; ++i) st_sc1nt(p + i * 1024, v);
        }
    }
}

static float timed(void (*launch)(void *), void *arg)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) launch(arg);
    (void)hipDeviceSynchronize();
    float best = 1e9f, sum = 0; const int R = 20;
    for (int r = 0; r < R; ++r) {
        (void)hipEventRecord(e0); launch(arg); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); sum += ms; if (ms < best) best = ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return sum / R;
}

struct Args { char *out; size_t n_kib; unsigned kib, iters, spin, grid; size_t lds; int mode; };
static void go(void *a_)
{
    Args *a = (Args *)a_;
    if (a->mode == 0) hipLaunchKernelGGL((k<0>), dim3(a->grid), dim3(64), a->lds, 0, a->out, a->n_kib, a->kib, a->iters, a->spin);
    else              hipLaunchKernelGGL((k<1>), dim3(a->grid), dim3(64), a->lds, 0, a->out, a->n_kib, a->kib, a->iters, a->spin);
}

int main()
{
    const size_t n_kib = (size_t)1 << 20;      // 1 GiB
    char *out;
    (void)hipMalloc(&out, n_kib * 1024);
    for (unsigned spin : {0u, 40u, 120u}) {
        {   // the short-lived stream: 2 KiB per wave, 16 KiB charged
            Args a{out, n_kib, 2, 1, spin, (unsigned)(n_kib / 2), 16384, 1};
            float ms = timed(go, &a);
            printf("short-lived 2 KiB/wave, 10 waves/CU            spin %3u : %.4f ms %.3f TB/s\n", spin, ms, (double)n_kib * 1024 / ms / 1e9);
        }
        {   // blocked long-lived: 28 x 1 KiB per wave, full occupancy
            Args a{out, n_kib, 1, 28, spin, (unsigned)((n_kib + 27) / 28), 0, 1};
            float ms = timed(go, &a);
            printf("blocked 28 x 1 KiB/wave, no LDS cap            spin %3u : %.4f ms %.3f TB/s\n", spin, ms, (double)n_kib * 1024 / ms / 1e9);
        }
        for (unsigned per_cu : {8u, 10u, 12u, 16u, 20u, 24u, 32u})
            for (unsigned kib : {1u, 2u, 4u, 7u}) {
                size_t lds = (160 * 1024 / per_cu) & ~(size_t)1023;      // exactly per_cu one-wave workgroups fit a CU
                if (per_cu == 32) lds = 0;                               // the wave-slot limit itself
                Args a{out, n_kib, kib, 0, spin, 256 * per_cu, lds, 0};
                float ms = timed(go, &a);
                printf("persistent lockstep %2u waves/CU chunk %u KiB (window %6.2f MiB) spin %3u : %.4f ms %.3f TB/s\n", per_cu, kib, 256.0 * per_cu * kib / 1024, spin, ms, (double)n_kib * 1024 / ms / 1e9);
                fflush(stdout);
            }
    }
    return 0;
}
