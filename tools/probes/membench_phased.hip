// membench_phased (round 5): can a copy-pattern stream beat the 1 KiB-interleaved copy of xHipMemCeilingDev
// (6.7-6.8 TB/s; reads alone 7.3-7.5, writes alone 7.2-7.4) by separating its reads from its writes IN TIME?
//   mode 0  the library's copy shape: one-wave workgroups, 2 KiB per wave, register staged, loads then stores (the baseline row)
//   mode 1  per-workgroup phases: every wave of a W-wave workgroup fills C KiB of LDS by LDS-DMA (loads only), the workgroup meets at a
//           barrier, then every wave drains its C KiB (ds_read -> stores only): bursts of W x C KiB per workgroup instead of 1 KiB
//   mode 2  the same, with the phases of odd XCDs started half a phase late (s_sleep): the chip's read and write bursts interleave at
//           XCD granularity (workgroup id mod 8 = XCD under round-robin dispatch)
//   mode 3  chip-wide phases on the wall clock: every wave loads only while bit `k` of s_memrealtime (100 MHz) is 0 and stores only
//           while it is 1 -- the whole chip reads for 2^k / 100 us, then writes
//   mode 4  pipelined ring, no phases: two C KiB slots per wave, the next chunk's DMA in flight while the current one drains
// Arithmetic-free, nt loads, "sc1 nt" stores, every access 1 KiB-linear.  Each wave moves `iters` chunks of C KiB; `lds` per wave caps the
// resident waves.  build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/membench_phased tools/probes/membench_phased.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st_sc1nt_s(char *base, unsigned off, v4i v)
{
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1 nt\n\ts_nop 1" :: "v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void dma_1k(const char *base, unsigned off, unsigned lds)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(lds) : "memory");
}
__device__ __forceinline__ unsigned long long realtime()
{
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

template <int C>   // KiB per chunk
__global__ __launch_bounds__(1024) void phased(const char *__restrict__ in, char *__restrict__ out, size_t n_kib, unsigned iters, int mode,
                                               unsigned lds_per_wave, unsigned phase_bit, unsigned stagger_sleep)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const unsigned wave_in_wg = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
    const size_t first = wave * iters * C;                       // this wave's run of chunks, in KiB
    char *slot = lds + wave_in_wg * lds_per_wave;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)slot;
    const unsigned off = (unsigned)lane * 16u;
    if (mode == 2 && (blockIdx.x & 1)) for (unsigned s = 0; s < stagger_sleep; ++s) __builtin_amdgcn_s_sleep(64);
    auto fill = [&](size_t kib, unsigned slot_off) {
#pragma unroll
        for (int i = 0; i < C; ++i)
            if (kib + i < n_kib) dma_1k(in + (kib + i) * 1024, off, lds0 + slot_off + 1024u * i);
    };
    auto drain = [&](size_t kib, unsigned slot_off) {
        v4i a[C];
#pragma unroll
        for (int i = 0; i < C; ++i) a[i] = *reinterpret_cast<const v4i *>(slot + slot_off + 1024 * i + lane * 16);
#pragma unroll
        for (int i = 0; i < C; ++i)
            if (kib + i < n_kib) st_sc1nt_s(out + (kib + i) * 1024, off, a[i]);
    };
    if (mode == 4) {                                            // ring of two slots
        if (first < n_kib) fill(first, 0);
        for (unsigned it = 0; it < iters; ++it) {
            const size_t kib = first + (size_t)it * C;
            if (kib >= n_kib) break;
            const bool more = it + 1 < iters && kib + C < n_kib;
            if (more) fill(kib + C, ((it + 1) & 1) * C * 1024);
            // younger than this chunk's DMA, in issue order: the previous chunk's C stores (when there was one) and the next chunk's C DMA (when there is one)
            const int younger = (more ? 1 : 0) + (it ? 1 : 0);
            if (younger == 2)      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * C) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(C) : "memory");
            else                   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            drain(kib, (it & 1) * C * 1024);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        return;
    }
    for (unsigned it = 0; it < iters; ++it) {
        const size_t kib = first + (size_t)it * C;
        if (mode == 3) while ((realtime() >> phase_bit) & 1ull) __builtin_amdgcn_s_sleep(8);      // read phase: bit clear
        if (kib < n_kib) fill(kib, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (mode == 1 || mode == 2) __syncthreads();             // the whole workgroup has finished loading
        if (mode == 3) while (!((realtime() >> phase_bit) & 1ull)) __builtin_amdgcn_s_sleep(8);   // write phase: bit set
        if (kib < n_kib) drain(kib, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (mode == 1 || mode == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }   // ... and storing
    }
}

__global__ __launch_bounds__(64) void baseline(const char *__restrict__ in, char *__restrict__ out, size_t n_kib)
{
    extern __shared__ __attribute__((aligned(16))) char cap[];
    const int lane = threadIdx.x & 63;
    const size_t p = (size_t)blockIdx.x * 2;
    if (p >= n_kib) return;
    const v4i a = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(in + lane * 16 + p * 1024));
    const v4i b = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(in + lane * 16 + (p + 1) * 1024));
    st_sc1nt_s(out + p * 1024, lane * 16, a);
    st_sc1nt_s(out + (p + 1) * 1024, lane * 16, b);
}

static float time_launch(void (*launch)(void *), void *ctx)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 25; ++i) launch(ctx);
    (void)hipDeviceSynchronize();
    float ms[21];
    for (int r = 0; r < 21; ++r) {
        (void)hipEventRecord(e0); launch(ctx); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms[r], e0, e1);
    }
    for (int i = 0; i < 21; ++i) for (int j = i + 1; j < 21; ++j) if (ms[j] < ms[i]) { float t = ms[i]; ms[i] = ms[j]; ms[j] = t; }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms[10];
}

struct Cfg { const char *in; char *out; size_t n_kib; int c; unsigned iters, wpw, lds_per_wave, phase_bit, stagger; int mode; };

static void launch_phased(void *p)
{
    const Cfg &c = *(const Cfg *)p;
    const size_t per_wave = (size_t)c.iters * c.c, waves = (c.n_kib + per_wave - 1) / per_wave;
    const unsigned grid = (unsigned)((waves + c.wpw - 1) / c.wpw);
    const size_t lds = (size_t)c.wpw * c.lds_per_wave;
#define L(C) if (lds > 65536) (void)hipFuncSetAttribute((const void *)phased<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
             hipLaunchKernelGGL(phased<C>, dim3(grid), dim3(64 * c.wpw), lds, 0, c.in, c.out, c.n_kib, c.iters, c.mode, c.lds_per_wave, c.phase_bit, c.stagger)
    if (c.c == 2) { L(2); } else if (c.c == 4) { L(4); } else if (c.c == 8) { L(8); } else { L(16); }
#undef L
}
static void launch_baseline(void *p)
{
    const Cfg &c = *(const Cfg *)p;
    hipLaunchKernelGGL(baseline, dim3((unsigned)((c.n_kib + 1) / 2)), dim3(64), 8192, 0, c.in, c.out, c.n_kib);
}

int main(int argc, char **argv)
{
    const size_t n_kib = (size_t)2 << 20;                        // 2 GiB in, 2 GiB out
    char *in, *out;
    if (hipMalloc(&in, n_kib * 1024) != hipSuccess || hipMalloc(&out, n_kib * 1024) != hipSuccess) return 1;
    (void)hipMemset(in, 0x5a, n_kib * 1024);
    Cfg c{in, out, n_kib, 2, 1, 1, 8192, 9, 0, 0};
    const double bytes = 2.0 * n_kib * 1024;
    const int rounds = argc > 1 ? atoi(argv[1]) : 1;
    for (int rnd = 0; rnd < rounds; ++rnd) {
        float ms = time_launch(launch_baseline, &c);
        printf("baseline copy (1-wave workgroups, 2 KiB per wave, 20 waves per CU): %.4f ms  %.3f TB/s\n", ms, bytes / ms / 1e9);
        // correctness of the phased kernels once: out == in
        const int modes[] = {1, 2, 3, 4};
        for (int mi = 0; mi < 4; ++mi) {
            c.mode = modes[mi];
            for (int cc = 2; cc <= 16; cc *= 2)
                for (unsigned wpw : {1u, 4u, 8u, 16u})
                    for (unsigned iters : {1u, 4u, 16u}) {
                        const unsigned need = (unsigned)cc * 1024u * (c.mode == 4 ? 2u : 1u);
                        for (unsigned lds : {need, 2 * need}) {
                            if ((size_t)lds * wpw > 160 * 1024 || lds * wpw < 4096) continue;
                            if (c.mode == 3 && (wpw != 4 || iters < 4)) continue;
                            if ((c.mode == 1 || c.mode == 2) && wpw == 1) continue;
                            c.c = cc; c.wpw = wpw; c.iters = iters; c.lds_per_wave = lds;
                            for (unsigned pb : {8u, 9u, 10u}) {
                                if (c.mode != 3 && pb != 9u) continue;
                                c.phase_bit = pb;
                                // stagger: half of a workgroup's phase, estimated as W x C KiB at ~25 GB/s per CU -> sleeps of 64 clocks
                                c.stagger = (unsigned)((double)wpw * cc * 1024 / 25e9 * 2.2e9 / 64 / 2);
                                ms = time_launch(launch_phased, &c);
                                printf("mode %d C %2d KiB waves/wg %2u iters %2u lds/wave %6u%s : %.4f ms  %.3f TB/s\n", c.mode, cc, wpw, iters, lds,
                                       c.mode == 3 ? (pb == 8 ? " phase 2.56us" : pb == 9 ? " phase 5.12us" : " phase 10.2us") : "", ms, bytes / ms / 1e9);
                                fflush(stdout);
                            }
                        }
                    }
        }
    }
    // every mode really copies: pattern in, zeros out, one launch, three windows compared
    static unsigned char pat[1 << 20], probe[1 << 20];
    for (size_t i = 0; i < sizeof pat; ++i) pat[i] = (unsigned char)(i * 2654435761u >> 13);
    const size_t win[3] = {0, (n_kib / 2) * 1024 + 4096, n_kib * 1024 - sizeof pat};
    for (int k = 0; k < 3; ++k) (void)hipMemcpy(in + win[k], pat, sizeof pat, hipMemcpyHostToDevice);
    for (int mode = 1; mode <= 4; ++mode) {
        (void)hipMemset(out, 0, n_kib * 1024);
        c.mode = mode; c.c = 4; c.wpw = 4; c.iters = 5; c.lds_per_wave = 8192; c.phase_bit = 9; c.stagger = 10;
        launch_phased(&c);
        (void)hipDeviceSynchronize();
        for (int k = 0; k < 3; ++k) {
            (void)hipMemcpy(probe, out + win[k], sizeof pat, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < sizeof pat; ++i) if (probe[i] != pat[i]) { printf("mode %d: COPY WRONG at window %d byte %zu\n", mode, k, i); return 2; }
        }
    }
    printf("all four modes copy correctly\n");
    return 0;
}
