// membench_lds_dma: does fetching a tile straight into LDS (global_load_lds_dwordx4, no VGPR round trip) move the
// copy / read ceilings of the staged kernels' shape?  One-wave workgroups, one 2 KiB tile per wave, dynamic LDS cap.
//   reg : global_load_dwordx4 nt x2 -> ds_write_b128 x2 -> ds_read_b128 x2 -> global_store_dwordx4 sc1 nt x2 (the kernels today)
//   dma : global_load_lds_dwordx4 [nt] x2                -> ds_read_b128 x2 -> global_store_dwordx4 sc1 nt x2
//   *_read : the same without the stores (sum, never written)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/membench_lds_dma tools/probes/membench_lds_dma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4i ld_nt(const char *p) { v4i v; asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void st_sc1nt(char *p, v4i v) { asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory"); }

// 1 KiB per wave-instruction: lane's 16 bytes at src land at lds_dst (wave-uniform byte address) + lane * 16
template <bool NT>
__device__ __forceinline__ void dma16(const char *src, unsigned lds_dst)
{
    unsigned keep;
    if (NT) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
    else    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}

// MODE 0 reg copy, 1 dma copy, 2 dma-nt copy, 3 reg read, 4 dma read, 5 dma-nt read
template <int MODE, int TPW>
__global__ __launch_bounds__(64) void k(const char *in, char *out, size_t tiles)
{
    extern __shared__ __attribute__((aligned(16))) char slot[];
    const int lane = threadIdx.x;
    const unsigned slot_addr = (unsigned)(uintptr_t)slot;                 // LDS byte address of the workgroup's slot (wave-uniform)
    int acc = 0;
    for (int i = 0; i < TPW; ++i) {
        const size_t t = (size_t)blockIdx.x * TPW + i;
        if (t >= tiles) return;
        const char *src = in + t * 2048 + lane * 16;
        char *dst = out + t * 2048 + lane * 16;
        if (MODE == 0 || MODE == 3) {
            const v4i a = ld_nt(src), b = ld_nt(src + 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *reinterpret_cast<v4i *>(slot + lane * 16) = a;
            *reinterpret_cast<v4i *>(slot + 1024 + lane * 16) = b;
        } else {
            dma16<MODE == 2 || MODE == 5>(src, slot_addr);
            dma16<MODE == 2 || MODE == 5>(src + 1024, slot_addr + 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_wave_barrier();
        // read back rotated by one lane so that the LDS pass cannot be elided (the data lands as a permutation of the tile: fine for a bandwidth probe)
        const v4i a = *reinterpret_cast<const v4i *>(slot + ((lane + 1) & 63) * 16);
        const v4i b = *reinterpret_cast<const v4i *>(slot + 1024 + ((lane + 1) & 63) * 16);
        __builtin_amdgcn_wave_barrier();
        if (MODE >= 3) { acc += a[0] + a[3] + b[1] + b[2]; continue; }
        st_sc1nt(dst, a);
        st_sc1nt(dst + 1024, b);
    }
    if (MODE >= 3 && acc == 0x12345678) out[blockIdx.x] = 1;
}

template <int MODE, int TPW>
static void run(const char *label, const char *in, char *out, size_t tiles, size_t lds)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = (unsigned)((tiles + TPW - 1) / TPW);
    for (int i = 0; i < 150; ++i) hipLaunchKernelGGL((k<MODE, TPW>), dim3(grid), dim3(64), lds, 0, in, out, tiles);
    hipDeviceSynchronize();
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 20; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, TPW>), dim3(grid), dim3(64), lds, 0, in, out, tiles);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    const double bytes = (double)tiles * 2048 * (MODE >= 3 ? 1 : 2);
    printf("%-22s tiles/wave=%d lds/wave=%5zu : mean %.4f ms (min %.4f)  %.3f TB/s\n", label, TPW, lds, sum / 20, best, bytes / (sum / 20) / 1e9);
    fflush(stdout);
}

int main()
{
    const size_t tiles = 1 << 20;
    char *in, *out;
    hipMalloc(&in, tiles * 2048); hipMalloc(&out, tiles * 2048);
    hipMemset(in, 0x5a, tiles * 2048); hipMemset(out, 0, tiles * 2048);
    for (size_t lds : {(size_t)2048, (size_t)4096, (size_t)8192}) {
        run<0, 1>("reg copy (nt)", in, out, tiles, lds);
        run<1, 1>("lds-dma copy", in, out, tiles, lds);
        run<2, 1>("lds-dma copy nt", in, out, tiles, lds);
        run<3, 1>("reg read (nt)", in, out, tiles, lds);
        run<4, 1>("lds-dma read", in, out, tiles, lds);
        run<5, 1>("lds-dma read nt", in, out, tiles, lds);
    }
    run<0, 2>("reg copy (nt)", in, out, tiles, 4096);
    run<2, 2>("lds-dma copy nt", in, out, tiles, 4096);
    // did the DMA land where the register path lands?  compare the two outputs
    hipMemset(in, 0, tiles * 2048);
    unsigned *h = (unsigned *)malloc(1 << 20);
    for (size_t i = 0; i < (1 << 18); ++i) h[i] = (unsigned)(i * 2654435761u);
    hipMemcpy(in, h, 1 << 20, hipMemcpyHostToDevice);
    char *o2; hipMalloc(&o2, 1 << 20);
    hipLaunchKernelGGL((k<0, 1>), dim3(512), dim3(64), 2048, 0, in, out, (size_t)512);
    hipLaunchKernelGGL((k<2, 1>), dim3(512), dim3(64), 2048, 0, in, o2, (size_t)512);
    hipDeviceSynchronize();
    unsigned *a = (unsigned *)malloc(1 << 20), *b = (unsigned *)malloc(1 << 20);
    hipMemcpy(a, out, 1 << 20, hipMemcpyDeviceToHost); hipMemcpy(b, o2, 1 << 20, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < (1 << 18); ++i) bad += a[i] != b[i];
    printf("lds-dma output equals the register path: %s (%zu differing words)\n", bad ? "NO" : "yes", bad);
    return 0;
}
