// membench6: cache-policy bits (sc0 / sc1 / nt) on line-dense 16 B/lane loads and stores, in the
// launch shape of the staged kernels (one-wave workgroups, 8 KiB dynamic LDS per wave, one 2 KiB
// tile per wave).  modes: copy, read-only (sum), write-only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));

#define LD(NAME, BITS) __device__ __forceinline__ v4i NAME(const char *p) { v4i v; asm volatile("global_load_dwordx4 %0, %1, off " BITS : "=v"(v) : "v"(p) : "memory"); return v; }
#define ST(NAME, BITS) __device__ __forceinline__ void NAME(char *p, v4i v) { asm volatile("global_store_dwordx4 %0, %1, off " BITS :: "v"(p), "v"(v) : "memory"); }
LD(ld_0, "") LD(ld_nt, "nt") LD(ld_sc1, "sc1") LD(ld_sc0sc1, "sc0 sc1") LD(ld_sc0sc1nt, "sc0 sc1 nt") LD(ld_sc0, "sc0") LD(ld_sc1nt, "sc1 nt") LD(ld_sc0nt, "sc0 nt")
ST(st_0, "") ST(st_nt, "nt") ST(st_sc1, "sc1") ST(st_sc0sc1, "sc0 sc1") ST(st_sc0sc1nt, "sc0 sc1 nt") ST(st_sc0, "sc0") ST(st_sc1nt, "sc1 nt") ST(st_sc0nt, "sc0 nt")

template <int L, int S, int MODE>
__global__ __launch_bounds__(64) void k(const char *in, char *out, size_t tiles, unsigned spread = 1)
{
    extern __shared__ char pad[];
    const size_t t = spread > 1 ? (size_t)(blockIdx.x % spread) * (tiles / spread) + blockIdx.x / spread : (size_t)blockIdx.x;
    if (t >= tiles) return;
    const char *src = in + t * 2048 + threadIdx.x * 16;
    char *dst = out + t * 2048 + threadIdx.x * 16;
    v4i a = {1, 2, 3, 4}, b = {5, 6, 7, 8};
    if (MODE != 2) {
        if (L == 0) { a = ld_0(src); b = ld_0(src + 1024); }
        if (L == 1) { a = ld_nt(src); b = ld_nt(src + 1024); }
        if (L == 2) { a = ld_sc1(src); b = ld_sc1(src + 1024); }
        if (L == 3) { a = ld_sc0sc1(src); b = ld_sc0sc1(src + 1024); }
        if (L == 4) { a = ld_sc0sc1nt(src); b = ld_sc0sc1nt(src + 1024); }
        if (L == 5) { a = ld_sc0(src); b = ld_sc0(src + 1024); }
        if (L == 6) { a = ld_sc1nt(src); b = ld_sc1nt(src + 1024); }
        if (L == 7) { a = ld_sc0nt(src); b = ld_sc0nt(src + 1024); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 1) {
        const int s = a[0] + a[1] + a[2] + a[3] + b[0] + b[1] + b[2] + b[3];
        if (s == 0x12345678) out[t] = 1;                       // never true for the fill pattern
        return;
    }
    if (S == 0) { st_0(dst, a); st_0(dst + 1024, b); }
    if (S == 1) { st_nt(dst, a); st_nt(dst + 1024, b); }
    if (S == 2) { st_sc1(dst, a); st_sc1(dst + 1024, b); }
    if (S == 3) { st_sc0sc1(dst, a); st_sc0sc1(dst + 1024, b); }
    if (S == 4) { st_sc0sc1nt(dst, a); st_sc0sc1nt(dst + 1024, b); }
    if (S == 5) { st_sc0(dst, a); st_sc0(dst + 1024, b); }
    if (S == 6) { st_sc1nt(dst, a); st_sc1nt(dst + 1024, b); }
    if (S == 7) { st_sc0nt(dst, a); st_sc0nt(dst + 1024, b); }
}

// plain copy, 16 bytes per lane per instruction, ITER instructions per lane, TPB threads per workgroup, no LDS
template <int TPB, int ITER>
__global__ __launch_bounds__(TPB) void kc(const char *in, char *out, size_t bytes)
{
    const size_t wave = ((size_t)blockIdx.x * TPB + threadIdx.x) >> 6;
    const size_t base = wave * (size_t)(ITER * 1024) + (threadIdx.x & 63) * 16;
    if (base + (ITER - 1) * 1024 + 16 > bytes) return;
    v4i v[ITER];
#pragma unroll
    for (int i = 0; i < ITER; ++i) v[i] = ld_nt(in + base + i * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < ITER; ++i) st_sc1nt(out + base + i * 1024, v[i]);
}
template <int TPB, int ITER>
static void runc(const char *in, char *out, size_t bytes)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t waves = bytes / (ITER * 1024), wgs = (waves + TPB / 64 - 1) / (TPB / 64);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((kc<TPB, ITER>), dim3((unsigned)wgs), dim3(TPB), 0, 0, in, out, bytes);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        if (rep && ms < best) best = ms;
    }
    printf("plain copy tpb=%3d, %d KiB per wave, no LDS : %.4f ms  %.3f TB/s\n", TPB, ITER, best, 2.0 * bytes / best / 1e9);
    fflush(stdout);
}

// conv-like pattern: a workgroup of 8 waves owns 8 tiles (2 KiB each) that lie PITCH bytes apart; every wave
// instruction takes one 128-byte line of each of the 8 tiles (lane = 8 * tile + chunk)
template <int PITCH_KB>
__global__ __launch_bounds__(512) void kc2(const char *in, char *out, size_t bytes)
{
    const size_t pitch = (size_t)PITCH_KB * 1024, tiles_per_row = pitch / 2048;
    const size_t g = blockIdx.x, band = g / tiles_per_row, col = g % tiles_per_row;      // band = 8 rows of `pitch` bytes
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane >> 3, q = lane & 7;
    const size_t base = (band * 8 + r) * pitch + col * 2048 + q * 16;
    if ((band * 8 + 8) * pitch > bytes) return;
    const v4i a = ld_nt(in + base + (2 * w) * 128), b = ld_nt(in + base + (2 * w + 1) * 128);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_sc1nt(out + base + (2 * w) * 128, a);
    st_sc1nt(out + base + (2 * w + 1) * 128, b);
}
template <int PITCH_KB>
static void runc2(const char *in, char *out, size_t bytes)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t wgs = bytes / (8 * 2048);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((kc2<PITCH_KB>), dim3((unsigned)wgs), dim3(512), 0, 0, in, out, bytes);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        if (rep && ms < best) best = ms;
    }
    printf("8-tile workgroups, tiles %4d KiB apart, line per tile per instruction : %.4f ms  %.3f TB/s\n", PITCH_KB, best, 2.0 * bytes / best / 1e9);
    fflush(stdout);
}

static const char *names[8] = {"-", "nt", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc0", "sc1 nt", "sc0 nt"};

template <int L, int S, int MODE>
static void run(const char *in, char *out, size_t tiles, size_t lds, unsigned spread = 1)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<L, S, MODE>), dim3((unsigned)tiles), dim3(64), lds, 0, in, out, tiles, spread);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        if (rep && ms < best) best = ms;
    }
    const double bytes = (MODE == 0 ? 4096.0 : 2048.0) * tiles;
    printf("%-5s spread=%5u load[%-10s] store[%-10s] lds=%5zu : %.4f ms  %.3f TB/s\n", MODE == 0 ? "copy" : (MODE == 1 ? "read" : "write"), spread,
           MODE == 2 ? "" : names[L], MODE == 1 ? "" : names[S], lds, best, bytes / best / 1e9);
    fflush(stdout);
}

int main()
{
    const size_t tiles = 1 << 20, bytes = tiles * 2048;
    char *in, *out;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(in, 1, bytes); hipMemset(out, 0, bytes); hipDeviceSynchronize();
    if (getenv("PLAIN_ONLY")) {
        for (int rep = 0; rep < 2; ++rep) {
            runc<64, 1>(in, out, bytes); runc<64, 2>(in, out, bytes); runc<64, 4>(in, out, bytes);
            runc<256, 1>(in, out, bytes); runc<256, 2>(in, out, bytes); runc<256, 4>(in, out, bytes);
            runc<512, 1>(in, out, bytes); runc<1024, 1>(in, out, bytes);
            run<1, 6, 0>(in, out, tiles, 8192, 1);
            runc2<2>(in, out, bytes); runc2<16>(in, out, bytes); runc2<64>(in, out, bytes); runc2<1024>(in, out, bytes);
        }
        return 0;
    }
    if (getenv("SPREAD_ONLY")) {
        for (int rep = 0; rep < 2; ++rep)
            for (unsigned sp : {1u, 2u, 8u, 64u, 256u, 1024u, 4096u, 16384u, 65536u}) { run<1, 6, 0>(in, out, tiles, 8192, sp); }
        for (unsigned sp : {1u, 64u, 4096u}) { run<1, 0, 1>(in, out, tiles, 8192, sp); run<0, 6, 2>(in, out, tiles, 8192, sp); }
        return 0;
    }
    for (size_t lds : {(size_t)8192}) {
#define COPY(L, S) run<L, S, 0>(in, out, tiles, lds)
        COPY(0, 0); COPY(1, 1); COPY(0, 1); COPY(1, 0);
        COPY(0, 2); COPY(0, 3); COPY(0, 4); COPY(0, 5); COPY(0, 6); COPY(0, 7);
        COPY(2, 1); COPY(3, 1); COPY(4, 1); COPY(5, 1); COPY(6, 1); COPY(7, 1);
        COPY(1, 4); COPY(4, 4); COPY(1, 6); COPY(6, 6); COPY(1, 7);
        COPY(0, 0); COPY(1, 1);
#define RD(L) run<L, 0, 1>(in, out, tiles, lds)
        RD(0); RD(1); RD(2); RD(3); RD(4); RD(5); RD(6); RD(7);
#define WR(S) run<0, S, 2>(in, out, tiles, lds)
        WR(0); WR(1); WR(2); WR(3); WR(4); WR(5); WR(6); WR(7);
    }
    return 0;
}
