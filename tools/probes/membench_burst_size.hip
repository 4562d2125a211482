// membench2.hip -- developer probe #2: sweep launch geometry / buffer placement / per-wave
// burst size for a streaming copy with the DCT kernel's lane->address pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// BPI blocks (2 KiB each, contiguous) per wave-iteration; PAT 0 linear / 1 dct
template <int PAT, int BPI, int TPB>
__global__ __launch_bounds__(TPB) void copy_k(const char* __restrict__ in, char* __restrict__ out, size_t ngrp) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    size_t o0, o1;
    if (PAT == 0) { o0 = lane * 16; o1 = o0 + 1024; } else { o0 = (lane & 31) * 64 + (lane >> 5) * 32; o1 = o0 + 16; }
    for (size_t g = wave; g < ngrp; g += nw) {
        v4i a[BPI], c[BPI];
#pragma unroll
        for (int i = 0; i < BPI; ++i) { a[i] = *(const v4i*)(in + (g * BPI + i) * 2048 + o0); c[i] = *(const v4i*)(in + (g * BPI + i) * 2048 + o1); }
#pragma unroll
        for (int i = 0; i < BPI; ++i) { a[i][0] ^= 1; c[i][1] += 3; *(v4i*)(out + (g * BPI + i) * 2048 + o0) = a[i]; *(v4i*)(out + (g * BPI + i) * 2048 + o1) = c[i]; }
    }
}

typedef void (*kern_t)(const char*, char*, size_t);
template <int PAT, int BPI> kern_t pick_tpb(int tpb) {
    switch (tpb) { case 64: return copy_k<PAT, BPI, 64>; case 128: return copy_k<PAT, BPI, 128>; case 512: return copy_k<PAT, BPI, 512>; case 1024: return copy_k<PAT, BPI, 1024>; default: return copy_k<PAT, BPI, 256>; }
}
template <int PAT> kern_t pick_bpi(int bpi, int tpb) {
    switch (bpi) { case 2: return pick_tpb<PAT, 2>(tpb); case 4: return pick_tpb<PAT, 4>(tpb); case 8: return pick_tpb<PAT, 8>(tpb); default: return pick_tpb<PAT, 1>(tpb); }
}

int main() {
    const size_t nblk = 1 << 20;
    char *in, *outbase; const size_t slack = 64 << 20;
    CK(hipMalloc(&in, nblk * 2048)); CK(hipMalloc(&outbase, nblk * 2048 + slack));
    CK(hipMemset(in, 1, nblk * 2048)); CK(hipMemset(outbase, 0, nblk * 2048 + slack));
    printf("in=%p out=%p\n", (void*)in, (void*)outbase);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* tag, int pat, int bpi, int tpb, int waves_per_cu, size_t out_off) {
        kern_t k = pat ? pick_bpi<1>(bpi, tpb) : pick_bpi<0>(bpi, tpb);
        int wgs = 256 * waves_per_cu * 64 / tpb; if (wgs < 1) wgs = 1;
        char* out = outbase + out_off;
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(tpb), 0, 0, in, out, nblk / bpi);
        CK(hipEventRecord(e0, 0)); const int reps = 20;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(tpb), 0, 0, in, out, nblk / bpi);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-10s pat=%d bpi=%d tpb=%4d waves/cu=%2d off=%9zu : %.3f ms %5.2f TB/s\n", tag, pat, bpi, tpb, waves_per_cu, out_off, ms, 4096.0 * nblk / ms * 1e3 / 1e12);
    };
    for (int pat = 0; pat < 2; ++pat)
        for (int w : {4, 6, 8, 12, 16, 24, 32}) run("waves", pat, 1, 256, w, 0);
    for (int pat = 0; pat < 2; ++pat)
        for (int bpi : {2, 4, 8}) for (int w : {4, 8, 16}) run("burst", pat, bpi, 256, w, 0);
    for (int tpb : {64, 128, 512, 1024}) for (int w : {8, 16}) run("tpb", 1, 1, tpb, w, 0);
    for (size_t off : {(size_t)0, (size_t)256, (size_t)1024, (size_t)4096, (size_t)(64 << 10), (size_t)(1 << 20) + 4096, (size_t)(2 << 20), (size_t)(3 << 20) + (192 << 10), (size_t)(32 << 20)})
        for (int w : {8, 16}) run("offset", 1, 1, 256, w, off);
    return 0;
}
