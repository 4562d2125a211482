// membench.hip -- developer probe: what does the memory system give a 2 KiB-per-wave
// streaming kernel for different lane->address patterns?  (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool NT> __device__ __forceinline__ v4i ld(const void* p) { const v4i* q = (const v4i*)p; if (NT) return __builtin_nontemporal_load(q); return *q; }
template <bool NT> __device__ __forceinline__ void st(void* p, v4i v) { v4i* q = (v4i*)p; if (NT) __builtin_nontemporal_store(v, q); else *q = v; }

// PAT 0: linear  (lane*16, +1024)     PAT 1: dct pattern (64*(l&31)+32*(l>>5), +16)
// PAT 2: 32B/lane (lane*32, +16)      PAT 3: sector pattern (64*(l&31)+16*(l>>5), +32)
template <int PAT> __device__ __forceinline__ void offs(int lane, size_t& o0, size_t& o1) {
    if (PAT == 0) { o0 = lane * 16; o1 = o0 + 1024; }
    if (PAT == 1) { o0 = (lane & 31) * 64 + (lane >> 5) * 32; o1 = o0 + 16; }
    if (PAT == 2) { o0 = lane * 32; o1 = o0 + 16; }
    if (PAT == 3) { o0 = (lane & 31) * 64 + (lane >> 5) * 16; o1 = o0 + 32; }
}

template <int PAT, bool NTL, bool NTS, int DEPTH>
__global__ __launch_bounds__(256) void copy_k(const char* __restrict__ in, char* __restrict__ out, size_t nblk) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    size_t o0, o1; offs<PAT>(lane, o0, o1);
    if (DEPTH == 1) {
        for (size_t b = wave; b < nblk; b += nw) {
            v4i a = ld<NTL>(in + b * 2048 + o0), c = ld<NTL>(in + b * 2048 + o1);
            a[0] ^= 1; c[1] += 3;
            st<NTS>(out + b * 2048 + o0, a); st<NTS>(out + b * 2048 + o1, c);
        }
    } else {
        size_t b = wave; if (b >= nblk) return;
        v4i a = ld<NTL>(in + b * 2048 + o0), c = ld<NTL>(in + b * 2048 + o1);
        for (;;) {
            size_t nb = b + nw; v4i na = a, nc = c;
            if (nb < nblk) { na = ld<NTL>(in + nb * 2048 + o0); nc = ld<NTL>(in + nb * 2048 + o1); }
            a[0] ^= 1; c[1] += 3;
            st<NTS>(out + b * 2048 + o0, a); st<NTS>(out + b * 2048 + o1, c);
            if (nb >= nblk) break; b = nb; a = na; c = nc;
        }
    }
}

// read-only and write-only variants
template <int PAT, bool NT> __global__ __launch_bounds__(256) void read_k(const char* __restrict__ in, char* __restrict__ out, size_t nblk) {
    const int lane = threadIdx.x & 63; const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    size_t o0, o1; offs<PAT>(lane, o0, o1); v4i acc = {0,0,0,0};
    for (size_t b = wave; b < nblk; b += nw) { acc += ld<NT>(in + b * 2048 + o0); acc += ld<NT>(in + b * 2048 + o1); }
    if (acc[0] == 0x12345 && acc[1] == 0x777) st<false>(out + o0, acc);
}
template <int PAT, bool NT> __global__ __launch_bounds__(256) void write_k(const char* __restrict__ in, char* __restrict__ out, size_t nblk) {
    const int lane = threadIdx.x & 63; const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; const size_t nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    size_t o0, o1; offs<PAT>(lane, o0, o1); v4i v = {lane, 1, 2, 3};
    for (size_t b = wave; b < nblk; b += nw) { st<NT>(out + b * 2048 + o0, v); st<NT>(out + b * 2048 + o1, v); }
    (void)in;
}

typedef void (*kern_t)(const char*, char*, size_t);
struct Case { const char* name; kern_t k; double bytes_per_blk; };

int main(int argc, char** argv) {
    size_t nblk = 1 << 20; if (argc > 1) nblk = atol(argv[1]);
    char *in, *out; CK(hipMalloc(&in, nblk * 2048)); CK(hipMalloc(&out, nblk * 2048));
    CK(hipMemset(in, 1, nblk * 2048)); CK(hipMemset(out, 0, nblk * 2048));
    std::vector<Case> cases = {
        {"copy lin  ld/st   d1", copy_k<0,false,false,1>, 4096}, {"copy lin  ld/st   d2", copy_k<0,false,false,2>, 4096},
        {"copy lin  ntl/nts d1", copy_k<0,true,true,1>, 4096},   {"copy lin  ntl/nts d2", copy_k<0,true,true,2>, 4096},
        {"copy lin  ld/nts  d2", copy_k<0,false,true,2>, 4096},  {"copy lin  ntl/st  d2", copy_k<0,true,false,2>, 4096},
        {"copy dct  ld/st   d1", copy_k<1,false,false,1>, 4096}, {"copy dct  ld/st   d2", copy_k<1,false,false,2>, 4096},
        {"copy dct  ntl/nts d2", copy_k<1,true,true,2>, 4096},   {"copy dct  ld/nts  d2", copy_k<1,false,true,2>, 4096}, {"copy dct  ntl/st  d2", copy_k<1,true,false,2>, 4096},
        {"copy 32B  ld/st   d2", copy_k<2,false,false,2>, 4096}, {"copy sect ld/st   d2", copy_k<3,false,false,2>, 4096},
        {"copy sect ntl/nts d2", copy_k<3,true,true,2>, 4096},
        {"read lin  ld", read_k<0,false>, 2048}, {"read lin  nt", read_k<0,true>, 2048}, {"read dct  ld", read_k<1,false>, 2048}, {"read sect ld", read_k<3,false>, 2048},
        {"write lin st", write_k<0,false>, 2048}, {"write lin nt", write_k<0,true>, 2048}, {"write dct st", write_k<1,false>, 2048}, {"write sect st", write_k<3,false>, 2048},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int wgs_list[] = {2, 4, 8};
    for (auto& c : cases) {
        printf("%-22s", c.name);
        for (int wg : wgs_list) {
            dim3 grid(256 * wg), block(256);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(c.k, grid, block, 0, 0, in, out, nblk);
            CK(hipEventRecord(e0, 0));
            const int reps = 20;
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(c.k, grid, block, 0, 0, in, out, nblk);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
            printf("  wg%d: %.3f ms %5.2f TB/s", wg, ms, c.bytes_per_blk * nblk / ms * 1e3 / 1e12);
        }
        printf("\n");
    }
    return 0;
}
