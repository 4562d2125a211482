// isabench_sad.hip -- developer probe: issue cost of the byte-SAD family on gfx950 (inline asm, 8 independent accumulators,
// 64 instructions per loop trip), for the SAD motion search: v_sad_u8, v_sad_hi_u8, v_qsad_pk_u16_u8 (four SADs of four
// bytes at byte offsets 0..3, packed 16-bit accumulate), v_mqsad_pk_u16_u8, v_mqsad_u32_u8.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/isabench_sad tools/probes/isabench_sad.hip       run: tools/probes/isabench_sad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

#define BODY32(INS)                                                                                 \
    for (int it = 0; it < iters; ++it) {                                                            \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                             \
            asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                    \
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                         : "v"(c), "s"(sc), "v"(q));                                                \
        }                                                                                           \
    }
#define BODY64(INS)                                                                                 \
    for (int it = 0; it < iters; ++it) {                                                            \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                             \
            asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                    \
                         : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) \
                         : "v"(c), "s"(sc), "v"(q));                                                \
        }                                                                                           \
    }
#define I_SAD8(i)     "v_sad_u8 %" #i ", %8, %9, %" #i "\n"
#define I_SADHI8(i)   "v_sad_hi_u8 %" #i ", %8, %9, %" #i "\n"
#define I_QSAD(i)     "v_qsad_pk_u16_u8 %" #i ", %10, %9, %" #i "\n"
#define I_QSADV(i)    "v_qsad_pk_u16_u8 %" #i ", %10, %8, %" #i "\n"
#define I_MQSAD(i)    "v_mqsad_pk_u16_u8 %" #i ", %10, %9, %" #i "\n"
#define I_PKMIN(i)    "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define I_MIN(i)      "v_min_u32 %" #i ", %" #i ", %8\n"

template <int OP> __global__ __launch_bounds__(256) void k(unsigned long long *out, unsigned seed, int iters)
{
    unsigned a[8];
    unsigned long long w[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 7u + i * 13u + seed; w[i] = (unsigned long long)a[i] * 0x100000001ull; }
    unsigned c = seed ^ 0x5555u;
    unsigned long long q = (unsigned long long)(threadIdx.x * 0x01010101u) << 16 | seed;
    unsigned sc = __builtin_amdgcn_readfirstlane(seed * 77u + 5u);
    if (OP == 0) BODY32(I_SAD8)
    if (OP == 1) BODY32(I_SADHI8)
    if (OP == 2) BODY64(I_QSAD)
    if (OP == 3) BODY64(I_QSADV)
    if (OP == 4) BODY64(I_MQSAD)
    if (OP == 5) BODY32(I_PKMIN)
    if (OP == 6) BODY32(I_MIN)
    unsigned long long s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef void (*kt)(unsigned long long *, unsigned, int);
int main()
{
    unsigned long long *out;
    CK(hipMalloc(&out, 256 * 2048 * 4 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[] = {"v_sad_u8 (vgpr,sgpr)", "v_sad_hi_u8 (vgpr,sgpr)", "v_qsad_pk_u16_u8 (v64,sgpr)", "v_qsad_pk_u16_u8 (v64,vgpr)", "v_mqsad_pk_u16_u8 (v64,sgpr)", "v_pk_min_u16", "v_min_u32"};
    kt ks[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>};
    const int iters = 1000, n_ops = sizeof(ks) / sizeof(ks[0]);
    for (int wps = 4; wps >= 1; wps /= 2) {                          // waves per SIMD: 4, 2, 1
        const int wgs = 256 * wps;
        printf("---- %d wave(s) per SIMD ----\n", wps);
        for (int o = 0; o < n_ops; ++o) {
            hipLaunchKernelGGL(ks[o], dim3(wgs), dim3(256), 0, 0, out, 1u, 200);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(ks[o], dim3(wgs), dim3(256), 0, 0, out, 1u, iters);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_simd = (double)iters * 64 * (wgs * 4.0 / 1024.0);
            printf("%-32s %.3f ms  -> %.2f cycles per wave64 issue at 2.4 GHz\n", names[o], ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
        }
    }
    // semantics check of v_qsad_pk_u16_u8 on one lane (host restatement)
    return 0;
}
