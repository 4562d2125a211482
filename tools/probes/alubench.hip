// alubench.hip -- developer probe: issue rate of the VALU ops the kernels lean on (cycles per wave64 instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)
template <int OP> __global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed, int iters) {
    unsigned a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 7u + i * 13u + seed;
    unsigned b = seed * 3u + threadIdx.x, c = seed ^ 0x5555u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = a[i] + b;
                if (OP == 1) a[i] = __builtin_amdgcn_sad_u16(a[i], b, c) ;
                if (OP == 2) a[i] = __builtin_amdgcn_perm(a[i], b, 0x05040100u + c);
                if (OP == 3) a[i] = (a[i] << 8) + b;
                if (OP == 4) a[i] = a[i] * b;
                if (OP == 5) a[i] = __builtin_amdgcn_alignbit(a[i], b, c & 31);
                if (OP == 6) { auto r = __builtin_amdgcn_permlane32_swap(a[i], b, false, false); a[i] = r[0]; b = r[1]; }
                if (OP == 7) a[i] = __builtin_amdgcn_sad_u8(a[i], b, c);
                if (OP == 8) a[i] = (int)a[i] >> 4;
                if (OP == 9) a[i] = a[i] < b ? a[i] : b;
                if (OP == 10) { typedef short v2s __attribute__((ext_vector_type(2))); v2s r = __builtin_amdgcn_cvt_pk_i16((int)a[i], (int)b); a[i] = __builtin_bit_cast(unsigned, r) + c; }
                if (OP == 11) { int v = (int)a[i]; a[i] = (unsigned)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)) + c; }
                if (OP == 12) { typedef unsigned short v2u __attribute__((ext_vector_type(2))); v2u x = __builtin_bit_cast(v2u, a[i]), y = __builtin_bit_cast(v2u, b), z = __builtin_bit_cast(v2u, c); a[i] = __builtin_bit_cast(unsigned, (v2u)(x * y + z)); }
                if (OP == 13) a[i] = __builtin_amdgcn_udot4(a[i], b, c, false);
                if (OP == 14) a[i] = __builtin_amdgcn_ubfe(a[i], 5u, 8u) + b;
            }
        }
    }
    unsigned s = 0; for (int i = 0; i < 8; ++i) s += a[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = s + b;
}
typedef void (*kt)(unsigned*, unsigned, int);
int main() {
    unsigned* out; CK(hipMalloc(&out, 256 * 2048 * 4 * 4)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[] = {"v_add_u32", "v_sad_u16", "v_perm_b32", "v_lshl_add_u32", "v_mul_lo_u32", "v_alignbit_b32", "v_permlane32_swap", "v_sad_u8", "v_ashrrev_i32", "v_min_u32", "v_cvt_pk_i16_i32 (+add)", "v_med3_i32 (+add)", "v_pk_mad_u16", "v_dot4_u32_u8", "v_bfe_u32 (+add)"};
    kt ks[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>};
    const int iters = 2000, wgs = 256 * 8;       // 8 WGs of 256 per CU = 8 waves per SIMD
    for (int o = 0; o < 15; ++o) {
        hipLaunchKernelGGL(ks[o], dim3(wgs), dim3(256), 0, 0, out, 1u, 10);
        CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(ks[o], dim3(wgs), dim3(256), 0, 0, out, 1u, iters); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double instr_per_simd = (double)iters * 64 * (wgs * 4.0 / 1024.0);   // wave-instructions per SIMD
        printf("%-18s %.3f ms  -> %.2f cycles per wave64 instruction at 2.4 GHz (%.2f at 2.0)\n", names[o], ms, ms * 1e-3 * 2.4e9 / instr_per_simd, ms * 1e-3 * 2.0e9 / instr_per_simd);
    }
    return 0;
}
