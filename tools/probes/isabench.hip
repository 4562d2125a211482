// isabench.hip -- developer probe: issue cost of single gfx950 VALU / DS instructions, written as inline asm so that
// the compiler cannot fold, reorder or strength-reduce them (tools/probes/alubench.hip's C-level loops lost several ops
// that way).  8 independent destination registers per op, 64 instructions per loop trip, 8 waves per SIMD.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/isabench tools/probes/isabench.hip       run: tools/probes/isabench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define BODY(INS)                                                                                   \
    for (int it = 0; it < iters; ++it) {                                                            \
        _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                             \
            asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                    \
                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b) \
                         : "v"(c), "s"(sc), "v"(d));                                                \
        }                                                                                           \
    }

#define I_ADD(i)      "v_add_u32 %" #i ", %" #i ", %9\n"
#define I_XOR(i)      "v_xor_b32 %" #i ", %" #i ", %9\n"
#define I_XORS(i)     "v_xor_b32 %" #i ", %10, %" #i "\n"
#define I_MIN(i)      "v_min_u32 %" #i ", %" #i ", %9\n"
#define I_MIN3(i)     "v_min3_u32 %" #i ", %" #i ", %9, %11\n"
#define I_LSHR(i)     "v_lshrrev_b32 %" #i ", 2, %" #i "\n"
#define I_LSHLOR(i)   "v_lshl_or_b32 %" #i ", %" #i ", 16, %9\n"
#define I_ANDOR(i)    "v_and_or_b32 %" #i ", %" #i ", %9, %11\n"
#define I_BFI(i)      "v_bfi_b32 %" #i ", %9, %" #i ", %11\n"
#define I_BFE(i)      "v_bfe_u32 %" #i ", %" #i ", 2, 16\n"
#define I_PERM(i)     "v_perm_b32 %" #i ", %" #i ", %9, %11\n"
#define I_PACK(i)     "v_pack_b32_f16 %" #i ", %" #i ", %9\n"
#define I_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %9, 8\n"
#define I_ALIGNBYTE(i) "v_alignbyte_b32 %" #i ", %" #i ", %9, 1\n"
#define I_SAD16(i)    "v_sad_u16 %" #i ", %9, %11, %" #i "\n"
#define I_SAD16S(i)   "v_sad_u16 %" #i ", %9, %10, %" #i "\n"
#define I_SAD8S(i)    "v_sad_u8 %" #i ", %9, %10, %" #i "\n"
#define I_SADU32(i)   "v_sad_u32 %" #i ", %9, %10, %" #i "\n"
#define I_ADD3(i)     "v_add3_u32 %" #i ", %" #i ", %9, %11\n"
#define I_LSHLADD(i)  "v_lshl_add_u32 %" #i ", %" #i ", 16, %9\n"
#define I_CNDMASK(i)  "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define I_PKSUB(i)    "v_pk_sub_i16 %" #i ", %" #i ", %9\n"
#define I_PKMAX(i)    "v_pk_max_i16 %" #i ", %" #i ", %9\n"
#define I_PKADD(i)    "v_pk_add_u16 %" #i ", %" #i ", %9\n"
#define I_MOV(i)      "v_mov_b32 %" #i ", %9\n"
#define I_MOVDPP(i)   "v_mov_b32_dpp %" #i ", %9 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_SWAP(i)     "v_permlane32_swap_b32 %" #i ", %8\n"
#define I_MADU16(i)   "v_mad_u32_u16 %" #i ", %9, %11, %" #i "\n"
#define I_BPERM(i)    "ds_bpermute_b32 %" #i ", %11, %" #i "\n"
#define I_SUBABS(i)   "v_sub_u32 %" #i ", %" #i ", %9\n v_max_i32 %" #i ", %" #i ", %9\n"
#define I_CMPCND(i)   "v_cmp_lt_u32 vcc, %" #i ", %9\n v_cndmask_b32 %" #i ", %9, %" #i ", vcc\n"
#define I_DOT2(i)     "v_dot2_i32_i16 %" #i ", %9, %11, %" #i "\n"
#define I_MSAD(i)     "v_msad_u8 %" #i ", %9, %10, %" #i "\n"
#define I_DSMIN(i)    "ds_min_u32 %11, %" #i " offset:" #i "024\n"
#define I_DSWRITE(i)  "ds_write_b32 %11, %" #i " offset:" #i "024\n"
#define I_DSMIN_SAD(i) "ds_min_u32 %11, %" #i " offset:" #i "024\n v_sad_u16 %" #i ", %9, %10, %" #i "\n v_sad_u16 %" #i ", %9, %10, %" #i "\n v_sad_u16 %" #i ", %9, %10, %" #i "\n v_sad_u16 %" #i ", %9, %10, %" #i "\n v_sad_u16 %" #i ", %9, %10, %" #i "\n v_sad_u16 %" #i ", %9, %10, %" #i "\n v_sad_u16 %" #i ", %9, %10, %" #i "\n v_sad_u16 %" #i ", %9, %10, %" #i "\n"

template <int OP> __global__ __launch_bounds__(256) void k(unsigned *out, unsigned seed, int iters)
{
    __shared__ unsigned lds_buf[4096];
    if (seed == 12345u) out[0] = lds_buf[threadIdx.x];
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 7u + i * 13u + seed;
    unsigned b = seed * 3u + threadIdx.x, c = seed ^ 0x5555u, d = (threadIdx.x ^ 32u) * 4u;
    unsigned sc = __builtin_amdgcn_readfirstlane(seed * 77u + 5u);
    if (OP == 0) BODY(I_ADD)
    if (OP == 1) BODY(I_XOR)
    if (OP == 2) BODY(I_XORS)
    if (OP == 3) BODY(I_MIN)
    if (OP == 4) BODY(I_MIN3)
    if (OP == 5) BODY(I_LSHR)
    if (OP == 6) BODY(I_LSHLOR)
    if (OP == 7) BODY(I_ANDOR)
    if (OP == 8) BODY(I_BFI)
    if (OP == 9) BODY(I_BFE)
    if (OP == 10) BODY(I_PERM)
    if (OP == 11) BODY(I_PACK)
    if (OP == 12) BODY(I_ALIGNBIT)
    if (OP == 13) BODY(I_ALIGNBYTE)
    if (OP == 14) BODY(I_SAD16)
    if (OP == 15) BODY(I_SAD16S)
    if (OP == 16) BODY(I_SAD8S)
    if (OP == 17) BODY(I_SADU32)
    if (OP == 18) BODY(I_ADD3)
    if (OP == 19) BODY(I_LSHLADD)
    if (OP == 20) BODY(I_CNDMASK)
    if (OP == 21) BODY(I_PKSUB)
    if (OP == 22) BODY(I_PKMAX)
    if (OP == 23) BODY(I_PKADD)
    if (OP == 24) BODY(I_MOV)
    if (OP == 25) BODY(I_MOVDPP)
    if (OP == 26) BODY(I_SWAP)
    if (OP == 27) BODY(I_MADU16)
    if (OP == 28) { BODY(I_BPERM) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if (OP == 29) BODY(I_SUBABS)
    if (OP == 30) BODY(I_CMPCND)
    if (OP == 31) BODY(I_DOT2)
    if (OP == 32) BODY(I_MSAD)
    if (OP == 33) { d = threadIdx.x * 4u; BODY(I_DSMIN) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if (OP == 34) { d = threadIdx.x * 4u; BODY(I_DSWRITE) asm volatile("s_waitcnt lgkmcnt(0)"); }
    if (OP == 35) { d = threadIdx.x * 4u; BODY(I_DSMIN_SAD) asm volatile("s_waitcnt lgkmcnt(0)"); }
    unsigned s = b;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef void (*kt)(unsigned *, unsigned, int);
int main(int argc, char **argv)
{
    unsigned *out;
    CK(hipMalloc(&out, 256 * 2048 * 4 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char *names[] = {"v_add_u32", "v_xor_b32", "v_xor_b32 (sgpr)", "v_min_u32", "v_min3_u32", "v_lshrrev_b32", "v_lshl_or_b32", "v_and_or_b32",
                           "v_bfi_b32", "v_bfe_u32", "v_perm_b32", "v_pack_b32_f16", "v_alignbit_b32", "v_alignbyte_b32", "v_sad_u16 (vgpr,vgpr)",
                           "v_sad_u16 (vgpr,sgpr)", "v_sad_u8 (vgpr,sgpr)", "v_sad_u32 (vgpr,sgpr)", "v_add3_u32", "v_lshl_add_u32", "v_cndmask_b32",
                           "v_pk_sub_i16", "v_pk_max_i16", "v_pk_add_u16", "v_mov_b32", "v_mov_b32_dpp", "v_permlane32_swap", "v_mad_u32_u16",
                           "ds_bpermute_b32", "v_sub_u32+v_max_i32 (2 instr)", "v_cmp+v_cndmask (2 instr)", "v_dot2_i32_i16", "v_msad_u8", "ds_min_u32 (lane-consecutive)", "ds_write_b32 (lane-consecutive)", "ds_min_u32 + 8 v_sad_u16 (9 instr)"};
    kt ks[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>, k<15>, k<16>, k<17>, k<18>, k<19>, k<20>,
               k<21>, k<22>, k<23>, k<24>, k<25>, k<26>, k<27>, k<28>, k<29>, k<30>, k<31>, k<32>, k<33>, k<34>, k<35>};
    const int iters = 1000, n_ops = sizeof(ks) / sizeof(ks[0]);
    for (int wps = 4; wps >= 2; wps /= 2) {                          // waves per SIMD: 8, 4, 2, 1
        const int wgs = 256 * wps;
        printf("---- %d wave(s) per SIMD ----\n", wps);
        for (int o = (argc > 1 ? atoi(argv[1]) : 0); o < n_ops; ++o) {
            hipLaunchKernelGGL(ks[o], dim3(wgs), dim3(256), 0, 0, out, 1u, 200);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(ks[o], dim3(wgs), dim3(256), 0, 0, out, 1u, iters);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_simd = (double)iters * 64 * (wgs * 4.0 / 1024.0);   // asm groups issued per SIMD
            printf("%-30s %.3f ms  -> %.2f cycles per wave64 issue at 2.4 GHz\n", names[o], ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
        }
    }
    return 0;
}
