// Developer probe: what ds_read_b64_tr_b8 (gfx950) returns.  Every lane l reads 8 bytes at its OWN address; LDS byte i holds i (two passes: low
// and high part), so the output shows which LDS byte each result byte came from, i.e. which (lane, element).
// hipcc --offload-arch=gfx950 -O2 -o tools/probes/lds_tr8_read_test tools/probes/lds_tr8_read_test.hip && tools/probes/lds_tr8_read_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint32_t *out, int mode, int pass)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[2048];
    const int l = threadIdx.x;
    for (int i = l; i < 2048; i += 64) lds[i] = (uint8_t)(pass ? i >> 8 : i);
    __syncthreads();
    // mode 0: lane l -> bytes 8l .. 8l+7
    // mode 1: lane l -> 8 bytes of a 32-byte-pitch image: row l (l < 64), byte column 0  -> address 32 l
    // mode 2: 32-byte pitch, lane l: row (l & 31), column 8 * (l >> 5)
    const unsigned a = mode == 0 ? 8u * l : mode == 1 ? 32u * (l & 31) + 8u * (l >> 5) : 32u * (l >> 1) + 8u * (l & 1);
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    u2 v;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(a + (unsigned)(reinterpret_cast<uintptr_t>(lds) & 0xFFFFu))) : "memory");
    out[2 * l] = v.x; out[2 * l + 1] = v.y;
}
int main()
{
    uint32_t *d, h[2][128];
    hipMalloc(&d, sizeof h[0]);
    for (int mode = 0; mode < 3; ++mode) {
        for (int pass = 0; pass < 2; ++pass) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode, pass);
            hipMemcpy(h[pass], d, sizeof h[0], hipMemcpyDeviceToHost);
        }
        printf("mode %d (source LDS byte index of each of the 8 result bytes; mode 0: /8 = source lane, %%8 = element)\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 8; ++j) {
                const unsigned lo = (h[0][2 * l + j / 4] >> (8 * (j & 3))) & 0xFF, hi = (h[1][2 * l + j / 4] >> (8 * (j & 3))) & 0xFF;
                const unsigned i = hi * 256 + lo;
                if (mode == 0) printf(" (%2u,%u)", i / 8, i % 8); else printf(" r%2u c%2u |", i / 32, i % 32);
            }
            printf("\n");
        }
    }
    return 0;
}
