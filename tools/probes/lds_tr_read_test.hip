// Developer probe: what ds_read_b64_tr_b16 (gfx950) returns.  Every lane l reads 8 bytes at its OWN address (here: element 4*l of an
// array holding its own index in every 16-bit element); the output shows which (lane, element) each result half-word came from.
// hipcc --offload-arch=gfx950 -O2 -o lds_tr_read_test lds_tr_read_test.hip && ./lds_tr_read_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint32_t *out, int mode)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    // mode 0: lane l -> elements 4l .. 4l+3 (so value = 4 * source_lane + source_element)
    // mode 1: lane l -> row (l >> 2) of a 64-byte-pitch image, column quad (l & 3): address = (l >> 2) * 32 + (l & 3) * 4 elements
    const unsigned addr_elems = mode == 0 ? 4u * l : (unsigned)((l >> 2) * 32 + (l & 3) * 4);
    const unsigned addr = (unsigned)(uintptr_t)lds + addr_elems * 2u;   // LDS byte address (low 32 bits of the generic pointer are the LDS offset)
    uint32_t lo, hi;
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    u2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(addr_elems * 2u + (unsigned)(reinterpret_cast<uintptr_t>(lds) & 0xFFFFu))) : "memory");
    lo = v.x; hi = v.y;
    out[2 * l] = lo; out[2 * l + 1] = hi;
    (void)addr;
}
int main()
{
    uint32_t *d, h[128];
    hipMalloc(&d, sizeof h);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) {
            const unsigned e[4] = {h[2 * l] & 0xFFFF, h[2 * l] >> 16, h[2 * l + 1] & 0xFFFF, h[2 * l + 1] >> 16};
            if (mode == 0) printf("lane %2d: (%2u,%u) (%2u,%u) (%2u,%u) (%2u,%u)\n", l, e[0] / 4, e[0] % 4, e[1] / 4, e[1] % 4, e[2] / 4, e[2] % 4, e[3] / 4, e[3] % 4);
            else printf("lane %2d: r%2u c%2u | r%2u c%2u | r%2u c%2u | r%2u c%2u\n", l, e[0] / 32, e[0] % 32, e[1] / 32, e[1] % 32, e[2] / 32, e[2] % 32, e[3] / 32, e[3] % 32);
        }
    }
    return 0;
}
