// hostpipe_probe: why does the host-pointer batch API move only ~27 GB/s each way over a link that gives 57 one way / 48 both ways?
// Variants of the staging pipeline (32 MiB chunks of a 256 MiB batch, a trivial kernel between the copies):
//   A  one stream per staging slot: H2D, kernel, D2H in stream order (what the library did in rounds 1-3)
//   B  one stream per ENGINE: all H2D copies on one stream, all kernels on a second, all D2H copies on a third, ordered by events
//   C  B for PAGEABLE buffers with a second host thread: hipMemcpyAsync from / to pageable memory blocks the calling thread while the runtime stages
//      the copy, so one thread can keep only one direction busy -- the uploads stay on the caller's thread, the downloads get their own
//   each with pinned (hipHostMalloc), pageable and pageable + hipHostRegister host buffers, 2..4 slots, 8..64 MiB chunks.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/hostpipe_probe tools/probes/hostpipe_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void touch(const v4i *in, v4i *out, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { v4i v = in[i]; v[0] ^= 1; out[i] = v; } }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double run(char variant, const char *in, char *out, size_t bytes, size_t chunk, int slots)
{
    std::vector<void *> din(slots), dout(slots);
    std::vector<hipStream_t> st(slots);
    std::vector<hipEvent_t> e_up(slots), e_k(slots), e_down(slots);
    hipStream_t s_up, s_k, s_down;
    CK(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking));
    for (int i = 0; i < slots; ++i) {
        CK(hipMalloc(&din[i], chunk)); CK(hipMalloc(&dout[i], chunk)); CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
        CK(hipEventCreateWithFlags(&e_up[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e_k[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e_down[i], hipEventDisableTiming));
    }
    double best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        std::vector<bool> used(slots, false);
        int k = 0;
        for (size_t done = 0; done < bytes; done += chunk, k = (k + 1) % slots) {
            const size_t cnt = bytes - done < chunk ? bytes - done : chunk;
            const unsigned grid = (unsigned)((cnt / 16 + 255) / 256);
            if (variant == 'A') {
                CK(hipStreamSynchronize(st[k]));
                CK(hipMemcpyAsync(din[k], in + done, cnt, hipMemcpyHostToDevice, st[k]));
                hipLaunchKernelGGL(touch, dim3(grid), dim3(256), 0, st[k], (const v4i *)din[k], (v4i *)dout[k], cnt / 16);
                CK(hipMemcpyAsync(out + done, dout[k], cnt, hipMemcpyDeviceToHost, st[k]));
            } else {
                if (used[k]) CK(hipEventSynchronize(e_down[k]));          // the slot's previous chunk has left
                CK(hipMemcpyAsync(din[k], in + done, cnt, hipMemcpyHostToDevice, s_up));
                CK(hipEventRecord(e_up[k], s_up));
                CK(hipStreamWaitEvent(s_k, e_up[k], 0));
                hipLaunchKernelGGL(touch, dim3(grid), dim3(256), 0, s_k, (const v4i *)din[k], (v4i *)dout[k], cnt / 16);
                CK(hipEventRecord(e_k[k], s_k));
                CK(hipStreamWaitEvent(s_down, e_k[k], 0));
                CK(hipMemcpyAsync(out + done, dout[k], cnt, hipMemcpyDeviceToHost, s_down));
                CK(hipEventRecord(e_down[k], s_down));
                used[k] = true;
            }
        }
        CK(hipDeviceSynchronize());
        const double dt = now() - t0;
        if (dt < best) best = dt;
    }
    for (int i = 0; i < slots; ++i) { CK(hipFree(din[i])); CK(hipFree(dout[i])); CK(hipStreamDestroy(st[i])); }
    CK(hipStreamDestroy(s_up)); CK(hipStreamDestroy(s_k)); CK(hipStreamDestroy(s_down));
    return best;
}

static double run_c(const char *in, char *out, size_t bytes, size_t chunk, int slots)
{
    std::vector<void *> din(slots), dout(slots);
    std::vector<hipEvent_t> e_k(slots);
    hipStream_t s_up, s_k, s_down;
    CK(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_k, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking));
    for (int i = 0; i < slots; ++i) { CK(hipMalloc(&din[i], chunk)); CK(hipMalloc(&dout[i], chunk)); CK(hipEventCreateWithFlags(&e_k[i], hipEventDisableTiming)); }
    const size_t n_chunks = (bytes + chunk - 1) / chunk;
    double best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipDeviceSynchronize());
        std::atomic<long> launched{0}, drained{0};
        const double t0 = now();
        std::thread down([&]() {
            for (size_t i = 0; i < n_chunks; ++i) {
                while (launched.load(std::memory_order_acquire) <= (long)i) std::this_thread::yield();
                const int k = (int)(i % slots);
                const size_t done = i * chunk, cnt = bytes - done < chunk ? bytes - done : chunk;
                CK(hipStreamWaitEvent(s_down, e_k[k], 0));
                CK(hipMemcpyAsync(out + done, dout[k], cnt, hipMemcpyDeviceToHost, s_down));
                CK(hipStreamSynchronize(s_down));
                drained.store((long)i + 1, std::memory_order_release);
            }
        });
        for (size_t i = 0; i < n_chunks; ++i) {
            const int k = (int)(i % slots);
            const size_t done = i * chunk, cnt = bytes - done < chunk ? bytes - done : chunk;
            while ((long)i - drained.load(std::memory_order_acquire) >= slots) std::this_thread::yield();      // the slot's previous chunk has left
            CK(hipMemcpyAsync(din[k], in + done, cnt, hipMemcpyHostToDevice, s_up));
            CK(hipStreamSynchronize(s_up));
            hipLaunchKernelGGL(touch, dim3((unsigned)((cnt / 16 + 255) / 256)), dim3(256), 0, s_k, (const v4i *)din[k], (v4i *)dout[k], cnt / 16);
            CK(hipEventRecord(e_k[k], s_k));
            launched.store((long)i + 1, std::memory_order_release);
        }
        down.join();
        CK(hipDeviceSynchronize());
        const double dt = now() - t0;
        if (dt < best) best = dt;
    }
    for (int i = 0; i < slots; ++i) { CK(hipFree(din[i])); CK(hipFree(dout[i])); }
    CK(hipStreamDestroy(s_up)); CK(hipStreamDestroy(s_k)); CK(hipStreamDestroy(s_down));
    return best;
}

int main()
{
    const size_t bytes = (size_t)256 << 20;
    char *pin_in, *pin_out;
    CK(hipHostMalloc((void **)&pin_in, bytes, hipHostMallocDefault)); CK(hipHostMalloc((void **)&pin_out, bytes, hipHostMallocDefault));
    char *pg_in = (char *)aligned_alloc(4096, bytes), *pg_out = (char *)aligned_alloc(4096, bytes);
    memset(pin_in, 1, bytes); memset(pin_out, 0, bytes); memset(pg_in, 1, bytes); memset(pg_out, 0, bytes);
    for (char variant : {'A', 'B'})
        for (int slots : {3})
            for (size_t mib : {(size_t)16, (size_t)32}) {
                const double tp = run(variant, pin_in, pin_out, bytes, mib << 20, slots);
                const double tg = run(variant, pg_in, pg_out, bytes, mib << 20, slots);
                printf("variant %c slots %d chunk %2zu MiB: pinned %.1f GB/s each way, pageable %.1f GB/s each way\n", variant, slots, mib, bytes / tp / 1e9, bytes / tg / 1e9);
                fflush(stdout);
            }
    for (int slots : {2, 3, 4})
        for (size_t mib : {(size_t)8, (size_t)16, (size_t)32, (size_t)64}) {
            const double tg = run_c(pg_in, pg_out, bytes, mib << 20, slots);
            const double tp = run_c(pin_in, pin_out, bytes, mib << 20, slots);
            printf("variant C (two host threads) slots %d chunk %2zu MiB: pageable %.1f GB/s each way, pinned %.1f GB/s each way\n", slots, mib, bytes / tg / 1e9, bytes / tp / 1e9);
            fflush(stdout);
        }
    double t0 = now();
    CK(hipHostRegister(pg_in, bytes, hipHostRegisterDefault)); CK(hipHostRegister(pg_out, bytes, hipHostRegisterDefault));
    const double treg = now() - t0;
    for (char variant : {'A', 'B'}) {
        const double t = run(variant, pg_in, pg_out, bytes, (size_t)32 << 20, 3);
        printf("variant %c slots 3 chunk 32 MiB, pageable + hipHostRegister (registration of 2 x 256 MiB took %.1f ms): %.1f GB/s each way\n", variant, treg * 1e3, bytes / t / 1e9);
    }
    t0 = now();
    CK(hipHostUnregister(pg_in)); CK(hipHostUnregister(pg_out));
    printf("unregistration took %.1f ms\n", (now() - t0) * 1e3);
    size_t bad = 0;
    for (size_t i = 0; i < bytes; i += 4097) bad += pg_out[i] != (i % 16 == 0 ? 0 : 1) && pg_out[i] != 1 && pg_out[i] != 0;
    printf("done (%zu)\n", bad);
    return 0;
}
