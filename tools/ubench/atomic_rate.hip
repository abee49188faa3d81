// Microbenchmark behind the sample-reservation scheme of k_wf_shade (round 3): how many returning 64-bit atomic adds per
// microsecond does the chip sustain when every wave adds to ONE counter, and when the waves are spread over K counters on
// separate 128-byte lines?  (One counter: ~90 / us, MI355X_MICROARCH.md; a wave-level reservation per 64 path vertices needs
// ~250 / us.)  Also: the latency of one dependent atomic add per wave on an otherwise idle chip and under load.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_rate tools/ubench/atomic_rate.hip && /tmp/atomic_rate   (on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// Every wave: `iters` dependent atomic adds (the next one is issued when the previous value has come back), counter chosen
// by the wave's global index modulo K; `work` fused multiply-adds per lane between two atomics stand in for the shading.
__global__ void __launch_bounds__(256, 4) k_atomic(unsigned long long* ctr, uint32_t K, int iters, int work, double* sink) {
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long* mine = ctr + size_t(wave % K) * 16;  // 128 B apart
    unsigned long long acc = 0;
    double x = double(lane) * 1e-3 + 1.0;
    for (int i = 0; i < iters; i++) {
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(mine, 19ull);
        base = __shfl(base, 0);
        acc += base;
        for (int w = 0; w < work; w++) x = x * 1.0000001 + double(base & 1ull) * 1e-9;
    }
    if (acc == 0x12345ull || x == 123.456) sink[0] = x + double(acc);
}

static int run(unsigned long long* d_ctr, double* d_sink, int blocks, uint32_t K, int iters, int work) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipMemset(d_ctr, 0, 4096 * 128));
    hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, d_ctr, K, 50, work, d_sink);
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k_atomic, dim3(blocks), dim3(256), 0, 0, d_ctr, K, iters, work, d_sink);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double n = double(blocks) * 4.0 * iters;
    std::printf("blocks %5d  K %5u  work %5d fma: %9.3f ms  %9.1f atomics/us  %8.2f us per dependent atomic (+work) per wave\n", blocks, K, work, ms,
                n / (ms * 1e3), ms * 1e3 / iters);
    return 0;
}

int main() {
    unsigned long long* d_ctr = nullptr;
    double* d_sink = nullptr;
    CHECK(hipMalloc(&d_ctr, 4096 * 128));
    CHECK(hipMalloc(&d_sink, 64));
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    std::printf("%d CUs\n", cus);
    // latency: one wave, then one block per CU
    if (run(d_ctr, d_sink, 1, 1, 2000, 0)) return 1;
    const int full = cus * 4;
    const uint32_t ks[] = {1, 2, 4, 8, 16, 32, 64, 128, 256, 1024, 4096};
    for (uint32_t k : ks)
        if (run(d_ctr, d_sink, full, k, 400, 0)) return 1;
    // with ~16 us of arithmetic between two atomics of a wave (the shade kernel's rhythm): is the latency hidden?
    for (uint32_t k : {1u, 8u, 64u, 512u})
        if (run(d_ctr, d_sink, full, k, 100, 2000)) return 1;
    if (run(d_ctr, d_sink, full, 4096, 100, 2000)) return 1;
    return 0;
}
