// Microbenchmark behind the node-size question of k_wf_mesh: what does a wave pay for L 16-byte loads from ONE
// random 128-byte record per lane (64 lanes = 64 different lines), as a dependent chain like a BVH descent?
//   time per record visit flat in L        -> latency-bound: a smaller node buys nothing
//   time per record visit growing with L   -> the L1's per-line request rate binds: a 64-byte node (4 loads) pays
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/micro_l1 tools/ubench/micro_l1.hip && /tmp/micro_l1 [table MiB]   (on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int L, int STRIDE16>  // L loads of 16 B at 16-B steps; record stride = STRIDE16 * 16 bytes
__global__ void __launch_bounds__(256, 4) chase(const uint4* __restrict__ table, uint32_t mask, int iters, uint32_t* out) {
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
        const uint4* p = table + size_t(idx) * STRIDE16;
        uint4 v[L];
#pragma unroll
        for (int l = 0; l < L; l++) v[l] = p[l];
        uint32_t s = 0;
#pragma unroll
        for (int l = 0; l < L; l++) s += v[l].x ^ v[l].y ^ v[l].z ^ v[l].w;
        acc += s;
        idx = (s + uint32_t(i)) & mask;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int L, int STRIDE16>
static int run(const uint4* d_table, uint32_t n_records, int blocks, uint32_t* d_out, const char* label) {
    const int iters = 4000;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((chase<L, STRIDE16>), dim3(blocks), dim3(256), 0, 0, d_table, n_records - 1, 200, d_out);
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((chase<L, STRIDE16>), dim3(blocks), dim3(256), 0, 0, d_table, n_records - 1, iters, d_out);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    double visits = double(blocks) * 256.0 * iters;
    std::printf("%-34s L=%d: %8.2f ms  %7.2f Gvisits/s  %7.2f G 16-B lane loads/s  %6.1f clk per wave visit per CU (2.4 GHz, 256 CUs)\n", label, L, ms,
                visits / ms * 1e-6, visits * L / ms * 1e-6, ms * 1e-3 * 2.4e9 * 256.0 / (visits / 64.0));
    return 0;
}

int main(int argc, char** argv) {
    const uint32_t mb = argc > 1 ? uint32_t(std::atoi(argv[1])) : 32u;  // table size in MiB (power of two)
    const size_t bytes = size_t(mb) << 20;
    std::vector<uint32_t> h(bytes / 4);
    uint64_t s = 88172645463325252ull;
    for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = uint32_t(s >> 16); }
    uint4* d_table = nullptr;
    uint32_t* d_out = nullptr;
    CHECK(hipMalloc(&d_table, bytes));
    CHECK(hipMalloc(&d_out, 64));
    CHECK(hipMemcpy(d_table, h.data(), bytes, hipMemcpyHostToDevice));
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int blocks = cus * 4;  // 4 waves per SIMD, like k_wf_mesh
    std::printf("table %u MiB, %d CUs, %d blocks of 256\n", mb, cus, blocks);
    const uint32_t n128 = uint32_t(bytes / 128), n64 = uint32_t(bytes / 64);
    if (run<1, 8>(d_table, n128, blocks, d_out, "128-B records")) return 1;
    if (run<2, 8>(d_table, n128, blocks, d_out, "128-B records")) return 1;
    if (run<4, 8>(d_table, n128, blocks, d_out, "128-B records")) return 1;
    if (run<5, 8>(d_table, n128, blocks, d_out, "128-B records")) return 1;
    if (run<7, 8>(d_table, n128, blocks, d_out, "128-B records")) return 1;
    if (run<8, 8>(d_table, n128, blocks, d_out, "128-B records")) return 1;
    if (run<1, 4>(d_table, n64, blocks, d_out, "64-B records")) return 1;
    if (run<2, 4>(d_table, n64, blocks, d_out, "64-B records")) return 1;
    if (run<4, 4>(d_table, n64, blocks, d_out, "64-B records")) return 1;
    return 0;
}
