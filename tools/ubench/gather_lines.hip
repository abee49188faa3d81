// Micro-benchmark: rate at which a CU fleet can fetch random 128-byte lines of a table, in the two access
// shapes a BVH traversal can use:
//   shape 0 "lane":  every lane fetches ITS OWN line with 7 x 16-B loads (what k_wf_mesh does for a BvhNode4f)
//   shape 1 "coop":  8 lanes fetch one line together, 16 B each (64 lanes = 8 lines per instruction)
// and with dependent (next index from the loaded data) or independent (hash of a counter) addresses.
// Prints G lines/s and TB/s for a few table sizes.  Build: hipcc --offload-arch=gfx950 -O3 -o gather_lines gather_lines.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

__device__ inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int SHAPE, bool DEP>
__global__ void __launch_bounds__(256) k_gather(const float4* __restrict__ table, uint32_t n_lines, uint32_t iters, float* out) {
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    if (SHAPE == 0) {
        uint32_t idx = mix32(gtid * 2654435761u + 12345u) % n_lines;
        for (uint32_t it = 0; it < iters; it++) {
            const float4* p = table + size_t(idx) * 8;
            float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4], f = p[5], g = p[6];
            float s = a.x + b.y + c.z + d.w + e.x + f.y + g.z;
            acc += s;
            uint32_t nxt = DEP ? (__float_as_uint(g.w) ^ (it * 0x9E3779B9u) ^ gtid) : (gtid * 2654435761u + it * 0x9E3779B9u);
            idx = mix32(nxt) % n_lines;
        }
    } else if (SHAPE == 2) {
        // every lane fetches its own 64-byte half line (4 x 16 B): a quantised 64-B node
        uint32_t idx = mix32(gtid * 2654435761u + 12345u) % (n_lines * 2);
        for (uint32_t it = 0; it < iters; it++) {
            const float4* p = table + size_t(idx) * 4;
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z + d.w;
            uint32_t nxt = DEP ? (__float_as_uint(d.x) ^ (it * 0x9E3779B9u) ^ gtid) : (gtid * 2654435761u + it * 0x9E3779B9u);
            idx = mix32(nxt) % (n_lines * 2);
        }
    } else if (SHAPE == 3) {
        // every lane fetches 80 contiguous bytes at a random 80-byte record (5 x 16 B): a TriRec<double>
        const uint32_t n_rec = uint32_t(size_t(n_lines) * 128 / 80);
        uint32_t idx = mix32(gtid * 2654435761u + 12345u) % n_rec;
        for (uint32_t it = 0; it < iters; it++) {
            const float4* p = table + size_t(idx) * 5;
            float4 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
            acc += a.x + b.y + c.z + d.w + e.x;
            uint32_t nxt = DEP ? (__float_as_uint(e.y) ^ (it * 0x9E3779B9u) ^ gtid) : (gtid * 2654435761u + it * 0x9E3779B9u);
            idx = mix32(nxt) % n_rec;
        }
    } else {
        // group of 8 lanes shares one line per instruction; 8 instructions cover the 8 lines of the group's 8 lanes
        const uint32_t part = threadIdx.x & 7u;
        const uint32_t grp = gtid >> 3;
        uint32_t seed = mix32(grp * 2654435761u + 777u);
        for (uint32_t it = 0; it < iters; it++) {
            float s = 0.f;
            float w = 0.f;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t idx = mix32(seed + uint32_t(k) * 0x85ebca6bu) % n_lines;
                float4 v = table[size_t(idx) * 8 + part];
                s += v.x + v.y;
                w = v.w;
            }
            acc += s;
            uint32_t nxt = DEP ? (__float_as_uint(w) ^ (it * 0x9E3779B9u) ^ grp) : (grp * 2654435761u + it * 0x9E3779B9u);
            // the group must agree on the next seed: take lane 0 of the group
            nxt = __shfl(int(nxt), int(threadIdx.x & 63u) & ~7);
            seed = mix32(nxt);
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int SHAPE, bool DEP>
double run(const float4* table, uint32_t n_lines, int blocks_per_cu, uint32_t iters, float* out) {
    int n_cu = 0;
    CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    dim3 grid(n_cu * blocks_per_cu), block(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_gather<SHAPE, DEP>), grid, block, 0, 0, table, n_lines, iters / 8, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather<SHAPE, DEP>), grid, block, 0, 0, table, n_lines, iters, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    double lines = double(grid.x) * 256.0 * iters * (SHAPE == 0 ? 1.0 : 1.0);  // shape 1: 8 lines per 8 lanes per iteration
    return lines / (ms * 1e-3);
}

int main(int argc, char** argv) {
    const bool calib = argc > 1;  // `gather_lines calib`: two tables, one kernel each (for a PMC pass with known byte counts)
    const size_t sizes_all[] = {2, 17, 87, 220, 1024};
    const size_t sizes_calib[] = {87, 1024};
    const size_t* sizes_mb = calib ? sizes_calib : sizes_all;
    const size_t n_sizes = calib ? 2 : 5;
    float* out; CHECK(hipMalloc(&out, 4));
    for (size_t si = 0; si < n_sizes; si++) {
        const size_t mb = sizes_mb[si];
        size_t bytes = mb << 20;
        uint32_t n_lines = uint32_t(bytes / 128);
        float4* table; CHECK(hipMalloc(&table, bytes));
        std::vector<uint32_t> h(bytes / 4);
        uint32_t s = 1u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s; }
        CHECK(hipMemcpy(table, h.data(), bytes, hipMemcpyHostToDevice));
        if (calib) {
            double b = run<0, true>(table, n_lines, 4, 800, out);
            std::printf("calib: table %zu MB, lane shape, dependent: warm-up launch 100 iterations then 800 iterations x %d threads x 128 B = %.3f GB: %.1f G lines/s\n", mb, 256 * 4 * 256, 256.0 * 4 * 256 * 800 * 128 / 1e9, b / 1e9);
            CHECK(hipFree(table));
            continue;
        }
        for (int bpc : {4, 8}) {
            double a = run<0, false>(table, n_lines, bpc, 2000, out);
            double b = run<0, true>(table, n_lines, bpc, 2000, out);
            double c = run<1, false>(table, n_lines, bpc, 2000, out);
            double d = run<1, true>(table, n_lines, bpc, 2000, out);
            double e = run<2, true>(table, n_lines, bpc, 2000, out);
            double f = run<3, true>(table, n_lines, bpc, 2000, out);
            std::printf("table %5zu MB, %d blocks/CU: 64-B pieces dep %.1f G/s (%.2f TB/s);  80-B records dep %.1f G/s (%.2f TB/s)\n", mb, bpc, e / 1e9, e * 64 / 1e12, f / 1e9, f * 80 / 1e12);
            std::printf("table %5zu MB, %d blocks/CU (%d waves/SIMD): lane-shape indep %.1f G lines/s (%.2f TB/s), dep %.1f (%.2f);  coop-shape indep %.1f (%.2f), dep %.1f (%.2f)\n",
                        mb, bpc, bpc, a / 1e9, a * 128 / 1e12, b / 1e9, b * 128 / 1e12, c / 1e9, c * 128 / 1e12, d / 1e9, d * 128 / 1e12);
            std::fflush(stdout);
        }
        CHECK(hipFree(table));
    }
    return 0;
}
