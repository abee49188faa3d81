// rt_kernels.hip — render kernels and the C ABI (include/rt_mi355.h) of librt_mi355.so.
// gfx950 only: 64-lane waves, per-lane traversal stack in LDS, HIP events on the launch stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "rt_compile.h"
#include "rt_device.h"
#include "rt_wavefront.h"

namespace rt {

// Packed owned row r -> image row y (RtRenderParams row partition).
template <typename R> RT_DEV uint32_t row_to_y(const ParamsView<R>& prm, uint32_t r) {
    if (prm.band_rows == 0 || prm.n_parts <= 1) return r;
    uint32_t band = r / prm.band_rows;
    return (band * prm.n_parts + prm.part) * prm.band_rows + (r % prm.band_rows);
}

// ---------------------------------------------------------------------------------------------
// Megakernel: one lane owns one pixel and walks its samples in the reference's order
// (replica tid, then sy, sx: camera.rs:197,217-218), one bounce per loop trip.  A lane whose path
// ended starts its next sample in the same trip, so the wave stays converged on
// world_test -> shade and no lane idles while it still has samples.  Sums are accumulated
// per pixel in the reference's order, so the result does not depend on scheduling.
// ---------------------------------------------------------------------------------------------
template <typename R, bool STATS, bool TEX>
__global__ void __launch_bounds__(256) k_megakernel(SceneView<R> sc, CameraView<R> cam, ParamsView<R> prm,
                                                    double* __restrict__ out, DeviceCounters* counters) {
    extern __shared__ int lds_stack[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tiles_x = (cam.width + 15u) / 16u;
    const uint32_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const uint32_t px = bx * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t row = by * 16u + (wave >> 1) * 8u + (lane >> 3);
    if (px >= cam.width || row >= prm.owned_rows) return;
    const uint32_t py = row_to_y(prm, row);
    int* stack = lds_stack + threadIdx.x;
    const int stride = int(blockDim.x);

    const uint32_t S = cam.sqrt_spt;
    const uint32_t per_replica = S * S;
    const uint32_t total = per_replica * cam.thread_count;
    const uint64_t pixel_index = uint64_t(py) * cam.width + px;

    LaneCounters cnt;
    double acc[3] = {0.0, 0.0, 0.0};  // buf += thread_buf, camera.rs:247-253
    double col[3] = {0.0, 0.0, 0.0};  // `color` of the current replica, camera.rs:215-229
    uint32_t sample = 0;              // next sample to start
    uint32_t in_replica = 0;
    bool alive = false;
    PathState<R> ps;
    Rng rng;
    rng.s = 0;

    for (;;) {
        if (!alive) {
            if (sample == total) break;
            uint32_t tid = sample / per_replica;
            uint32_t st = sample - tid * per_replica;
            uint32_t sy = st / S, sx = st - sy * S;
            rng.key(prm.seed, tid, pixel_index, st);
            ps.ray = get_ray(cam, px, py, sx, sy, rng);
            ps.throughput = mk<R>(1, 1, 1);
            ps.radiance = mk<R>(0, 0, 0);
            ps.depth = cam.max_depth;
            alive = true;
            sample++;
        }
        bool cont = false;
        if (ps.depth != 0) {
            Best<R> best;
            world_test<R, STATS, TEX>(sc, ps.ray, R(0.001), best, stack, stride, cnt, &rng);  // TEX variant = full feature set (+ volumes)
            cont = shade<R, STATS, TEX>(sc, prm, ps, best, rng, cnt);
            ps.depth--;
            if (cont && ps.depth == 0) {  // ray_color(depth == 0) returns black without tracing (camera.rs:290)
                end_black(ps);
                cont = false;
            }
        }
        if (!cont) {
            alive = false;
            col[0] += double(ps.radiance.x);
            col[1] += double(ps.radiance.y);
            col[2] += double(ps.radiance.z);
            if (++in_replica == per_replica) {
                in_replica = 0;
                for (int k = 0; k < 3; k++) {
                    acc[k] += col[k] / prm.spp;  // color /= samples_per_pixel (total), camera.rs:229
                    col[k] = 0.0;
                }
            }
        }
    }
    double* o = out + (size_t(row) * cam.width + px) * 4;
    o[0] = acc[0];
    o[1] = acc[1];
    o[2] = acc[2];
    o[3] = 0.0;
    if (STATS) {
        atomicAdd(&counters->rays, (unsigned long long)cnt.rays);
        atomicAdd(&counters->mesh_rays, (unsigned long long)cnt.mesh_rays);
        atomicAdd(&counters->node_visits, (unsigned long long)cnt.node_visits);
        atomicAdd(&counters->tri_tests, (unsigned long long)cnt.tri_tests);
        atomicAdd(&counters->prim_tests, (unsigned long long)cnt.prim_tests);
    }
}

// Diagnostic probe: one sample traced by one lane, with a per-bounce record
// (17 doubles: t, pos xyz, material, op type, triangle slot, 0, normal xyz, ray origin xyz, ray dir xyz).
// Used by the parity tests to localise differences.
template <typename R>
__global__ void k_trace_sample(SceneView<R> sc, CameraView<R> cam, ParamsView<R> prm, uint32_t tid, uint32_t px, uint32_t py,
                               uint32_t sx, uint32_t sy, double* rgb, double* trace, uint32_t max_bounces, uint32_t* n_out) {
    extern __shared__ int lds_stack[];
    if (threadIdx.x != 0) return;
    LaneCounters cnt;
    Rng rng;
    rng.key(prm.seed, tid, uint64_t(py) * cam.width + px, sy * cam.sqrt_spt + sx);
    PathState<R> ps;
    ps.ray = get_ray(cam, px, py, sx, sy, rng);
    ps.throughput = mk<R>(1, 1, 1);
    ps.radiance = mk<R>(0, 0, 0);
    ps.depth = cam.max_depth;
    uint32_t n = 0;
    while (ps.depth != 0) {
        Best<R> best;
        world_test<R, false, true>(sc, ps.ray, R(0.001), best, lds_stack, int(blockDim.x), cnt, &rng);
        if (n < max_bounces) {
            double* t = trace + 17 * n;
            for (int k = 0; k < 17; k++) t[k] = 0;
            t[0] = double(best.t);
            t[4] = -1; t[5] = -1; t[6] = double(best.tri);
            if (best.pc >= 0) {
                HitInfo<R> h = resolve_hit<R, true>(sc, ps.ray, best);
                t[1] = double(h.pos.x); t[2] = double(h.pos.y); t[3] = double(h.pos.z);
                t[4] = double(h.material);
                t[5] = double(sc.ops[best.pc].type);
                t[8] = double(h.normal.x); t[9] = double(h.normal.y); t[10] = double(h.normal.z);
            }
            t[11] = double(ps.ray.o.x); t[12] = double(ps.ray.o.y); t[13] = double(ps.ray.o.z);
            t[14] = double(ps.ray.d.x); t[15] = double(ps.ray.d.y); t[16] = double(ps.ray.d.z);
        }
        n++;
        bool cont = shade<R, false, true>(sc, prm, ps, best, rng, cnt);  // the general (interpreter) texture path
        ps.depth--;
        if (cont && ps.depth == 0) end_black(ps);
        if (!cont) break;
    }
    rgb[0] = double(ps.radiance.x); rgb[1] = double(ps.radiance.y); rgb[2] = double(ps.radiance.z);
    *n_out = n;
}

// ---------------------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int set_err(int st, const std::string& msg) {
    g_err = msg;
    return st;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return set_err(RT_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

struct DeviceBuffers {
    std::vector<void*> allocs;
    ~DeviceBuffers() {
        for (void* p : allocs) (void)hipFree(p);
    }
    template <typename T> int upload(const std::vector<T>& v, const T** out) {
        *out = nullptr;
        size_t bytes = (v.empty() ? 1 : v.size()) * sizeof(T);
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, bytes));
        allocs.push_back(p);
        if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        *out = static_cast<const T*>(p);
        return RT_OK;
    }
};

template <typename R> R round_down(double x);
template <typename R> R round_up(double x);
template <> double round_down<double>(double x) { return x; }
template <> double round_up<double>(double x) { return x; }
template <> float round_down<float>(double x) {
    float f = float(x);
    if (double(f) > x) f = nextafterf(f, -INFINITY);
    return f;
}
template <> float round_up<float>(double x) {
    float f = float(x);
    if (double(f) < x) f = nextafterf(f, INFINITY);
    return f;
}

template <typename R, size_t N> void cast_arr(R (&dst)[N], const double (&src)[N]) {
    for (size_t i = 0; i < N; i++) dst[i] = R(src[i]);
}

// Scene tables in arithmetic type R on the device.
template <typename R>
struct DeviceScene {
    DeviceBuffers buf;
    SceneView<R> view{};

    int build(const CompiledScene& cs) {
        std::vector<Bounds<R>> bounds(cs.bounds.size());
        for (size_t i = 0; i < bounds.size(); i++) {
            // list/bvh bounds are part of the reference's semantics (inverted boxes cull, B-8): nearest rounding
            cast_arr(bounds[i].lo, cs.bounds[i].lo);
            cast_arr(bounds[i].hi, cs.bounds[i].hi);
        }
        std::vector<Xform<R>> xforms(cs.xforms.size());
        for (size_t i = 0; i < xforms.size(); i++) {
            cast_arr(xforms[i].m, cs.xforms[i].m);
            cast_arr(xforms[i].inv, cs.xforms[i].inv);
        }
        std::vector<SpherePrim<R>> spheres(cs.spheres.size());
        for (size_t i = 0; i < spheres.size(); i++) {
            cast_arr(spheres[i].center, cs.spheres[i].center);
            spheres[i].radius = R(cs.spheres[i].radius);
            spheres[i].material = cs.spheres[i].material;
            spheres[i]._pad = 0;
        }
        std::vector<PlanePrim<R>> planes(cs.planes.size());
        for (size_t i = 0; i < planes.size(); i++) {
            const auto& s = cs.planes[i];
            cast_arr(planes[i].corner, s.corner); cast_arr(planes[i].normal, s.normal);
            cast_arr(planes[i].u, s.u); cast_arr(planes[i].v, s.v);
            cast_arr(planes[i].inv_u, s.inv_u); cast_arr(planes[i].inv_v, s.inv_v);
            planes[i].area = R(s.area);
            planes[i].material = s.material;
            planes[i].backface = s.backface;
        }
        std::vector<SunPrim<R>> suns(cs.suns.size());
        for (size_t i = 0; i < suns.size(); i++) {
            cast_arr(suns[i].direction, cs.suns[i].direction);
            suns[i].material = cs.suns[i].material;
            suns[i]._pad = 0;
        }
        std::vector<BvhNode<R>> nodes(cs.nodes.size());
        for (size_t i = 0; i < nodes.size(); i++) {
            const BuildNode& s = cs.nodes[i];
            BvhNode<R>& n = nodes[i];
            // Conservative boxes: outward rounding plus a few ulps so that the slab arithmetic
            // never culls a triangle the exact test would hit.
            auto pad = [](double lo, double hi, R* olo, R* ohi) {
                if (!(lo <= hi)) { *olo = R(lo); *ohi = R(hi); return; }  // empty child box
                double m = std::fmax(std::fabs(lo), std::fabs(hi));
                double e = 8.0 * double(std::numeric_limits<R>::epsilon()) * std::fmax(m, hi - lo);
                *olo = round_down<R>(lo - e);
                *ohi = round_up<R>(hi + e);
            };
            for (int a = 0; a < 3; a++) {
                pad(s.lo0[a], s.hi0[a], &n.lo0[a], &n.hi0[a]);
                pad(s.lo1[a], s.hi1[a], &n.lo1[a], &n.hi1[a]);
            }
            n.c0 = s.c0;
            n.c1 = s.c1;
        }
        // 4-wide f32 nodes: pad by 2^-19 x (largest |coordinate| of the mesh), round outward
        std::vector<BvhNode4f> nodes4(cs.nodes4.size());
        {
            size_t inst = 0;
            std::vector<std::pair<uint32_t, double>> pads;  // (node4_base, pad) per distinct mesh, ascending base
            for (const MeshInst& mi : cs.meshes) {
                const Bounds<double>& b = cs.mesh_bounds[inst++];
                double S = 0.0;
                for (int a = 0; a < 3; a++) S = std::fmax(S, std::fmax(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
                if (!std::isfinite(S)) S = 0.0;
                pads.emplace_back(mi.node4_base, S * (1.0 / 524288.0));
            }
            std::sort(pads.begin(), pads.end());
            size_t pi = 0;
            for (size_t i = 0; i < nodes4.size(); i++) {
                while (pi + 1 < pads.size() && pads[pi + 1].first <= i) pi++;
                const double m = pads.empty() ? 0.0 : pads[pi].second;
                const BuildNode4& sn = cs.nodes4[i];
                BvhNode4f& dn = nodes4[i];
                for (int k = 0; k < 4; k++) {
                    float* lo[3] = {&dn.lox[k], &dn.loy[k], &dn.loz[k]};
                    float* hi[3] = {&dn.hix[k], &dn.hiy[k], &dn.hiz[k]};
                    for (int a = 0; a < 3; a++) {
                        if (!(sn.lo[k][a] <= sn.hi[k][a])) { *lo[a] = INFINITY; *hi[a] = -INFINITY; }
                        else { *lo[a] = round_down<float>(sn.lo[k][a] - m); *hi[a] = round_up<float>(sn.hi[k][a] + m); }
                    }
                    dn.child[k] = sn.child[k];
                    dn._pad[k] = 0;
                }
            }
        }
        // Quantised nodes: the padded child boxes (as in BvhNode4f: 2^-19 S, rounded outward to f32) on a per-node 8-bit
        // grid, rounded outward on the grid.  k_wf_mesh evaluates t = fma(q, cell * iv, fma(org, iv, -o * iv)): cell is a
        // power of two (cell * iv exact), so against the f32-node test (fma(plane, iv, -o * iv)) there is one more rounding,
        // of a value bounded by 2 S |iv|; with it the error is < 2.5 x 2^-23 S |iv| per plane, inside the padding.
        std::vector<BvhNode4q> nodes4q(cs.nodes4.size());
        std::vector<BvhNode4q> group_nodes;
        {
            auto pad_of = [&]() {
                std::vector<std::pair<uint32_t, double>> pads;
                size_t inst = 0;
                for (const MeshInst& mi : cs.meshes) {
                    const Bounds<double>& b = cs.mesh_bounds[inst++];
                    double S = 0.0;
                    for (int a = 0; a < 3; a++) S = std::fmax(S, std::fmax(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
                    if (!std::isfinite(S)) S = 0.0;
                    pads.emplace_back(mi.node4_base, S * (1.0 / 524288.0));
                }
                std::sort(pads.begin(), pads.end());
                return pads;
            };
            // one node: W children, words = W / 4 grid words per axis
            auto quantise = [](int W, const double (*lo)[3], const double (*hi)[3], const int32_t* child, double m, float* org_out,
                               float* cell_out, uint32_t* qlo, uint32_t* qhi, int32_t* child_out) -> bool {
                const int words = W / 4;
                for (int a = 0; a < 3; a++) {
                    double flo[8], fhi[8];
                    bool real[8];
                    double lo_min = INFINITY, hi_max = -INFINITY;
                    for (int k = 0; k < W; k++) {
                        real[k] = child[k] != kEmptyChild && lo[k][a] <= hi[k][a];
                        if (!real[k]) continue;
                        flo[k] = double(round_down<float>(lo[k][a] - m));
                        fhi[k] = double(round_up<float>(hi[k][a] + m));
                        lo_min = std::fmin(lo_min, flo[k]);
                        hi_max = std::fmax(hi_max, fhi[k]);
                    }
                    const bool any = lo_min <= hi_max && std::isfinite(lo_min) && std::isfinite(hi_max);
                    const double org = any ? lo_min : 0.0;  // an f32 value
                    const double ext = any ? hi_max - org : 0.0;
                    int e = -100;
                    if (ext > 0.0) {
                        e = int(std::ceil(std::log2(ext / 255.0)));
                        while (ext / std::ldexp(1.0, e) > 255.0) e++;
                        while (e > -100 && ext / std::ldexp(1.0, e - 1) <= 255.0) e--;
                        e = std::max(-100, std::min(e, 120));
                    }
                    const double cell = std::ldexp(1.0, e);
                    org_out[a] = float(org);
                    cell_out[a] = float(cell);
                    for (int w = 0; w < words; w++) { qlo[a * words + w] = 0; qhi[a * words + w] = 0; }
                    for (int k = 0; k < W; k++) {
                        uint32_t ql = 255u, qh = 0u;  // empty child: lo > hi, never entered
                        if (any && real[k]) {
                            double l = std::floor((flo[k] - org) / cell), h = std::ceil((fhi[k] - org) / cell);
                            ql = uint32_t(std::max(0.0, std::min(l, 255.0)));
                            qh = uint32_t(std::max(0.0, std::min(h, 255.0)));
                            if (!(org + ql * cell <= flo[k] && org + qh * cell >= fhi[k])) return false;  // e > 120: coordinates beyond 1e38
                        }
                        qlo[a * words + (k >> 2)] |= ql << (8 * (k & 3));
                        qhi[a * words + (k >> 2)] |= qh << (8 * (k & 3));
                    }
                }
                for (int k = 0; k < W; k++) child_out[k] = child[k];
                return true;
            };
            const auto pads4 = pad_of();
            size_t pi = 0;
            for (size_t i = 0; i < nodes4q.size(); i++) {
                while (pi + 1 < pads4.size() && pads4[pi + 1].first <= i) pi++;
                const BuildNode4& sn = cs.nodes4[i];
                BvhNode4q& qn = nodes4q[i];
                if (!quantise(4, sn.lo, sn.hi, sn.child, pads4.empty() ? 0.0 : pads4[pi].second, qn.org, qn.cell, qn.qlo, qn.qhi, qn.child))
                    return set_err(RT_E_UNSUPPORTED, "BVH node does not fit the 8-bit grid");
            }
            // the primitive groups' BVHs: same node format, same padding rule (2^-19 x the largest |coordinate| of the group's box)
            group_nodes.resize(cs.group_nodes4.size());
            for (size_t g = 0; g < cs.groups.size(); g++) {
                const GroupRec<double>& gr = cs.groups[g];
                double S = 0.0;
                for (int a = 0; a < 3; a++) S = std::fmax(S, std::fmax(std::fabs(gr.lo[a]), std::fabs(gr.hi[a])));
                if (!std::isfinite(S)) return set_err(RT_E_UNSUPPORTED, "primitive group with an unbounded box");
                const size_t end = g + 1 < cs.groups.size() ? cs.groups[g + 1].root : cs.group_nodes4.size();
                for (size_t i = gr.root; i < end; i++) {
                    const BuildNode4& sn = cs.group_nodes4[i];
                    BvhNode4q& qn = group_nodes[i];
                    if (!quantise(4, sn.lo, sn.hi, sn.child, S * (1.0 / 524288.0), qn.org, qn.cell, qn.qlo, qn.qhi, qn.child))
                        return set_err(RT_E_UNSUPPORTED, "group BVH node does not fit the 8-bit grid");
                }
            }
        }
        std::vector<Bounds<R>> mesh_bounds(cs.mesh_bounds.size());
        for (size_t i = 0; i < mesh_bounds.size(); i++)
            for (int a = 0; a < 3; a++) {  // outward: this box only decides which rays are queued for the mesh
                mesh_bounds[i].lo[a] = round_down<R>(cs.mesh_bounds[i].lo[a]);
                mesh_bounds[i].hi[a] = round_up<R>(cs.mesh_bounds[i].hi[a]);
            }
        std::vector<TriRec<R>> tris(cs.tris.size());
        for (size_t i = 0; i < tris.size(); i++) {
            cast_arr(tris[i].v0, cs.tris[i].v0); cast_arr(tris[i].e1, cs.tris[i].e1); cast_arr(tris[i].e2, cs.tris[i].e2);
            tris[i]._pad = R(0);
        }
        std::vector<TriAttr<R>> attrs(cs.attrs.size());
        for (size_t i = 0; i < attrs.size(); i++) {
            const auto& s = cs.attrs[i];
            cast_arr(attrs[i].n0, s.n0); cast_arr(attrs[i].n1, s.n1); cast_arr(attrs[i].n2, s.n2);
            cast_arr(attrs[i].uv0, s.uv0); cast_arr(attrs[i].uv1, s.uv1); cast_arr(attrs[i].uv2, s.uv2);
            attrs[i].has_uv = s.has_uv;
            attrs[i]._pad = 0;
        }
        std::vector<MaterialParams<R>> mparams(cs.material_params.size());
        for (size_t i = 0; i < mparams.size(); i++) {
            mparams[i].ior = R(cs.material_params[i].ior);
            mparams[i].inv_ior = R(cs.material_params[i].inv_ior);
            auto r0_of = [](R x) { R r0 = (R(1) - x) / (R(1) + x); return r0 * r0; };  // utils.rs:32-33, in R like reflectance()
            mparams[i].inv_ior_r = R(1) / mparams[i].ior;
            mparams[i].r0_glossy = r0_of(mparams[i].inv_ior);
            mparams[i].r0_front = r0_of(mparams[i].inv_ior_r);
            mparams[i].r0_back = r0_of(mparams[i].ior);
        }
        std::vector<TextureRec<R>> textures(cs.textures.size());
        for (size_t i = 0; i < textures.size(); i++) {
            const auto& s = cs.textures[i];
            textures[i].type = s.type; textures[i].aux = s.aux; textures[i].data = s.data;
            textures[i].width = s.width; textures[i].height = s.height; textures[i]._pad = 0;
            cast_arr(textures[i].v, s.v);
            textures[i].scale = R(s.scale);
        }
        std::vector<VolumeRec<R>> volumes(cs.volumes.size());
        for (size_t i = 0; i < volumes.size(); i++) {
            volumes[i].neg_inv_density = R(cs.volumes[i].neg_inv_density);
            volumes[i].material = cs.volumes[i].material;
            volumes[i]._pad = 0;
        }
        std::vector<R> perlin_vec(cs.perlin_vec.size());
        for (size_t i = 0; i < perlin_vec.size(); i++) perlin_vec[i] = R(cs.perlin_vec[i]);
        int st;
        if ((st = buf.upload(cs.ops, &view.ops)) != RT_OK) return st;
        if ((st = buf.upload(bounds, &view.bounds)) != RT_OK) return st;
        if ((st = buf.upload(cs.chain_offsets, &view.chain_offsets)) != RT_OK) return st;
        if ((st = buf.upload(cs.chain_items, &view.chain_items)) != RT_OK) return st;
        if ((st = buf.upload(xforms, &view.xforms)) != RT_OK) return st;
        if ((st = buf.upload(spheres, &view.spheres)) != RT_OK) return st;
        if ((st = buf.upload(planes, &view.planes)) != RT_OK) return st;
        if ((st = buf.upload(suns, &view.suns)) != RT_OK) return st;
        if ((st = buf.upload(cs.meshes, &view.meshes)) != RT_OK) return st;
        if ((st = buf.upload(volumes, &view.volumes)) != RT_OK) return st;
        if ((st = buf.upload(nodes, &view.nodes)) != RT_OK) return st;
        if ((st = buf.upload(nodes4, &view.nodes4)) != RT_OK) return st;
        if ((st = buf.upload(nodes4q, &view.nodes4q)) != RT_OK) return st;
        if ((st = buf.upload(mesh_bounds, &view.mesh_bounds)) != RT_OK) return st;
        if ((st = buf.upload(cs.mesh_ops, &view.mesh_ops)) != RT_OK) return st;
        view.n_mesh_ops = int32_t(cs.mesh_ops.size());
        std::vector<MeshOpRec<R>> mesh_op_recs(cs.mesh_ops.size());
        for (size_t m = 0; m < mesh_op_recs.size(); m++) {
            const Op& op = cs.ops[size_t(cs.mesh_ops[m])];
            const MeshInst& mi = cs.meshes[size_t(op.arg)];
            MeshOpRec<R>& r = mesh_op_recs[m];
            r.pc = cs.mesh_ops[m];
            r.chain = op.chain;
            r.node4_base = mi.node4_base;
            const int32_t cb = cs.chain_offsets[size_t(op.chain)], ce = cs.chain_offsets[size_t(op.chain) + 1];
            r.flags = (mi.flags & 0xFFFFu) | (uint32_t(std::min(ce - cb, 0xFFFF)) << 16);
            for (int a = 0; a < 3; a++) { r.lo[a] = mesh_bounds[size_t(op.arg)].lo[a]; r.hi[a] = mesh_bounds[size_t(op.arg)].hi[a]; }
            for (int k = 0; k < 12; k++) r.inv[k] = ce - cb == 1 ? xforms[size_t(cs.chain_items[size_t(cb)])].inv[k] : R(0);
        }
        if ((st = buf.upload(mesh_op_recs, &view.mesh_op_recs)) != RT_OK) return st;
        std::vector<GroupRec<R>> groups(cs.groups.size());
        for (size_t g = 0; g < groups.size(); g++) {
            groups[g].root = cs.groups[g].root;
            for (int a = 0; a < 3; a++) { groups[g].lo[a] = round_down<R>(cs.groups[g].lo[a]); groups[g].hi[a] = round_up<R>(cs.groups[g].hi[a]); }
        }
        if ((st = buf.upload(groups, &view.groups)) != RT_OK) return st;
        if ((st = buf.upload(group_nodes, &view.group_nodes)) != RT_OK) return st;
        if ((st = buf.upload(cs.group_prims, &view.group_prims)) != RT_OK) return st;
        if ((st = buf.upload(cs.group_guards, &view.group_guards)) != RT_OK) return st;
        view.n_group_nodes = int32_t(group_nodes.size());
        view.group_stack_levels = int32_t(cs.max_group_stack);
        if ((st = buf.upload(tris, &view.tris)) != RT_OK) return st;
        if ((st = buf.upload(attrs, &view.attrs)) != RT_OK) return st;
        if ((st = buf.upload(cs.materials, &view.materials)) != RT_OK) return st;
        if ((st = buf.upload(mparams, &view.material_params)) != RT_OK) return st;
        if ((st = buf.upload(textures, &view.textures)) != RT_OK) return st;
        if ((st = buf.upload(cs.texels, &view.texels)) != RT_OK) return st;
        if ((st = buf.upload(perlin_vec, &view.perlin_vec)) != RT_OK) return st;
        if ((st = buf.upload(cs.perlin_perm, &view.perlin_perm)) != RT_OK) return st;
        if ((st = buf.upload(cs.lights, &view.lights)) != RT_OK) return st;
        // the same small tables once more, packed for LDS staging: once in k_wf_prims' order, once in k_wf_shade's
        {
            struct Tbl { const void* data; size_t bytes; };
            const Tbl tbl[ST_COUNT] = {
                {cs.ops.data(), cs.ops.size() * sizeof(Op)}, {bounds.data(), bounds.size() * sizeof(Bounds<R>)},
                {cs.chain_offsets.data(), cs.chain_offsets.size() * 4}, {cs.chain_items.data(), cs.chain_items.size() * 4},
                {xforms.data(), xforms.size() * sizeof(Xform<R>)}, {spheres.data(), spheres.size() * sizeof(SpherePrim<R>)},
                {planes.data(), planes.size() * sizeof(PlanePrim<R>)}, {suns.data(), suns.size() * sizeof(SunPrim<R>)},
                {cs.meshes.data(), cs.meshes.size() * sizeof(MeshInst)}, {cs.materials.data(), cs.materials.size() * sizeof(MaterialRec)},
                {mparams.data(), mparams.size() * sizeof(MaterialParams<R>)}, {textures.data(), textures.size() * sizeof(TextureRec<R>)},
                {cs.lights.data(), cs.lights.size() * sizeof(LightRec)}};
            auto pack = [&](const int (&order)[ST_COUNT], SmallLayout& L, const char** out) -> int {
                std::vector<char> blob;
                for (int k : order) {
                    const size_t off = (blob.size() + 15) & ~size_t(15);
                    blob.resize(off + tbl[k].bytes);
                    if (tbl[k].bytes) std::memcpy(blob.data() + off, tbl[k].data, tbl[k].bytes);
                    L.begin[k] = uint32_t(off);
                    L.end[k] = uint32_t(off + tbl[k].bytes);
                }
                blob.resize((blob.size() + 15) & ~size_t(15));
                L.total_bytes = uint32_t(blob.size());
                return buf.upload(blob, out);
            };
            const int order_prims[ST_COUNT] = {ST_OPS, ST_CHAIN_OFFSETS, ST_CHAIN_ITEMS, ST_XFORMS, ST_MESHES, ST_SUNS, ST_PLANES, ST_BOUNDS, ST_SPHERES,
                                               ST_MATERIALS, ST_MATERIAL_PARAMS, ST_LIGHTS, ST_TEXTURES};
            const int order_shade[ST_COUNT] = {ST_OPS, ST_CHAIN_OFFSETS, ST_CHAIN_ITEMS, ST_XFORMS, ST_MESHES, ST_SUNS, ST_LIGHTS, ST_PLANES, ST_MATERIALS,
                                               ST_MATERIAL_PARAMS, ST_TEXTURES, ST_SPHERES, ST_BOUNDS};
            if ((st = pack(order_prims, view.lay, &view.small_blob)) != RT_OK) return st;
            if ((st = pack(order_shade, view.lay_shade, &view.small_blob_shade)) != RT_OK) return st;
        }
        view.n_lights = cs.n_top_lights;
        view.inv_n_lights = R(1) / R(cs.n_top_lights);
        view.lights_is_list = cs.lights_is_list;
        view.stop_on_zero_weight = cs.zero_weight_stop ? 1 : 0;
        if (const char* e = std::getenv("RT_ZERO_WEIGHT_STOP")) view.stop_on_zero_weight = std::atoi(e) != 0;  // experiments
        view.stack_entries = int32_t(std::max(cs.max_bvh_depth + 2, cs.max_bvh4_stack + 1));
        view.n_ops = int32_t(cs.ops.size());
        // The uploads above went through the null stream (small pageable copies may return once
        // staged); the render kernels run on a NON-BLOCKING stream that is not ordered against it.
        HIP_TRY(hipDeviceSynchronize());
        return RT_OK;
    }
};

}  // namespace rt

struct RtScene {
    int device = 0;
    rt::CompiledScene compiled;
    // Both precisions are materialised lazily on first use.
    std::unique_ptr<rt::DeviceScene<double>> f64;
    std::unique_ptr<rt::DeviceScene<float>> f32;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    rt::DeviceCounters* d_counters = nullptr;
    RtRenderStats stats{};
    int32_t* tail_flag = nullptr;  // rt_scene_set_tail_flag: set to 1 when a render stops filling the GPU (frame pipelining)
    // wavefront pipeline resources (allocated on first use, reused between renders)
    struct Wavefront {
        uint32_t capacity = 0;
        size_t real_size = 0;          // sizeof(R) the pool was allocated for
        std::vector<void*> allocs;
        void* pool_view = nullptr;     // host copy of WfPool<R>
        void* pool_dev = nullptr;      // the same descriptor in device memory (k_wf_shade re-reads the array bases from it)
        void* pool2_view = nullptr;    // the second pool, 3/4 of the slots: destination of the first tail compaction (k_wf_compact),
        void* pool2_dev = nullptr;     // after which the two take turns
        uint32_t* queue[2] = {nullptr, nullptr};
        uint32_t* mesh_queue = nullptr;
        void* mesh_spill = nullptr;        // k_wf_mesh: stack levels beyond the LDS part
        size_t mesh_spill_bytes = 0;
        rt::WfCounters* d_ctr = nullptr;
        rt::WfCounters* h_ctr = nullptr;   // pinned
        double* sample_L = nullptr;
        size_t sample_L_bytes = 0;
        double* acc = nullptr;
        size_t acc_bytes = 0;
        std::vector<hipEvent_t> events;
    } wf;
};

namespace rt {

template <typename R> CameraView<R> make_camera_view(const RtCameraDesc& c, const RtRenderParams& p) {
    CameraView<R> v{};
    for (int i = 0; i < 3; i++) {
        v.position[i] = R(c.position[i]);
        v.first_pixel[i] = R(c.first_pixel[i]);
        v.pdu[i] = R(c.pixel_delta_u[i]);
        v.pdv[i] = R(c.pixel_delta_v[i]);
        v.basis_u[i] = R(c.basis_u[i]);
        v.basis_v[i] = R(c.basis_v[i]);
    }
    v.aperture_radius = R(c.aperture_radius);
    v.inv_sqrt_spt = R(1.0 / double(p.sqrt_spt));  // camera.rs:52
    v.has_aperture = int32_t(c.has_aperture);
    v.width = c.image_width;
    v.height = c.image_height;
    v.sqrt_spt = p.sqrt_spt;
    v.thread_count = p.thread_count;
    v.max_depth = p.max_depth;
    return v;
}

template <typename R> ParamsView<R> make_params_view(const RtRenderParams& p, uint32_t owned) {
    ParamsView<R> v{};
    v.light_bias = R(p.light_bias);
    for (int i = 0; i < 3; i++) v.background[i] = p.has_background ? R(p.background[i]) : R(0);
    v.seed = p.seed;
    v.band_rows = p.band_rows;
    v.n_parts = p.n_parts;
    v.part = p.part;
    v.owned_rows = owned;
    v.spp = double(p.sqrt_spt) * double(p.sqrt_spt) * double(p.thread_count);  // camera.rs:50-51
    v.inv_spp = 1.0 / v.spp;
    return v;
}

template <typename R>
int render_typed(RtScene* s, DeviceScene<R>& ds, const RtCameraDesc& cam, const RtRenderParams& p, uint32_t owned,
                 double* d_out, hipStream_t stream) {
    CameraView<R> cv = make_camera_view<R>(cam, p);
    ParamsView<R> pv = make_params_view<R>(p, owned);
    const uint32_t tiles_x = (cam.image_width + 15u) / 16u, tiles_y = (owned + 15u) / 16u;
    const dim3 grid(tiles_x * tiles_y), block(256);
    const size_t lds = size_t(ds.view.stack_entries) * 256 * sizeof(int);
    if (lds > 160 * 1024) return set_err(RT_E_UNSUPPORTED, "mesh BVH too deep for the LDS traversal stack");
    HIP_TRY(hipMemsetAsync(s->d_counters, 0, sizeof(DeviceCounters), stream));
    HIP_TRY(hipEventRecord(s->ev0, stream));
    // full-feature variant: texture interpreter (lerp / image / noise / channel / normal maps) and volumes
    const bool tex = s->compiled.needs_tex_interpreter || !s->compiled.volumes.empty();
#define RT_LAUNCH_MEGA(ST, TX) hipLaunchKernelGGL((k_megakernel<R, ST, TX>), grid, block, lds, stream, ds.view, cv, pv, d_out, s->d_counters)
    if (p.collect_stats) { if (tex) RT_LAUNCH_MEGA(true, true); else RT_LAUNCH_MEGA(true, false); }
    else { if (tex) RT_LAUNCH_MEGA(false, true); else RT_LAUNCH_MEGA(false, false); }
#undef RT_LAUNCH_MEGA
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev1, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    DeviceCounters hc{};
    HIP_TRY(hipMemcpy(&hc, s->d_counters, sizeof hc, hipMemcpyDeviceToHost));
    RtRenderStats& st = s->stats;
    st = RtRenderStats{};
    st.kernel_ms = ms;
    st.traversal_kernel_ms = ms;
    st.n_launches = 1;
    st.pipeline_used = RT_PIPELINE_MEGAKERNEL;
    st.samples = uint64_t(cam.image_width) * owned * uint64_t(pv.spp);
    st.rays = hc.rays;
    st.mesh_rays = hc.mesh_rays;
    st.node_visits = hc.node_visits;
    st.tri_tests = hc.tri_tests;
    st.prim_tests = hc.prim_tests;
    st.bytes_node = sizeof(BvhNode<R>);
    st.bytes_tri = sizeof(TriRec<R>);
    st.bytes_attr = sizeof(TriAttr<R>);
    st.bytes_state = 0;
    return RT_OK;
}

// ---------------------------------------------------------------------------------------------
// Wavefront pipeline driver
// ---------------------------------------------------------------------------------------------
static uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    return uint32_t(std::strtoul(v, nullptr, 10));
}

// Releases the path pool and its queues and marks the pool as absent, so that a failed re-allocation can never be
// mistaken for a valid pool by a later render (and nothing is freed twice by rt_scene_destroy).
static void wf_release_pool(RtScene::Wavefront& w) {
    w.capacity = 0;
    w.real_size = 0;
    for (void* p : w.allocs) (void)hipFree(p);
    w.allocs.clear();
    ::operator delete(w.pool_view);
    w.pool_view = nullptr;
    ::operator delete(w.pool2_view);
    w.pool2_view = nullptr;
    if (w.pool_dev) (void)hipFree(w.pool_dev);
    w.pool_dev = nullptr;
    if (w.pool2_dev) (void)hipFree(w.pool2_dev);
    w.pool2_dev = nullptr;
    for (int q = 0; q < 2; q++) {
        if (w.queue[q]) (void)hipFree(w.queue[q]);
        w.queue[q] = nullptr;
    }
    if (w.mesh_queue) (void)hipFree(w.mesh_queue);
    w.mesh_queue = nullptr;
}

template <typename R>
int wf_ensure(RtScene* s, uint32_t capacity) {
    RtScene::Wavefront& w = s->wf;
    if (w.capacity == capacity && w.real_size == sizeof(R)) return RT_OK;
    if (const uint32_t limit = env_u32("RT_WF_FAKE_OOM_ABOVE", 0); limit && capacity > limit)  // tests: the out-of-memory path of render_wavefront
        return set_err(RT_E_NOMEM, "path pool does not fit in device memory (RT_WF_FAKE_OOM_ABOVE)");
    wf_release_pool(w);
    auto* pool = new WfPool<R>();
    w.pool_view = pool;
    pool->capacity = capacity;
    auto* pool2 = new WfPool<R>();
    w.pool2_view = pool2;
    pool2->capacity = std::max<uint32_t>(64u, uint32_t((uint64_t(capacity) * 3 + 3) / 4));  // a compaction happens below RT_WF_COMPACT_PCT <= 75 % of the addressed slots
    auto alloc = [&](size_t bytes, void** out) -> int {
        const hipError_t e = hipMalloc(out, bytes);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();  // not sticky: the caller retries with a smaller pool
            return set_err(RT_E_NOMEM, "path pool does not fit in device memory");
        }
        HIP_TRY(e);
        w.allocs.push_back(*out);
        return RT_OK;
    };
    auto build_pool = [&](WfPool<R>* pl) -> int {
        const size_t cap = pl->capacity;
        R** reals[] = {&pl->ox, &pl->oy, &pl->oz, &pl->dx, &pl->dy, &pl->dz, &pl->tr, &pl->tg, &pl->tb, &pl->ht, &pl->hu, &pl->hv};
        for (R** r : reals)
            if (int st = alloc(cap * sizeof(R), reinterpret_cast<void**>(r))) return st;
        if (int st = alloc(cap * 8, reinterpret_cast<void**>(&pl->rng))) return st;
        if (int st = alloc(cap * 8, reinterpret_cast<void**>(&pl->sample))) return st;
        if (int st = alloc(cap * 4, reinterpret_cast<void**>(&pl->depth))) return st;
        if (int st = alloc(cap * 4, reinterpret_cast<void**>(&pl->hpc))) return st;
        if (int st = alloc(cap * 4, reinterpret_cast<void**>(&pl->htri))) return st;
        return RT_OK;
    };
    auto build = [&]() -> int {
        if (int st = build_pool(pool)) return st;
        if (int st = build_pool(pool2)) return st;
        HIP_TRY(hipMalloc(&w.pool2_dev, sizeof(WfPool<R>)));
        HIP_TRY(hipMemcpy(w.pool2_dev, pool2, sizeof(WfPool<R>), hipMemcpyHostToDevice));
        for (int q = 0; q < 2; q++) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.queue[q]), size_t(capacity) * 4));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.mesh_queue), size_t(capacity) * 4));
        HIP_TRY(hipMalloc(&w.pool_dev, sizeof(WfPool<R>)));
        HIP_TRY(hipMemcpy(w.pool_dev, pool, sizeof(WfPool<R>), hipMemcpyHostToDevice));
        if (!w.d_ctr) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.d_ctr), sizeof(WfCounters)));
        if (!w.h_ctr) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&w.h_ctr), sizeof(WfCounters)));
        if (w.events.empty()) {
            w.events.resize(136);  // 4 per iteration, up to 32 iterations between host checks, + 2 for the stand-alone prims launch
            for (auto& e : w.events) e = nullptr;
            for (auto& e : w.events) HIP_TRY(hipEventCreate(&e));
        }
        return RT_OK;
    };
    if (int st = build()) {
        wf_release_pool(w);  // e.g. out of memory half way: leave no half-built pool behind
        return st;
    }
    w.capacity = capacity;
    w.real_size = sizeof(R);
    return RT_OK;
}

template <typename R>
int render_wavefront(RtScene* s, DeviceScene<R>& ds, const RtCameraDesc& cam, const RtRenderParams& p, uint32_t owned,
                     double* d_out, hipStream_t stream) {
    CameraView<R> cv = make_camera_view<R>(cam, p);
    ParamsView<R> pv = make_params_view<R>(p, owned);
    const uint64_t npix = uint64_t(cam.image_width) * owned;
    const uint32_t strata = p.sqrt_spt * p.sqrt_spt;
    const uint32_t T = p.thread_count;
    const uint64_t per_replica = uint64_t(strata) * npix;
    // Pool size.  Every launch of the persistent mesh kernel ends with a drain of ~0.4 ms (the longest remaining traversals:
    // dependent fetches) and the streaming kernels run better in few large launches, so fewer, larger launches win; against that
    // stands the tail: the pool is what drains at the end of a replica group, over ~20 ever smaller iterations.  Round 2 (tail at
    // 2.5 x its work's worth): best size 64 M slots at 1.44 G samples, growing with the square root of the work.  Round 3's tail
    // compaction (k_wf_compact) halved the tail's price and moved the optimum up - Msamples/s, same box (profiles/r03/tail_compaction.txt):
    //   C4 1.44 G samples:  64 M 1256-1287, 96 M 1289-1293, 128 M 1312-1332, 160 M 1319-1338, 192 M 1326-1337, 256 M 1300-1311
    //   C3 0.96 G: 52 M 4841, 80 M 4907, 96 M 4977, 112 M 4890      C1 0.25 G: 26 M 1962, 48 M 2028, 64 M 2061-2079, 96 M 2120, 128 M 2129
    //   C2 0.16 G (no mesh): 16 M 1681, 22 M 1692, 32 M 1668, 44 M 1670      one of 8 ranks' share of C4: 23 M 157 ms, 46 M / 92 M 151, 128 M 153
    // -> 128 M slots at 1.44 G samples, with the square root of the work below it (within 1-3 % of each workload's best).
    uint32_t capacity;
    {
        const double total_samples = double(per_replica) * double(T);
        double c = 134217728.0 * std::sqrt(total_samples / 1.44e9);
        c = std::fmin(std::fmax(c, 1048576.0), 134217728.0);
        capacity = env_u32("RT_WF_POOL", uint32_t(c) & ~0xFFFFFu);
    }
    if (capacity > (1u << 28)) capacity = 1u << 28;  // the kernels address pool arrays through 32-bit byte offsets (rt_wavefront.h, at())
    if (uint64_t(capacity) > per_replica * T) capacity = uint32_t(per_replica * T);
    if (capacity < 64) capacity = 64;
    // the pool is a matter of speed, not of correctness: when device memory is short (other scenes of a frame pipeline, other
    // processes on the card) a smaller one renders the same frame
    for (;;) {
        const int st = wf_ensure<R>(s, capacity);
        if (st == RT_OK) break;
        if (st != RT_E_NOMEM || capacity <= (1u << 20) || std::getenv("RT_WF_POOL")) return st;
        capacity = std::max<uint32_t>(1u << 20, (capacity / 2) & ~0xFFFFFu);
    }
    RtScene::Wavefront& w = s->wf;
    WfPool<R> pool = *static_cast<WfPool<R>*>(w.pool_view);   // the pool the kernels are working on (changes at a tail compaction)
    const WfPool<R> pool_a = pool, pool_b = *static_cast<WfPool<R>*>(w.pool2_view);
    const void* pool_dev_cur = w.pool_dev;

    // per-sample radiance buffer: as many replicas per group as the memory budget allows
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    size_t budget = size_t(env_u32("RT_WF_SAMPLE_GB", 64)) << 30;
    size_t avail = free_b + w.sample_L_bytes;
    if (budget > avail / 2) budget = avail / 2;
    uint64_t bytes_per_replica = per_replica * 24ull;
    uint32_t group = uint32_t(std::min<uint64_t>(T, std::max<uint64_t>(1, budget / bytes_per_replica)));
    group = (T + (T + group - 1) / group - 1) / ((T + group - 1) / group);  // same number of groups, equal sizes (9 + 1 -> 5 + 5)
    if (bytes_per_replica > avail) return set_err(RT_E_NOMEM, "per-sample radiance buffer of one replica does not fit in device memory");
    size_t need = size_t(bytes_per_replica) * group;
    if (w.sample_L_bytes < need) {
        if (w.sample_L) (void)hipFree(w.sample_L);
        w.sample_L = nullptr;
        w.sample_L_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.sample_L), need));
        w.sample_L_bytes = need;
    }
    const bool multi_group = group < T;
    if (multi_group && w.acc_bytes < npix * 24) {
        if (w.acc) (void)hipFree(w.acc);
        w.acc = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&w.acc), npix * 24));
        w.acc_bytes = npix * 24;
    }

    const size_t lds = size_t(ds.view.stack_entries) * 256 * sizeof(int);
    if (lds > 160 * 1024) return set_err(RT_E_UNSUPPORTED, "mesh BVH too deep for the LDS traversal stack");
    int n_cu = 0, blocks_per_cu = 0;
    HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, s->device));
    const bool stats = p.collect_stats != 0;
    const bool vol = !s->compiled.volumes.empty();  // volume ops: combined intersect kernel, VOL variant
    if (vol) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, k_wf_intersect<R, false, true>, 256, lds));
    else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, k_wf_intersect<R, false, false>, 256, lds));
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    // which kernels serve this scene (rt_compile.cpp plan_wavefront); RT_WF_SPLIT=0 (tests): the combined kernel for every scene
    const WavefrontPlan plan = plan_wavefront(s->compiled);
    const int n_mesh_ops = int(s->compiled.mesh_ops.size());
    const bool use_split = env_u32("RT_WF_SPLIT", 1) != 0 && plan.split;
    const bool prims_only = use_split && n_mesh_ops == 0;
    const bool split = use_split && n_mesh_ops > 0;
    const bool multi_mesh = plan.multi_mesh || env_u32("RT_WF_MESH_MULTI", 0) != 0;  // the general form of k_wf_mesh (env: A/B on single-mesh scenes, tests)
    // k_wf_mesh keeps (child, entry distance) pairs: a shallow LDS part (occupancy) + a global spill part
    // BVH node format of k_wf_mesh: 1 = 4-wide quantised (BvhNode4q, 64 B, default), 0 = 4-wide f32 (BvhNode4f, 128 B; A/B control).
    // An 8-wide quantised node (a third fewer visits) was slower: profiles/r02/ab/node_width_and_size.txt.
    const int node_kind = env_u32("RT_WF_NODES", 1) != 0 ? 1 : 0;
    const int mesh_levels = int(s->compiled.max_bvh4_stack) + 1;
    const int lds_levels = std::min<int>(mesh_levels, int(env_u32("RT_WF_LDS_LEVELS", 12)));
    const size_t lds_mesh = size_t(lds_levels) * 256 * sizeof(uint2) + 4 * kMeshWaveLds<R>;
    if (split) {
#define RT_MESH_OCC(ST, ND) do { if (multi_mesh) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, k_wf_mesh<R, ST, ND, true>, 256, lds_mesh)); \
                                 else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, k_wf_mesh<R, ST, ND, false>, 256, lds_mesh)); } while (0)
        if (stats) { if (node_kind == 1) RT_MESH_OCC(true, 1); else RT_MESH_OCC(true, 0); }
        else { if (node_kind == 1) RT_MESH_OCC(false, 1); else RT_MESH_OCC(false, 0); }
#undef RT_MESH_OCC
        if (blocks_per_cu < 1) blocks_per_cu = 1;
        blocks_per_cu = std::min<int>(blocks_per_cu, int(env_u32("RT_WF_MESH_BLOCKS", 64)));  // experiments: occupancy scaling
    }
    const uint32_t isect_blocks = uint32_t(n_cu) * uint32_t(blocks_per_cu);
    if (split) {
        size_t need = size_t(std::max(mesh_levels - lds_levels, 1)) * isect_blocks * 256 * sizeof(uint2);
        if (need > w.mesh_spill_bytes) {
            if (w.mesh_spill) (void)hipFree(w.mesh_spill);
            w.mesh_spill = nullptr;
            w.mesh_spill_bytes = 0;
            HIP_TRY(hipMalloc(&w.mesh_spill, need));
            w.mesh_spill_bytes = need;
        }
    }
    const uint32_t refill_min = env_u32("RT_WF_REFILL", 32);  // measured optimum (64 = no refill: -20 %)
    const uint32_t inner_min = env_u32("RT_WF_INNER_MIN", 16);
    // small tables staged in LDS by the prims / shade kernels when they fit
    // small tables staged in LDS by the prims / shade kernels: the longest prefix of whole tables (in the kernel's own table order)
    // that fits the budget; 32 KB keeps four workgroups per CU resident next to the queue lists
    const bool lds_tables = env_u32("RT_LDS_TABLES", 1) != 0;
    const uint32_t lds_budget = env_u32("RT_LDS_BUDGET", 32u * 1024u);
    auto staged_prefix = [&](const SmallLayout& L) -> uint32_t {
        if (!lds_tables) return 0u;
        if (L.total_bytes <= lds_budget) return L.total_bytes;
        uint32_t best = 0;  // tables are packed back to back in staging order: a table fits iff its end does
        for (int k = 0; k < ST_COUNT; k++) {
            bool prefix_ok = L.end[k] <= lds_budget;
            if (prefix_ok && L.end[k] > best) {
                // every table that starts before this one ends must fit as well (it does: ends are monotone along the order)
                best = L.end[k];
            }
        }
        return (best + 15u) & ~15u;
    };
    // re-built primitive groups as 4-wide BVHs inside k_wf_prims (OP_GROUP): nodes + a per-lane stack in LDS; scenes whose
    // groups need more than that LDS (> 24 KB of nodes, > 16 stack levels) keep the op form.  RT_WF_GROUPS=0: A/B, tests.
    const uint32_t group_levels = uint32_t(ds.view.group_stack_levels);
    const size_t group_node_bytes = size_t(ds.view.n_group_nodes) * sizeof(BvhNode4q);
    const bool groups = (split || prims_only) && plan.groups && env_u32("RT_WF_GROUPS", 1) != 0;
    const size_t lds_groups = groups ? size_t(group_levels) * 256 * 8 + group_node_bytes : 0;
    uint32_t staged_prims = staged_prefix(ds.view.lay);
    const uint32_t staged_shade = staged_prefix(ds.view.lay_shade);
    if (groups && lds_groups + staged_prims > 44u * 1024u) staged_prims = 0;  // three workgroups per CU with the group data: tables from global memory
    const int lds_prims = staged_prims == 0 ? 0 : (staged_prims == ds.view.lay.total_bytes ? 1 : 2);          // kernel variant: none / all / prefix
    // k_wf_shade: all or nothing (a staged prefix read through flat instructions was 3 % slower than global memory on the default scene),
    // and only while five workgroups still fit a CU's 160 KB next to its lists (<= 23 KB of tables; RT_LDS_SHADE_MAX overrides)
    const uint32_t shade_tables_max = env_u32("RT_LDS_SHADE_MAX", 32u * 1024u - kShadeListBytes);
    const int lds_shade = (staged_shade != 0 && staged_shade == ds.view.lay_shade.total_bytes && staged_shade <= shade_tables_max) ? 1 : (env_u32("RT_LDS_SHADE_PREFIX", 0) && staged_shade ? 2 : 0);
    // tail compaction: RT_WF_COMPACT=0 keeps the paths where they are (A/B, tests), RT_WF_COMPACT_MIN = fewest paths worth a launch
    const bool compact_tail = env_u32("RT_WF_COMPACT", 1) != 0;
    const uint32_t compact_min = std::max<uint32_t>(1u, env_u32("RT_WF_COMPACT_MIN", 1024));
    const uint32_t compact_pct = std::min<uint32_t>(75u, std::max<uint32_t>(1u, env_u32("RT_WF_COMPACT_PCT", 50)));  // ... when at most this share of the addressed slots is alive.  50 / 62 / 75: the tail's iterations of C2 take 13.0 / 12.4 / 11.9 ms, but the paths thin out by ~22 % per iteration, so 75 compacts after nearly every one (13 copies of 0.3 ms per C2 frame, 6 with 50); whole frames are equal within the noise
    uint32_t n_compactions = 0;
    const bool iter_log = env_u32("RT_WF_ITER_LOG", 0) != 0;
    const bool trace_pool = env_u32("RT_WF_TRACE", 0) != 0;  // debug: dump the first pool slots after every iteration
    const uint32_t check_every = trace_pool ? 1u : std::min<uint32_t>(32u, std::max<uint32_t>(1u, env_u32("RT_WF_CHECK", 8)));  // 4 timing events per iteration, 128 events
    const bool tex = s->compiled.needs_tex_interpreter;
    const size_t shade_lds_pad = env_u32("RT_WF_SHADE_LDS_PAD", 0);  // experiments: fewer resident blocks of the shade kernel

    HIP_TRY(hipMemsetAsync(s->d_counters, 0, sizeof(DeviceCounters), stream));
    HIP_TRY(hipEventRecord(s->ev0, stream));
    // HIP-event sums per kernel of the iteration loop: 4 events per iteration (before prims / intersect, after it,
    // after the mesh kernel, after shade); slot 0 = prims, 1 = traversal (mesh or combined intersect), 2 = shade
    double phase_ms[3] = {0.0, 0.0, 0.0};
    uint32_t isect_launches = 0, n_groups = 0;
    for (uint32_t t0 = 0; t0 < T; t0 += group) {
        n_groups++;
        uint32_t nrep = std::min(group, T - t0);
        WfGroup<R> grp{};
        grp.total = per_replica * nrep;
        grp.npix = npix;
        grp.per_replica = per_replica;
        grp.inv_per_replica = 1.0 / double(per_replica);
        grp.inv_npix = 1.0 / double(npix);
        grp.inv_width = 1.0 / double(cam.image_width);
        grp.tid0 = t0;
        grp.strata = strata;
        if (grp.total >= (1ull << 51)) return set_err(RT_E_UNSUPPORTED, "more than 2^51 samples in one replica group");
        uint32_t first = uint32_t(std::min<uint64_t>(capacity, grp.total));
        pool = pool_a;
        pool_dev_cur = w.pool_dev;
        bool on_b = false;
        pool.capacity = first;  // slots in use by this group: the kernels address slots directly while all of them are queued
        WfCounters init{};
        init.n_in = first;
        init.n_out = 0;
        init.cursor = 0;
        init.n_mesh = 0;
        init.next_sample = first;
        *w.h_ctr = init;
        HIP_TRY(hipMemcpyAsync(w.d_ctr, w.h_ctr, sizeof(WfCounters), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL((k_wf_generate<R>), dim3((first + 255) / 256), dim3(256), 0, stream, pool, first, grp, cv, pv, w.queue[0]);
        int qi = 0;
        uint32_t upper = first;  // upper bound of the queue length (never grows: slots are reused in place)
#define RT_LAUNCH_PRIMS(ST, L, VL, GR) hipLaunchKernelGGL((k_wf_prims<R, ST, L, VL, GR>), dim3((upper + WF_CHUNK - 1) / WF_CHUNK), dim3(256), (L ? size_t(staged_prims) : size_t(0)) + (GR ? lds_groups : size_t(0)) + (WF_CHUNK + 4) * 4, stream, ds.view, pool, w.queue[qi], w.mesh_queue, w.d_ctr, s->d_counters, staged_prims, group_levels)
#define RT_LAUNCH_PRIMS_ANY() do { if (vol) { if (stats) RT_LAUNCH_PRIMS(true, 0, true, false); else RT_LAUNCH_PRIMS(false, 0, true, false); } \
                                   else if (groups) { if (stats) { if (lds_prims == 2) RT_LAUNCH_PRIMS(true, 2, false, true); else if (lds_prims == 1) RT_LAUNCH_PRIMS(true, 1, false, true); else RT_LAUNCH_PRIMS(true, 0, false, true); } \
                                                      else { if (lds_prims == 2) RT_LAUNCH_PRIMS(false, 2, false, true); else if (lds_prims == 1) RT_LAUNCH_PRIMS(false, 1, false, true); else RT_LAUNCH_PRIMS(false, 0, false, true); } } \
                                   else if (stats) { if (lds_prims == 1) RT_LAUNCH_PRIMS(true, 1, false, false); else if (lds_prims == 2) RT_LAUNCH_PRIMS(true, 2, false, false); else RT_LAUNCH_PRIMS(true, 0, false, false); } \
                                   else { if (lds_prims == 1) RT_LAUNCH_PRIMS(false, 1, false, false); else if (lds_prims == 2) RT_LAUNCH_PRIMS(false, 2, false, false); else RT_LAUNCH_PRIMS(false, 0, false, false); } } while (0)
#define RT_LAUNCH_MESH_M(ST, ND, MU, QUEUE, NPTR, CPTR) hipLaunchKernelGGL((k_wf_mesh<R, ST, ND, MU>), dim3(isect_blocks), dim3(256), lds_mesh, stream, ds.view, pool, QUEUE, w.d_ctr, s->d_counters, refill_min, inner_min, static_cast<uint2*>(w.mesh_spill), lds_levels, NPTR, CPTR)
#define RT_LAUNCH_MESH_V(ST, ND, QUEUE, NPTR, CPTR) do { if (multi_mesh) RT_LAUNCH_MESH_M(ST, ND, true, QUEUE, NPTR, CPTR); else RT_LAUNCH_MESH_M(ST, ND, false, QUEUE, NPTR, CPTR); } while (0)
#define RT_LAUNCH_MESH(QUEUE, NPTR, CPTR)                                                                                                         \
    do {                                                                                                                                          \
        if (stats) { if (node_kind == 1) RT_LAUNCH_MESH_V(true, 1, QUEUE, NPTR, CPTR); else RT_LAUNCH_MESH_V(true, 0, QUEUE, NPTR, CPTR); } \
        else { if (node_kind == 1) RT_LAUNCH_MESH_V(false, 1, QUEUE, NPTR, CPTR); else RT_LAUNCH_MESH_V(false, 0, QUEUE, NPTR, CPTR); }       \
    } while (0)
        for (;;) {
            size_t ev = 0;
            // near the end of the last group the host looks after every second iteration, so that the tail is seen when it starts
            const bool near_end = s->tail_flag && t0 + nrep >= T && w.h_ctr->next_sample + 4ull * pool.capacity >= grp.total;
            // tail compaction wants to see the queue length of every iteration once the samples have run out, and to notice
            // within two iterations that they have (a quarter of the slots restarts per iteration)
            const bool all_started = w.h_ctr->next_sample >= grp.total;
            const bool closing = compact_tail && w.h_ctr->next_sample + 2ull * pool.capacity >= grp.total;
            const uint32_t check_now = (compact_tail && all_started) ? 1u : ((near_end || closing) ? std::min<uint32_t>(check_every, 2u) : check_every);
            for (uint32_t k = 0; k < check_now; k++) {
                HIP_TRY(hipEventRecord(w.events[ev++], stream));
                if (split || prims_only) {
                    RT_LAUNCH_PRIMS_ANY();
                    HIP_TRY(hipEventRecord(w.events[ev++], stream));
                    if (!prims_only) RT_LAUNCH_MESH(w.mesh_queue, &w.d_ctr->n_mesh, &w.d_ctr->cursor);
                    HIP_TRY(hipEventRecord(w.events[ev++], stream));
                } else {
                    HIP_TRY(hipEventRecord(w.events[ev++], stream));
#define RT_LAUNCH_ISECT(ST, VL) hipLaunchKernelGGL((k_wf_intersect<R, ST, VL>), dim3(isect_blocks), dim3(256), lds, stream, ds.view, pool, w.queue[qi], w.d_ctr, s->d_counters, refill_min)
                    if (stats) { if (vol) RT_LAUNCH_ISECT(true, true); else RT_LAUNCH_ISECT(true, false); }
                    else { if (vol) RT_LAUNCH_ISECT(false, true); else RT_LAUNCH_ISECT(false, false); }
#undef RT_LAUNCH_ISECT
                    HIP_TRY(hipEventRecord(w.events[ev++], stream));
                }
#define RT_LAUNCH_SHADE(ST, L, TX) hipLaunchKernelGGL((k_wf_shade<R, ST, L, TX>), dim3((upper + WF_CHUNK - 1) / WF_CHUNK), dim3(256), (L ? size_t(staged_shade) : size_t(0)) + kShadeListBytes + shade_lds_pad, stream, ds.view, cv, pv, pool, grp, w.queue[qi], w.queue[qi ^ 1], w.d_ctr, w.sample_L, s->d_counters, static_cast<const WfPool<R>*>(pool_dev_cur), staged_shade)
                if (tex) {  // interpreter variant: tables from global memory (rare scenes, fewer instantiations)
                    if (stats) RT_LAUNCH_SHADE(true, 0, true); else RT_LAUNCH_SHADE(false, 0, true);
                } else if (stats) { if (lds_shade == 1) RT_LAUNCH_SHADE(true, 1, false); else if (lds_shade == 2) RT_LAUNCH_SHADE(true, 2, false); else RT_LAUNCH_SHADE(true, 0, false); }
                else { if (lds_shade == 1) RT_LAUNCH_SHADE(false, 1, false); else if (lds_shade == 2) RT_LAUNCH_SHADE(false, 2, false); else RT_LAUNCH_SHADE(false, 0, false); }
#undef RT_LAUNCH_SHADE
                hipLaunchKernelGGL(k_wf_advance, dim3(1), dim3(1), 0, stream, w.d_ctr);
                HIP_TRY(hipEventRecord(w.events[ev++], stream));
                qi ^= 1;
                isect_launches++;
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(w.h_ctr, w.d_ctr, sizeof(WfCounters), hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            for (size_t e = 0; e + 3 < ev; e += 4) {
                float it_ms[3];
                for (int ph = 0; ph < 3; ph++) {
                    it_ms[ph] = 0.f;
                    HIP_TRY(hipEventElapsedTime(&it_ms[ph], w.events[e + ph], w.events[e + ph + 1]));
                    phase_ms[ph] += it_ms[ph];
                }
                if (iter_log)  // RT_WF_ITER_LOG=1 (with RT_WF_CHECK=1 the queue length printed is the one of this very iteration)
                    std::fprintf(stderr, "[wf iter] group %u: <= %u paths queued: prims %.3f ms, traversal %.3f ms, shade %.3f ms\n", n_groups, upper,
                                 it_ms[0], it_ms[1], it_ms[2]);
            }
            if (trace_pool) {
                const uint32_t n = std::min<uint32_t>(std::min(first, pool.capacity), env_u32("RT_WF_TRACE", 0));
                std::vector<R> a[12];
                R* src[12] = {pool.ox, pool.oy, pool.oz, pool.dx, pool.dy, pool.dz, pool.tr, pool.tg, pool.tb, pool.ht, pool.hu, pool.hv};
                for (int k = 0; k < 12; k++) { a[k].resize(n); HIP_TRY(hipMemcpy(a[k].data(), src[k], n * sizeof(R), hipMemcpyDeviceToHost)); }
                std::vector<uint64_t> rngs(n), smp(n);
                std::vector<uint32_t> dep(n), qn(first);
                std::vector<int32_t> hpc(n);
                HIP_TRY(hipMemcpy(rngs.data(), pool.rng, n * 8, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(smp.data(), pool.sample, n * 8, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(dep.data(), pool.depth, n * 4, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(hpc.data(), pool.hpc, n * 4, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(qn.data(), w.queue[qi], size_t(w.h_ctr->n_in) * 4, hipMemcpyDeviceToHost));
                std::fprintf(stderr, "[wf trace] iteration %u: %u paths queued:", isect_launches, w.h_ctr->n_in);
                for (uint32_t k = 0; k < w.h_ctr->n_in && k < 64; k++) std::fprintf(stderr, " %u", qn[k]);
                std::fprintf(stderr, "\n");
                for (uint32_t k = 0; k < n; k++)
                    std::fprintf(stderr, "  slot %u sample %llu depth %u o %.17g %.17g %.17g d %.17g %.17g %.17g thr %.6g %.6g %.6g hit t %.17g pc %d rng %016llx\n", k,
                                 (unsigned long long)smp[k], dep[k], double(a[0][k]), double(a[1][k]), double(a[2][k]), double(a[3][k]), double(a[4][k]),
                                 double(a[5][k]), double(a[6][k]), double(a[7][k]), double(a[8][k]), double(a[9][k]), hpc[k], (unsigned long long)rngs[k]);
            }
            upper = w.h_ctr->n_in;
            // every sample of the last group has been started and slots are running empty: from here on this render cannot
            // fill the GPU any more, the next frame's render (another RtScene, another stream) may start underneath it
            if (s->tail_flag && t0 + nrep >= T && upper < first) __atomic_store_n(s->tail_flag, 1, __ATOMIC_RELEASE);
            if (upper == 0) break;
            // ---- tail compaction (rt_wavefront.h k_wf_compact): fewer than half of the addressed slots are alive and none will
            //      restart: the live paths move to slots 0 .. upper-1 of the other pool, which becomes the pool ----
            if (compact_tail && w.h_ctr->next_sample >= grp.total && upper >= compact_min && uint64_t(upper) * 100 <= uint64_t(pool.capacity) * compact_pct) {
                WfPool<R> dst = on_b ? pool_a : pool_b;
                hipLaunchKernelGGL((k_wf_compact<R>), dim3((upper + 255) / 256), dim3(256), 0, stream, pool, dst, w.queue[qi], upper);
                on_b = !on_b;
                pool = dst;
                pool.capacity = upper;  // n_in == capacity: the kernels address slot i for entry i again
                pool_dev_cur = on_b ? w.pool2_dev : w.pool_dev;
                n_compactions++;
            }
        }
#undef RT_LAUNCH_PRIMS_ANY
#undef RT_LAUNCH_PRIMS
#undef RT_LAUNCH_MESH
#undef RT_LAUNCH_MESH_V
#undef RT_LAUNCH_MESH_M
        hipLaunchKernelGGL(k_wf_resolve, dim3(uint32_t((npix + 255) / 256)), dim3(256), 0, stream, w.sample_L, w.acc, npix, strata, nrep,
                           pv.spp, int(t0 == 0), d_out, int(t0 + nrep >= T));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev1, stream));
    HIP_TRY(hipStreamSynchronize(stream));
#ifdef RT_SHADE_STAMPS
    {
        unsigned long long h[16];
        HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_shade_stamps), sizeof h));
        const double trips = double(h[8] ? h[8] : 1);
        const char* names[7] = {"loop head / previous trip's tail", "state loads + resolve_hit", "throughput / rng loads + shade_hit", "stores + list appends",
                                "barrier after phase 1", "phase 2 (restarts)", "barrier + phase 3 (queue)"};
        std::fprintf(stderr, "[k_wf_shade stamps] %llu waves, %llu wave-trips (cumulative over the process)\n", h[9], h[8]);
        for (int k = 0; k < 7; k++) std::fprintf(stderr, "  %-40s %10.0f clk per wave-trip\n", names[k], double(h[k]) / trips);
    }
#endif
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    DeviceCounters hc{};
    HIP_TRY(hipMemcpy(&hc, s->d_counters, sizeof hc, hipMemcpyDeviceToHost));
    RtRenderStats& st = s->stats;
    st = RtRenderStats{};
    st.kernel_ms = ms;
    st.traversal_kernel_ms = prims_only ? 0.0 : phase_ms[1];
    st.prims_kernel_ms = (split || prims_only) ? phase_ms[0] : 0.0;
    st.shade_kernel_ms = phase_ms[2];
    st.n_launches = isect_launches;
    st.n_iterations = isect_launches;
    st.n_replica_groups = n_groups;
    st.n_tail_compactions = n_compactions;
    if (stats && split && env_u32("RT_WF_DEBUG", 0)) {
        auto pct = [](unsigned long long lanes, unsigned long long waves) { return waves ? 100.0 * double(lanes) / (64.0 * double(waves)) : 0.0; };
        std::fprintf(stderr,
                     "[k_wf_mesh] rays %llu  node code: %llu wave iterations, %.1f %% lanes active;  triangle code: %llu, %.1f %%;  "
                     "refill: %llu, %.1f %%;  stack entries culled on pop %llu\n",
                     hc.mesh_rays, hc.node_wave_iters, pct(hc.node_visits, hc.node_wave_iters), hc.tri_wave_iters,
                     pct(hc.tri_tests, hc.tri_wave_iters), hc.refill_wave_iters, pct(hc.refill_lanes, hc.refill_wave_iters), hc.pops_culled);
    }
    st.pipeline_used = RT_PIPELINE_WAVEFRONT;
    st.samples = npix * uint64_t(pv.spp);
    st.rays = hc.rays;
    st.mesh_rays = hc.mesh_rays;
    st.node_visits = hc.node_visits;
    st.tri_tests = hc.tri_tests;
    st.prim_tests = hc.prim_tests;
    st.bytes_node = split ? (node_kind == 0 ? sizeof(BvhNode4f) : sizeof(BvhNode4q)) : sizeof(BvhNode<R>);
    st.bytes_tri = sizeof(TriRec<R>);
    st.bytes_attr = sizeof(TriAttr<R>);
    // path state moved by the DOMINANT kernel per ray it traverses: ray (6 R) + bound/op read (R + 4)
    // + hit written when a triangle wins (3 R + 8) + queue entry (4)
    st.bytes_state = 6 * sizeof(R) + sizeof(R) + 4 + 3 * sizeof(R) + 8 + 4;
    // k_wf_prims per ray: ray in (6 R), closest hit out (3 R + 8); k_wf_shade per ray: ray + hit + throughput + rng + depth +
    // sample index in, ray + throughput + rng + depth out (a path that ends writes 24 B of radiance instead and restarts)
    st.bytes_state_prims = 6 * sizeof(R) + 3 * sizeof(R) + 8;
    st.bytes_state_shade = (6 + 3 + 3) * sizeof(R) + 8 + 8 + 4 + 8 + (6 + 3) * sizeof(R) + 8 + 4;
    if (!split) st.mesh_rays = hc.rays;  // combined kernel: every ray's state passes through it
    return RT_OK;
}

static int validate_render_args(const RtCameraDesc* camera, const RtRenderParams* params) {
    if (params->sqrt_spt == 0 || params->thread_count == 0) return set_err(RT_E_INVALID, "sqrt_spt and thread_count must be positive");
    if (params->band_rows != 0 && params->n_parts > 1 && params->part >= params->n_parts) return set_err(RT_E_INVALID, "part >= n_parts");
    if (camera->image_width == 0 || camera->image_height == 0) return set_err(RT_E_INVALID, "empty image");
    if (uint64_t(params->sqrt_spt) * params->sqrt_spt * params->thread_count > 0xFFFFFFFFull)
        return set_err(RT_E_UNSUPPORTED, "more than 2^32 samples per pixel");
    if (params->precision != RT_PRECISION_F64 && params->precision != RT_PRECISION_F32) return set_err(RT_E_INVALID, "unknown precision");
    return RT_OK;
}

static uint32_t owned_rows(uint32_t h, const RtRenderParams* p) {
    if (p->band_rows == 0 || p->n_parts <= 1) return h;
    uint32_t n = 0;
    for (uint32_t y = 0; y < h; y++)
        if ((y / p->band_rows) % p->n_parts == p->part) n++;
    return n;
}

}  // namespace rt

// Probe of fuzzy_reflection (rt_device.h): the routine with its fuzz == 0 shortcut next to the plain expression of
// metal.rs:33-35 / glossy.rs:66-68, same inputs, same generator state (tests/test_gpu_parity.py).
namespace rt {
__global__ void k_debug_fuzzy_reflection(uint32_t n, const double* __restrict__ reflected, const double* __restrict__ fuzz,
                                         const unsigned long long* __restrict__ state, double* __restrict__ out, unsigned long long* __restrict__ state_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3<double> r = mk<double>(reflected[3 * i], reflected[3 * i + 1], reflected[3 * i + 2]);
    Rng a, b;
    a.s = b.s = state[i];
    const V3<double> da = fuzzy_reflection(r, fuzz[i], a);
    const V3<double> db = r + random_unit<double>(b) * fuzz[i] * length(r);
    out[6 * i + 0] = da.x; out[6 * i + 1] = da.y; out[6 * i + 2] = da.z;
    out[6 * i + 3] = db.x; out[6 * i + 4] = db.y; out[6 * i + 5] = db.z;
    state_out[2 * i] = a.s; state_out[2 * i + 1] = b.s;
}
}  // namespace rt

extern "C" {

const char* rt_last_error(void) { return rt::g_err.c_str(); }

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rt_scene_create(const RtSceneDesc* desc, int device, RtScene** out) {
    using namespace rt;
    if (!out) return set_err(RT_E_INVALID, "rt_scene_create: out is NULL");
    *out = nullptr;
    std::unique_ptr<RtScene> s(new (std::nothrow) RtScene);
    if (!s) return set_err(RT_E_NOMEM, "out of memory");
    std::string err;
    CompileOptions opt;
    const char* builder = std::getenv("RT_BVH_BUILDER");  // "device" / "host" override the scene's flag
    opt.bvh_on_device = desc && (desc->flags & RT_SCENE_BVH_ON_DEVICE) != 0;
    if (builder && !std::strcmp(builder, "device")) opt.bvh_on_device = true;
    if (builder && !std::strcmp(builder, "host")) opt.bvh_on_device = false;
    if (const char* e = std::getenv("RT_PRIM_REBUILD")) opt.rebuild_prim_groups = std::atoi(e) != 0;  // A/B, tests
    int n = 0;
    if (opt.bvh_on_device) {  // the device builder needs its device before the scene is compiled
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return set_err(RT_E_DEVICE, "no HIP device available");
        if (device < 0 || device >= n) return set_err(RT_E_INVALID, "device index out of range");
        HIP_TRY(hipSetDevice(device));
    }
    int st = compile_scene(desc, &s->compiled, &err, opt);
    if (st != RT_OK) return set_err(st, err);
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return set_err(RT_E_DEVICE, "no HIP device available");
    if (device < 0 || device >= n) return set_err(RT_E_INVALID, "device index out of range");
    s->device = device;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&s->ev0));
    HIP_TRY(hipEventCreate(&s->ev1));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_counters), sizeof(DeviceCounters)));
    *out = s.release();
    return RT_OK;
}

void rt_scene_destroy(RtScene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    s->f64.reset();
    s->f32.reset();
    rt::wf_release_pool(s->wf);
    if (s->wf.mesh_spill) (void)hipFree(s->wf.mesh_spill);
    if (s->wf.d_ctr) (void)hipFree(s->wf.d_ctr);
    if (s->wf.h_ctr) (void)hipHostFree(s->wf.h_ctr);
    if (s->wf.sample_L) (void)hipFree(s->wf.sample_L);
    if (s->wf.acc) (void)hipFree(s->wf.acc);
    for (hipEvent_t e : s->wf.events)
        if (e) (void)hipEventDestroy(e);
    if (s->d_counters) (void)hipFree(s->d_counters);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

uint32_t rt_owned_rows(uint32_t image_height, const RtRenderParams* params) {
    if (!params) return 0;
    return rt::owned_rows(image_height, params);
}

static int render_device_impl(const RtScene* scene, const RtCameraDesc* camera, const RtRenderParams* params,
                              double* d_rgba_out, void* stream) {
    using namespace rt;
    if (!scene || !camera || !params || !d_rgba_out) return set_err(RT_E_INVALID, "rt_render_device: NULL argument");
    if (int v = validate_render_args(camera, params)) return v;
    RtScene* s = const_cast<RtScene*>(scene);  // stats + lazily built tables; the scene data itself is immutable
    HIP_TRY(hipSetDevice(s->device));
    uint32_t owned = owned_rows(camera->image_height, params);
    if (owned == 0) return RT_OK;
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : s->stream;
    // AUTO: the wavefront scheduler (since its state accesses are coalesced streams it beats the per-pixel
    // megakernel on every scene measured, with or without meshes); RT_AUTO_MEGA_NO_MESH=1 restores the old rule
    bool has_mesh = !s->compiled.meshes.empty();
    bool wavefront = params->pipeline == RT_PIPELINE_WAVEFRONT ||
                     (params->pipeline == RT_PIPELINE_AUTO && (has_mesh || env_u32("RT_AUTO_MEGA_NO_MESH", 0) == 0));
    if (params->max_depth == 0) wavefront = false;  // every sample is black (camera.rs:290): nothing to schedule
    if (params->precision == RT_PRECISION_F32) {
        if (!s->f32) {
            auto ds = std::make_unique<DeviceScene<float>>();
            int r = ds->build(s->compiled);
            if (r != RT_OK) return r;
            s->f32 = std::move(ds);
        }
        if (wavefront) return render_wavefront<float>(s, *s->f32, *camera, *params, owned, d_rgba_out, st);
        return render_typed<float>(s, *s->f32, *camera, *params, owned, d_rgba_out, st);
    }
    if (!s->f64) {
        auto ds = std::make_unique<DeviceScene<double>>();
        int r = ds->build(s->compiled);
        if (r != RT_OK) return r;
        s->f64 = std::move(ds);
    }
    if (wavefront) return render_wavefront<double>(s, *s->f64, *camera, *params, owned, d_rgba_out, st);
    return render_typed<double>(s, *s->f64, *camera, *params, owned, d_rgba_out, st);
}

int rt_render_device(const RtScene* scene, const RtCameraDesc* camera, const RtRenderParams* params,
                     double* d_rgba_out, void* stream) {
    int r = render_device_impl(scene, camera, params, d_rgba_out, stream);
    // whoever waits for this render's tail (rt_scene_set_tail_flag) is released at the latest here, errors included
    if (scene && scene->tail_flag) __atomic_store_n(scene->tail_flag, 1, __ATOMIC_RELEASE);
    return r;
}

int rt_scene_set_tail_flag(RtScene* scene, int32_t* flag) {
    if (!scene) return rt::set_err(RT_E_INVALID, "rt_scene_set_tail_flag: NULL scene");
    scene->tail_flag = flag;
    return RT_OK;
}

static int render_host_impl(const RtScene* scene, const RtCameraDesc* camera, const RtRenderParams* params, double* rgba_out) {
    using namespace rt;
    if (!scene || !camera || !params || !rgba_out) return set_err(RT_E_INVALID, "rt_render: NULL argument");
    if (int v = validate_render_args(camera, params)) return v;
    HIP_TRY(hipSetDevice(scene->device));
    uint32_t owned = owned_rows(camera->image_height, params);
    size_t bytes = size_t(owned) * camera->image_width * 4 * sizeof(double);
    if (bytes == 0) return RT_OK;
    double* d_out = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_out), bytes));
    int st = rt_render_device(scene, camera, params, d_out, nullptr);
    if (st == RT_OK) {
        hipError_t e = hipMemcpy(rgba_out, d_out, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) st = set_err(RT_E_DEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e));
    }
    (void)hipFree(d_out);
    return st;
}

int rt_render(const RtScene* scene, const RtCameraDesc* camera, const RtRenderParams* params, double* rgba_out) {
    int r = render_host_impl(scene, camera, params, rgba_out);
    if (scene && scene->tail_flag) __atomic_store_n(scene->tail_flag, 1, __ATOMIC_RELEASE);  // also on the early error returns
    return r;
}

int rt_debug_fuzzy_reflection(int device, uint32_t n, const double* reflected, const double* fuzz, const uint64_t* state, double* out, uint64_t* state_out) {
    using namespace rt;
    if (!n || !reflected || !fuzz || !state || !out || !state_out) return set_err(RT_E_INVALID, "rt_debug_fuzzy_reflection: NULL argument");
    HIP_TRY(hipSetDevice(device));
    DeviceBuffers buf;
    auto alloc = [&](size_t bytes, void** p) -> int { HIP_TRY(hipMalloc(p, bytes)); buf.allocs.push_back(*p); return RT_OK; };
    void *d_r = nullptr, *d_f = nullptr, *d_s = nullptr, *d_o = nullptr, *d_so = nullptr;
    int st;
    if ((st = alloc(size_t(n) * 24, &d_r)) || (st = alloc(size_t(n) * 8, &d_f)) || (st = alloc(size_t(n) * 8, &d_s)) ||
        (st = alloc(size_t(n) * 48, &d_o)) || (st = alloc(size_t(n) * 16, &d_so))) return st;
    HIP_TRY(hipMemcpy(d_r, reflected, size_t(n) * 24, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_f, fuzz, size_t(n) * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_s, state, size_t(n) * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_fuzzy_reflection, dim3((n + 255) / 256), dim3(256), 0, nullptr, n, static_cast<const double*>(d_r), static_cast<const double*>(d_f),
                       static_cast<const unsigned long long*>(d_s), static_cast<double*>(d_o), static_cast<unsigned long long*>(d_so));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, d_o, size_t(n) * 48, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(state_out, d_so, size_t(n) * 16, hipMemcpyDeviceToHost));
    return RT_OK;
}

// Diagnostic: traces ONE sample on the device and returns its radiance plus a per-bounce record
// (17 doubles, see k_trace_sample).  Returns the bounce count.
int rt_debug_trace_sample(const RtScene* scene, const RtCameraDesc* camera, const RtRenderParams* params,
                          uint32_t tid, uint32_t x, uint32_t y, uint32_t sx, uint32_t sy,
                          double* rgb_out, double* trace_out, uint32_t max_bounces) {
    using namespace rt;
    if (!scene || !camera || !params || !rgb_out || !trace_out) return set_err(RT_E_INVALID, "NULL argument");
    RtScene* s = const_cast<RtScene*>(scene);
    HIP_TRY(hipSetDevice(s->device));
    double* d_buf = nullptr;
    size_t n_d = 3 + size_t(max_bounces) * 17 + 1;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_buf), n_d * sizeof(double)));
    HIP_TRY(hipMemset(d_buf, 0, n_d * sizeof(double)));
    uint32_t* d_n = reinterpret_cast<uint32_t*>(d_buf + 3 + size_t(max_bounces) * 17);
    if (params->precision == RT_PRECISION_F32) {
        if (!s->f32) { auto ds = std::make_unique<DeviceScene<float>>(); int r = ds->build(s->compiled); if (r != RT_OK) return r; s->f32 = std::move(ds); }
        size_t lds = size_t(s->f32->view.stack_entries) * 64 * sizeof(int);
        hipLaunchKernelGGL((k_trace_sample<float>), dim3(1), dim3(64), lds, s->stream, s->f32->view, make_camera_view<float>(*camera, *params),
                           make_params_view<float>(*params, camera->image_height), tid, x, y, sx, sy, d_buf, d_buf + 3, max_bounces, d_n);
    } else {
        if (!s->f64) { auto ds = std::make_unique<DeviceScene<double>>(); int r = ds->build(s->compiled); if (r != RT_OK) return r; s->f64 = std::move(ds); }
        size_t lds = size_t(s->f64->view.stack_entries) * 64 * sizeof(int);
        hipLaunchKernelGGL((k_trace_sample<double>), dim3(1), dim3(64), lds, s->stream, s->f64->view, make_camera_view<double>(*camera, *params),
                           make_params_view<double>(*params, camera->image_height), tid, x, y, sx, sy, d_buf, d_buf + 3, max_bounces, d_n);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    std::vector<double> h(n_d);
    HIP_TRY(hipMemcpy(h.data(), d_buf, n_d * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(d_buf);
    std::memcpy(rgb_out, h.data(), 3 * sizeof(double));
    std::memcpy(trace_out, h.data() + 3, size_t(max_bounces) * 17 * sizeof(double));
    uint32_t n;
    std::memcpy(&n, h.data() + 3 + size_t(max_bounces) * 17, sizeof n);
    return int(n);
}

int rt_scene_info(const RtSceneDesc* desc, uint32_t* flags_out) {
    using namespace rt;
    if (!desc || !flags_out) return set_err(RT_E_INVALID, "rt_scene_info: NULL argument");
    CompiledScene cs;
    std::string err;
    int st = compile_scene(desc, &cs, &err, CompileOptions());
    if (st != RT_OK) return set_err(st, err);
    *flags_out = (cs.zero_weight_stop ? RT_SCENE_INFO_ZERO_WEIGHT_STOP : 0u) | (cs.needs_tex_interpreter ? RT_SCENE_INFO_TEX_INTERPRETER : 0u) |
                 (cs.volumes.empty() ? 0u : RT_SCENE_INFO_VOLUMES);
    return RT_OK;
}

int rt_scene_mesh_stats(const RtSceneDesc* desc, uint64_t out[8]) {
    using namespace rt;
    if (!desc || !out) return set_err(RT_E_INVALID, "rt_scene_mesh_stats: NULL argument");
    CompiledScene cs;
    std::string err;
    CompileOptions opt;
    if (const char* e = std::getenv("RT_PRIM_REBUILD")) opt.rebuild_prim_groups = std::atoi(e) != 0;
    int st = compile_scene(desc, &cs, &err, opt);
    if (st != RT_OK) return set_err(st, err);
    out[0] = cs.tris.size();
    out[1] = cs.nodes.size();
    out[2] = cs.nodes4.size();
    out[3] = cs.max_bvh_depth;
    out[4] = cs.max_bvh4_stack;
    out[5] = cs.ops.size();
    out[6] = cs.n_rebuilt_groups;
    out[7] = cs.n_rebuilt_prims;
    return RT_OK;
}

int rt_scene_program(const RtSceneDesc* desc, int32_t* ops_out, uint32_t capacity, uint32_t* n_ops_out, uint64_t info[8]) {
    using namespace rt;
    if (!desc || !n_ops_out || !info) return set_err(RT_E_INVALID, "rt_scene_program: NULL argument");
    CompiledScene cs;
    std::string err;
    CompileOptions opt;
    if (const char* e = std::getenv("RT_PRIM_REBUILD")) opt.rebuild_prim_groups = std::atoi(e) != 0;
    int st = compile_scene(desc, &cs, &err, opt);
    if (st != RT_OK) return set_err(st, err);
    *n_ops_out = uint32_t(cs.ops.size());
    if (ops_out)
        for (size_t i = 0; i < cs.ops.size() && i < capacity; i++) {
            ops_out[4 * i + 0] = cs.ops[i].type; ops_out[4 * i + 1] = cs.ops[i].arg;
            ops_out[4 * i + 2] = cs.ops[i].skip; ops_out[4 * i + 3] = cs.ops[i].chain;
        }
    const WavefrontPlan plan = plan_wavefront(cs);
    info[0] = cs.mesh_ops.size();
    info[1] = cs.groups.size();
    info[2] = cs.group_nodes4.size();
    info[3] = cs.max_group_stack;
    info[4] = cs.lights.size();
    info[5] = cs.volumes.size();
    info[6] = (plan.split ? 1u : 0u) | (plan.vol_prims ? 2u : 0u) | (plan.multi_mesh ? 4u : 0u) | (plan.groups ? 8u : 0u);
    info[7] = cs.group_prims.size();
    return RT_OK;
}

int rt_get_stats(const RtScene* scene, RtRenderStats* out) {
    if (!scene || !out) return rt::set_err(RT_E_INVALID, "rt_get_stats: NULL argument");
    *out = scene->stats;
    return RT_OK;
}

}  // extern "C"
