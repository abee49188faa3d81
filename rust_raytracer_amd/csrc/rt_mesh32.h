// rt_mesh32.h — two-stage closest hit for the deferred mesh (included by rt_kernels.hip after rt_wavefront.h).
//
// k_wf_mesh (rt_wavefront.h) walks the 4-wide f32 BVH and tests every triangle of every visited leaf EXACTLY, in R.
// In f64 that test dominates the kernel's registers (124 VGPRs = 4 waves per SIMD in a kernel that spends 59 % of its
// wave cycles waiting for memory) and half of its vector instructions, although nine tests out of ten are misses.
// Here the search is split:
//
//   k_wf_mesh32       the same traversal, but every triangle meets a CONSERVATIVE f32 test first.  It can only say
//                     "certainly missed" (with an error bound on every quantity it compares, below) or "candidate";
//                     a candidate that is certainly a hit also tightens the f32 search bound with an UPPER bound of
//                     its distance.  Candidates are appended, in traversal order, to a small per-path list.
//                     No f64 state per lane: more resident waves for the pointer chase, 48-B instead of 80-B
//                     triangle fetches, f32 instead of f64 arithmetic for the misses.
//   k_wf_mesh_exact   runs the exact test of mesh.rs:62-107 in R over each path's candidates in that same order, with
//                     the interval and tie rules of k_wf_mesh: the closest hit is the one k_wf_mesh finds, bit for bit
//                     (tests: every mesh scene, both variants against the megakernel and the oracle).
//   k_wf_mesh         still runs, on the paths whose list overflowed (kMeshCandCap) — a handful per million.
//
// Why the result is the same.  Let X be any triangle the exact test accepts for a path, at distance t*.  (1) The f32
// bound `tmax32` only ever shrinks to an upper bound of the distance of a triangle that the exact test certainly
// accepts, so it never drops below the closest accepted distance; with the conservative node boxes (rt_scene.h,
// BvhNode4f) the leaf of X is visited.  (2) The f32 test never classifies X as missed (error analysis below).
// (3) X is appended unless the list is full, in which case the path goes to the exact kernel.  (4) The order of the
// visited leaves does not depend on how tight the bound is (sorted pushes, culling only removes entries), so the
// candidates arrive in the order in which k_wf_mesh would have tested them and `strictly nearer wins` picks the same
// triangle among equal distances.
//
// Error analysis of the f32 test (u = 2^-24).  Inputs: d~ = fl(d), v0~ = fl(v0), e~ = fl(e) (relative error u each),
// oc~ = fl(oc) with oc = o + d t_shift the entry point into the mesh box (computed in R), so |oc|, |v0| <= S, the
// mesh extent.  b~ = fl(oc~ - v0~) differs from oc - v0 by at most u (|b~| + 2 S) per component.  With
// Dm = |d|_inf, L >= max(|e1|_inf, |e2|_inf), Bm = |b~|_inf, every product of the Moeller-Trumbore quantities
//   p = d x e2, det = e1 . p, nu = b . p, q = b x e1, nv = d . q, nt = e2 . q
// carries at most 3 relative errors of u plus its own rounding; summing the worst cases term by term gives
//   |det~ - det| <= 48 u L^2 Dm,   |nu~ - nu|, |nv~ - nv| <= u L Dm (54 Bm + 12 S),   |nt~ - nt| <= u L^2 (54 Bm + 12 S).
// The kernel uses K = 2^-17 = 128 u:  E_det = K L^2 Dm,  E_n = K L Dm (Bm + S),  E_t = K L^2 (Bm + S): more than twice the
// bounds above, which also swallows the second-order terms, the rounding of the bounds themselves and the (2^-53-level)
// rounding of the exact test whose verdict is being predicted.  The exact test accepts only if det >= eps > 0,
// 0 <= nu <= det, nv >= 0, nu + nv <= det (in its own rounding) and t in (t_lo, t_max); so
//   det~ < -E_det                                  => det < 0                     => rejected there too
//   det~ > E_det and one of  nu~ < -E_n,  nu~ - det~ > E_n + E_det,  nv~ < -E_n,  nu~ + nv~ - det~ > 2 E_n + E_det
//                                                  => u < 0, u > 1, v < 0 or u + v > 1  => rejected there too
//   det~ > E_det and nt~ - E_t > tmax32 (det~ + E_det)  => t' > tmax32 >= closest accepted distance: cannot be nearer
//   det~ > E_det and nt~ + E_t < 0                 => the hit lies in front of the entry point: outside the mesh box
// and everything else is a candidate.  Meshes with hit_back_faces keep the one-stage kernel.
#pragma once
#include "rt_wavefront.h"

#ifndef RT_MESH32_WAVES
#define RT_MESH32_WAVES 6
#endif

namespace rt {

constexpr uint32_t kMeshCandCap = 6;            // candidates kept per path; more: the path is re-done by k_wf_mesh
constexpr uint32_t kMeshCandOverflow = 0xFFFFFFFFu;
constexpr uint32_t kMesh32WaveLds = 1024u + 64u * 4u;  // per wave: pair table (512 x u16) + one float result per lane

template <typename R, bool STATS>
__global__ void __launch_bounds__(256, RT_MESH32_WAVES) k_wf_mesh32(SceneView<R> sc, WfPool<R> pool, const uint32_t* __restrict__ mesh_queue,
                                                                     WfCounters* __restrict__ ctr, DeviceCounters* counters,
                                                                     uint32_t refill_min, uint32_t inner_min, int32_t mesh_pc,
                                                                     uint2* __restrict__ spill, int lds_levels) {
    extern __shared__ uint2 lds_stack2[];
    MeshStack stk;
    stk.lds = (LdsU64*)(lds_stack2 + threadIdx.x);
    const uint32_t lane = threadIdx.x & 63u;
    char* wave_area = reinterpret_cast<char*>(lds_stack2 + size_t(lds_levels) * 256) + (threadIdx.x >> 6) * kMesh32WaveLds;
    uint16_t* pair_tbl = reinterpret_cast<uint16_t*>(wave_area);
    float* res = reinterpret_cast<float*>(wave_area + 1024);  // < 0: missed; else candidate, value = upper bound of t' (+inf: none)
    stk.spill = spill + (size_t(blockIdx.x) * blockDim.x + threadIdx.x);
    stk.lds_levels = lds_levels;
    stk.spill_stride = gridDim.x * blockDim.x;
    const uint32_t n = ctr->n_mesh;
    const Op mop = sc.ops[mesh_pc];
    const MeshInst mi = sc.meshes[mop.arg];
    const Bounds<R> rb = sc.mesh_bounds[mop.arg];
    const BvhNode4f* nodes = sc.nodes4 + mi.node4_base;
    const TriRec32* tris = sc.tris32 + mi.tri_base;
    const float S = __uint_as_float(mi.extent_bits);  // >= every |coordinate| of the mesh box
    const float K = 7.62939453125e-6f;                 // 2^-17

    LaneCounters cnt;
    uint32_t w_node = 0, w_tri = 0, w_refill = 0, l_refill = 0, l_culled = 0, l_cand = 0, l_over = 0;
    bool has = false;
    bool exhausted = false;
    WaveRange range;
    uint32_t slot = 0;
    // f32 ray: origin moved onto the mesh box, t' = t - t_shift measured from there
    float ocx = 0.f, ocy = 0.f, ocz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f, dmax = 0.f;
    float ivx = 0.f, ivy = 0.f, ivz = 0.f, oix = 0.f, oiy = 0.f, oiz = 0.f, tmax32 = 0.f;
    uint32_t nearx = 0, neary = 1, nearz = 2;
    uint32_t n_cand = 0;
    int32_t node = 0;
    int sp = 0;

    auto pop_next = [&]() {
        for (;;) {
            if (sp == 0) {
                has = false;
                at(pool.cn, slot) = n_cand;
                return;
            }
            sp--;
            uint2 e = stk.get(sp);
            if (__uint_as_float(e.y) <= tmax32) {
                node = int32_t(e.x);
                return;
            }
            if (STATS) l_culled++;
        }
    };

    for (;;) {
        // ---- refill ----
        unsigned long long idle = __ballot(!has);
        uint32_t n_idle = uint32_t(__popcll(idle));
        if (!exhausted && n_idle >= refill_min) {
            uint32_t my = 0;
            if (STATS) w_refill++;
            if (wave_fetch(range, idle, &ctr->cursor, n, exhausted, my)) {
                if (STATS) l_refill++;
                slot = mesh_queue[my];
                const R big = sizeof(R) == 8 ? R(1e150) : R(1e18);
                Ray<R> wray = make_ray(mk<R>(at(pool.ox, slot), at(pool.oy, slot), at(pool.oz, slot)), mk<R>(at(pool.dx, slot), at(pool.dy, slot), at(pool.dz, slot)));
                Ray<R> ray = ray_in_chain(sc, wray, mop.chain);
                V3<R> inv = {fabs(ray.inv.x) > big ? copysign(big, ray.inv.x) : ray.inv.x,
                             fabs(ray.inv.y) > big ? copysign(big, ray.inv.y) : ray.inv.y,
                             fabs(ray.inv.z) > big ? copysign(big, ray.inv.z) : ray.inv.z};
                R e0x = (rb.lo[0] - ray.o.x) * inv.x, e1x = (rb.hi[0] - ray.o.x) * inv.x;
                R e0y = (rb.lo[1] - ray.o.y) * inv.y, e1y = (rb.hi[1] - ray.o.y) * inv.y;
                R e0z = (rb.lo[2] - ray.o.z) * inv.z, e1z = (rb.hi[2] - ray.o.z) * inv.z;
                R t_shift = fmax(fmax(fmin(e0x, e1x), fmin(e0y, e1y)), fmax(fmin(e0z, e1z), R(0)));
                if (!(t_shift < Lim<R>::inf())) t_shift = R(0);
                V3<R> oc = ray.o + ray.d * t_shift;
                const float big32 = 1e18f;
                ocx = float(oc.x); ocy = float(oc.y); ocz = float(oc.z);
                dx = float(ray.d.x); dy = float(ray.d.y); dz = float(ray.d.z);
                dmax = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
                ivx = 1.0f / dx; ivy = 1.0f / dy; ivz = 1.0f / dz;
                ivx = fabsf(ivx) > big32 ? copysignf(big32, ivx) : ivx;
                ivy = fabsf(ivy) > big32 ? copysignf(big32, ivy) : ivy;
                ivz = fabsf(ivz) > big32 ? copysignf(big32, ivz) : ivz;
                oix = ocx * ivx; oiy = ocy * ivy; oiz = ocz * ivz;
                nearx = ivx < 0.0f ? 3u : 0u;
                neary = ivy < 0.0f ? 4u : 1u;
                nearz = ivz < 0.0f ? 5u : 2u;
                // bound left by the other primitives (same rule as k_wf_mesh: equal t goes to the op that comes first)
                R bound = at(pool.ht, slot);
                int32_t bpc = at(pool.hpc, slot);
                R t_max = (bpc > mesh_pc && bound < Lim<R>::inf()) ? nextafter(bound, Lim<R>::inf()) : bound;
                tmax32 = f32_at_least(t_max - t_shift);
                n_cand = 0;
                node = 0;
                sp = 0;
                has = true;
                if (STATS) cnt.mesh_rays++;
            }
        }
        if (__ballot(has) == 0ull) {
            if (exhausted) break;
            continue;
        }
        // ---- inner nodes (as in k_wf_mesh) ----
        for (;;) {
            unsigned long long inner = __ballot(has && node >= 0);
            if (inner == 0ull) break;
            if (uint32_t(__popcll(inner)) < inner_min && __ballot(has && node < 0) != 0ull) break;
            if (STATS) w_node++;
            if (has && node >= 0) {
                const float4* nd = reinterpret_cast<const float4*>(nodes + node);
                if (STATS) cnt.node_visits++;
                const float4 nx = nd[nearx], fx = nd[3u - nearx];
                const float4 ny = nd[neary], fy = nd[5u - neary];
                const float4 nz = nd[nearz], fz = nd[7u - nearz];
                const int4 cc = *reinterpret_cast<const int4*>(nd + 6);
                float nr[4];
                int32_t ch[4] = {cc.x, cc.y, cc.z, cc.w};
                const float nxa[4] = {nx.x, nx.y, nx.z, nx.w}, fxa[4] = {fx.x, fx.y, fx.z, fx.w};
                const float nya[4] = {ny.x, ny.y, ny.z, ny.w}, fya[4] = {fy.x, fy.y, fy.z, fy.w};
                const float nza[4] = {nz.x, nz.y, nz.z, nz.w}, fza[4] = {fz.x, fz.y, fz.z, fz.w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float tn = fmaxf(fmaxf(fmaf(nxa[k], ivx, -oix), fmaf(nya[k], ivy, -oiy)), fmaxf(fmaf(nza[k], ivz, -oiz), 0.0f));
                    float tf = fminf(fminf(fmaf(fxa[k], ivx, -oix), fmaf(fya[k], ivy, -oiy)), fminf(fmaf(fza[k], ivz, -oiz), tmax32));
                    bool h = (tn <= tf) && ch[k] != kEmptyChild;
                    nr[k] = h ? tn : __builtin_huge_valf();
                }
#define RT_CE(a, b)                                                   \
    if (nr[a] > nr[b]) {                                              \
        float tn_ = nr[a]; nr[a] = nr[b]; nr[b] = tn_;                \
        int32_t tc_ = ch[a]; ch[a] = ch[b]; ch[b] = tc_;              \
    }
                RT_CE(0, 1) RT_CE(2, 3) RT_CE(0, 2) RT_CE(1, 3) RT_CE(1, 2)
#undef RT_CE
                const float miss = __builtin_huge_valf();
                if (nr[0] < miss) {
                    if (nr[3] < miss) { stk.put(sp, ch[3], nr[3]); sp++; }
                    if (nr[2] < miss) { stk.put(sp, ch[2], nr[2]); sp++; }
                    if (nr[1] < miss) { stk.put(sp, ch[1], nr[1]); sp++; }
                    node = ch[0];
                } else {
                    pop_next();
                }
            }
        }
        // ---- leaves: flattened conservative f32 tests ----
        {
            const bool leaf = has && node < 0;
            const uint32_t code = uint32_t(~node);
            const uint32_t first = leaf ? (code >> 3) : 0u, count = leaf ? ((code & 7u) + 1u) : 0u;
            uint32_t pre = 0, total = 0;
#pragma unroll
            for (int bit = 0; bit < 4; bit++) {
                unsigned long long m = __ballot(((count >> bit) & 1u) != 0u);
                pre += lane_prefix(m) << bit;
                total += uint32_t(__popcll(m)) << bit;
            }
            if (total != 0u) {
                for (uint32_t j = 0; j < 8u; j++) {
                    if (__ballot(j < count) == 0ull) break;
                    if (j < count) pair_tbl[pre + j] = uint16_t(lane | (j << 8));
                }
                __builtin_amdgcn_wave_barrier();
                for (uint32_t c0 = 0; c0 < total; c0 += 64u) {
                    if (STATS) w_tri++;
                    const uint32_t w = c0 + lane;
                    const bool act = w < total;
                    const uint32_t e = act ? uint32_t(pair_tbl[w]) : 0u;
                    const int owner = int(e & 0xFFu);
                    const uint32_t k = e >> 8;
                    const float pox = __shfl(ocx, owner), poy = __shfl(ocy, owner), poz = __shfl(ocz, owner);
                    const float pdx = __shfl(dx, owner), pdy = __shfl(dy, owner), pdz = __shfl(dz, owner);
                    const float pdm = __shfl(dmax, owner), ptm = __shfl(tmax32, owner);
                    const uint32_t pfirst = uint32_t(__shfl(int(first), owner));
                    float r = -1.0f;  // missed
                    if (act) {
                        const float4* tr = reinterpret_cast<const float4*>(tris + pfirst + k);
                        const float4 a = tr[0], bb = tr[1], c = tr[2];  // (v0, L), (e1, -), (e2, -)
                        if (STATS) cnt.tri_tests++;
                        const float bx = pox - a.x, by = poy - a.y, bz = poz - a.z;
                        const float Bm = fmaxf(fmaxf(fabsf(bx), fabsf(by)), fabsf(bz));
                        const float L = a.w;
                        // p = d x e2, det = e1 . p, nu = b . p
                        const float px = pdy * c.z - pdz * c.y, py = pdz * c.x - pdx * c.z, pz = pdx * c.y - pdy * c.x;
                        const float det = bb.x * px + bb.y * py + bb.z * pz;
                        const float nu = bx * px + by * py + bz * pz;
                        // q = b x e1, nv = d . q, nt = e2 . q
                        const float qx = by * bb.z - bz * bb.y, qy = bz * bb.x - bx * bb.z, qz = bx * bb.y - by * bb.x;
                        const float nv = pdx * qx + pdy * qy + pdz * qz;
                        const float nt = c.x * qx + c.y * qy + c.z * qz;
                        const float KL = K * L;
                        const float Edet = KL * L * pdm, En = KL * pdm * (Bm + S), Et = KL * L * (Bm + S);
                        bool missed;
                        if (det < -Edet) missed = true;                 // back face
                        else if (!(det > Edet)) missed = false;         // grazing: the sign of det is not certain
                        else
                            missed = nu < -En || nu - det > En + Edet || nv < -En || nu + nv - det > 2.0f * En + Edet ||
                                     nt - Et > ptm * (det + Edet) * 1.000002f || nt + Et < 0.0f;
                        if (!missed) {
                            r = __builtin_huge_valf();  // candidate; a certain hit also bounds the search
                            // certainly accepted by the exact test: inside the triangle by the margins, and farther than the
                            // self-intersection threshold t_lo = 0.001 even if no shift was applied (t' <= t)
                            const bool certain = det > 2.0f * Edet && nu > En && nv > En && det - nu - nv > 2.0f * En + Edet &&
                                                 nt - Et > 0.0010001f * (det + Edet) * 1.000002f;
                            if (certain) r = (nt + Et) / (det - Edet) * 1.000002f;
                        }
                    }
                    res[lane] = r;
                    __builtin_amdgcn_wave_barrier();
                    if (leaf) {
                        const int jlo = max(0, int(c0) - int(pre));
                        const int jhi = min(int(count), int(c0) + 64 - int(pre));
                        for (int j = jlo; j < jhi; j++) {
                            const float rj = res[int(pre) + j - int(c0)];
                            if (rj < 0.0f) continue;
                            if (n_cand < kMeshCandCap) at(pool.ctri + size_t(n_cand) * pool.cand_stride, slot) = mi.tri_base + first + uint32_t(j);
                            n_cand++;
                            if (STATS) l_cand++;
                            tmax32 = fminf(tmax32, rj);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if (leaf) {
                    if (n_cand > kMeshCandCap) {  // list full: the exact kernel re-does this path from scratch
                        n_cand = kMeshCandOverflow;
                        if (STATS) l_over++;
                        sp = 0;
                    }
                    pop_next();
                }
            }
        }
    }
    if (STATS) {
        atomicAdd(&counters->mesh_rays, (unsigned long long)cnt.mesh_rays);
        atomicAdd(&counters->node_visits, (unsigned long long)cnt.node_visits);
        atomicAdd(&counters->tri_tests, (unsigned long long)cnt.tri_tests);
        atomicAdd(&counters->refill_lanes, (unsigned long long)l_refill);
        atomicAdd(&counters->pops_culled, (unsigned long long)l_culled);
        atomicAdd(&counters->candidates, (unsigned long long)l_cand);
        atomicAdd(&counters->cand_overflows, (unsigned long long)l_over);
        uint32_t wn = w_node, wt = w_tri, wr = w_refill;  // wave-uniform
        if ((threadIdx.x & 63u) == 0) {
            atomicAdd(&counters->node_wave_iters, (unsigned long long)wn);
            atomicAdd(&counters->tri_wave_iters, (unsigned long long)wt);
            atomicAdd(&counters->refill_wave_iters, (unsigned long long)wr);
        }
    }
}

// Exact tests over the candidate lists (mesh.rs:62-107, the interval and tie rules of k_wf_mesh).  One lane per queued
// path; paths whose list overflowed are collected for k_wf_mesh.
template <typename R, bool STATS>
__global__ void __launch_bounds__(256) k_wf_mesh_exact(SceneView<R> sc, WfPool<R> pool, const uint32_t* __restrict__ mesh_queue,
                                                       uint32_t* __restrict__ fallback_queue, WfCounters* __restrict__ ctr,
                                                       DeviceCounters* counters, int32_t mesh_pc) {
    __shared__ uint32_t fb_list[WF_CHUNK];
    __shared__ uint32_t lc[2];
    if (threadIdx.x < 2) lc[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t n = ctr->n_mesh;
    const uint32_t begin = blockIdx.x * WF_CHUNK;
    const uint32_t end = min(n, begin + WF_CHUNK);
    const R t_lo = R(0.001);
    const Op mop = sc.ops[mesh_pc];
    const MeshInst mi = sc.meshes[mop.arg];
    const bool hit_back = (mi.flags & RT_MESH_HIT_BACK_FACES) != 0;
    uint32_t n_tests = 0;
    for (uint32_t base = begin; base < end; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        bool overflow = false;
        uint32_t slot = 0;
        if (i < end) {
            slot = mesh_queue[i];
            const uint32_t nc = at(pool.cn, slot);
            overflow = nc == kMeshCandOverflow;
            if (!overflow && nc != 0u) {
                Ray<R> wray = make_ray(mk<R>(at(pool.ox, slot), at(pool.oy, slot), at(pool.oz, slot)), mk<R>(at(pool.dx, slot), at(pool.dy, slot), at(pool.dz, slot)));
                const Ray<R> ray = ray_in_chain(sc, wray, mop.chain);
                R bound = at(pool.ht, slot);
                int32_t bpc = at(pool.hpc, slot);
                R t_max = (bpc > mesh_pc && bound < Lim<R>::inf()) ? nextafter(bound, Lim<R>::inf()) : bound;
                R hit_u = R(0), hit_v = R(0);
                int32_t hit_tri = -1;
                for (uint32_t k = 0; k < nc; k++) {
                    const uint32_t tslot = at(pool.ctri + size_t(k) * pool.cand_stride, slot);
                    const TriRec<R>& tr = sc.tris[tslot];
                    if (STATS) n_tests++;
                    V3<R> edge1 = ld3(tr.e1), edge2 = ld3(tr.e2);
                    V3<R> ray_x_edge2 = cross(ray.d, edge2);
                    R det = dot(edge1, ray_x_edge2);
                    R dd = hit_back ? fabs(det) : det;
                    if (dd < Lim<R>::eps()) continue;
                    R inv_det = R(1) / det;
                    V3<R> b = ray.o - ld3(tr.v0);
                    R u = dot(b, ray_x_edge2) * inv_det;
                    if (u < R(0) || u > R(1)) continue;
                    V3<R> b_x_edge1 = cross(b, edge1);
                    R v = dot(ray.d, b_x_edge1) * inv_det;
                    if (v < R(0) || u + v > R(1)) continue;
                    R t = dot(edge2, b_x_edge1) * inv_det;
                    if (t <= t_lo || t_max <= t) continue;
                    t_max = t; hit_u = u; hit_v = v; hit_tri = int32_t(tslot);
                }
                if (hit_tri >= 0) {
                    at(pool.ht, slot) = t_max; at(pool.hu, slot) = hit_u; at(pool.hv, slot) = hit_v;
                    at(pool.hpc, slot) = mesh_pc; at(pool.htri, slot) = hit_tri;
                }
            }
        }
        lds_append(overflow, slot, fb_list, &lc[0]);
    }
    __syncthreads();
    const uint32_t n_list = lc[0];
    if (threadIdx.x == 0 && n_list) lc[1] = atomicAdd(&ctr->n_fallback, n_list);
    __syncthreads();
    const uint32_t qb = lc[1];
    for (uint32_t j = threadIdx.x; j < n_list; j += blockDim.x) fallback_queue[qb + j] = fb_list[j];
    if (STATS) {
        for (int off = 32; off > 0; off >>= 1) n_tests += __shfl_down(n_tests, off);
        if ((threadIdx.x & 63u) == 0 && n_tests) atomicAdd(&counters->exact_tests, (unsigned long long)n_tests);
    }
}

}  // namespace rt
