// rt_wavefront.h — iterative wavefront scheduler (included by rt_kernels.hip).
//
// The reference's recursion `ray_color -> world.test -> scatter -> ray_color` (camera.rs:282-332)
// becomes a loop over a POOL of paths kept in HBM as structure-of-arrays:
//
//   k_wf_generate   camera rays for the first P samples                       (camera.rs:260-280)
//   repeat until no path is alive:
//     k_wf_intersect   closest hit of every queued path: persistent waves pull rays from the
//                      queue with one wave-aggregated atomic and REFILL idle lanes while the
//                      other lanes are still walking the BVH (per-lane stack in LDS)
//     k_wf_shade       emission / scatter / light-biased mixture pdf for every hit; a path that
//                      ends writes its radiance to the per-sample buffer and restarts IN PLACE
//                      on the next unrendered sample; surviving slots are compacted into the
//                      next queue with ballot + mbcnt prefix sums (one atomic per wave)
//   k_wf_resolve    per pixel: sum the sample radiances in the reference's order
//                   (replica, sy, sx: camera.rs:217-229, 247-253) -> frame
//
// Every sample owns one slot of the per-sample radiance buffer (24 B), so the frame does not
// depend on scheduling: bit-identical between runs, partitions and pipelines.
#pragma once
#include "rt_device.h"

// Minimum waves per SIMD requested from the register allocator (second __launch_bounds__
// argument); tuned on MI355X, see DESIGN.md.
#ifndef RT_ISECT_WAVES
#define RT_ISECT_WAVES 3
#endif
#ifndef RT_MESH_WAVES
#define RT_MESH_WAVES 4
#endif
namespace rt {

template <typename R>
struct WfPool {
    uint32_t capacity;
    R *ox, *oy, *oz, *dx, *dy, *dz;  // current ray, world space
    R *tr, *tg, *tb;                 // throughput: product of the weights so far (the radiance exists only at the terminal)
    uint64_t* rng;                   // stream state
    uint64_t* sample;                // sample index within the current replica group
    uint32_t* depth;                 // remaining depth (the `depth` argument of ray_color)
    R *ht, *hu, *hv;                 // closest hit: t, (u, v)
    int32_t *hpc, *htri;             // op that produced it (-1 none), triangle slot
};

// Element `slot` of a pool array through a 32-bit BYTE offset.  `base + zext(offset)` lets the compiler address every
// array of one element size with ONE offset VGPR and keep the array bases in SGPRs (global_load ... v_off, s[base:base+1]);
// with 64-bit index arithmetic it kept a VGPR pair per array alive from the loads to the stores (22 VGPRs in k_wf_shade).
// Pool sizes are capped at 2^28 slots by the driver, so the offset cannot wrap.
template <typename T> RT_DEV T& at(T* base, uint32_t slot) {
    return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + slot * uint32_t(sizeof(T)));
}

// The same for a base pointer that was LOADED from memory (the device copy of the pool descriptor): the compiler cannot know
// its address space and would emit flat_store; the cast says "global".
template <typename T> RT_DEV void put_global(T* base, uint32_t slot, T v) {
    typedef __attribute__((address_space(1))) char GChar;
    typedef __attribute__((address_space(1))) T GT;
    *reinterpret_cast<GT*>(reinterpret_cast<GChar*>((GT*)base) + slot * uint32_t(sizeof(T))) = v;
}

struct WfCounters {
    uint32_t n_in;        // entries of the current queue
    uint32_t n_out;       // entries appended to the next queue
    uint32_t cursor;      // next queue entry to hand out (persistent intersect / mesh kernel)
    uint32_t n_mesh;      // entries of the mesh queue (paths whose ray enters a deferred mesh's box)
    unsigned long long next_sample;  // next sample (within the group) to start
};

// Sample s of a replica group -> (replica, stratum, owned pixel).  Pixels run fastest so that a
// wave starts neighbouring pixels of one stratum: coherent primary rays, coalesced buffers.
template <typename R>
struct WfGroup {
    uint64_t total;        // samples in this group = n_replicas * S*S * npix  (< 2^51, checked by the driver)
    uint64_t npix;         // owned pixels
    uint64_t per_replica;  // S*S * npix
    double inv_per_replica, inv_npix, inv_width;  // reciprocals rounded to nearest: quotient ESTIMATES, made exact in div_by
    uint32_t tid0;         // first replica of the group
    uint32_t strata;       // S*S
};

// floor(a / b) and the remainder for a < 2^51: the reciprocal estimate is off by at most one, the remainder test makes it
// exact (integers throughout: nothing here can move a pixel).  A generic 64-bit division is ~120 instructions on gfx950
// and the restart of a finished path needs three of them.
RT_DEV uint64_t div_by(uint64_t a, uint64_t b, double inv_b, uint64_t& rem) {
    uint64_t q = uint64_t(double(a) * inv_b);
    int64_t r = int64_t(a - q * b);
    if (r < 0) { q--; r += int64_t(b); }
    else if (r >= int64_t(b)) { q++; r -= int64_t(b); }
    rem = uint64_t(r);
    return q;
}

// Camera ray (origin, direction) and RNG state of sample s, in registers (camera.rs:260-280 through camera_ray).
template <typename R>
RT_DEV void wf_new_sample(uint64_t s, const WfGroup<R>& grp, const CameraView<R>& cam, const ParamsView<R>& prm, V3<R>& o, V3<R>& d, Rng& rng) {
    uint64_t rem, pix, px64;
    const uint32_t tid_local = uint32_t(div_by(s, grp.per_replica, grp.inv_per_replica, rem));
    const uint32_t st = uint32_t(div_by(rem, grp.npix, grp.inv_npix, pix));
    const uint32_t row = uint32_t(div_by(pix, cam.width, grp.inv_width, px64));
    const uint32_t px = uint32_t(px64);
    uint32_t py;
    if (prm.band_rows == 0 || prm.n_parts <= 1) py = row;
    else py = ((row / prm.band_rows) * prm.n_parts + prm.part) * prm.band_rows + (row % prm.band_rows);
    const uint32_t S = cam.sqrt_spt;
    const uint32_t sy = st / S, sx = st - sy * S;
    rng.key(prm.seed, grp.tid0 + tid_local, uint64_t(py) * cam.width + px, st);
    camera_ray(cam, px, py, sx, sy, rng, o, d);
}

template <typename R>
__global__ void __launch_bounds__(256) k_wf_generate(WfPool<R> pool, uint32_t count, WfGroup<R> grp, CameraView<R> cam,
                                                     ParamsView<R> prm, uint32_t* __restrict__ queue) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    V3<R> o, d;
    Rng rng;
    wf_new_sample(uint64_t(i), grp, cam, prm, o, d, rng);
    at(pool.ox, i) = o.x; at(pool.oy, i) = o.y; at(pool.oz, i) = o.z;
    at(pool.dx, i) = d.x; at(pool.dy, i) = d.y; at(pool.dz, i) = d.z;
    at(pool.tr, i) = R(1); at(pool.tg, i) = R(1); at(pool.tb, i) = R(1);
    at(pool.rng, i) = rng.s;
    at(pool.sample, i) = uint64_t(i);
    at(pool.depth, i) = cam.max_depth;
    queue[i] = i;
}

// Tail compaction.  Once every sample of a replica group has been started, finished slots stay empty and the queue is a
// thinning, near-random subset of the pool: the kernels then read and write whole 128-byte lines for the one or two live
// slots in them, and their time stops following the number of paths (k_wf_shade takes 1.3 ms for 18 M and for 4 M paths of a
// 22 M pool; a fifth of a small frame is spent this way, profiles/r03/tail_compaction.txt).  Whenever fewer than half the
// addressed slots are alive, this kernel copies the live paths - ray, weight, depth, generator, sample index; the hit record is
// recomputed by the next search - into slots 0 .. n-1 of the OTHER pool, which becomes the pool: n == capacity again, i.e. the
// identity order with unit-stride accesses.  A path does not care which slot holds it (its generator and its sample index
// travel with it), so frames do not change.
template <typename R>
__global__ void __launch_bounds__(256) k_wf_compact(WfPool<R> src, WfPool<R> dst, const uint32_t* __restrict__ queue, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = queue[i];
    at(dst.ox, i) = at(src.ox, s); at(dst.oy, i) = at(src.oy, s); at(dst.oz, i) = at(src.oz, s);
    at(dst.dx, i) = at(src.dx, s); at(dst.dy, i) = at(src.dy, s); at(dst.dz, i) = at(src.dz, s);
    at(dst.tr, i) = at(src.tr, s); at(dst.tg, i) = at(src.tg, s); at(dst.tb, i) = at(src.tb, s);
    at(dst.rng, i) = at(src.rng, s);
    at(dst.sample, i) = at(src.sample, s);
    at(dst.depth, i) = at(src.depth, s);
}

// Number of lanes below `lane` whose bit is set in `mask` (v_mbcnt_lo/hi).
RT_DEV uint32_t lane_prefix(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

// A single global word sustains only ~90 atomics/us on MI355X (MI355X_MICROARCH.md, "dequeue"), so
// one atomic per WAVE on a queue tail (16 M paths = 260 k waves = 3 ms per launch) was the
// bottleneck of every kernel here.  Queue traffic is therefore aggregated per WORKGROUP through
// LDS lists (chunked kernels: one global atomic per WF_CHUNK entries) and the persistent kernels
// reserve WF_BATCH entries per atomic.
constexpr uint32_t WF_CHUNK = 2048;  // queue entries handled by one workgroup of the chunked kernels (three LDS lists of this size in k_wf_shade)
constexpr uint32_t WF_BATCH = 256;   // queue entries a wave of a persistent kernel reserves at once

// Appends `value` of the lanes with `pred` to an LDS list: ballot + mbcnt prefix, one LDS atomic per wave.
template <typename T> RT_DEV void lds_append(bool pred, T value, T* list, uint32_t* count) {
    unsigned long long m = __ballot(pred);
    if (m) {
        int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if (int(threadIdx.x & 63u) == leader) base = atomicAdd(count, uint32_t(__popcll(m)));
        base = __shfl(base, leader);
        if (pred) list[base + lane_prefix(m)] = value;
    }
}

// Wave-level queue reader of the persistent kernels: hands out entries [cur, end) of a reserved batch.
struct WaveRange {
    uint32_t cur = 0, end = 0;
};
// Gives every idle lane (bit set in `idle`) a queue index if one is available; returns true in
// `take` lanes.  Sets `exhausted` when the queue has been handed out completely.
RT_DEV bool wave_fetch(WaveRange& r, unsigned long long idle, uint32_t* cursor, uint32_t n, bool& exhausted, uint32_t& my) {
    const uint32_t lane = threadIdx.x & 63u;
    if (r.cur >= r.end) {
        uint32_t base = 0;
        int leader = __ffsll((long long)idle) - 1;
        if (int(lane) == leader) base = atomicAdd(cursor, WF_BATCH);
        base = __shfl(base, leader);
        if (base >= n) {
            exhausted = true;
            return false;
        }
        r.cur = base;
        r.end = min(base + WF_BATCH, n);
    }
    uint32_t avail = r.end - r.cur;
    uint32_t rank = lane_prefix(idle);
    bool take = ((idle >> lane) & 1ull) && rank < avail;
    my = r.cur + rank;
    uint32_t n_idle = uint32_t(__popcll(idle));
    r.cur += n_idle < avail ? n_idle : avail;
    return take;
}

// ---------------------------------------------------------------------------------------------
// Intersect: world.test for every queued path.  Persistent waves; each lane is a small state
// machine (scene program counter + BVH traversal state), idle lanes are refilled from the queue.
// ---------------------------------------------------------------------------------------------
// VOL: the program contains volume ops (OP_VOL_*): the search then carries the path's RNG (Volume::test draws the
// free-flight distance in the middle of it, volume.rs:47) and a second search state for the boundary tests.
template <typename R, bool STATS, bool VOL>
__global__ void __launch_bounds__(256, VOL ? 2 : RT_ISECT_WAVES) k_wf_intersect(SceneView<R> sc, WfPool<R> pool, const uint32_t* __restrict__ queue,
                                                      WfCounters* __restrict__ ctr, DeviceCounters* counters, uint32_t refill_min) {
    extern __shared__ int lds_stack[];
    int* stack = lds_stack + threadIdx.x;
    const int stride = int(blockDim.x);
    const uint32_t n = ctr->n_in;
    const R t_lo_outer = R(0.001);
    R t_lo = t_lo_outer;     // changes only inside a volume's boundary searches
    VolFrames<R> vol;        // VOL: the callers' search states during boundary searches
    Rng rng;
    rng.s = 0;

    LaneCounters cnt;
    bool has = false;        // this lane holds a ray
    bool in_mesh = false;    // ... and is inside a mesh BVH
    bool exhausted = false;  // wave-uniform: the queue has been handed out completely
    WaveRange range;
    uint32_t slot = 0;
    Ray<R> wray{}, cur{};
    Best<R> best{};
    int32_t pc = 0;
    // mesh traversal state
    int32_t node = 0;
    int sp = 0;
    V3<R> inv{}, oi{};
    const BvhNode<R>* nodes = nullptr;
    const TriRec<R>* tris = nullptr;
    uint32_t tri_base = 0;
    bool hit_back = false;

    for (;;) {
        // ---- refill idle lanes ----
        unsigned long long idle = __ballot(!has);
        uint32_t n_idle = uint32_t(__popcll(idle));
        if (!exhausted && n_idle >= refill_min) {
            uint32_t my = 0;
            if (wave_fetch(range, idle, &ctr->cursor, n, exhausted, my)) {
                slot = n == pool.capacity ? my : queue[my];  // full pool: identity order (see k_wf_shade)
                wray = make_ray(mk<R>(at(pool.ox, slot), at(pool.oy, slot), at(pool.oz, slot)), mk<R>(at(pool.dx, slot), at(pool.dy, slot), at(pool.dz, slot)));
                cur = wray;
                best.t = Lim<R>::inf(); best.pc = -1; best.tri = -1; best.u = R(0); best.v = R(0);
                pc = 0;
                has = true;
                in_mesh = false;
                if constexpr (VOL) { rng.s = at(pool.rng, slot); t_lo = t_lo_outer; vol.depth = 0; }
                if (STATS) cnt.rays++;
            }
        }
        if (__ballot(has) == 0ull) {
            if (exhausted) break;
            continue;  // nothing held but the queue may still have entries: fetch again
        }
        // ---- scene program until the next mesh (or the end) ----
        if (has && !in_mesh) {
            for (;;) {
                const Op op = sc.ops[pc];
                if (op.type == OP_END) {
                    at(pool.ht, slot) = best.t; at(pool.hu, slot) = best.u; at(pool.hv, slot) = best.v;
                    at(pool.hpc, slot) = best.pc; at(pool.htri, slot) = best.tri;
                    if constexpr (VOL) at(pool.rng, slot) = rng.s;
                    has = false;
                    break;
                }
                if (op.type == OP_MESH) {
                    const MeshInst& mi = sc.meshes[op.arg];
                    nodes = sc.nodes + mi.node_base;
                    tris = sc.tris + mi.tri_base;
                    tri_base = mi.tri_base;
                    hit_back = (mi.flags & RT_MESH_HIT_BACK_FACES) != 0;
                    const R big = sizeof(R) == 8 ? R(1e150) : R(1e18);  // see mesh_traverse
                    inv = {fabs(cur.inv.x) > big ? copysign(big, cur.inv.x) : cur.inv.x,
                           fabs(cur.inv.y) > big ? copysign(big, cur.inv.y) : cur.inv.y,
                           fabs(cur.inv.z) > big ? copysign(big, cur.inv.z) : cur.inv.z};
                    oi = cur.o * inv;
                    node = 0;
                    sp = 0;
                    in_mesh = true;
                    if (STATS) cnt.mesh_rays++;
                    break;
                }
                switch (op.type) {
                    case OP_VOL_BEGIN:
                        if constexpr (VOL) vol_begin(vol, best, t_lo);
                        break;
                    case OP_VOL_MID:
                        if constexpr (VOL) {
                            if (vol_mid(vol, best, t_lo)) {
                                pc = op.skip;
                                continue;
                            }
                        }
                        break;
                    case OP_VOL_END:
                        if constexpr (VOL) vol_end(sc, vol, best, t_lo, cur, op, pc, rng);
                        break;
                    case OP_BOUNDS:
                        if (!test_bounding_box(sc.bounds[op.arg], cur, t_lo, best.t)) {
                            pc = op.skip;
                            continue;
                        }
                        break;
                    case OP_XFORM_PUSH: {
                        const Xform<R>& x = sc.xforms[op.arg];
                        cur = make_ray(xform_apply(x.inv, cur.o, R(1)), xform_apply(x.inv, cur.d, R(0)));
                        break;
                    }
                    case OP_XFORM_POP:
                        cur = ray_in_chain(sc, wray, op.chain);
                        break;
                    case OP_SPHERE: {
                        R t;
                        if (STATS) cnt.prim_tests++;
                        if (sphere_test<R, true>(sc.spheres[op.arg], cur, t_lo, best.t, t) && hit_takes_over(sc, t, op, best)) { best.t = t; best.pc = pc; }
                        break;
                    }
                    case OP_PLANE: {
                        R t, u, v;
                        if (STATS) cnt.prim_tests++;
                        if (plane_test<R, true>(sc.planes[op.arg], cur, t_lo, best.t, t, u, v) && hit_takes_over(sc, t, op, best)) { best.t = t; best.pc = pc; best.u = u; best.v = v; }
                        break;
                    }
                    case OP_SKY:
                        if (STATS) cnt.prim_tests++;
                        if (!(Lim<R>::inf() > best.t)) { best.t = Lim<R>::inf(); best.pc = pc; }
                        break;
                    case OP_SUN: {
                        if (STATS) cnt.prim_tests++;
                        const SunPrim<R>& s = sc.suns[op.arg];
                        V3<R> unit_dir = to_unit(cur.d);
                        if (!(fabs(dot(ld3(s.direction), unit_dir) - R(1)) > R(0.001)) && !(Lim<R>::max() >= best.t)) {
                            best.t = Lim<R>::max();
                            best.pc = pc;
                        }
                        break;
                    }
                    default: break;
                }
                pc++;
            }
        }
        // ---- BVH traversal: all lanes inside a mesh step together until too few remain ----
        while (true) {
            unsigned long long walking = __ballot(in_mesh);
            if (walking == 0ull) break;
            if (in_mesh) {
                bool pop = true;
                if (node >= 0) {
                    const BvhNode<R>& nd = nodes[node];
                    if (STATS) cnt.node_visits++;
                    R t0x = nd.lo0[0] * inv.x - oi.x, t1x = nd.hi0[0] * inv.x - oi.x;
                    R t0y = nd.lo0[1] * inv.y - oi.y, t1y = nd.hi0[1] * inv.y - oi.y;
                    R t0z = nd.lo0[2] * inv.z - oi.z, t1z = nd.hi0[2] * inv.z - oi.z;
                    R near0 = fmax(fmax(fmin(t0x, t1x), fmin(t0y, t1y)), fmax(fmin(t0z, t1z), t_lo));
                    R far0 = fmin(fmin(fmax(t0x, t1x), fmax(t0y, t1y)), fmin(fmax(t0z, t1z), best.t));
                    R s0x = nd.lo1[0] * inv.x - oi.x, s1x = nd.hi1[0] * inv.x - oi.x;
                    R s0y = nd.lo1[1] * inv.y - oi.y, s1y = nd.hi1[1] * inv.y - oi.y;
                    R s0z = nd.lo1[2] * inv.z - oi.z, s1z = nd.hi1[2] * inv.z - oi.z;
                    R near1 = fmax(fmax(fmin(s0x, s1x), fmin(s0y, s1y)), fmax(fmin(s0z, s1z), t_lo));
                    R far1 = fmin(fmin(fmax(s0x, s1x), fmax(s0y, s1y)), fmin(fmax(s0z, s1z), best.t));
                    int32_t c0 = nd.c0, c1 = nd.c1;
                    bool h0 = (near0 <= far0) && c0 != kEmptyChild;
                    bool h1 = (near1 <= far1) && c1 != kEmptyChild;
                    if (h0 && h1) {
                        bool first0 = near0 <= near1;
                        stack[sp * stride] = first0 ? c1 : c0;
                        sp++;
                        node = first0 ? c0 : c1;
                        pop = false;
                    } else if (h0) { node = c0; pop = false; }
                    else if (h1) { node = c1; pop = false; }
                } else {
                    uint32_t code = uint32_t(~node);
                    uint32_t first = code >> 3, count = (code & 7u) + 1u;
                    for (uint32_t i = 0; i < count; i++) {
                        const TriRec<R>& tr = tris[first + i];
                        if (STATS) cnt.tri_tests++;
                        V3<R> edge1 = ld3(tr.e1), edge2 = ld3(tr.e2);
                        V3<R> ray_x_edge2 = cross(cur.d, edge2);
                        R det = dot(edge1, ray_x_edge2);
                        R dd = hit_back ? fabs(det) : det;
                        if (dd < Lim<R>::eps()) continue;
                        R inv_det = R(1) / det;
                        V3<R> b = cur.o - ld3(tr.v0);
                        R u = dot(b, ray_x_edge2) * inv_det;
                        if (u < R(0) || u > R(1)) continue;
                        V3<R> b_x_edge1 = cross(b, edge1);
                        R v = dot(cur.d, b_x_edge1) * inv_det;
                        if (v < R(0) || u + v > R(1)) continue;
                        R t = dot(edge2, b_x_edge1) * inv_det;
                        if (t <= t_lo || best.t <= t) continue;
                        best.t = t; best.pc = pc; best.tri = int32_t(tri_base + first + i); best.u = u; best.v = v;
                    }
                }
                if (pop) {
                    if (sp == 0) { in_mesh = false; pc++; }
                    else { sp--; node = stack[sp * stride]; }
                }
            }
            // leave the traversal loop when enough lanes could do other work (finish / refill)
            uint32_t still = uint32_t(__popcll(__ballot(in_mesh)));
            if (still == 0u) break;
            if (!exhausted && 64u - still >= refill_min) break;
        }
    }
    if (STATS) {
        atomicAdd(&counters->rays, (unsigned long long)cnt.rays);
        atomicAdd(&counters->mesh_rays, (unsigned long long)cnt.mesh_rays);
        atomicAdd(&counters->node_visits, (unsigned long long)cnt.node_visits);
        atomicAdd(&counters->tri_tests, (unsigned long long)cnt.tri_tests);
        atomicAdd(&counters->prim_tests, (unsigned long long)cnt.prim_tests);
    }
}

// ---------------------------------------------------------------------------------------------
// Split intersect (every scene without volumes; any number of mesh instances):
//   k_wf_prims  every lane runs the same scene program over spheres / quads / sky / sun (uniform
//               control flow); the mesh ops are deferred: paths whose object-space ray enters a
//               mesh's root box with the interval left by the primitives visited so far are appended
//               (ballot + prefix sum) to the mesh queue, once.
//   k_wf_mesh   persistent waves that do nothing but BVH traversal, "while-while": all lanes
//               descend inner nodes until each holds a leaf, then all test triangles; idle lanes
//               are refilled from the mesh queue.  A lane serves the mesh ops of its path one after the
//               other, in program order (SceneView::mesh_ops), with the interval the earlier ones left.
// Closest-hit semantics are those of the in-order program: the nearest t wins and, at exactly equal
// t, the op that comes first in the reference's visiting order (its tests use strict `t < closest`).
// ---------------------------------------------------------------------------------------------
// f32 value that is certainly >= x (x finite or +inf): round to nearest, then add a relative margin.
RT_DEV float f32_at_least(double x) {
    float f = float(x);
    return f + fabsf(f) * 9.5367431640625e-7f + 1e-30f;  // 2^-20 relative
}
RT_DEV float f32_at_least(float x) { return x + fabsf(x) * 9.5367431640625e-7f + 1e-30f; }

typedef __attribute__((address_space(3))) unsigned long long LdsU64;

// ---------------------------------------------------------------------------------------------
// Re-built primitive groups (OP_GROUP): closest hit of the ray with the group's spheres / quads through the group's 4-wide
// quantised BVH - the node format, the conservative f32 slab test on a ray clipped to the group's box, the nearest-first
// order and the (child, entry distance) stack entries culled on pop are those of k_wf_mesh; the leaves hold primitives
// instead of triangles and are tested by the lane itself with the exact tests of the scene program (sphere.rs:40-62,
// plane.rs:66-89, ties by rank).  The default scene's 455-sphere field costs a ray ~5 node steps and ~3 sphere tests this
// way, against ~50 serial ops of the skip-pointer form (each op a dependent fetch: 101 of the scene's 177 ms per step).
// ---------------------------------------------------------------------------------------------
struct GroupCtx {
    const BvhNode4q* nodes;  // SceneView::group_nodes, or their copy in LDS
    LdsU64* stack;           // this lane's column of the workgroup's LDS stack: entry k at stack[k * 256]
};

template <typename R, bool STATS>
RT_DEV void group_search(const SceneView<R>& sc, const GroupCtx& gc, const GroupRec<R>& g, const Ray<R>& cur, R t_lo, Best<R>& best, LaneCounters& cnt) {
    const R big = sizeof(R) == 8 ? R(1e150) : R(1e18);
    const V3<R> inv = {fabs(cur.inv.x) > big ? copysign(big, cur.inv.x) : cur.inv.x,
                       fabs(cur.inv.y) > big ? copysign(big, cur.inv.y) : cur.inv.y,
                       fabs(cur.inv.z) > big ? copysign(big, cur.inv.z) : cur.inv.z};
    // entry into the group's box (>= 0) and exit; the culling origin is o + d * t_shift, so that its f32 image is no larger than the box
    const R e0x = (g.lo[0] - cur.o.x) * inv.x, e1x = (g.hi[0] - cur.o.x) * inv.x;
    const R e0y = (g.lo[1] - cur.o.y) * inv.y, e1y = (g.hi[1] - cur.o.y) * inv.y;
    const R e0z = (g.lo[2] - cur.o.z) * inv.z, e1z = (g.hi[2] - cur.o.z) * inv.z;
    R t_shift = fmax(fmax(fmin(e0x, e1x), fmin(e0y, e1y)), fmax(fmin(e0z, e1z), R(0)));
    const R t_exit = fmin(fmin(fmax(e0x, e1x), fmax(e0y, e1y)), fmax(e0z, e1z));
    const R eps = Lim<R>::eps() * R(16);  // a miss only if it is one with a few ulps of slack (a NaN compares false: the group is searched)
    if ((t_shift - fabs(t_shift) * eps > t_exit + fabs(t_exit) * eps) || (t_shift - fabs(t_shift) * eps > best.t)) return;
    if (!(t_shift < Lim<R>::inf())) t_shift = R(0);
    const V3<R> oc = cur.o + cur.d * t_shift;
    const float big32 = 1e18f;
    float ivx = 1.0f / float(cur.d.x), ivy = 1.0f / float(cur.d.y), ivz = 1.0f / float(cur.d.z);
    ivx = fabsf(ivx) > big32 ? copysignf(big32, ivx) : ivx;
    ivy = fabsf(ivy) > big32 ? copysignf(big32, ivy) : ivy;
    ivz = fabsf(ivz) > big32 ? copysignf(big32, ivz) : ivz;
    const float oix = float(oc.x) * ivx, oiy = float(oc.y) * ivy, oiz = float(oc.z) * ivz;
    const bool negx = ivx < 0.0f, negy = ivy < 0.0f, negz = ivz < 0.0f;
    float tmax32 = f32_at_least(best.t - t_shift);
    int32_t node = int32_t(g.root);
    int sp = 0;
    for (;;) {
        bool pop = false;
        if (node >= 0) {
            if (STATS) cnt.node_visits++;
            const float miss = __builtin_huge_valf();
            const uint4* nd = reinterpret_cast<const uint4*>(gc.nodes + node);
            const uint4 h0 = nd[0], h1 = nd[1], h2 = nd[2];
            const int4 cc = *reinterpret_cast<const int4*>(nd + 3);
            float nr[4];
            int32_t ch[4] = {cc.x, cc.y, cc.z, cc.w};
            const float ax = __uint_as_float(h0.w) * ivx, ay = __uint_as_float(h1.x) * ivy, az = __uint_as_float(h1.y) * ivz;
            const float bx = fmaf(__uint_as_float(h0.x), ivx, -oix), by = fmaf(__uint_as_float(h0.y), ivy, -oiy), bz = fmaf(__uint_as_float(h0.z), ivz, -oiz);
            const uint32_t qnx = negx ? h2.y : h1.z, qfx = negx ? h1.z : h2.y;
            const uint32_t qny = negy ? h2.z : h1.w, qfy = negy ? h1.w : h2.z;
            const uint32_t qnz = negz ? h2.w : h2.x, qfz = negz ? h2.x : h2.w;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float nxk = float((qnx >> (8 * k)) & 0xFFu), nyk = float((qny >> (8 * k)) & 0xFFu), nzk = float((qnz >> (8 * k)) & 0xFFu);
                const float fxk = float((qfx >> (8 * k)) & 0xFFu), fyk = float((qfy >> (8 * k)) & 0xFFu), fzk = float((qfz >> (8 * k)) & 0xFFu);
                const float tn = fmaxf(fmaxf(fmaf(nxk, ax, bx), fmaf(nyk, ay, by)), fmaxf(fmaf(nzk, az, bz), 0.0f));
                const float tf = fminf(fminf(fmaf(fxk, ax, bx), fmaf(fyk, ay, by)), fminf(fmaf(fzk, az, bz), tmax32));
                nr[k] = ((tn <= tf) && ch[k] != kEmptyChild) ? tn : miss;
            }
#define RT_CE(a, b)                                                   \
    if (nr[a] > nr[b]) {                                              \
        float tn_ = nr[a]; nr[a] = nr[b]; nr[b] = tn_;                \
        int32_t tc_ = ch[a]; ch[a] = ch[b]; ch[b] = tc_;              \
    }
            RT_CE(0, 1) RT_CE(2, 3) RT_CE(0, 2) RT_CE(1, 3) RT_CE(1, 2)
#undef RT_CE
            if (nr[0] < miss) {
                // farthest first, so that the nearest remaining child is popped first
                if (nr[3] < miss) { gc.stack[sp * 256] = (static_cast<unsigned long long>(__float_as_uint(nr[3])) << 32) | uint32_t(ch[3]); sp++; }
                if (nr[2] < miss) { gc.stack[sp * 256] = (static_cast<unsigned long long>(__float_as_uint(nr[2])) << 32) | uint32_t(ch[2]); sp++; }
                if (nr[1] < miss) { gc.stack[sp * 256] = (static_cast<unsigned long long>(__float_as_uint(nr[1])) << 32) | uint32_t(ch[1]); sp++; }
                node = ch[0];
            } else {
                pop = true;
            }
        } else {
            const uint32_t code = uint32_t(~node);
            const uint32_t first = code >> 3, count = (code & 7u) + 1u;
            for (uint32_t i = 0; i < count; i++) {
                const GroupPrimRef ref = sc.group_prims[first + i];
                bool reachable = true;  // the reference's ancestor boxes that do not contain this primitive still stand in front of it
                for (int32_t k = 0; k < ref.guard_count; k++)
                    reachable = reachable && test_bounding_box(sc.bounds[sc.group_guards[ref.guard_first + k]], cur, t_lo, best.t);
                if (!reachable) continue;
                const Op pop_ = sc.ops[ref.pc];
                if (STATS) cnt.prim_tests++;
                if (pop_.type == OP_SPHERE) {
                    R t;
                    if (sphere_test<R, true>(sc.spheres[pop_.arg], cur, t_lo, best.t, t) && hit_takes_over(sc, t, pop_, best)) { best.t = t; best.pc = ref.pc; }
                } else {
                    R t, u, v;
                    if (plane_test<R, true>(sc.planes[pop_.arg], cur, t_lo, best.t, t, u, v) && hit_takes_over(sc, t, pop_, best)) { best.t = t; best.pc = ref.pc; best.u = u; best.v = v; }
                }
            }
            tmax32 = f32_at_least(best.t - t_shift);
            pop = true;
        }
        if (pop) {
            for (;;) {
                if (sp == 0) return;
                sp--;
                const unsigned long long e = gc.stack[sp * 256];
                if (__uint_as_float(uint32_t(e >> 32)) <= tmax32) {
                    node = int32_t(uint32_t(e));
                    break;
                }
            }
        }
    }
}

// The scene program over everything except the deferred mesh ops: closest hit of `wray` with the spheres /
// quads / sky / sun in `best`; returns true if the ray also has to visit a mesh (its object-space ray enters a
// mesh's box inside the interval the primitives visited before it left).  Every lane walks the same program.
// VOL: the program contains volumes (OP_VOL_*: two boundary searches, then the free-flight draw from the path's RNG,
// volume.rs:33-71, as in world_test / k_wf_intersect) whose boundaries are made of spheres and quads, and every mesh op
// comes AFTER the last volume: a volume's draw depends on the closest hit so far (volume.rs:40-43), so a mesh in front
// of it cannot be deferred (such scenes, and meshes inside a boundary, use the combined kernel).
// GROUPS: OP_GROUP ops are served by group_search (their op form behind them is skipped); otherwise they are no-ops.
template <typename R, bool STATS, bool VOL = false, bool GROUPS = false>
RT_DEV bool prims_search(const SceneView<R>& sc, const Ray<R>& wray, Best<R>& best, LaneCounters& cnt, Rng* rng = nullptr, const GroupCtx* gc = nullptr) {
    const R t_lo_outer = R(0.001);
    R t_lo = t_lo_outer;     // changes only inside a volume's boundary searches
    VolFrames<R> vol;        // VOL: the callers' search states during boundary searches
    Ray<R> cur = wray;
    best.t = Lim<R>::inf(); best.pc = -1; best.tri = -1; best.u = R(0); best.v = R(0);
    bool to_mesh = false;
    int32_t pc = 0;
    if (STATS) cnt.rays++;
    for (;;) {
        const Op op = sc.ops[pc];
        if (op.type == OP_END) break;
        switch (op.type) {
            case OP_VOL_BEGIN:
                if constexpr (VOL) vol_begin(vol, best, t_lo);
                break;
            case OP_VOL_MID:
                if constexpr (VOL) {
                    if (vol_mid(vol, best, t_lo)) {
                        pc = op.skip;
                        continue;
                    }
                }
                break;
            case OP_VOL_END:
                if constexpr (VOL) vol_end(sc, vol, best, t_lo, cur, op, pc, *rng);
                break;
            case OP_GROUP:
                if constexpr (GROUPS) {
                    group_search<R, STATS>(sc, *gc, sc.groups[op.arg], cur, t_lo, best, cnt);
                    pc = op.skip;
                    continue;
                }
                break;
            case OP_BOUNDS:
                if (!test_bounding_box(sc.bounds[op.arg], cur, t_lo, best.t)) {
                    pc = op.skip;
                    continue;
                }
                break;
            case OP_XFORM_PUSH: {
                const Xform<R>& x = sc.xforms[op.arg];
                cur = make_ray(xform_apply(x.inv, cur.o, R(1)), xform_apply(x.inv, cur.d, R(0)));
                break;
            }
            case OP_XFORM_POP:
                cur = ray_in_chain(sc, wray, op.chain);
                break;
            case OP_SPHERE: {
                R t;
                if (STATS) cnt.prim_tests++;
                if (sphere_test<R, true>(sc.spheres[op.arg], cur, t_lo, best.t, t) && hit_takes_over(sc, t, op, best)) { best.t = t; best.pc = pc; }
                break;
            }
            case OP_PLANE: {
                R t, u, v;
                if (STATS) cnt.prim_tests++;
                if (plane_test<R, true>(sc.planes[op.arg], cur, t_lo, best.t, t, u, v) && hit_takes_over(sc, t, op, best)) { best.t = t; best.pc = pc; best.u = u; best.v = v; }
                break;
            }
            case OP_MESH: {  // deferred to k_wf_mesh
                // Does the ray enter the mesh's box inside (t_lo, best.t]?  Decided HERE, with the closest hit so far (ops behind
                // the mesh can only shorten the interval, and the test only culls: conservative), so that no copy of the
                // object-space ray has to stay alive to the end of the program.
                const Bounds<R>& rb = sc.mesh_bounds[op.arg];
                const R big = sizeof(R) == 8 ? R(1e150) : R(1e18);
                V3<R> inv = {fabs(cur.inv.x) > big ? copysign(big, cur.inv.x) : cur.inv.x,
                             fabs(cur.inv.y) > big ? copysign(big, cur.inv.y) : cur.inv.y,
                             fabs(cur.inv.z) > big ? copysign(big, cur.inv.z) : cur.inv.z};
                // a few ulps of slack on the box: this test must never be stricter than the traversal
                const R eps = Lim<R>::eps() * R(16);
                R t0x = (rb.lo[0] - fabs(rb.lo[0]) * eps - cur.o.x) * inv.x, t1x = (rb.hi[0] + fabs(rb.hi[0]) * eps - cur.o.x) * inv.x;
                R t0y = (rb.lo[1] - fabs(rb.lo[1]) * eps - cur.o.y) * inv.y, t1y = (rb.hi[1] + fabs(rb.hi[1]) * eps - cur.o.y) * inv.y;
                R t0z = (rb.lo[2] - fabs(rb.lo[2]) * eps - cur.o.z) * inv.z, t1z = (rb.hi[2] + fabs(rb.hi[2]) * eps - cur.o.z) * inv.z;
                R tn = fmax(fmax(fmin(t0x, t1x), fmin(t0y, t1y)), fmax(fmin(t0z, t1z), t_lo));
                R tf = fmin(fmin(fmax(t0x, t1x), fmax(t0y, t1y)), fmin(fmax(t0z, t1z), best.t));
                tf = tf + fabs(tf) * eps;
                to_mesh = to_mesh || ((tn <= tf) && rb.lo[0] <= rb.hi[0]);
                break;
            }
            case OP_SKY:
                if (STATS) cnt.prim_tests++;
                if (!(Lim<R>::inf() > best.t)) { best.t = Lim<R>::inf(); best.pc = pc; }
                break;
            case OP_SUN: {
                if (STATS) cnt.prim_tests++;
                const SunPrim<R>& s = sc.suns[op.arg];
                V3<R> unit_dir = to_unit(cur.d);
                if (!(fabs(dot(ld3(s.direction), unit_dir) - R(1)) > R(0.001)) && !(Lim<R>::max() >= best.t)) {
                    best.t = Lim<R>::max();
                    best.pc = pc;
                }
                break;
            }
            default: break;
        }
        pc++;
    }
    return to_mesh;
}

#ifndef RT_PRIMS_WAVES
#define RT_PRIMS_WAVES 5  // 95 VGPRs without scratch since the mesh-box test moved to the mesh op (round 2); 4 waves before: 171 -> 157 ms per step
#endif
// LDS: 0 = tables in global memory, 1 = all small tables staged in LDS, 2 = a prefix of them (see scene_tables_to_lds)
// GROUPS: the scene has re-built primitive groups (OP_GROUP): their BVH nodes and a per-lane traversal stack of `group_levels`
// entries live in LDS behind the mesh list.
template <typename R, bool STATS, int LDS, bool VOL, bool GROUPS>
__global__ void __launch_bounds__(256, VOL ? 2 : (GROUPS ? 3 : RT_PRIMS_WAVES)) k_wf_prims(SceneView<R> sc_g, WfPool<R> pool, const uint32_t* __restrict__ queue,
                                                  uint32_t* __restrict__ mesh_queue, WfCounters* __restrict__ ctr,
                                                  DeviceCounters* counters, uint32_t staged, uint32_t group_levels) {
    extern __shared__ __align__(16) char lds_raw[];
    uint32_t* mesh_list = reinterpret_cast<uint32_t*>(lds_raw);  // [WF_CHUNK]
    uint32_t* lc = mesh_list + WF_CHUNK;                        // [0] list length, [1] queue base
    char* tables = reinterpret_cast<char*>(lc + 4);
    GroupCtx gc{};
    if constexpr (GROUPS) {
        gc.stack = (LdsU64*)(reinterpret_cast<unsigned long long*>(tables) + threadIdx.x);
        tables += size_t(group_levels) * 256 * 8;
        uint4* dst = reinterpret_cast<uint4*>(tables);
        const uint4* src = reinterpret_cast<const uint4*>(sc_g.group_nodes);
        const uint32_t n16 = uint32_t(sc_g.n_group_nodes) * uint32_t(sizeof(BvhNode4q) / 16);
        for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = src[i];
        gc.nodes = reinterpret_cast<const BvhNode4q*>(dst);
        tables += size_t(n16) * 16;
    }
    if (threadIdx.x < 4) lc[threadIdx.x] = 0;
    SceneView<R> sc = sc_g;
    if constexpr (LDS == 1) sc = scene_tables_to_lds<R, true>(sc_g, sc_g.lay, sc_g.small_blob, tables, staged);
    else if constexpr (LDS == 2) sc = scene_tables_to_lds<R, false>(sc_g, sc_g.lay, sc_g.small_blob, tables, staged);
    else __syncthreads();
    const uint32_t n = ctr->n_in;
    const bool full = n == pool.capacity;
    const uint32_t begin = blockIdx.x * WF_CHUNK;
    const uint32_t end = min(n, begin + WF_CHUNK);
    LaneCounters cnt;
    for (uint32_t base = begin; base < end; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        bool to_mesh = false;
        uint32_t slot = 0;
        if (i < end) {
            slot = full ? i : queue[i];
            const Ray<R> wray = make_ray(mk<R>(at(pool.ox, slot), at(pool.oy, slot), at(pool.oz, slot)), mk<R>(at(pool.dx, slot), at(pool.dy, slot), at(pool.dz, slot)));
            Best<R> best;
            if constexpr (VOL) {  // Volume::test draws the free-flight distance from the path's stream (volume.rs:47)
                Rng rng;
                rng.s = at(pool.rng, slot);
                to_mesh = prims_search<R, STATS, true, false>(sc, wray, best, cnt, &rng);
                at(pool.rng, slot) = rng.s;
            } else {
                to_mesh = prims_search<R, STATS, false, GROUPS>(sc, wray, best, cnt, nullptr, &gc);
            }
            at(pool.ht, slot) = best.t; at(pool.hu, slot) = best.u; at(pool.hv, slot) = best.v;
            at(pool.hpc, slot) = best.pc; at(pool.htri, slot) = best.tri;
        }
        lds_append(to_mesh, slot, mesh_list, &lc[0]);
    }
    __syncthreads();
    const uint32_t n_list = lc[0];
    if (threadIdx.x == 0 && n_list) lc[1] = atomicAdd(&ctr->n_mesh, n_list);  // ONE global atomic per workgroup
    __syncthreads();
    const uint32_t qb = lc[1];
    for (uint32_t j = threadIdx.x; j < n_list; j += blockDim.x) mesh_queue[qb + j] = mesh_list[j];
    if (STATS) {
        uint32_t rays = cnt.rays, prims = cnt.prim_tests;
        for (int off = 32; off > 0; off >>= 1) { rays += __shfl_down(rays, off); prims += __shfl_down(prims, off); }
        if ((threadIdx.x & 63u) == 0 && rays) {
            atomicAdd(&counters->rays, (unsigned long long)rays);
            atomicAdd(&counters->prim_tests, (unsigned long long)prims);
        }
    }
}

// Per-lane traversal stack of k_wf_mesh: entries are (child reference, f32 entry distance of its box).
// The first `lds_levels` levels live in LDS (`[level][lane]`, conflict-free 8-B accesses), deeper levels in
// a private global spill area (`[level][global lane]`, coalesced).  A shallow LDS part keeps 4 blocks per CU
// resident; the spill part is touched by a few percent of the pushes (worst case = BVH4 max_stack).
// Measured (headline scene): 12 LDS levels 742 ms/step, 8: +5 %, 24: +20 %; 5 waves/SIMD spills and is slower.
// The LDS part is addressed through an LDS-qualified pointer: with plain (generic) pointers hipcc merges the two stores of
// `put` into one store through a selected pointer and then fails in the backend ("Illegal instruction detected:
// V_CMP_NE_U32 0, src_shared_base") once a node step has more than a few puts (the 8-wide node has seven).
struct MeshStack {
    LdsU64* lds;       // + threadIdx.x
    uint2* spill;      // + global lane
    int lds_levels;
    uint32_t spill_stride;
    RT_DEV void put(int sp, int32_t child, float dist) const {
        if (sp < lds_levels) lds[sp * 256] = (static_cast<unsigned long long>(__float_as_uint(dist)) << 32) | uint32_t(child);
        else spill[size_t(sp - lds_levels) * spill_stride] = make_uint2(uint32_t(child), __float_as_uint(dist));
    }
    RT_DEV uint2 get(int sp) const {
        if (sp < lds_levels) {
            unsigned long long e = lds[sp * 256];
            return make_uint2(uint32_t(e), uint32_t(e >> 32));
        }
        return spill[size_t(sp - lds_levels) * spill_stride];
    }
};

template <typename R> constexpr uint32_t kMeshWaveLds = 1024u + 3u * 64u * uint32_t(sizeof(R));  // per wave, see k_wf_mesh

// NODE: 0 = 4-wide f32 nodes (BvhNode4f, 128 B), 1 = 4-wide quantised nodes (BvhNode4q, 64 B)
// MULTI: the program has more than one mesh op: a lane serves the mesh ops of its path one after the other (per-lane mesh
// cursor, one op entered per trip with wave-uniform records).  false: the one mesh op's record sits in SGPRs for the whole
// kernel and none of that bookkeeping exists - the general form costs the single-mesh headline scene 16 % more vector
// instructions and 4 % of the kernel's time (profiles/r03/ab/multi_mesh_kernel.txt), so both are kept.
template <typename R, bool STATS, int NODE, bool MULTI>
__global__ void __launch_bounds__(256, RT_MESH_WAVES) k_wf_mesh(SceneView<R> sc, WfPool<R> pool, const uint32_t* __restrict__ mesh_queue,
                                                                 WfCounters* __restrict__ ctr, DeviceCounters* counters,
                                                                 uint32_t refill_min, uint32_t inner_min,
                                                                 uint2* __restrict__ spill, int lds_levels,
                                                                 const uint32_t* __restrict__ n_ptr, uint32_t* __restrict__ cursor_ptr) {
    // n_ptr / cursor_ptr: length and hand-out cursor of `mesh_queue` (&ctr->n_mesh / &ctr->cursor)
    extern __shared__ uint2 lds_stack2[];
    MeshStack stk;
    stk.lds = (LdsU64*)(lds_stack2 + threadIdx.x);
    // wave-private LDS behind the stack: pair table (512 x u16) + one result slot per lane (t, u, v)
    const uint32_t lane = threadIdx.x & 63u;
    char* wave_area = reinterpret_cast<char*>(lds_stack2 + size_t(lds_levels) * 256) + (threadIdx.x >> 6) * kMeshWaveLds<R>;
    uint16_t* pair_tbl = reinterpret_cast<uint16_t*>(wave_area);
    R* res_t = reinterpret_cast<R*>(wave_area + 1024);
    R* res_u = res_t + 64;
    R* res_v = res_u + 64;
    stk.spill = spill + (size_t(blockIdx.x) * blockDim.x + threadIdx.x);
    stk.lds_levels = lds_levels;
    stk.spill_stride = gridDim.x * blockDim.x;
    const uint32_t n = *n_ptr;
    const R t_lo = R(0.001);
    const BvhNode4f* nodes = sc.nodes4;     // child references and leaf triangle slots are absolute: one table for every mesh
    const BvhNode4q* nodesq = sc.nodes4q;
    const TriRec<R>* tris = sc.tris;
    const uint32_t n_mesh_ops = uint32_t(sc.n_mesh_ops);
    const R big = sizeof(R) == 8 ? R(1e150) : R(1e18);
    const MeshOpRec<R>* mrecs = sc.mesh_op_recs;
    // Loads record m (wave-uniform index) with scalar loads, field by field (no copy constructor from an address space).
    auto load_rec = [&](uint32_t m, MeshOpRec<R>& rb) {
        const auto* rec = as_const_mem(mrecs) + m;
        rb.pc = rec->pc; rb.chain = rec->chain; rb.node4_base = rec->node4_base; rb.flags = rec->flags;
#pragma unroll
        for (int a = 0; a < 3; a++) { rb.lo[a] = rec->lo[a]; rb.hi[a] = rec->hi[a]; }
#pragma unroll
        for (int k = 0; k < 12; k++) rb.inv[k] = rec->inv[k];
    };
    MeshOpRec<R> rb0;  // !MULTI: the one mesh op, for the whole kernel
    if constexpr (!MULTI) load_rec(0u, rb0);
    LaneCounters cnt;
    uint32_t w_node = 0, w_tri = 0, w_refill = 0, l_refill = 0, l_culled = 0;  // STATS: see DeviceCounters
    bool has = false;        // this lane is inside a mesh's BVH
    bool pending = false;    // this lane holds a path whose mesh op `mcur` has not been entered yet
    bool exhausted = false;
    WaveRange range;
    uint32_t slot = 0;
    // bits 0-14: index into sc.mesh_ops of the mesh being traversed / tried next; bits 15-29: 1 + index of the mesh op that holds
    // the closest triangle found for this path so far (0: none); bit 31: the current mesh hits back faces
    uint32_t mcur = 0;
    V3<R> o{}, d{};          // object-space ray, exact: used by the triangle tests
    R t_max = R(0), hit_u = R(0), hit_v = R(0), t_shift = R(0);
    int32_t hit_tri = -1;
    // f32 culling ray: origin moved onto the mesh box (so |origin| <= mesh extent), t measured from there
    float ivx = 0.f, ivy = 0.f, ivz = 0.f, oix = 0.f, oiy = 0.f, oiz = 0.f, tmax32 = 0.f;
    int32_t node = 0;        // >= 0 inner node, < 0 leaf
    int sp = 0;

    // The lane has finished every mesh of its path: the closest triangle, if one beat the other primitives' hit, is the path's hit.
    auto finish_path = [&]() {
        if constexpr (MULTI) {
            const uint32_t hm = (mcur >> 15) & 0x7FFFu;
            if (hm != 0u) {
                at(pool.ht, slot) = t_max; at(pool.hu, slot) = hit_u; at(pool.hv, slot) = hit_v;
                at(pool.hpc, slot) = sc.mesh_ops[hm - 1u]; at(pool.htri, slot) = hit_tri;
            }
        } else {
            if (hit_tri >= 0) {  // a triangle beat the other primitives' hit
                at(pool.ht, slot) = t_max; at(pool.hu, slot) = hit_u; at(pool.hv, slot) = hit_v;
                at(pool.hpc, slot) = rb0.pc; at(pool.htri, slot) = hit_tri;
            }
        }
    };
    // Pops entries until one whose box can still contain a closer hit is found (entry distance <= current
    // bound); a lane whose stack runs empty has finished this mesh and goes on to the path's next mesh op.
    // (The end-of-mesh bookkeeping stands BEHIND the pop loop: inside it, it was if-converted into every trip of the loop -
    // 3.3 G culled pops per headline step - and cost the kernel 12 % more vector instructions than the single-mesh form.)
    auto pop_next = [&]() {
        if constexpr (MULTI) {
            bool found_entry = false;
            while (sp > 0) {
                sp--;
                const uint2 e = stk.get(sp);
                if (__uint_as_float(e.y) <= tmax32) {
                    node = int32_t(e.x);
                    found_entry = true;
                    break;
                }
                if (STATS) l_culled++;
            }
            if (!found_entry) {
                has = false;
                mcur = (mcur & 0x7FFFFFFFu) + 1u;
                if ((mcur & 0x7FFFu) < n_mesh_ops) pending = true;
                else finish_path();
            }
        } else {
            for (;;) {
                if (sp == 0) {
                    has = false;
                    finish_path();
                    return;
                }
                sp--;
                const uint2 e = stk.get(sp);
                if (__uint_as_float(e.y) <= tmax32) {
                    node = int32_t(e.x);
                    return;
                }
                if (STATS) l_culled++;
            }
        }
    };
    // Enters the mesh of record `rb` with this lane's path: object-space ray, search bound, entry into the mesh's box, f32
    // culling ray.  Returns false if the ray misses the box inside its interval (MULTI only: k_wf_prims has already asked that
    // question for the single mesh).
    auto enter_mesh = [&](const MeshOpRec<R>& rb) -> bool {
        const int32_t mpc = rb.pc;
        Ray<R> wray = make_ray(mk<R>(at(pool.ox, slot), at(pool.oy, slot), at(pool.oz, slot)), mk<R>(at(pool.dx, slot), at(pool.dy, slot), at(pool.dz, slot)));
        Ray<R> ray;
        const uint32_t n_chain = rb.flags >> 16;
        if (n_chain == 0u) ray = wray;
        else if (n_chain == 1u) ray = make_ray(xform_apply(rb.inv, wray.o, R(1)), xform_apply(rb.inv, wray.d, R(0)));  // transform.rs:124-127
        else ray = ray_in_chain_uniform(sc, wray, rb.chain);
        o = ray.o;
        d = ray.d;
        V3<R> inv = mk<R>(R(fabs(ray.inv.x) > big ? copysign(big, ray.inv.x) : ray.inv.x),
                          R(fabs(ray.inv.y) > big ? copysign(big, ray.inv.y) : ray.inv.y),
                          R(fabs(ray.inv.z) > big ? copysign(big, ray.inv.z) : ray.inv.z));
        // The other primitives' closest hit bounds the search.  At exactly equal t the op that comes first in
        // program order wins: if that is this mesh, t == bound must be accepted.  A triangle of an EARLIER mesh
        // always wins a tie (strict bound, like the reference's shrinking interval: list.rs:58-74).
        const R bound = at(pool.ht, slot);
        const int32_t bpc = at(pool.hpc, slot);
        const R excl = (bpc > mpc && bound < Lim<R>::inf()) ? nextafter(bound, Lim<R>::inf()) : bound;
        if constexpr (MULTI) t_max = ((mcur >> 15) & 0x7FFFu) != 0u ? fmin(t_max, excl) : excl;
        else t_max = excl;
        // entry into the mesh box (>= 0) and exit; the culling origin is o + d * t_shift
        R e0x = (rb.lo[0] - o.x) * inv.x, e1x = (rb.hi[0] - o.x) * inv.x;
        R e0y = (rb.lo[1] - o.y) * inv.y, e1y = (rb.hi[1] - o.y) * inv.y;
        R e0z = (rb.lo[2] - o.z) * inv.z, e1z = (rb.hi[2] - o.z) * inv.z;
        t_shift = fmax(fmax(fmin(e0x, e1x), fmin(e0y, e1y)), fmax(fmin(e0z, e1z), R(0)));
        if constexpr (MULTI) {
            R t_exit = fmin(fmin(fmax(e0x, e1x), fmax(e0y, e1y)), fmax(e0z, e1z));
            // a miss only if it is one with a few ulps of slack on both ends (a NaN compares false: the mesh is entered)
            const R eps = Lim<R>::eps() * R(16);
            if ((t_shift - fabs(t_shift) * eps > t_exit + fabs(t_exit) * eps) || (t_shift - fabs(t_shift) * eps > t_max) || !(rb.lo[0] <= rb.hi[0])) return false;
        }
        if (!(t_shift < Lim<R>::inf())) t_shift = R(0);
        V3<R> oc = o + d * t_shift;
        const float big32 = 1e18f;
        float dx32 = float(d.x), dy32 = float(d.y), dz32 = float(d.z);
        ivx = 1.0f / dx32; ivy = 1.0f / dy32; ivz = 1.0f / dz32;
        ivx = fabsf(ivx) > big32 ? copysignf(big32, ivx) : ivx;
        ivy = fabsf(ivy) > big32 ? copysignf(big32, ivy) : ivy;
        ivz = fabsf(ivz) > big32 ? copysignf(big32, ivz) : ivz;
        oix = float(oc.x) * ivx; oiy = float(oc.y) * ivy; oiz = float(oc.z) * ivz;
        tmax32 = f32_at_least(t_max - t_shift);
        node = int32_t(rb.node4_base);
        sp = 0;
        return true;
    };

    for (;;) {
        if constexpr (MULTI) {
            // ---- refill: lanes without a path take a queue entry; lanes between two meshes of their path enter the next one ----
            const unsigned long long idle = __ballot(!has);
            const unsigned long long waiting = __ballot(pending);
            if ((uint32_t(__popcll(idle)) >= refill_min && (!exhausted || waiting != 0ull)) || (waiting != 0ull && idle == ~0ull)) {
                const unsigned long long want = __ballot(!has && !pending);
                if (!exhausted && want != 0ull) {
                    uint32_t my = 0;
                    if (STATS) w_refill++;
                    if (wave_fetch(range, want, cursor_ptr, n, exhausted, my)) {
                        if (STATS) l_refill++;
                        slot = mesh_queue[my];
                        mcur = 0;
                        hit_tri = -1;
                        pending = true;
                    }
                }
                // Enter the path's next mesh: its object-space ray against the mesh's box, inside the interval that the other
                // primitives (k_wf_prims) and the meshes visited before left.  Lanes whose ray misses the box try the op after it.
                // One mesh op per trip - the one the first waiting lane wants - so that its record (bounds, transform) is
                // wave-uniform: scalar loads into SGPRs.  (Per-lane records cost every refill four dependent vector-memory
                // round trips.)
                for (;;) {
                    const unsigned long long pend = __ballot(pending);
                    if (pend == 0ull) break;
                    const uint32_t m = uint32_t(__builtin_amdgcn_readfirstlane(int(__shfl(int(mcur & 0x7FFFu), __ffsll((long long)pend) - 1))));
                    MeshOpRec<R> rb;
                    load_rec(m, rb);
                    if (pending && (mcur & 0x7FFFu) == m) {
                        if (!enter_mesh(rb)) {
                            mcur++;
                            if ((mcur & 0x7FFFu) >= n_mesh_ops) { pending = false; finish_path(); }
                        } else {
                            if (rb.flags & RT_MESH_HIT_BACK_FACES) mcur |= 0x80000000u;
                            has = true;
                            pending = false;
                            if (STATS) cnt.mesh_rays++;
                        }
                    }
                }
            }
            if (__ballot(has) == 0ull) {
                if (exhausted && __ballot(pending) == 0ull) break;
                continue;
            }
        } else {
            // ---- refill ----
            const unsigned long long idle = __ballot(!has);
            if (!exhausted && uint32_t(__popcll(idle)) >= refill_min) {
                uint32_t my = 0;
                if (STATS) w_refill++;
                if (wave_fetch(range, idle, cursor_ptr, n, exhausted, my)) {
                    if (STATS) l_refill++;
                    slot = mesh_queue[my];
                    hit_tri = -1;
                    enter_mesh(rb0);
                    has = true;
                    if (STATS) cnt.mesh_rays++;
                }
            }
            if (__ballot(has) == 0ull) {
                if (exhausted) break;
                continue;
            }
        }
        // ---- inner nodes: descend until (nearly) every lane holds a leaf or has finished ----
        for (;;) {
            unsigned long long inner = __ballot(has && node >= 0);
            if (inner == 0ull) break;
            // a few stragglers do not keep a wave full of ready leaves waiting
            if (uint32_t(__popcll(inner)) < inner_min && __ballot(has && node < 0) != 0ull) break;
            if (STATS) w_node++;
            if (has && node >= 0) {
                if (STATS) cnt.node_visits++;
                const float miss = __builtin_huge_valf();
                float nr[4];
                int32_t ch[4];
                if constexpr (NODE == 1) {
                    // four 16-B loads; plane = org + q * cell, so t = q * (cell * iv) + (org * iv - o * iv)
                    const uint4* nd = reinterpret_cast<const uint4*>(nodesq + node);
                    const uint4 h0 = nd[0], h1 = nd[1], h2 = nd[2];
                    const int4 cc = *reinterpret_cast<const int4*>(nd + 3);
                    ch[0] = cc.x; ch[1] = cc.y; ch[2] = cc.z; ch[3] = cc.w;
                    const float ax = __uint_as_float(h0.w) * ivx, ay = __uint_as_float(h1.x) * ivy, az = __uint_as_float(h1.y) * ivz;
                    const float bx = fmaf(__uint_as_float(h0.x), ivx, -oix), by = fmaf(__uint_as_float(h0.y), ivy, -oiy), bz = fmaf(__uint_as_float(h0.z), ivz, -oiz);
                    // with lo <= hi the nearer plane of an axis is `lo` for a non-negative inverse direction, `hi` otherwise:
                    // lo * iv vs hi * iv are then already ordered and the per-box min/max disappear
                    const bool negx = ivx < 0.0f, negy = ivy < 0.0f, negz = ivz < 0.0f;
                    const uint32_t qnx = negx ? h2.y : h1.z, qfx = negx ? h1.z : h2.y;
                    const uint32_t qny = negy ? h2.z : h1.w, qfy = negy ? h1.w : h2.z;
                    const uint32_t qnz = negz ? h2.w : h2.x, qfz = negz ? h2.x : h2.w;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float nxk = float((qnx >> (8 * k)) & 0xFFu), nyk = float((qny >> (8 * k)) & 0xFFu), nzk = float((qnz >> (8 * k)) & 0xFFu);
                        const float fxk = float((qfx >> (8 * k)) & 0xFFu), fyk = float((qfy >> (8 * k)) & 0xFFu), fzk = float((qfz >> (8 * k)) & 0xFFu);
                        float tn = fmaxf(fmaxf(fmaf(nxk, ax, bx), fmaf(nyk, ay, by)), fmaxf(fmaf(nzk, az, bz), 0.0f));
                        float tf = fminf(fminf(fmaf(fxk, ax, bx), fmaf(fyk, ay, by)), fminf(fmaf(fzk, az, bz), tmax32));
                        bool h = (tn <= tf) && ch[k] != kEmptyChild;
                        nr[k] = h ? tn : miss;
                    }
                } else {
                    const float4* nd = reinterpret_cast<const float4*>(nodes + node);
                    const uint32_t nearx = ivx < 0.0f ? 3u : 0u, neary = ivy < 0.0f ? 4u : 1u, nearz = ivz < 0.0f ? 5u : 2u;  // float4 index of the near planes
                    const float4 nx = nd[nearx], fx = nd[3u - nearx];
                    const float4 ny = nd[neary], fy = nd[5u - neary];
                    const float4 nz = nd[nearz], fz = nd[7u - nearz];
                    const int4 cc = *reinterpret_cast<const int4*>(nd + 6);
                    ch[0] = cc.x; ch[1] = cc.y; ch[2] = cc.z; ch[3] = cc.w;
                    const float nxa[4] = {nx.x, nx.y, nx.z, nx.w}, fxa[4] = {fx.x, fx.y, fx.z, fx.w};
                    const float nya[4] = {ny.x, ny.y, ny.z, ny.w}, fya[4] = {fy.x, fy.y, fy.z, fy.w};
                    const float nza[4] = {nz.x, nz.y, nz.z, nz.w}, fza[4] = {fz.x, fz.y, fz.z, fz.w};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        // explicit FMAs: the translation unit is built with -ffp-contract=off for the f64 parity arithmetic, but
                        // this f32 test only culls (its rounding is inside the boxes' padding either way): 24 fewer instructions
                        float tn = fmaxf(fmaxf(fmaf(nxa[k], ivx, -oix), fmaf(nya[k], ivy, -oiy)), fmaxf(fmaf(nza[k], ivz, -oiz), 0.0f));
                        float tf = fminf(fminf(fmaf(fxa[k], ivx, -oix), fmaf(fya[k], ivy, -oiy)), fminf(fmaf(fza[k], ivz, -oiz), tmax32));
                        bool h = (tn <= tf) && ch[k] != kEmptyChild;
                        nr[k] = h ? tn : miss;
                    }
                }
                // sort the four (entry distance, child) pairs, nearest first (5 compare-exchanges)
#define RT_CE(a, b)                                                   \
    if (nr[a] > nr[b]) {                                              \
        float tn_ = nr[a]; nr[a] = nr[b]; nr[b] = tn_;                \
        int32_t tc_ = ch[a]; ch[a] = ch[b]; ch[b] = tc_;              \
    }
                RT_CE(0, 1) RT_CE(2, 3) RT_CE(0, 2) RT_CE(1, 3) RT_CE(1, 2)
#undef RT_CE
                if (nr[0] < miss) {
                    // farthest first, so that the nearest remaining child is popped first
                    if (nr[3] < miss) { stk.put(sp, ch[3], nr[3]); sp++; }
                    if (nr[2] < miss) { stk.put(sp, ch[2], nr[2]); sp++; }
                    if (nr[1] < miss) { stk.put(sp, ch[1], nr[1]); sp++; }
                    node = ch[0];
                } else {
                    pop_next();
                }
            }
        }
        // ---- leaves: the (lane, triangle) pairs of all lanes that hold a leaf are FLATTENED over the wave, so
        //      that 64 triangle tests run per pass whatever the leaf sizes are (a per-lane loop ran at 33 % lane
        //      utilisation: leaves hold 1..4 triangles and a third of the lanes hold none).  Every pass: pair w ->
        //      (owner lane, k) through a wave-private LDS table, the owner's ray through cross-lane reads, one exact
        //      test in R, result into LDS; the owners then take their results in k order with the reference's
        //      interval rule, which makes the outcome identical to the sequential loop (mesh.rs:62-107). ----
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
                    const V3<R> po = {__shfl(o.x, owner), __shfl(o.y, owner), __shfl(o.z, owner)};
                    const V3<R> pd = {__shfl(d.x, owner), __shfl(d.y, owner), __shfl(d.z, owner)};
                    uint32_t pfirst;
                    bool hit_back;
                    if constexpr (MULTI) {
                        const uint32_t pfirst_hb = uint32_t(__shfl(int(first | (mcur & 0x80000000u)), owner));  // bit 31: the owner's mesh hits back faces
                        pfirst = pfirst_hb & 0x7FFFFFFFu;
                        hit_back = (pfirst_hb >> 31) != 0u;
                    } else {
                        pfirst = uint32_t(__shfl(int(first), owner));
                        hit_back = (rb0.flags & RT_MESH_HIT_BACK_FACES) != 0u;
                    }
                    R rt = Lim<R>::inf(), ru = R(0), rv = R(0);  // t = +inf: "no hit" (fails `t_max <= t` at the owner)
                    if (act) {
                        const TriRec<R>& tr = tris[pfirst + k];
                        if (STATS) cnt.tri_tests++;
                        V3<R> edge1 = ld3(tr.e1), edge2 = ld3(tr.e2);
                        V3<R> ray_x_edge2 = cross(pd, edge2);
                        R det = dot(edge1, ray_x_edge2);
                        R dd = hit_back ? fabs(det) : det;
                        if (!(dd < Lim<R>::eps())) {
                            R inv_det = R(1) / det;
                            V3<R> b = po - ld3(tr.v0);
                            R u = dot(b, ray_x_edge2) * inv_det;
                            if (!(u < R(0) || u > R(1))) {
                                V3<R> b_x_edge1 = cross(b, edge1);
                                R v = dot(pd, b_x_edge1) * inv_det;
                                if (!(v < R(0) || u + v > R(1))) {
                                    rt = dot(edge2, b_x_edge1) * inv_det;
                                    ru = u;
                                    rv = v;
                                }
                            }
                        }
                    }
                    res_t[lane] = rt; res_u[lane] = ru; res_v[lane] = rv;
                    __builtin_amdgcn_wave_barrier();
                    if (leaf) {
                        const int jlo = max(0, int(c0) - int(pre));
                        const int jhi = min(int(count), int(c0) + 64 - int(pre));
                        for (int j = jlo; j < jhi; j++) {
                            const int idx = int(pre) + j - int(c0);
                            const R t = res_t[idx];
                            if (t <= t_lo || t_max <= t) continue;
                            t_max = t; hit_u = res_u[idx]; hit_v = res_v[idx];
                            hit_tri = int32_t(first + uint32_t(j));
                            if constexpr (MULTI) mcur = (mcur & 0xC0007FFFu) | (((mcur & 0x7FFFu) + 1u) << 15);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                if (leaf) {
                    tmax32 = f32_at_least(t_max - t_shift);
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
        uint32_t wn = w_node, wt = w_tri, wr = w_refill;  // wave-uniform
        if ((threadIdx.x & 63u) == 0) {
            atomicAdd(&counters->node_wave_iters, (unsigned long long)wn);
            atomicAdd(&counters->tri_wave_iters, (unsigned long long)wt);
            atomicAdd(&counters->refill_wave_iters, (unsigned long long)wr);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Shade: one path vertex per lane (camera.rs:295-331), regeneration in place and queue compaction.
// ---------------------------------------------------------------------------------------------
// Waves per SIMD asked of the register allocator for k_wf_shade.  The lean variant with every small table in LDS (LDS == 1:
// the headline scenes) fits 95 VGPRs without scratch since round 3 (sample-index split by reciprocals, Schlick constants from
// the host) and runs at 5 waves: -4.5 % (C4) to -7.5 % (default scene) of the kernel's time against 4 waves
// (profiles/r03/ab/shade_five_waves.txt).  It fits with NOTHING to spare: tools/kernel_regs.py after every change to the
// shading code - at 96 VGPRs + 32 B of scratch the gain is gone.  The variants that read the tables from global memory and the
// counting variants need 16-48 B of scratch at that size and stay at 4 (<= 128 VGPRs).  History: 106 VGPRs since the inverse
// trigonometric functions of the UV maps are called out of line (uv_acos / uv_atan2 in rt_device.h: inlined, their polynomial
// coefficients sat in 50 VGPRs for the whole kernel); 166 and 3 waves before that, 188-197 and 2 waves in round 1.  The
// texture-interpreter variants need 174-189: no cap.
constexpr uint32_t kShadeListBytes = 2u * WF_CHUNK * 2u + 8u * 4u;  // k_wf_shade's LDS in front of the staged tables: two lists of 16-bit entries, counters
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 5
#endif
#define RT_SHADE_BOUNDS __launch_bounds__(256, TEX ? 1 : (!STATS ? RT_SHADE_WAVES : 4))

// Diagnostic build (-DRT_SHADE_STAMPS, tools/gpu_shade_stamps.sh): where a wave of k_wf_shade spends its cycles.  s_memtime stamps
// around the sections of a trip, summed per wave and added to g_shade_stamps at the end; never compiled into the product.
#ifdef RT_SHADE_STAMPS
__device__ unsigned long long g_shade_stamps[16];
#define RT_STAMP(k)                                                     \
    do {                                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
        stamp_acc[k] += now_ - stamp_t;                                 \
        stamp_t = now_;                                                 \
    } while (0)
#else
#define RT_STAMP(k) do { } while (0)
#endif

template <typename R, bool STATS, int LDS, bool TEX>
__global__ void RT_SHADE_BOUNDS k_wf_shade(SceneView<R> sc_g, CameraView<R> cam, ParamsView<R> prm, WfPool<R> pool, WfGroup<R> grp,
                                           const uint32_t* __restrict__ queue_in, uint32_t* __restrict__ queue_out,
                                           WfCounters* __restrict__ ctr, double* __restrict__ sample_L, DeviceCounters* counters,
                                           const WfPool<R>* __restrict__ pool_dev, uint32_t staged) {
    extern __shared__ __align__(16) char lds_raw[];
    // the lists hold positions inside this workgroup's chunk (16 bits): 8 KB instead of 16, so that up to 23 KB of tables fit the
    // 32 KB that five workgroups per CU leave each other
    uint16_t* alive_list = reinterpret_cast<uint16_t*>(lds_raw);  // [WF_CHUNK] entries that go to the next queue
    uint16_t* dead_list = alive_list + WF_CHUNK;                 // [WF_CHUNK] entries whose path ended
    uint32_t* lc = reinterpret_cast<uint32_t*>(dead_list + WF_CHUNK);  // [0] n_alive [1] n_dead [2,3] sample base [4] queue base
    char* tables = reinterpret_cast<char*>(lc + 8);
    if (threadIdx.x < 8) lc[threadIdx.x] = 0;
    SceneView<R> sc = sc_g;
    if constexpr (LDS == 1) sc = scene_tables_to_lds<R, true>(sc_g, sc_g.lay_shade, sc_g.small_blob_shade, tables, staged);
    else if constexpr (LDS == 2) sc = scene_tables_to_lds<R, false>(sc_g, sc_g.lay_shade, sc_g.small_blob_shade, tables, staged);
    else __syncthreads();
    const uint32_t n = ctr->n_in;
    // While samples remain every finished path restarts in place, so ALL slots are queued: the order of
    // the queue is then irrelevant and slot = queue position makes every state access coalesced (the
    // compacted queue is a near-random permutation after a few iterations: 64 lines per wave load).
    const bool full = n == pool.capacity;
    const uint32_t begin = blockIdx.x * WF_CHUNK;
    const uint32_t end = min(n, begin + WF_CHUNK);
    LaneCounters cnt;
#ifdef RT_SHADE_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_t = __builtin_amdgcn_s_memtime();
#endif
    // ---- phase 1: one path vertex per lane, chunk by chunk ----
    for (uint32_t base = begin; base < end; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const bool active = i < end;
        bool alive = false;
        uint32_t slot = 0;
        RT_STAMP(0);
        if (active) {
            slot = full ? i : queue_in[i];
            PathState<R> ps;
            ps.ray = make_ray(mk<R>(at(pool.ox, slot), at(pool.oy, slot), at(pool.oz, slot)), mk<R>(at(pool.dx, slot), at(pool.dy, slot), at(pool.dz, slot)));
            Best<R> best;
            best.t = at(pool.ht, slot); best.u = at(pool.hu, slot); best.v = at(pool.hv, slot);
            best.pc = at(pool.hpc, slot); best.tri = at(pool.htri, slot);
            // Resolve the hit BEFORE the rest of the path state is loaded: the compiler otherwise hoists those loads
            // above resolve_hit's loops (texture walk, transform chain) and keeps more values live across them (9 VGPRs that
            // the kernel does not have at 5 waves per SIMD).  Requesting them with the ray and parking them in LDS meanwhile -
            // one memory round trip less per trip - was measured: -3.5 % of the kernel on the default scene, +1..2 % on
            // light_test / two_meshes, +-0 on the headline (profiles/r03/ab/shade_park_state.txt); not kept.
            HitInfo<R> hit{};
            if (best.pc >= 0) hit = resolve_hit<R, TEX>(sc, ps.ray, best);
            asm volatile("" ::: "memory");
            RT_STAMP(1);
            ps.throughput = mk<R>(at(pool.tr, slot), at(pool.tg, slot), at(pool.tb, slot));
            ps.depth = at(pool.depth, slot);
            Rng rng;
            rng.s = at(pool.rng, slot);
            ps.radiance = mk<R>(0, 0, 0);
            bool cont;
            if (best.pc < 0) {  // camera.rs:331 background
                ps.radiance = ps.throughput * ld3(prm.background);
                cont = false;
            } else {
                cont = shade_hit<R, STATS, TEX>(sc, prm, ps, hit, rng, cnt);
            }
            ps.depth--;
            RT_STAMP(2);
            // The array bases are re-read HERE from a copy of the pool descriptor in global memory (scalar loads): with
            // the kernel-argument copy the compiler kept the eleven load addresses alive as VGPR pairs across the
            // whole shading code to reuse them for these stores (22 VGPRs of a kernel that is occupancy-bound).
            asm volatile("" ::: "memory");
            const WfPool<R>& pw = *pool_dev;
            alive = cont && ps.depth != 0;  // depth == 0: ray_color returns black without tracing (camera.rs:290)
            if (!alive) {
                if (cont) end_black(ps);  // depth exhausted: the next ray_color call returns black (camera.rs:290)
                uint64_t s = at(pool.sample, slot);
                sample_L[3 * s + 0] = double(ps.radiance.x);
                sample_L[3 * s + 1] = double(ps.radiance.y);
                sample_L[3 * s + 2] = double(ps.radiance.z);
                // This slot restarts in phase 2 on a new camera sample.  The part of that state which does not depend on the
                // sample is stored HERE, by the same store instructions as the surviving lanes' values: those arrays then get
                // their 128-byte lines whole in one go.  (Round 2 wrote every array in two parts - 70 % of a line's slots
                // here, the rest in phase 2 after the line had left the L2 - and the L2 wrote back 1.8 x the bytes;
                // profiles/r03/ab/shade_inplace_restart.txt has the counters and why the remaining arrays stay split.)
                ps.throughput = mk<R>(1, 1, 1);
                ps.depth = cam.max_depth;
                if (!cam.has_aperture) ps.ray.o = ld3(cam.position);  // camera.rs:265-275: the origin moves only with an aperture
            }
            put_global(pw.tr, slot, ps.throughput.x); put_global(pw.tg, slot, ps.throughput.y); put_global(pw.tb, slot, ps.throughput.z);
            put_global(pw.depth, slot, ps.depth);
            if (alive || !cam.has_aperture) { put_global(pw.ox, slot, ps.ray.o.x); put_global(pw.oy, slot, ps.ray.o.y); put_global(pw.oz, slot, ps.ray.o.z); }
            if (alive) {
                put_global(pw.dx, slot, ps.ray.d.x); put_global(pw.dy, slot, ps.ray.d.y); put_global(pw.dz, slot, ps.ray.d.z);
                put_global(pw.rng, slot, rng.s);
            }
        }
        lds_append(active && alive, uint16_t(i - begin), alive_list, &lc[0]);
        lds_append(active && !alive, uint16_t(i - begin), dead_list, &lc[1]);
        RT_STAMP(3);
    }
    __syncthreads();
    RT_STAMP(4);
    // ---- phase 2: finished paths restart IN PLACE on the next samples (one global atomic per workgroup; all lanes generate
    //      camera rays together, compacted over the workgroup: run inside phase 1 by the dead lanes themselves this code
    //      executes at 20-30 % lane utilisation in every trip and costs more than the split write saves) ----
    const uint32_t n_dead = lc[1];
    if (threadIdx.x == 0 && n_dead) {
        unsigned long long b0 = atomicAdd(&ctr->next_sample, (unsigned long long)n_dead);
        lc[2] = uint32_t(b0);
        lc[3] = uint32_t(b0 >> 32);
    }
    __syncthreads();
    const unsigned long long s_base = (unsigned long long)lc[2] | ((unsigned long long)lc[3] << 32);
    for (uint32_t j0 = 0; j0 < n_dead; j0 += blockDim.x) {
        const uint32_t j = j0 + threadIdx.x;
        bool restarted = false;
        uint32_t slot = 0;
        uint16_t entry = 0;
        if (j < n_dead) {
            const unsigned long long s2 = s_base + j;
            entry = dead_list[j];
            slot = full ? begin + entry : queue_in[begin + entry];
            if (s2 < grp.total) {
                V3<R> o, d;
                Rng rng;
                wf_new_sample(s2, grp, cam, prm, o, d, rng);
                if (cam.has_aperture) { at(pool.ox, slot) = o.x; at(pool.oy, slot) = o.y; at(pool.oz, slot) = o.z; }
                at(pool.dx, slot) = d.x; at(pool.dy, slot) = d.y; at(pool.dz, slot) = d.z;
                at(pool.rng, slot) = rng.s;
                at(pool.sample, slot) = uint64_t(s2);
                restarted = true;
            }
        }
        lds_append(restarted, entry, alive_list, &lc[0]);
    }
    RT_STAMP(5);
    __syncthreads();
    // ---- phase 3: surviving slots -> next queue (one global atomic per workgroup, coalesced copy) ----
    const uint32_t n_alive = lc[0];
    if (threadIdx.x == 0 && n_alive) lc[4] = atomicAdd(&ctr->n_out, n_alive);
    __syncthreads();
    const uint32_t qb = lc[4];
    for (uint32_t j = threadIdx.x; j < n_alive; j += blockDim.x) queue_out[qb + j] = full ? begin + alive_list[j] : queue_in[begin + alive_list[j]];
    RT_STAMP(6);
#ifdef RT_SHADE_STAMPS
    if ((threadIdx.x & 63u) == 0) {
        for (int k = 0; k < 7; k++) atomicAdd(&g_shade_stamps[k], stamp_acc[k]);
        atomicAdd(&g_shade_stamps[8], (unsigned long long)(begin < end ? (end - begin + 255u) / 256u : 0u));  // trips of this wave
        atomicAdd(&g_shade_stamps[9], 1ull);
    }
#endif
    if (STATS) {
        uint32_t rays = cnt.rays, prims = cnt.prim_tests;
        for (int off = 32; off > 0; off >>= 1) { rays += __shfl_down(rays, off); prims += __shfl_down(prims, off); }
        if ((threadIdx.x & 63u) == 0 && (prims | rays)) {
            atomicAdd(&counters->prim_tests, (unsigned long long)prims);
            if (rays) atomicAdd(&counters->rays, (unsigned long long)rays);
        }
    }
}

// Rotates the queue counters between iterations (device side, no host round trip).
__global__ void k_wf_advance(WfCounters* ctr) {
    ctr->n_in = ctr->n_out;
    ctr->n_out = 0;
    ctr->cursor = 0;
    ctr->n_mesh = 0;
}

// ---------------------------------------------------------------------------------------------
// Resolve: ordered sums.  sample_L is indexed by the sample number s = ((tid_local * S*S + st) * npix + pix).
// acc[pix] += (sum_st L) / spp for every replica of the group, in replica order (camera.rs:229,247-253).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_wf_resolve(const double* __restrict__ sample_L, double* __restrict__ acc, uint64_t npix,
                                                    uint32_t strata, uint32_t n_replicas, double spp, int first_group,
                                                    double* __restrict__ out, int last_group) {
    uint64_t pix = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    double a[3];
    for (int k = 0; k < 3; k++) a[k] = first_group ? 0.0 : acc[3 * pix + k];
    for (uint32_t t = 0; t < n_replicas; t++) {
        double col[3] = {0.0, 0.0, 0.0};
        for (uint32_t st = 0; st < strata; st++) {
            uint64_t s = (uint64_t(t) * strata + st) * npix + pix;
            col[0] += sample_L[3 * s + 0];
            col[1] += sample_L[3 * s + 1];
            col[2] += sample_L[3 * s + 2];
        }
        for (int k = 0; k < 3; k++) a[k] += col[k] / spp;
    }
    if (last_group) {
        out[4 * pix + 0] = a[0];
        out[4 * pix + 1] = a[1];
        out[4 * pix + 2] = a[2];
        out[4 * pix + 3] = 0.0;
    } else {
        for (int k = 0; k < 3; k++) acc[3 * pix + k] = a[k];
    }
}

}  // namespace rt
