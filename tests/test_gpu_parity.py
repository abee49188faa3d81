"""Parity of the HIP render path (through the C ABI, librt_mi355.so) with the CPU oracle.

Tolerances (stated here, used below):
  * f64 kernels vs oracle, same keyed RNG.  The kernels are built with -ffp-contract=off (Rust
    never fuses a*b+c) and share the deterministic sin/cos/ln of include/rt_detmath.h with the
    oracle, so every path takes bit-identical decisions; what remains is the order of the
    radiance products (iterative throughput vs the reference's recursion) — a few ulps.
    Bar: EVERY channel value within 1e-12 relative (absolute floor 1e-15), NaN pixels
    (reference 0/0 quirks) in exactly the same places.  Measured: max 1e-15.
  * f32 kernels vs oracle: individual paths may take different branches, so the bar is
    statistical: image mean within 1 %, >= 95 % of values within max(5 % relative, 0.02).
Everything here needs the GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle
from rust_raytracer_amd import api

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_frames.npz"))

SCENES = {
    "cornell": ["scenes/cornell", "-w=48", "-s=16", "--seed=1"],
    "light_test": ["scenes/light_test", "-w=60", "-s=16", "--seed=2"],
    "hollow_glass": ["tests/scenes/hollow_glass", "-w=48", "-s=16", "--seed=3"],
    "nested_transform": ["tests/scenes/nested_transform", "-w=48", "-s=8", "-t=2", "--seed=4"],
    "default": ["-w=60", "-s=16", "--seed=5"],
    "sun_sky": ["tests/scenes/sun_sky", "-w=48", "-s=16", "--seed=6"],
    "bvh_spheres": ["tests/scenes/bvh_spheres", "-w=48", "-s=16", "--seed=7"],
    "single_light": ["tests/scenes/single_light", "-w=37", "-s=9", "--seed=8"],     # width not a tile multiple
    "test": ["scenes/test", "-w=45", "-s=16", "--seed=9"],
    "tonemap_test": ["scenes/tonemap_test", "-w=40", "-s=16", "--seed=10"],
    "two_meshes": ["tests/scenes/two_meshes", "-w=48", "-s=16", "--seed=11"],   # >1 mesh op: k_wf_mesh serves a path's meshes one after the other
    # texture interpreter variants of the kernels (image / noise / lerp / channel textures, normal maps)
    "perlin": ["scenes/perlin", "-w=48", "-s=16", "--seed=12"],
    "earth": ["scenes/earth", "-w=48", "-s=16", "--seed=13"],          # JPEG texture on a sphere
    # the same at 320 x 180: one nearest-neighbour texel of the 2048-px map per ~0.4 pixel, so most samples sit next to a texel edge:
    # u = (atan2 + pi) / 2 pi and v = acos / pi must be the oracle's bits (include/rt_detmath.h det_atan2 / det_acos on both sides)
    "earth_dense": ["scenes/earth", "-w=320", "-s=4", "--seed=25"],
    "texture_test": ["scenes/texture_test", "-w=48", "-s=16", "--seed=14"],  # PNG albedo / roughness channel / normal map on a mesh
    "texture_mix": ["tests/scenes/texture_mix", "-w=48", "-s=16", "--seed=15"],  # every operator, every primitive's tangent frame
    # constant-density volumes (sphere / mesh / box boundaries); wavefront: combined intersect kernel, VOL variant (a mesh in a boundary)
    "smoke": ["tests/scenes/smoke", "-w=48", "-s=16", "--seed=16"],
    # the reference's cornell_smoke: box boundaries only -> the volumes run inside k_wf_prims (VOL variant)
    "cornell_smoke": ["scenes/cornell_smoke", "-w=48", "-s=16", "--seed=21"],
    # a volume inside another volume's boundary, a plain volume, and a mesh BEHIND them (k_wf_prims<VOL> + k_wf_mesh)
    "nested_volumes": ["tests/scenes/nested_volumes", "-w=48", "-s=16", "--seed=22"],
    # polished Metal / Glossy (fuzz 0: the kernels skip the normal samples of the random term) beside rough ones
    "polished": ["tests/scenes/polished", "-w=48", "-s=16", "--seed=26"],
    # ObjectLists three levels deep inside `lights` (with an empty list and a non-light member)
    "nested_lights": ["tests/scenes/nested_lights", "-w=48", "-s=16", "--seed=23"],
    # a texture expression with more live values than the interpreter's four registers (spilled stack)
    "deep_texture": ["tests/scenes/deep_texture", "-w=48", "-s=16", "--seed=24"],
    # an ObjectList (emissive box) inside `lights`: nested pdf_value / random
    "box_light": ["tests/scenes/box_light", "-w=48", "-s=16", "--seed=17"],
    # zero-weight vertices (black albedo, light samples below the horizon) in front of 0/0 vertices (one-sided light seen
    # from behind): the reference's 0 * NaN; 259 of these 2304 pixels are NaN ONLY through a zero-weight prefix
    # (tests/test_zero_weight.py), so a kernel that ends zero-weight paths early fails here
    "zero_weight_nan": ["tests/scenes/zero_weight_nan", "-w=48", "-s=16", "--seed=18"],
    # 48 spheres + quads in an object BVH, re-built as a SAH tree by the scene compiler (hollow balls with inverted boxes,
    # coincident duplicates, a list inside the bvh)
    "sphere_field": ["tests/scenes/sphere_field", "-w=60", "-s=16", "--seed=20"],
}


@pytest.fixture(scope="module")
def dev():
    lib = api.load_device_lib()
    assert lib.rt_device_count() >= 1, "no HIP device: the GPU tests must run on the MI355X box"
    return lib


def assert_f64_parity(gpu, ref):
    a, b = gpu[..., :3], ref[..., :3]
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    np.testing.assert_array_equal(nan_a, nan_b)
    fin = ~nan_b
    err = np.abs(a[fin] - b[fin])
    bad = err > np.maximum(1e-12 * np.abs(b[fin]), 1e-15)
    assert not bad.any(), f"{bad.mean():.4%} of values differ by more than 1e-12 relative (max abs err {err.max():.3e})"
    assert np.all(gpu[..., 3] == 0.0)


PIPELINES = {"mega": api.RT_PIPELINE_MEGAKERNEL, "wavefront": api.RT_PIPELINE_WAVEFRONT}


@pytest.mark.parametrize("pipeline", sorted(PIPELINES))
@pytest.mark.parametrize("name", sorted(SCENES))
def test_f64_matches_oracle(dev, name, pipeline):
    """Both schedulers (per-pixel megakernel, wavefront pool) against the oracle."""
    hs = api.HostScene(SCENES[name])
    scene = api.DeviceScene(hs.desc, 0)
    p = hs.params.copy()
    p.pipeline = PIPELINES[pipeline]
    ref, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
    gpu = scene.render(hs.camera, p)
    assert gpu.shape == ref.shape
    assert_f64_parity(gpu, ref)
    assert scene.stats().pipeline_used == PIPELINES[pipeline]


def test_pipelines_are_bit_identical_and_small_pool_works(dev, monkeypatch):
    """The wavefront scheduler writes every sample to its own slot and sums in the reference's
    order, so its frame equals the megakernel's bit for bit — whatever the pool size, including a
    pool far smaller than the sample count (many regenerations) and replica groups."""
    hs = api.HostScene(["scenes/light_test", "-w=80", "-s=48", "-t=3", "--seed=13"])
    scene = api.DeviceScene(hs.desc, 0)
    p = hs.params.copy()
    p.pipeline = api.RT_PIPELINE_MEGAKERNEL
    mega = scene.render(hs.camera, p)
    p.pipeline = api.RT_PIPELINE_WAVEFRONT
    np.testing.assert_array_equal(scene.render(hs.camera, p), mega)
    monkeypatch.setenv("RT_WF_POOL", "1000")
    np.testing.assert_array_equal(scene.render(hs.camera, p), mega)
    monkeypatch.setenv("RT_WF_REFILL", "64")
    np.testing.assert_array_equal(scene.render(hs.camera, p), mega)


def test_zero_weight_paths_are_traced_unless_the_light_set_is_safe(dev, monkeypatch):
    """camera.rs:310-314: `(L * att * s_pdf) / pdf` with a zero weight is NaN when L is.  (1) In a scene with a
    one-sided light the NaN mask must be the oracle's (also covered per pipeline by test_f64_matches_oracle) and the
    GPU must trace exactly as many rays as the reference does.  (2) In scenes the compiler classifies as safe the
    early end must not change a single bit: same frame with the early end switched off."""
    hs = api.HostScene(SCENES["zero_weight_nan"])
    assert not api.scene_info(hs.desc) & api.RT_SCENE_INFO_ZERO_WEIGHT_STOP
    ref, ost = pyoracle.render(hs.desc, hs.camera, hs.params)
    p = hs.params.copy()
    p.collect_stats = 1
    scene = api.DeviceScene(hs.desc, 0)
    gpu = scene.render(hs.camera, p)
    nan_ref = np.isnan(ref[..., :3]).any(axis=2)
    assert 0.3 < nan_ref.mean() < 0.8
    np.testing.assert_array_equal(np.isnan(gpu[..., :3]), np.isnan(ref[..., :3]))
    assert_f64_parity(gpu, ref)
    assert scene.stats().rays <= ost.rays          # only paths whose weight is NaN in every channel end early (their value is NaN whatever follows)
    for name in ("cornell", "light_test", "default"):
        hs = api.HostScene(SCENES[name])
        assert api.scene_info(hs.desc) & api.RT_SCENE_INFO_ZERO_WEIGHT_STOP
        monkeypatch.delenv("RT_ZERO_WEIGHT_STOP", raising=False)
        q = hs.params.copy()
        q.collect_stats = 1
        s1 = api.DeviceScene(hs.desc, 0)
        fast = s1.render(hs.camera, q)
        rays_fast = s1.stats().rays
        monkeypatch.setenv("RT_ZERO_WEIGHT_STOP", "0")   # read when the device tables are built
        s2 = api.DeviceScene(hs.desc, 0)
        exact = s2.render(hs.camera, q)
        _, ost = pyoracle.render(hs.desc, hs.camera, hs.params)
        assert s2.stats().rays == ost.rays and rays_fast <= ost.rays
        same = (fast == exact) | (np.isnan(fast) & np.isnan(exact))
        assert same.all()


@pytest.mark.parametrize("name", ["sphere_field", "default", "bvh_spheres"])
def test_rebuilt_primitive_groups_render_the_reference_tree_frame(dev, name, monkeypatch):
    """SURVEY 8 row f-4 (top-level BVH over primitives): object-BVH / list subtrees of spheres and quads are re-built as
    SAH trees (rt_compile.cpp).  The frame must be the one the reference's own tree gives, bit for bit (same reachable
    primitives incl. the inverted-box quirk B-8, first-visited-wins ties by rank), and the oracle's at the f64 bar; the
    re-built tree must need fewer primitive tests."""
    hs = api.HostScene(SCENES[name])
    frames, tests = {}, {}
    for rebuild in ("0", "1"):
        monkeypatch.setenv("RT_PRIM_REBUILD", rebuild)
        groups = api.scene_mesh_stats(hs.desc)["rebuilt_groups"]
        assert groups == (0 if rebuild == "0" or name == "bvh_spheres" else 1)      # bvh_spheres: 10 spheres, below the threshold
        scene = api.DeviceScene(hs.desc, 0)
        for pipe in (api.RT_PIPELINE_WAVEFRONT, api.RT_PIPELINE_MEGAKERNEL):
            p = hs.params.copy()
            p.pipeline = pipe
            p.collect_stats = 1
            frames[(rebuild, pipe)] = scene.render(hs.camera, p)
            tests[(rebuild, pipe)] = scene.stats().prim_tests
    first = frames[("0", api.RT_PIPELINE_WAVEFRONT)]
    for key, f in frames.items():
        same = (f == first) | (np.isnan(f) & np.isnan(first))
        assert same.all(), f"{key}: {int((~same).any(axis=2).sum())} pixels differ from the reference-tree frame"
    ref, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
    assert_f64_parity(first, ref)
    if name != "bvh_spheres":
        assert tests[("1", api.RT_PIPELINE_WAVEFRONT)] < 0.9 * tests[("0", api.RT_PIPELINE_WAVEFRONT)]   # sphere / quad tests incl. the light pdf re-intersections (box tests are not counted)


def test_replica_groups_forced(dev, monkeypatch):
    """The per-sample radiance buffer is rendered in replica GROUPS when it exceeds its budget (BASELINE config C5
    does: 69 GB against 64 GB).  Forced here with a zero budget: -t=3 becomes three groups of one replica, the
    per-pixel accumulator carries the sums between groups (k_wf_resolve first_group / last_group) — the frame must
    equal the single-group frame and the megakernel's bit for bit, and the oracle's at the f64 bar."""
    hs = api.HostScene(["scenes/light_test", "-w=80", "-s=27", "-t=3", "--seed=19"])
    assert hs.params.thread_count == 3 and hs.spp == 27
    scene = api.DeviceScene(hs.desc, 0)
    p = hs.params.copy()
    p.pipeline = api.RT_PIPELINE_WAVEFRONT
    one = scene.render(hs.camera, p)
    assert scene.stats().n_replica_groups == 1
    monkeypatch.setenv("RT_WF_SAMPLE_GB", "0")
    three = scene.render(hs.camera, p)
    assert scene.stats().n_replica_groups == 3
    np.testing.assert_array_equal(three, one)
    monkeypatch.setenv("RT_WF_POOL", "3000")       # pool smaller than one replica: regeneration inside every group
    np.testing.assert_array_equal(scene.render(hs.camera, p), one)
    assert scene.stats().n_replica_groups == 3
    monkeypatch.delenv("RT_WF_POOL")
    q = p.copy()
    q.band_rows, q.n_parts, q.part = 16, 2, 1      # groups + row partition together
    part = scene.render(hs.camera, q)
    np.testing.assert_array_equal(part, one[api.owned_rows(hs.height, q)])
    monkeypatch.delenv("RT_WF_SAMPLE_GB")
    p.pipeline = api.RT_PIPELINE_MEGAKERNEL
    np.testing.assert_array_equal(scene.render(hs.camera, p), one)
    ref, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
    assert_f64_parity(one, ref)


@pytest.mark.parametrize("name", sorted(GOLD.files))
def test_f64_matches_committed_golden_frames(dev, name):
    hs = api.HostScene(SCENES[name])
    gpu = api.DeviceScene(hs.desc, 0).render(hs.camera, hs.params)
    assert_f64_parity(gpu, GOLD[name])


@pytest.mark.parametrize("name", ["cornell", "light_test", "default", "nested_transform", "sun_sky", "texture_mix"])
def test_f32_is_statistically_equivalent(dev, name):
    args = [a for a in SCENES[name] if not a.startswith("-s=")] + ["-s=64"]
    hs = api.HostScene(args)
    ref, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
    p = hs.params.copy()
    p.precision = api.RT_PRECISION_F32
    gpu = api.DeviceScene(hs.desc, 0).render(hs.camera, p)
    a, b = gpu[..., :3], ref[..., :3]
    assert not np.isnan(a).any()
    assert abs(a.mean() - b.mean()) <= 0.01 * b.mean()
    close = np.abs(a - b) <= np.maximum(0.05 * np.abs(b), 0.02)
    assert close.mean() >= 0.95, f"only {close.mean():.3%} of f32 values are close to the f64 oracle"


@pytest.mark.parametrize("name", ["cornell", "hollow_glass", "default", "light_test", "two_meshes", "texture_mix", "smoke", "cornell_smoke", "nested_volumes"])
def test_every_kernel_variant_is_bit_identical(dev, name, monkeypatch):
    """The wavefront kernels exist in several template variants (counters on/off, small tables in
    LDS or global memory, split or combined intersect).  hipcc (ROCm 7.2) has produced wrong Dielectric
    code for individual variants of the large shade kernel when register allocation changed
    (`__launch_bounds__(256,3)`, out-of-line helpers), so every variant is checked against the
    megakernel bit for bit on scenes with glass, meshes and large tables."""
    hs = api.HostScene(SCENES[name])
    scene = api.DeviceScene(hs.desc, 0)
    p = hs.params.copy()
    p.pipeline = api.RT_PIPELINE_MEGAKERNEL
    mega = scene.render(hs.camera, p)
    p.pipeline = api.RT_PIPELINE_WAVEFRONT
    for stats in (0, 1):
        for lds in ("1", "0"):
            # (the round-2 variants that lost their A/B - primitive program fused into k_wf_shade, two-stage f32 mesh search -
            # are no longer part of the library: commit 231b3c2 has them, profiles/r02/ab/ their records)
            for split, nodeq in (("1", "1"), ("1", "0"), ("0", "1")):
                monkeypatch.setenv("RT_LDS_TABLES", lds)
                monkeypatch.setenv("RT_WF_NODES", nodeq)  # BVH nodes of k_wf_mesh: 1 = 64-B quantised (default), 0 = 128-B f32 (A/B control)
                monkeypatch.setenv("RT_WF_SPLIT", split)  # 0: combined intersect kernel for every scene
                p.collect_stats = stats
                wf = scene.render(hs.camera, p)
                same = (wf == mega) | (np.isnan(wf) & np.isnan(mega))
                assert same.all(), f"variant stats={stats} lds={lds} split={split} nodeq={nodeq}: {int((~same).any(axis=2).sum())} pixels differ"


def _bumpy_grid_obj(path, n, offset, scale, flat=False):
    """An n x n grid of quads (2 n^2 triangles) over [-1, 1]^2 with a sine bump, then scaled per axis and moved: the
    coordinates a quantised BVH node has to cover (large offsets, anisotropic and zero extents)."""
    lines = ["vt 0 0"]
    for j in range(n + 1):
        for i in range(n + 1):
            x, z = -1 + 2 * i / n, -1 + 2 * j / n
            y = 0.0 if flat else float(0.25 * np.sin(3.1 * x) * np.cos(2.3 * z) + 0.05 * np.sin(17 * x + 5 * z))
            lines.append(f"v {x * scale[0] + offset[0]!r} {y * scale[1] + offset[1]!r} {z * scale[2] + offset[2]!r}")
            lines.append("vn 0 1 0")
    idx = lambda i, j: j * (n + 1) + i + 1
    for j in range(n):
        for i in range(n):
            a, b, c, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
            lines.append(f"f {a}/1/{a} {c}/1/{c} {b}/1/{b}")
            lines.append(f"f {a}/1/{a} {d}/1/{d} {c}/1/{c}")
    path.write_text("\n".join(lines) + "\n")


@pytest.mark.parametrize("case", ["far_from_origin", "anisotropic", "flat", "tiny"])
def test_quantised_bvh_nodes_on_awkward_meshes(dev, tmp_path, monkeypatch, case):
    """k_wf_mesh culls with 64-B nodes whose child boxes are 8-bit grid coordinates (BvhNode4q).  The grid must contain
    the padded boxes whatever the mesh's coordinates look like: far from the origin (large origin, small cells), three
    orders of magnitude between the axes, zero extent on an axis, and small coordinates (0.02 units: at 1e-3 the
    REFERENCE's octree, restated by the oracle, never separates the triangles and runs out of memory).  Every frame must match the
    oracle (a box that is too small loses hits) and the frame of the unquantised f32 nodes bit for bit."""
    offset, scale, cam, target = {
        "far_from_origin": ((1000.0, 0.0, -2000.0), (1.0, 1.0, 1.0), "1000,1.5,-1997", "1000,0,-2000"),
        "anisotropic": ((0.0, 0.0, 0.0), (50.0, 0.02, 1.0), "0,20,30", "0,0,0"),
        "flat": ((0.0, 0.25, 0.0), (1.0, 1.0, 1.0), "0,1.5,3", "0,0.25,0"),
        "tiny": ((0.0, 0.0, 0.0), (0.02, 0.02, 0.02), "0,0.03,0.06", "0,0,0"),
    }[case]
    _bumpy_grid_obj(tmp_path / "grid.obj", 24, offset, scale, flat=(case == "flat"))
    s = max(scale)
    lx, ly, lz = offset[0] - 0.5 * s, offset[1] + 2.0 * s, offset[2] - 0.5 * s
    scene = tmp_path / "scene"
    scene.write_text(f"@config output_width = 40\n@config aspect_ratio = 1\n@config focal_length = 40\n"
                     f"@config camera_pos = {cam}\n@config camera_target = {target}\n"
                     f"grid: mesh grid.obj (glossy (constant 0.7,0.6,0.3) (constant 0.3))\n"
                     f"lamp: plane {lx!r},{ly!r},{lz!r} {s!r},0,0 0,0,{s!r} (emissive (constant 8,8,8)) backface\n"
                     f"sky: sky (constant 0.3,0.4,0.6)\nworld: list $grid $lamp $sky\nlights: list $lamp\n")
    hs = api.HostScene([str(scene), "-s=16", "--seed=21"])
    ref, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
    assert np.isfinite(ref[..., :3]).all() and ref[..., :3].std() > 0.01   # the mesh is in view and lit
    p = hs.params.copy()
    p.pipeline = api.RT_PIPELINE_WAVEFRONT
    p.collect_stats = 1
    frames = {}
    for nodes in ("1", "0"):
        monkeypatch.setenv("RT_WF_NODES", nodes)
        dscene = api.DeviceScene(hs.desc, 0)
        frames[nodes] = dscene.render(hs.camera, p)
        st = dscene.stats()
        assert st.mesh_rays > 0 and st.tri_tests > 0 and st.bytes_node == (64 if nodes == "1" else 128)
        assert_f64_parity(frames[nodes], ref)
    np.testing.assert_array_equal(frames["1"], frames["0"])


def test_frame_pipeline_frames_are_the_single_render_frames(dev):
    """api.FramePipeline (two device scenes, two host threads, two streams; frame k+1 starts under the tail of frame k,
    rt_scene_set_tail_flag) must produce, for every frame, exactly the frame one DeviceScene renders on its own — here five
    frames with different seeds and one with a different row partition, wavefront scheduler with a small pool so that
    every render has many iterations and a long tail."""
    import torch
    hs = api.HostScene(SCENES["light_test"])
    plist = []
    for k in range(5):
        p = hs.params.copy()
        p.pipeline = api.RT_PIPELINE_WAVEFRONT
        p.seed = 100 + k
        plist.append(p)
    single = api.DeviceScene(hs.desc, 0)
    want = [single.render(hs.camera, p) for p in plist]
    device = torch.device("cuda", 0)
    outs = [torch.zeros((hs.height, hs.width, 4), dtype=torch.float64, device=device) for _ in plist]
    streams = [torch.cuda.Stream(device) for _ in range(2)]
    os.environ["RT_WF_POOL"] = "4096"
    try:
        pipe = api.FramePipeline(hs.desc, 0, depth=2)
        stats = pipe.render_frames(hs.camera, plist, [o.data_ptr() for o in outs], [s.cuda_stream for s in streams])
    finally:
        del os.environ["RT_WF_POOL"]
    torch.cuda.synchronize()
    assert len(stats) == 5 and all(st.samples == hs.width * hs.height * hs.spp for st in stats)
    for k in range(5):
        got = outs[k].cpu().numpy()
        same = (got == want[k]) | (np.isnan(got) & np.isnan(want[k]))
        assert same.all(), f"frame {k} of the pipeline differs from the single render"
    # three frames in flight (what bench.py uses on a multi-GPU rank)
    for o in outs:
        o.zero_()
    three = api.FramePipeline(hs.desc, 0, depth=3)
    three.render_frames(hs.camera, plist, [o.data_ptr() for o in outs], [s.cuda_stream for s in streams] + [torch.cuda.Stream(device).cuda_stream])
    torch.cuda.synchronize()
    for k in range(5):
        got = outs[k].cpu().numpy()
        assert ((got == want[k]) | (np.isnan(got) & np.isnan(want[k]))).all(), f"frame {k} of the 3-deep pipeline differs"
    # depth 1 degenerates to one render after the other
    one = api.FramePipeline(hs.desc, 0, depth=1)
    one.render_frames(hs.camera, plist[:2], [o.data_ptr() for o in outs[3:5]], [streams[0].cuda_stream])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(outs[3].cpu().numpy(), want[0])
    np.testing.assert_array_equal(outs[4].cpu().numpy(), want[1])


def test_tail_flag_is_set_by_every_render_and_on_errors(dev):
    hs = api.HostScene(SCENES["cornell"])
    scene = api.DeviceScene(hs.desc, 0)
    lib = scene._lib
    flag = C.c_int32(0)
    assert lib.rt_scene_set_tail_flag(scene._h, C.addressof(flag)) == api.RT_OK
    for pipeline in (api.RT_PIPELINE_WAVEFRONT, api.RT_PIPELINE_MEGAKERNEL):
        p = hs.params.copy()
        p.pipeline = pipeline
        flag.value = 0
        scene.render(hs.camera, p)
        assert flag.value == 1
    bad = hs.params.copy()
    bad.sqrt_spt = 0
    flag.value = 0
    with pytest.raises(api.RtError):
        scene.render(hs.camera, bad)
    assert flag.value == 1                      # a waiting frame is released even when the render fails
    assert lib.rt_scene_set_tail_flag(scene._h, None) == api.RT_OK
    flag.value = 0
    scene.render(hs.camera, hs.params)
    assert flag.value == 0                      # cleared: the library no longer writes to it
    assert lib.rt_scene_set_tail_flag(None, None) == api.RT_E_INVALID


def test_stats_counters_and_collect_flag(dev):
    hs = api.HostScene(SCENES["light_test"])
    scene = api.DeviceScene(hs.desc, 0)
    p = hs.params.copy()
    p.collect_stats = 1
    with_stats = scene.render(hs.camera, p)
    st = scene.stats()
    assert st.samples == hs.width * hs.height * hs.spp
    assert st.rays >= st.samples and st.mesh_rays > 0 and st.node_visits > st.mesh_rays and st.tri_tests > 0
    assert st.kernel_ms > 0 and st.bytes_node == 64 and st.bytes_tri == 80   # 4-wide nodes quantised to 64 B
    assert st.pipeline_used == api.RT_PIPELINE_WAVEFRONT     # AUTO picks the wavefront scheduler for mesh scenes
    q = p.copy()
    q.pipeline = api.RT_PIPELINE_MEGAKERNEL
    scene.render(hs.camera, q)
    sm = scene.stats()
    # same rays; the wavefront mesh kernel walks the 4-wide tree (about half the node fetches of the
    # megakernel's BVH2) with padded f32 boxes (a few more leaves than the exact boxes)
    assert sm.rays == st.rays
    assert st.node_visits < sm.node_visits and st.tri_tests <= 1.3 * sm.tri_tests
    p.collect_stats = 0
    np.testing.assert_array_equal(scene.render(hs.camera, p), with_stats)   # counting never changes pixels
    _, ost = pyoracle.render(hs.desc, hs.camera, hs.params)
    # the GPU stops a path whose weight is exactly zero; the reference still traces it
    assert st.rays <= ost.rays


def test_single_sample_trace_matches_oracle(dev):
    """Bounce by bounce: hit distance and position of individual paths."""
    hs = api.HostScene(SCENES["default"])
    scene = api.DeviceScene(hs.desc, 0)
    for (x, y, sx, sy) in [(30, 20, 0, 0), (31, 14, 3, 2), (5, 35, 1, 1), (55, 3, 2, 3)]:
        orgb, otr = pyoracle.trace_sample(hs.desc, hs.camera, hs.params, 0, x, y, sx, sy)
        grgb, gtr = scene.trace_sample(hs.camera, hs.params, 0, x, y, sx, sy)
        np.testing.assert_allclose(grgb, orgb, rtol=1e-9, atol=1e-12, equal_nan=True)
        n = min(len(otr), len(gtr))          # the GPU may stop earlier on a zero weight
        assert n >= 1
        fin = np.isfinite(otr[:n, 0])
        np.testing.assert_allclose(gtr[:n][fin][:, :4], otr[:n][fin][:, :4], rtol=1e-9, atol=1e-9)


# ---------------------------------------------------------------- size-independent properties at full size
def test_full_size_sky_only_is_exact(dev):
    """1200x1200 @1024 spp (one replica, power-of-two count: sum and division are exact):
    every pixel is exactly L.  With the headline 10 x 100 split, col/1000 is rounded per replica
    (camera.rs:229), so the value is L within an ulp or two — and identical in every pixel."""
    hs = api.HostScene(["tests/scenes/sky_only", "-w=1200", "-s=1024", "-t=1"])
    assert hs.spp == 1024
    gpu = api.DeviceScene(hs.desc, 0).render(hs.camera, hs.params)
    assert gpu.shape == (1200, 1200, 4)
    assert np.all(gpu[..., 0] == 0.5) and np.all(gpu[..., 1] == 1.0) and np.all(gpu[..., 2] == 2.0)
    hs = api.HostScene(["tests/scenes/sky_only", "-w=1200", "-s=1000", "-t=10"])
    assert hs.spp == 1000
    gpu = api.DeviceScene(hs.desc, 0).render(hs.camera, hs.params)
    np.testing.assert_allclose(gpu[..., :3], np.broadcast_to([0.5, 1.0, 2.0], gpu[..., :3].shape), rtol=1e-15)
    assert np.all(gpu == gpu[0, 0])
    q = hs.params.copy()
    q.band_rows, q.n_parts, q.part = 1, 600, 7
    ref, _ = pyoracle.render(hs.desc, hs.camera, q)
    np.testing.assert_array_equal(gpu[api.owned_rows(hs.height, q)], ref)


def ensure_dragon():
    """The stand-in for scenes/resource/dragon_high.obj is generated on the box (it is too large to commit)."""
    obj = os.path.join(api.REPO_DIR, "scenes", "resource", "dragon_high.obj")
    if not os.path.exists(obj):
        tool = os.path.join(api.REPO_DIR, "tools", "gen_dragon")
        if not os.path.exists(tool):
            subprocess.run(["g++", "-std=c++17", "-O2", "-o", tool, tool + ".cpp"], check=True)
        tmp = obj + f".tmp{os.getpid()}"
        subprocess.run([tool, tmp], check=True)
        os.replace(tmp, obj)


def test_full_size_frame_row_partition_and_determinism(dev):
    """Headline scene (871 200-triangle mesh) at 1200x1200, low spp: rendering the frame in
    3 interleaved 16-row parts gives bit-identical pixels to the whole frame, and a second run
    is bit-identical to the first (per-pixel accumulation order is fixed)."""
    ensure_dragon()
    hs = api.HostScene(["scenes/cornell_dragon", "-w=1200", "-s=4", "--seed=12"])
    scene = api.DeviceScene(hs.desc, 0)
    full = scene.render(hs.camera, hs.params)
    again = scene.render(hs.camera, hs.params)
    np.testing.assert_array_equal(full, again)
    rebuilt = np.empty_like(full)
    p = hs.params.copy()
    p.band_rows, p.n_parts = 16, 3
    for part in range(3):
        p.part = part
        rows = api.owned_rows(hs.height, p)
        img = scene.render(hs.camera, p)
        assert img.shape[0] == len(rows) == api.load_device_lib().rt_owned_rows(hs.height, C.byref(p))
        rebuilt[rows] = img
    np.testing.assert_array_equal(rebuilt, full)
    # and a sparse sample of full-size pixels against the oracle (every 97th row)
    q = hs.params.copy()
    q.band_rows, q.n_parts, q.part = 1, 97, 5
    ref, _ = pyoracle.render(hs.desc, hs.camera, q)
    assert_f64_parity(full[api.owned_rows(hs.height, q)], ref)


def test_full_frame_of_the_headline_scene_is_bit_identical_between_pipelines(dev):
    """The whole 1200x1200 frame of the headline scene at 100 spp (4 replicas x 5x5 strata): the wavefront
    scheduler (pool refills, in-place regeneration, replica-ordered resolve) and the per-pixel megakernel must
    produce the same bits in every pixel."""
    ensure_dragon()
    hs = api.HostScene(["scenes/cornell_dragon", "-w=1200", "-s=100", "-t=4", "--seed=3"])
    assert hs.spp == 100
    scene = api.DeviceScene(hs.desc, 0)
    p = hs.params.copy()
    p.pipeline = api.RT_PIPELINE_WAVEFRONT
    wf = scene.render(hs.camera, p)
    p.pipeline = api.RT_PIPELINE_MEGAKERNEL
    mega = scene.render(hs.camera, p)
    same = (wf == mega) | (np.isnan(wf) & np.isnan(mega))
    assert same.all(), f"{int((~same).any(axis=2).sum())} of {1200 * 1200} pixels differ"
    assert np.isfinite(wf[..., :3]).mean() > 0.99


def test_headline_config_rows_match_oracle(dev):
    """BASELINE.json's headline configuration itself (cornell_dragon, 1200x1200, 1000 spp = 10 replicas x 10x10
    strata, 871 200 triangles), three full rows of it (every 400th): the HIP path through the C ABI against the
    oracle at the f64 bar, and the same rows cut out of a render that owns more rows (keyed RNG: a pixel's
    samples do not depend on the partition)."""
    ensure_dragon()
    hs = api.HostScene(["scenes/cornell_dragon", "-w=1200", "-s=1000", "-t=10", "--seed=1"])
    assert hs.spp == 1000 and hs.params.thread_count == 10 and hs.params.sqrt_spt == 10
    scene = api.DeviceScene(hs.desc, 0)
    q = hs.params.copy()
    q.band_rows, q.n_parts, q.part = 1, 400, 123
    rows = api.owned_rows(hs.height, q)
    assert list(rows) == [123, 523, 923]
    gpu = scene.render(hs.camera, q)
    ref, st = pyoracle.render(hs.desc, hs.camera, q)
    assert st.samples == 3 * 1200 * 1000
    assert_f64_parity(gpu, ref)
    q2 = hs.params.copy()
    q2.band_rows, q2.n_parts, q2.part = 1, 200, 123   # rows 123, 323, 523, 723, 923, 1123
    wider = scene.render(hs.camera, q2)
    np.testing.assert_array_equal(wider[[0, 2, 4]], gpu)


def test_big_mesh_row_matches_oracle(dev):
    """The headline scene with the stand-in surface tessellated four times finer (3 484 800 triangles: BVH nodes + triangle
    records = 314 MB, beyond the 256 MB Infinity Cache; bench workload c4_3m is profiled on it): one full row through the mesh
    (1200 px, 100 spp) against the oracle's octree over the same triangles at the f64 bar, both schedulers bit-identical."""
    import bench
    bench.ensure_big_dragon("c4_3m")
    hs = api.HostScene(["build/bigmesh/cornell_dragon_3m", "-w=1200", "-s=100", "-t=1", "--seed=1"])
    assert hs.spp == 100 and hs.desc.contents.meshes[0].n_triangles == 3484800
    scene = api.DeviceScene(hs.desc, 0)
    q = hs.params.copy()
    q.band_rows, q.n_parts, q.part = 1, 1200, 640
    gpu = scene.render(hs.camera, q)
    ref, st = pyoracle.render(hs.desc, hs.camera, q)
    assert st.samples == 1200 * 100
    assert_f64_parity(gpu, ref)
    q.pipeline = api.RT_PIPELINE_MEGAKERNEL
    np.testing.assert_array_equal(scene.render(hs.camera, q), gpu)


def test_c5_config_rows_of_one_rank_match_oracle(dev):
    """BASELINE.json's multi-GPU configuration C5 (cornell_dragon, 2400x2400, 4000 spp = 10 replicas x 20x20 strata,
    frame row-tiled over 8 GPUs), the share of rank 3 exactly as `bench.py --gpus 8` partitions it
    (dist.partition_params: 4-row bands, 300 rows per rank).  (1) Two full rows of that share (1070 and 2222) against
    the oracle at the f64 bar.  (2) The rank's whole share (2.9 G samples): its per-sample buffer (69 GB) is over the
    64 GB budget, so this is the replica-group path on the real configuration; the two rows cut out of it must be the
    same bits."""
    from rust_raytracer_amd import dist as rtdist
    ensure_dragon()
    hs = api.HostScene(["scenes/cornell_dragon", "-w=2400", "-s=4000", "-t=10", "--seed=1"])
    assert (hs.width, hs.height, hs.spp) == (2400, 2400, 4000) and hs.params.sqrt_spt == 20 and hs.params.thread_count == 10
    scene = api.DeviceScene(hs.desc, 0)
    share = rtdist.partition_params(hs.params, 8, 3, hs.height)
    assert (share.band_rows, share.n_parts, share.part) == (4, 8, 3)
    share_rows = api.owned_rows(hs.height, share)
    assert len(share_rows) == 300 and share_rows == rtdist.rows_of_part(hs.height, 8, 3)
    q = hs.params.copy()
    q.band_rows, q.n_parts, q.part = 1, 1152, 1070
    rows = api.owned_rows(hs.height, q)
    assert list(rows) == [1070, 2222] and all(r in share_rows for r in rows)
    gpu = scene.render(hs.camera, q)
    ref, st = pyoracle.render(hs.desc, hs.camera, q)
    assert st.samples == 2 * 2400 * 4000
    assert_f64_parity(gpu, ref)
    full = scene.render(hs.camera, share)
    assert full.shape == (300, 2400, 4)
    assert scene.stats().n_replica_groups >= 2
    np.testing.assert_array_equal(full[[share_rows.index(r) for r in rows]], gpu)


@pytest.mark.parametrize("name", ["light_test", "default", "two_meshes", "texture_test"])
def test_device_built_bvh_renders_the_same_frame(dev, name):
    """the mesh BVH built on the GPU (LBVH, rt_bvh_device.hip) instead of the host's binned SAH.
    The tree only culls, so the frame is the host-built one bit for bit — and the oracle's at the f64 bar."""
    hs = api.HostScene(SCENES[name])
    host_built = api.DeviceScene(hs.desc, 0).render(hs.camera, hs.params)
    hs.desc.contents.flags |= api.RT_SCENE_BVH_ON_DEVICE
    try:
        device_built = api.DeviceScene(hs.desc, 0).render(hs.camera, hs.params)
    finally:
        hs.desc.contents.flags &= ~api.RT_SCENE_BVH_ON_DEVICE
    np.testing.assert_array_equal(device_built, host_built)
    ref, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
    assert_f64_parity(device_built, ref)


def test_emission_linearity_on_gpu(dev, tmp_path):
    base = ("@config output_width = 64\n@config aspect_ratio = 1\n@config camera_pos = 0,1,5\n@config camera_target = 0,1,0\n"
            "floor: plane 0,0,0 4,0,0 0,0,-4 (lambertian (constant 0.7,0.7,0.7))\n"
            "ball: sphere 0,1,0 1 (glossy (constant 0.3,0.4,0.8) (constant 0.2))\n"
            "lamp: sphere 2,3,2 0.5 (emissive (constant {e},{e},{e}))\nworld: list $floor $ball $lamp\nlights: list $lamp\n")
    frames = []
    for e in (8, 16):
        f = tmp_path / f"s{e}"
        f.write_text(base.format(e=e))
        hs = api.HostScene([str(f), "-s=16", "--seed=9"])
        frames.append(api.DeviceScene(hs.desc, 0).render(hs.camera, hs.params))
    np.testing.assert_array_equal(frames[1], 2.0 * frames[0])


# ---------------------------------------------------------------- edge cases and errors
def test_empty_world_and_empty_lights(dev, tmp_path):
    f = tmp_path / "empty"
    f.write_text("@config output_width = 20\nworld: list\nlights: list\n")
    hs = api.HostScene([str(f), "-s=4", "-b=0.25,0.5,1"])
    ref, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
    gpu = api.DeviceScene(hs.desc, 0).render(hs.camera, hs.params)
    assert np.all(gpu[..., :3] == np.array([0.25, 0.5, 1.0]))     # every ray misses: background (camera.rs:331)
    np.testing.assert_array_equal(gpu, ref)


def test_diffuse_with_empty_light_list(dev, tmp_path):
    """ObjectList::random on an empty list returns (1,0,0) without a draw and pdf_value is 0
    (list.rs:80-100): light-biased samples get weight 0 or NaN exactly like the reference."""
    f = tmp_path / "nolights"
    f.write_text("@config output_width = 24\n@config camera_pos = 0,1,4\n@config camera_target = 0,0.5,0\n"
                 "floor: plane 0,0,0 4,0,0 0,0,-4 (lambertian (constant 0.7,0.7,0.7))\nsky: sky (constant 1,1,1)\n"
                 "world: list $floor $sky\nlights: list\n")
    hs = api.HostScene([str(f), "-s=16", "--seed=3"])
    ref, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
    gpu = api.DeviceScene(hs.desc, 0).render(hs.camera, hs.params)
    assert_f64_parity(gpu, ref)


def test_tiny_and_ragged_images(dev):
    for args in (["tests/scenes/single_light", "-w=1", "-s=4"], ["tests/scenes/single_light", "-w=17", "-r=0.37", "-s=4"],
                 ["scenes/cornell", "-w=5", "-s=1"]):
        hs = api.HostScene(args)
        ref, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
        gpu = api.DeviceScene(hs.desc, 0).render(hs.camera, hs.params)
        assert gpu.shape == (hs.height, hs.width, 4)
        assert_f64_parity(gpu, ref)


def test_max_depth_one_and_zero(dev):
    for depth in (0, 1, 2):
        hs = api.HostScene(["scenes/cornell", "-w=24", "-s=4", f"--max-depth={depth}"])
        ref, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
        gpu = api.DeviceScene(hs.desc, 0).render(hs.camera, hs.params)
        assert_f64_parity(gpu, ref)
        if depth == 0:
            assert np.all(gpu == 0.0)       # ray_color(depth = 0) = black (camera.rs:290)


def test_error_codes(dev):
    hs = api.HostScene(SCENES["cornell"])
    scene = api.DeviceScene(hs.desc, 0)
    p = hs.params.copy()
    p.sqrt_spt = 0
    with pytest.raises(api.RtError) as e:
        scene.render(hs.camera, p)
    assert e.value.status == api.RT_E_INVALID
    p = hs.params.copy()
    p.band_rows, p.n_parts, p.part = 16, 2, 2
    with pytest.raises(api.RtError) as e:
        scene.render(hs.camera, p)
    assert e.value.status == api.RT_E_INVALID
    with pytest.raises(api.RtError) as e:
        api.DeviceScene(hs.desc, 99)
    assert e.value.status == api.RT_E_INVALID
    lib = api.load_device_lib()
    assert lib.rt_render(None, None, None, None) == api.RT_E_INVALID
    assert b"NULL" in lib.rt_last_error()
    bad = api.RtSceneDesc()
    C.memmove(C.byref(bad), hs.desc, C.sizeof(api.RtSceneDesc))
    bad.world_root = 10 ** 6
    h = C.c_void_p()
    assert lib.rt_scene_create(C.byref(bad), 0, C.byref(h)) == api.RT_E_INVALID
    bad.world_root = hs.desc.contents.world_root
    bad.abi_version = 77
    assert lib.rt_scene_create(C.byref(bad), 0, C.byref(h)) == api.RT_E_INVALID


def test_render_device_keeps_frame_in_hbm(dev):
    """rt_render_device writes into caller-owned device memory on the caller's stream."""
    import torch
    hs = api.HostScene(SCENES["cornell"])
    scene = api.DeviceScene(hs.desc, 0)
    out = torch.full((hs.height, hs.width, 4), -1.0, dtype=torch.float64, device="cuda:0")
    stream = torch.cuda.Stream(device="cuda:0")
    with torch.cuda.stream(stream):
        scene.render_device(hs.camera, hs.params, out.data_ptr(), stream.cuda_stream)
    stream.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), scene.render(hs.camera, hs.params))


def test_specular_reflection_shortcut_is_bit_exact(dev):
    """Metal / Glossy reflect into `reflected + random_unit * fuzz * |reflected|`.  For fuzz == 0 the kernels return `reflected`
    and move the generator on by the six draws of the three normal samples (rt_device.h fuzzy_reflection) - unless a component
    of `reflected` is a zero (the sum can turn -0 into +0) or so large that the length overflows (0 * inf = NaN).  The device
    probe evaluates the routine and the plain expression on the same inputs: same bits, same generator state, for random and
    for special operands."""
    import ctypes as C
    lib = api.load_device_lib()
    lib.rt_debug_fuzzy_reflection.argtypes = [C.c_int, C.c_uint32] + [C.c_void_p] * 5
    lib.rt_debug_fuzzy_reflection.restype = C.c_int
    rng = np.random.default_rng(7)
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 1e-320, -1e-320, 1e149, 1e151, -1e151, 1e308, np.inf, -np.inf, np.nan, 3.7e-9])
    grid = np.stack(np.meshgrid(special, special, special, indexing="ij"), axis=-1).reshape(-1, 3)
    rand = rng.standard_normal((200000, 3)) * 10.0 ** rng.integers(-6, 6, (200000, 1))
    reflected = np.ascontiguousarray(np.concatenate([grid, grid, rand, rand[:20000]]), dtype=np.float64)
    fuzz = np.concatenate([np.zeros(len(grid)), np.full(len(grid), -0.0), np.zeros(len(rand)), rng.uniform(0.0, 1.0, 20000)])
    fuzz[len(grid) * 2 + 100000:len(grid) * 2 + 100050] = np.array([0.3, 1.0, 1e-300, np.nan, np.inf] * 10)
    n = len(reflected)
    state = rng.integers(0, 2 ** 63, n, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    out = np.zeros((n, 6), dtype=np.float64)
    state_out = np.zeros((n, 2), dtype=np.uint64)
    st = lib.rt_debug_fuzzy_reflection(0, n, reflected.ctypes.data, np.ascontiguousarray(fuzz).ctypes.data, state.ctypes.data,
                                       out.ctypes.data, state_out.ctypes.data)
    assert st == api.RT_OK, lib.rt_last_error().decode()
    a, b = out[:, :3], out[:, 3:]
    same = (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))
    assert same.all(), f"{int((~same).any(axis=1).sum())} of {n} directions differ, first: {reflected[np.flatnonzero((~same).any(axis=1))[0]]}"
    assert (state_out[:, 0] == state_out[:, 1]).all()
    # the shortcut was taken where it may be: the generator moved on although the direction is `reflected` itself
    plain = (fuzz == 0) & np.isfinite(reflected).all(axis=1) & (reflected != 0).all(axis=1) & (np.abs(reflected) < 1e150).all(axis=1)
    assert plain.sum() > 200000 and (a[plain].view(np.uint64) == reflected[plain].view(np.uint64)).all()
    six_draws = np.uint64((6 * 0x9E3779B97F4A7C15) % 2 ** 64)  # SplitMix64 adds its increment once per draw (wrapping)
    assert (state_out[:, 0] == state + six_draws).all()


@pytest.mark.parametrize("name", ["cornell", "light_test", "two_meshes", "cornell_smoke", "default"])
def test_tail_compaction_does_not_change_the_frame(dev, name, monkeypatch):
    """When a replica group's samples have all been started, the wavefront scheduler moves the surviving paths together
    (k_wf_compact: into slots 0 .. n-1 of a second pool, whenever fewer than half the addressed slots are alive), so that the last
    iterations keep unit-stride accesses.  A path carries its generator and sample index with it: the frame must be the one
    without compaction and the megakernel's, bit for bit - with the default pool, with a small one (several compactions, every
    kernel in both addressing modes) and with replica groups (a tail per group, back to the first pool each time)."""
    hs = api.HostScene(SCENES[name][:-2] + ["-s=64", "-t=3", "--seed=31"])
    scene = api.DeviceScene(hs.desc, 0)
    p = hs.params.copy()
    p.pipeline = api.RT_PIPELINE_MEGAKERNEL
    mega = scene.render(hs.camera, p)
    p.pipeline = api.RT_PIPELINE_WAVEFRONT
    for pool, sample_gb in (("0", None), ("16384", None), ("3000", "0")):
        frames = {}
        for compact in ("1", "0"):
            monkeypatch.setenv("RT_WF_COMPACT", compact)
            monkeypatch.setenv("RT_WF_COMPACT_MIN", "64")
            if pool != "0":
                monkeypatch.setenv("RT_WF_POOL", pool)
            if sample_gb is not None:
                monkeypatch.setenv("RT_WF_SAMPLE_GB", sample_gb)
            frames[compact] = scene.render(hs.camera, p)
            st = scene.stats()
            if compact == "1":
                assert st.n_tail_compactions >= (2 if pool != "0" else 1), (pool, st.n_tail_compactions)
                if sample_gb is not None:
                    assert st.n_replica_groups == 3 and st.n_tail_compactions >= 3
            else:
                assert st.n_tail_compactions == 0
        for compact, wf in frames.items():
            same = (wf == mega) | (np.isnan(wf) & np.isnan(mega))
            assert same.all(), f"pool {pool}, RT_WF_COMPACT={compact}: {int((~same).any(axis=2).sum())} pixels differ from the megakernel"
        monkeypatch.delenv("RT_WF_POOL", raising=False)
        monkeypatch.delenv("RT_WF_SAMPLE_GB", raising=False)


def test_a_pool_that_does_not_fit_is_halved(dev, monkeypatch):
    """The path pool is sized for speed (128 M slots at the headline's work).  When device memory is short - other scenes of a
    frame pipeline, other processes on the card - the scheduler halves it until it fits instead of failing; the frame is the
    same (RT_WF_FAKE_OOM_ABOVE makes every pool above that many slots fail like hipErrorOutOfMemory)."""
    hs = api.HostScene(["scenes/cornell", "-w=400", "-s=64", "--seed=3"])
    scene = api.DeviceScene(hs.desc, 0)
    p = hs.params.copy()
    p.pipeline = api.RT_PIPELINE_WAVEFRONT
    ref = scene.render(hs.camera, p)
    it_ref = scene.stats().n_iterations
    monkeypatch.setenv("RT_WF_FAKE_OOM_ABOVE", str(1 << 21))
    scene2 = api.DeviceScene(hs.desc, 0)
    out = scene2.render(hs.camera, p)
    assert scene2.stats().n_iterations > it_ref        # a smaller pool: more iterations
    assert ((out == ref) | (np.isnan(out) & np.isnan(ref))).all()
    monkeypatch.setenv("RT_WF_POOL", str(1 << 23))      # an explicit size is taken as given: the error surfaces
    scene3 = api.DeviceScene(hs.desc, 0)
    with pytest.raises(api.RtError) as e:
        scene3.render(hs.camera, p)
    assert e.value.status == api.RT_E_NOMEM
