"""Known-answer tests that pin the CPU oracle (oracle/oracle.cpp) to the reference's formulas.

The reference ships no tests and cannot be built here, so these values are derived by hand
from the cited source lines.  CPU only."""
import math

import numpy as np
import pytest

from oracle import pyoracle
from rust_raytracer_amd import api

INF = float("inf")


def scene(*args):
    return api.HostScene(list(args))


def dsl(tmp_path, text, *args, name="s"):
    f = tmp_path / name
    f.write_text(text)
    return scene(str(f), *args)


# ---------------------------------------------------------------- aabb.rs:50-87
@pytest.mark.parametrize("origin,dir,tmin,tmax,expect", [
    ((0, 0, -3), (0, 0, 1), 0.001, INF, True),       # straight through
    ((0, 0, -3), (0, 0, -1), 0.001, INF, False),     # pointing away: t_max < 0
    ((0, 0, -3), (0, 0, 1), 0.001, 1.5, False),      # interval ends before the box (t_min = 2)
    ((0, 0, -3), (0, 0, 1), 4.5, INF, False),        # interval starts after the box (t_max = 4)
    ((2, 0, -3), (0, 0, 1), 0.001, INF, False),      # parallel, outside the x slab
    ((0, 0, 0), (1, 1, 1), 0.001, INF, True),        # origin inside
    ((-3, -3, -3), (1, 1, 1), 0.001, INF, True),     # diagonal
    ((-3, -3, -3), (1, 1, 0.2), 0.001, INF, False),  # misses in z
])
def test_slab_truth_table(origin, dir, tmin, tmax, expect):
    assert pyoracle.test_bounding_box([-1, -1, -1, 1, 1, 1], origin, dir, tmin, tmax) == expect


def test_slab_inverted_box_never_hit():
    """Negative-radius spheres get inverted bounds (sphere.rs:28-29); alone in a BVH node they
    are unreachable (SURVEY B-8)."""
    assert not pyoracle.test_bounding_box([1, 1, 1, -1, -1, -1], (0, 0, -3), (0, 0, 1), 0.001, INF)


# ---------------------------------------------------------------- utils.rs, vec4.rs
def test_reflectance_schlick():
    assert pyoracle.reflectance(1.0, 1 / 1.5) == pytest.approx(0.04, rel=1e-15)
    assert pyoracle.reflectance(0.5, 1 / 1.5) == pytest.approx(0.07, rel=1e-14)
    assert pyoracle.reflectance(0.0, 1 / 1.5) == pytest.approx(1.0, rel=1e-15)


def test_onb_from_vec():
    u, v, w = pyoracle.onb_from_vec((0, 0, 1))
    np.testing.assert_array_equal(u, [-1, 0, 0])
    np.testing.assert_array_equal(v, [0, 1, 0])
    np.testing.assert_array_equal(w, [0, 0, 1])
    u, v, w = pyoracle.onb_from_vec((1, 0, 0))      # |x| > 0.9 picks a = (0,1,0)
    np.testing.assert_array_equal(v, [0, 0, 1])
    np.testing.assert_array_equal(u, [0, -1, 0])
    u, v, w = pyoracle.onb_from_vec((0, 0, 2))      # NOT normalised: u scales with |w| (sphere.rs:125 quirk)
    np.testing.assert_array_equal(u, [-2, 0, 0])
    np.testing.assert_array_equal(v, [0, 1, 0])


def test_refract_snell():
    s = math.sin(math.radians(30))
    v = (s, -math.cos(math.radians(30)), 0)
    r = pyoracle.refract(v, (0, 1, 0), 1 / 1.5)
    assert np.linalg.norm(r) == pytest.approx(1.0, rel=1e-14)
    assert r[0] == pytest.approx(s / 1.5, rel=1e-14) and r[1] < 0


# ---------------------------------------------------------------- sphere.rs:40-94
def test_sphere_hit_known_answer(tmp_path):
    hs = dsl(tmp_path, "m: lambertian (constant 1,1,1)\ns: sphere 0,0,0 1 $m\nworld: list $s\nlights: list $s\n")
    h = pyoracle.world_hit(hs.desc, (0, 0, -3), (0, 0, 1))
    assert h["t"] == 2.0 and h["front_face"]
    np.testing.assert_array_equal(h["normal"], [0, 0, -1])
    assert h["uv"] == (0.75, 0.5)                                  # sphere.rs:69-75
    # un-normalised direction: t scales inversely (SURVEY B-3)
    assert pyoracle.world_hit(hs.desc, (0, 0, -3), (0, 0, 4))["t"] == 0.5
    # from inside: far root, normal flipped to face the ray (object.rs:55-60)
    h = pyoracle.world_hit(hs.desc, (0, 0, 0), (0, 0, 1))
    assert h["t"] == 1.0 and not h["front_face"]
    np.testing.assert_array_equal(h["normal"], [0, 0, -1])
    assert pyoracle.world_hit(hs.desc, (0, 2, -3), (0, 0, 1)) is None
    # t_min is exclusive in ray-parameter units: a hit at exactly t = 2 is rejected by Interval(2, inf)
    assert pyoracle.world_hit(hs.desc, (0, 0, -3), (0, 0, 1), t_min=2.0)["t"] == 4.0


def test_sphere_light_pdf(tmp_path):
    hs = dsl(tmp_path, "m: emissive (constant 1,1,1)\ns: sphere 0,0,0 1 $m\nworld: list $s\nlights: list $s\n")
    # sphere.rs:106-121: 1 / (2 pi (1 - sqrt(1 - r^2/d^2))) when the ray hits, else 0
    expect = 1.0 / (2 * math.pi * (1 - math.sqrt(1 - 1 / 9)))
    assert pyoracle.lights_pdf_value(hs.desc, (0, 0, -3), (0, 0, 1)) == pytest.approx(expect, rel=1e-14)
    assert pyoracle.lights_pdf_value(hs.desc, (0, 0, -3), (0, 1, 0)) == 0.0
    d = pyoracle.lights_random(hs.desc, (0, 0, -3), seed=3, n=2000)
    # every sampled direction hits the sphere (cone sampling, sphere.rs:123-145; |dir|=3 is a unit ONB axis here)
    cosang = d[:, 2] / np.linalg.norm(d, axis=1)
    assert np.all(cosang >= math.sqrt(1 - 1 / 9) - 1e-12)


# ---------------------------------------------------------------- plane.rs
PLANE = "m: emissive (constant 1,1,1)\np: plane 0,0,0 1,0,0 0,0,-1 $m{flag}\nworld: list $p\nlights: list $p\n"


def test_plane_hit_uv_and_one_sidedness(tmp_path):
    hs = dsl(tmp_path, PLANE.format(flag=""))
    h = pyoracle.world_hit(hs.desc, (0.25, 2, -0.25), (0, -1, 0))
    assert h["t"] == 2.0 and h["front_face"]
    np.testing.assert_array_equal(h["normal"], [0, 1, 0])           # normalize(u x v)
    assert h["uv"] == (0.625, 0.625)                                # inv_u = u_unit*0.5/|u| from corner c-u-v
    assert pyoracle.world_hit(hs.desc, (0.25, -2, -0.25), (0, 1, 0)) is None   # culled from behind (plane.rs:69-76)
    assert pyoracle.world_hit(hs.desc, (1.5, 2, 0), (0, -1, 0)) is None        # outside the quad
    hs2 = dsl(tmp_path, PLANE.format(flag=" backface"), name="s2")
    h = pyoracle.world_hit(hs2.desc, (0.25, -2, -0.25), (0, 1, 0))
    assert h["t"] == 2.0 and not h["front_face"]
    np.testing.assert_array_equal(h["normal"], [0, -1, 0])


def test_plane_pdf_and_quarter_sampling(tmp_path):
    hs = dsl(tmp_path, PLANE.format(flag=""))
    # plane.rs:107-118 with area = 4|u x v| = 4 (plane.rs:38): t^2 |d|^2 / (|cos| area)
    assert pyoracle.lights_pdf_value(hs.desc, (0, 2, 0), (0, -1, 0)) == 1.0
    assert pyoracle.lights_pdf_value(hs.desc, (0, 2, 0), (0, -2, 0)) == 1.0          # t = 1, |d|^2 = 4
    assert pyoracle.lights_pdf_value(hs.desc, (0, -2, 0), (0, 1, 0)) == 0.0          # one-sided
    d = pyoracle.lights_random(hs.desc, (0, 2, 0), seed=5, n=4000)
    pts = d + np.array([0, 2, 0])
    # Quirk B-1: `corner + u*U + v*V` with half-vectors covers ONE QUARTER of the quad:
    # corner = (-1,0,1), so x in [-1,0], z in [0,1] — never the other three quarters.
    assert np.all((pts[:, 0] >= -1) & (pts[:, 0] <= 0) & (pts[:, 2] >= 0) & (pts[:, 2] <= 1))
    assert np.allclose(pts[:, 1], 0)
    assert abs(pts[:, 0].mean() + 0.5) < 0.03 and abs(pts[:, 2].mean() - 0.5) < 0.03


def test_light_list_mean_pdf_and_non_light_members(tmp_path):
    hs = dsl(tmp_path, "e: emissive (constant 1,1,1)\np: plane 0,0,0 1,0,0 0,0,-1 $e\ns: sphere 5,5,5 1 $e\n"
                       "t: transform $s t=1,0,0\nworld: list $p $s\nlights: list $p $s $t\n")
    # list.rs:80-89: mean over members; Transform::pdf_value is 0 (transform.rs:141-143)
    assert pyoracle.lights_pdf_value(hs.desc, (0, 2, 0), (0, -1, 0)) == pytest.approx(1.0 / 3.0, rel=1e-15)


# ---------------------------------------------------------------- mesh.rs:62-163
TRI_OBJ = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nvn 0 0 1\nvn 0 0 1\nf 1//1 2//2 3//3\n"


def test_triangle_moller_trumbore_and_backface_cull(tmp_path):
    (tmp_path / "t.obj").write_text(TRI_OBJ)
    hs = dsl(tmp_path, "m: lambertian (constant 1,1,1)\nt: mesh t.obj $m\nworld: list $t\nlights: list $t\n")
    h = pyoracle.world_hit(hs.desc, (0.25, 0.25, 1), (0, 0, -1))
    assert h["t"] == 1.0 and h["front_face"]                       # det = +1
    np.testing.assert_allclose(h["pos"], [0.25, 0.25, 0], atol=1e-15)
    np.testing.assert_array_equal(h["normal"], [0, 0, 1])
    assert pyoracle.world_hit(hs.desc, (0.25, 0.25, -1), (0, 0, 1)) is None   # det = -1 < EPSILON: culled
    assert pyoracle.world_hit(hs.desc, (0.75, 0.75, 1), (0, 0, -1)) is None   # u + v > 1
    # edge is inclusive: u = 0 exactly
    assert pyoracle.world_hit(hs.desc, (0.0, 0.5, 1), (0, 0, -1)) is not None


def test_transform_preserves_t_and_uses_M_for_normals(tmp_path):
    (tmp_path / "t.obj").write_text(TRI_OBJ)
    hs = dsl(tmp_path, "m: lambertian (constant 1,1,1)\nt: transform (mesh t.obj $m) s=2,2,1 t=0,0,5\n"
                       "world: list $t\nlights: list $t\n")
    h = pyoracle.world_hit(hs.desc, (0.5, 0.5, 8), (0, 0, -1))
    assert h["t"] == 3.0                                            # ray not re-normalised (transform.rs:124-127)
    np.testing.assert_allclose(h["pos"], [0.5, 0.5, 5.0], atol=1e-15)
    np.testing.assert_allclose(h["normal"], [0, 0, 1], atol=1e-15)  # normalize(M * n) (transform.rs:133)


# ---------------------------------------------------------------- octree.rs + aabb.rs bounds
def test_suzanne_octree_matches_reference_rule():
    hs = scene("scenes/light_test")
    st = pyoracle.octree_stats(hs.desc, 0)
    assert st == {"branches": 391, "leaves": 2738, "empty_leaves": 604, "refs": 42378, "max_depth": 7, "max_leaf": 50}
    d = hs.desc.contents
    mesh_node = [i for i in range(d.n_nodes) if d.nodes[i].type == api.RT_NODE_MESH][0]
    b = pyoracle.node_bounds(hs.desc, mesh_node)
    np.testing.assert_allclose(b, [-1.329186, -0.972822, -0.779266, 1.329186, 0.940236, 0.823441], atol=1e-6)


@pytest.mark.parametrize("args", [["scenes/cornell"], ["scenes/light_test"], ["tests/scenes/nested_transform"],
                                  ["tests/scenes/bvh_spheres"]])
def test_host_bounds_equal_oracle_bounds(args):
    """Two independent restatements of the bounding-box rules (host builder, oracle) must agree
    bit for bit, including the cumulative 0.001 padding of list.add (list.rs:52) and the
    w = -1 max-corner quirk in Transform::update_bounds (transform.rs:98-118)."""
    hs = scene(*args)
    d = hs.desc.contents
    reachable = 0
    for i in range(d.n_nodes):
        try:
            ob = pyoracle.node_bounds(hs.desc, i)
        except api.RtError:
            continue
        reachable += 1
        np.testing.assert_array_equal(ob, np.array(list(d.nodes[i].bounds)), err_msg=f"node {i}")
    assert reachable >= 5


def test_transform_bounds_quirk_is_reproduced():
    """scenes/cornell box: `t=82.5,165,82.5 ry=18 t=265,0,295`.  The max corner of the inner
    list's box carries w = -1, so the translation is SUBTRACTED for that corner: the resulting
    box reaches x = -264.99 although the box itself lives at x in [130, 423]."""
    hs = scene("scenes/cornell")
    d = hs.desc.contents
    t = [i for i in range(d.n_nodes) if d.nodes[i].type == api.RT_NODE_TRANSFORM][0]
    assert d.nodes[t].bounds[0] == pytest.approx(-264.99217948542525, rel=1e-12)


# ---------------------------------------------------------------- camera.rs:260-349
def test_get_ray_stratification_and_defocus_ring():
    hs = scene("scenes/light_test", "-w=60", "-s=16")      # f_number = 4: aperture on
    cam, p = hs.camera, hs.params
    S = p.sqrt_spt
    fp, du, dv = (np.array(list(x)) for x in (cam.first_pixel, cam.pixel_delta_u, cam.pixel_delta_v))
    pos, bu, bv = (np.array(list(x)) for x in (cam.position, cam.basis_u, cam.basis_v))
    for (x, y, sx, sy) in [(0, 0, 0, 0), (59, 39, 3, 3), (17, 5, 2, 1)]:
        r = pyoracle.get_ray(cam, p, 0, x, y, sx, sy)
        o, d = r[:3], r[3:]
        # Quirk B-2: the lens sample lies ON the aperture circle (normalised 2-D Gaussian).
        # The camera basis u = v_up x w is NOT normalised (camera.rs:102): |u| = |v| = sine of the
        # angle between the view direction and +y, so the circle's radius is aperture_radius * |u|.
        rel = o - pos
        assert abs(np.linalg.norm(bu) - np.linalg.norm(bv)) < 1e-15 and abs(bu @ bv) < 1e-15
        assert np.linalg.norm(rel) == pytest.approx(cam.aperture_radius * np.linalg.norm(bu), rel=1e-12)
        # the target point is inside stratum (sx, sy) of pixel (x, y)
        target = o + d
        off = target - (fp + du * x + dv * y)
        a = (off @ du) / (du @ du) + 0.5
        b = (off @ dv) / (dv @ dv) + 0.5
        assert sx / S <= a <= (sx + 1) / S and sy / S <= b <= (sy + 1) / S


# ---------------------------------------------------------------- RNG stream
def test_rng_is_keyed_and_reproducible():
    a = pyoracle.rng_raw(1, 0, 0, 0, 4)
    assert a == pyoracle.rng_raw(1, 0, 0, 0, 4)
    # golden values of the keyed SplitMix64 stream (pins oracle and kernels to the same generator)
    assert a == GOLDEN_RNG
    for other in [(2, 0, 0, 0), (1, 1, 0, 0), (1, 0, 1, 0), (1, 0, 0, 1)]:
        assert pyoracle.rng_raw(*other, 4) != a
    u = pyoracle.rng_uniforms(7, 3, 12345, 9, 20000)
    assert 0.0 <= u.min() and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005


def _splitmix_reference(seed, tid, pixel, stratum, n):
    M = (1 << 64) - 1

    def mix(z):
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)
    k = mix((seed + 0x9E3779B97F4A7C15 * (tid + 1)) & M)
    k = mix(k ^ ((pixel * 0xD1B54A32D192ED03 + 0x8CB92BA72F3D8DD7) & M))
    k = mix(k ^ ((stratum * 0xA0761D6478BD642F + 0xE7037ED1A0B428DB) & M))
    out = []
    for _ in range(n):
        k = (k + 0x9E3779B97F4A7C15) & M
        out.append(mix(k))
    return out


GOLDEN_RNG = _splitmix_reference(1, 0, 0, 0, 4)


def test_rng_matches_independent_python_implementation():
    for key in [(1, 0, 0, 0), (99, 7, 1439999, 99), (2**63, 3, 5, 8)]:
        assert pyoracle.rng_raw(*key, 6) == _splitmix_reference(*key, 6)


# ---------------------------------------------------------------- end-to-end, RNG independent
def test_sky_only_scene_is_exact():
    """Every sample returns exactly L (front-face emissive at t = inf): pixels are bit-exact for
    dyadic L whatever the RNG does."""
    hs = scene("tests/scenes/sky_only", "-s=16")
    img, st = pyoracle.render(hs.desc, hs.camera, hs.params)
    assert np.all(img[..., 0] == 0.5) and np.all(img[..., 1] == 1.0) and np.all(img[..., 2] == 2.0)
    assert np.all(img[..., 3] == 0.0)
    assert st.rays == hs.width * hs.height * hs.spp


def test_replicas_and_row_partition(tmp_path):
    hs = scene("tests/scenes/single_light", "-w=24", "-s=32", "-t=2")     # 2 replicas x 4x4 strata
    assert hs.spp == 32
    full, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
    p = hs.params.copy()
    p.band_rows, p.n_parts = 4, 3
    parts = []
    for part in range(3):
        p.part = part
        img, _ = pyoracle.render(hs.desc, hs.camera, p)
        rows = api.owned_rows(hs.height, p)
        assert img.shape[0] == len(rows)
        parts.append((rows, img))
    rebuilt = np.empty_like(full)
    for rows, img in parts:
        rebuilt[rows] = img
    np.testing.assert_array_equal(rebuilt, full)      # samples are keyed by global pixel index


def test_furnace_lambertian_sphere_mean(tmp_path):
    """Convex Lambertian sphere of albedo rho under a uniform sky L: E[pixel] = rho * L
    (no inter-reflection).  Checks CosinePDF, MixPDF weighting, sky pdf 1/4pi and
    scattering_pdf together (camera.rs:298-315)."""
    hs = dsl(tmp_path, "@config output_width = 16\n@config aspect_ratio = 1\n@config focal_length = 400\n"
                       "@config camera_pos = 0,0,10\n@config camera_target = 0,0,0\n"
                       "s: sphere 0,0,0 1 (lambertian (constant 0.5,0.25,0.8))\nsky: sky (constant 2,2,2)\n"
                       "world: list $s $sky\nlights: list $sky\n", "-s=256", "--seed=4")
    img, _ = pyoracle.render(hs.desc, hs.camera, hs.params)
    n = hs.width * hs.height * hs.spp
    mean = img[..., :3].mean(axis=(0, 1))
    expect = np.array([0.5, 0.25, 0.8]) * 2.0
    # per-sample sd is below ~1.5x the mean for this estimator; 5 sigma bound
    np.testing.assert_allclose(mean, expect, rtol=5 * 1.5 / math.sqrt(n))


def test_emission_is_linear(tmp_path):
    """Scaling every emitter by 2 scales the image by exactly 2 (power of two: exact in IEEE)."""
    base = ("@config output_width = 24\n@config aspect_ratio = 1\n@config camera_pos = 0,1,5\n@config camera_target = 0,1,0\n"
            "floor: plane 0,0,0 4,0,0 0,0,-4 (lambertian (constant 0.7,0.7,0.7))\n"
            "ball: sphere 0,1,0 1 (glossy (constant 0.3,0.4,0.8) (constant 0.2))\n"
            "lamp: sphere 2,3,2 0.5 (emissive (constant {e},{e},{e}))\nworld: list $floor $ball $lamp\nlights: list $lamp\n")
    a, _ = pyoracle.render(*_dp(dsl(tmp_path, base.format(e=8), "-s=16", "--seed=9", name="a")))
    b, _ = pyoracle.render(*_dp(dsl(tmp_path, base.format(e=16), "-s=16", "--seed=9", name="b")))
    np.testing.assert_array_equal(b, 2.0 * a)


def _dp(hs):
    _dp.keep = getattr(_dp, "keep", []) + [hs]
    return hs.desc, hs.camera, hs.params
