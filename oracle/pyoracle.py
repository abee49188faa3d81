"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE, NOT PRODUCT: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by anything under rust_raytracer_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from rust_raytracer_amd import api

ORACLE_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")


class OracleStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("node_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("prim_tests", C.c_uint64), ("samples", C.c_uint64), ("seconds", C.c_double),
                ("os_threads", C.c_uint32), ("_pad", C.c_uint32)]


_lib = None


def build() -> None:
    subprocess.run(["make", "-C", ORACLE_DIR, "liboracle.so"], check=True, capture_output=True)


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        dp = C.POINTER(C.c_double)
        lib.oracle_render.argtypes = [C.POINTER(api.RtSceneDesc), C.POINTER(api.RtCameraDesc),
                                      C.POINTER(api.RtRenderParams), C.c_void_p, C.POINTER(OracleStats)]
        lib.oracle_render.restype = C.c_int
        lib.oracle_node_bounds.argtypes = [C.POINTER(api.RtSceneDesc), C.c_uint32, dp]
        lib.oracle_node_bounds.restype = C.c_int
        lib.oracle_octree_stats.argtypes = [C.POINTER(api.RtSceneDesc), C.c_uint32, C.POINTER(C.c_uint64)]
        lib.oracle_octree_stats.restype = C.c_int
        lib.oracle_test_bounding_box.argtypes = [dp, dp, dp, C.c_double, C.c_double]
        lib.oracle_test_bounding_box.restype = C.c_int
        lib.oracle_world_hit.argtypes = [C.POINTER(api.RtSceneDesc), dp, dp, C.c_double, C.c_double, dp]
        lib.oracle_world_hit.restype = C.c_int
        lib.oracle_lights_pdf_value.argtypes = [C.POINTER(api.RtSceneDesc), dp, dp, dp]
        lib.oracle_lights_pdf_value.restype = C.c_int
        lib.oracle_lights_random.argtypes = [C.POINTER(api.RtSceneDesc), dp, C.c_uint64, C.c_uint32, dp]
        lib.oracle_lights_random.restype = C.c_int
        lib.oracle_detmath.argtypes = [C.c_double, dp]
        lib.oracle_detmath.restype = None
        lib.oracle_detmath_inv.argtypes = [C.c_double, C.c_double, dp]
        lib.oracle_detmath_inv.restype = None
        lib.oracle_texture_sample.argtypes = [C.POINTER(api.RtSceneDesc), C.c_uint32, C.c_double, C.c_double, dp, dp]
        lib.oracle_texture_sample.restype = C.c_int
        lib.oracle_reflectance.argtypes = [C.c_double, C.c_double]
        lib.oracle_reflectance.restype = C.c_double
        lib.oracle_onb_from_vec.argtypes = [dp, dp]
        lib.oracle_onb_from_vec.restype = None
        lib.oracle_refract.argtypes = [dp, dp, C.c_double, dp]
        lib.oracle_refract.restype = None
        lib.oracle_rng_uniforms.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, dp]
        lib.oracle_rng_uniforms.restype = None
        lib.oracle_rng_raw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32,
                                       C.POINTER(C.c_uint64)]
        lib.oracle_rng_raw.restype = None
        lib.oracle_get_ray.argtypes = [C.POINTER(api.RtCameraDesc), C.POINTER(api.RtRenderParams), C.c_uint32,
                                       C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, dp]
        lib.oracle_get_ray.restype = None
        lib.oracle_trace_sample.argtypes = [C.POINTER(api.RtSceneDesc), C.POINTER(api.RtCameraDesc),
                                            C.POINTER(api.RtRenderParams), C.c_uint32, C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.c_uint32, dp, dp, C.c_uint32]
        lib.oracle_trace_sample.restype = C.c_int
        lib.oracle_last_error.argtypes = []
        lib.oracle_last_error.restype = C.c_char_p
        _lib = lib
    return _lib


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def render(desc, camera: api.RtCameraDesc, params: api.RtRenderParams):
    """Returns (frame[(rows, W, 4) f64], OracleStats)."""
    lib = load()
    rows = len(api.owned_rows(camera.image_height, params))
    out = np.empty((rows, camera.image_width, 4), dtype=np.float64)
    stats = OracleStats()
    st = lib.oracle_render(desc, C.byref(camera), C.byref(params), out.ctypes.data, C.byref(stats))
    if st != 0:
        raise api.RtError(st, lib.oracle_last_error().decode())
    return out, stats


def node_bounds(desc, node: int) -> np.ndarray:
    lib = load()
    out = (C.c_double * 6)()
    st = lib.oracle_node_bounds(desc, node, out)
    if st != 0:
        raise api.RtError(st, lib.oracle_last_error().decode())
    return np.array(list(out))


def octree_stats(desc, mesh: int) -> dict:
    lib = load()
    out = (C.c_uint64 * 6)()
    st = lib.oracle_octree_stats(desc, mesh, out)
    if st != 0:
        raise api.RtError(st, lib.oracle_last_error().decode())
    keys = ["branches", "leaves", "empty_leaves", "refs", "max_depth", "max_leaf"]
    return dict(zip(keys, [int(x) for x in out]))


def test_bounding_box(bounds6, origin, dir, t_min, t_max) -> bool:
    lib = load()
    return bool(lib.oracle_test_bounding_box((C.c_double * 6)(*bounds6), _d3(origin), _d3(dir), t_min, t_max))


def world_hit(desc, origin, dir, t_min=0.001, t_max=float("inf")):
    lib = load()
    out = (C.c_double * 11)()
    r = lib.oracle_world_hit(desc, _d3(origin), _d3(dir), t_min, t_max, out)
    if r < 0:
        raise api.RtError(r, lib.oracle_last_error().decode())
    if r == 0:
        return None
    o = list(out)
    return {"t": o[0], "pos": o[1:4], "normal": o[4:7], "uv": (o[7], o[8]), "front_face": bool(o[9]),
            "material": int(o[10])}


def lights_pdf_value(desc, origin, dir) -> float:
    lib = load()
    out = C.c_double()
    st = lib.oracle_lights_pdf_value(desc, _d3(origin), _d3(dir), C.byref(out))
    if st != 0:
        raise api.RtError(st, lib.oracle_last_error().decode())
    return out.value


def rng_uniforms(seed, tid, pixel, stratum, n) -> np.ndarray:
    lib = load()
    out = (C.c_double * n)()
    lib.oracle_rng_uniforms(seed, tid, pixel, stratum, n, out)
    return np.array(list(out))


def rng_raw(seed, tid, pixel, stratum, n) -> list:
    lib = load()
    out = (C.c_uint64 * n)()
    lib.oracle_rng_raw(seed, tid, pixel, stratum, n, out)
    return [int(x) for x in out]


def get_ray(camera, params, tid, x, y, sx, sy) -> np.ndarray:
    lib = load()
    out = (C.c_double * 6)()
    lib.oracle_get_ray(C.byref(camera), C.byref(params), tid, x, y, sx, sy, out)
    return np.array(list(out))


def trace_sample(desc, camera, params, tid, x, y, sx, sy, max_bounces=64):
    """(rgb, trace[n, 17]) of one sample; see oracle.h oracle_trace_sample."""
    lib = load()
    rgb = (C.c_double * 3)()
    tr = (C.c_double * (17 * max_bounces))()
    n = lib.oracle_trace_sample(desc, C.byref(camera), C.byref(params), tid, x, y, sx, sy, rgb, tr, max_bounces)
    if n < 0:
        raise api.RtError(n, lib.oracle_last_error().decode())
    return np.array(list(rgb)), np.array(list(tr)).reshape(max_bounces, 17)[:min(n, max_bounces)]


def lights_random(desc, origin, seed, n) -> np.ndarray:
    lib = load()
    out = (C.c_double * (3 * n))()
    st = lib.oracle_lights_random(desc, _d3(origin), seed, n, out)
    if st != 0:
        raise api.RtError(st, lib.oracle_last_error().decode())
    return np.array(list(out)).reshape(n, 3)


def texture_sample(desc, tex, u, v, p=(0.0, 0.0, 0.0)) -> np.ndarray:
    lib = load()
    out = (C.c_double * 3)()
    st = lib.oracle_texture_sample(desc, tex, u, v, _d3(p), out)
    if st != 0:
        raise api.RtError(st, lib.oracle_last_error().decode())
    return np.array(list(out))


def reflectance(cos_theta, ior_ratio) -> float:
    return load().oracle_reflectance(cos_theta, ior_ratio)


def onb_from_vec(w) -> np.ndarray:
    out = (C.c_double * 9)()
    load().oracle_onb_from_vec(_d3(w), out)
    return np.array(list(out)).reshape(3, 3)  # rows = columns u, v, w


def refract(v, n, ior_ratio) -> np.ndarray:
    out = (C.c_double * 3)()
    load().oracle_refract(_d3(v), _d3(n), ior_ratio, out)
    return np.array(list(out))


def detmath_inv(y, x=1.0) -> np.ndarray:
    """(det_atan(y), det_atan2(y, x), det_acos(y)) of include/rt_detmath.h."""
    out = (C.c_double * 3)()
    load().oracle_detmath_inv(float(y), float(x), out)
    return np.array(list(out))


def detmath(x) -> np.ndarray:
    """(det_sin(x), det_cos(x), det_log(x)) of include/rt_detmath.h."""
    out = (C.c_double * 3)()
    load().oracle_detmath(float(x), out)
    return np.array(list(out))
