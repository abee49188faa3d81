"""ctypes mirror of include/rt_mi355.h and include/rt_host.h.

The product path is `librt_mi355.so` (HIP kernels behind the C ABI) plus `librt_host.so`
(CLI flags, scene DSL, OBJ loader, camera, output stage).  There is NO CPU fallback: if the
HIP library is missing, `load_device_lib()` raises.  The CPU oracle under oracle/ is test
infrastructure and is never loaded from this package.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Optional, Sequence

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)

RT_OK = 0
RT_E_INVALID, RT_E_UNSUPPORTED, RT_E_DEVICE, RT_E_NOMEM = -1, -2, -3, -4
RT_PRECISION_F64, RT_PRECISION_F32 = 0, 1
RT_PIPELINE_AUTO, RT_PIPELINE_MEGAKERNEL, RT_PIPELINE_WAVEFRONT = 0, 1, 2
RT_SCENE_BVH_ON_DEVICE = 1  # RtSceneDesc.flags

(RT_NODE_SPHERE, RT_NODE_PLANE, RT_NODE_MESH, RT_NODE_LIST, RT_NODE_TRANSFORM, RT_NODE_BVH,
 RT_NODE_SKY, RT_NODE_SUN, RT_NODE_VOLUME, RT_NODE_NULL) = range(1, 11)
(RT_MAT_LAMBERTIAN, RT_MAT_METAL, RT_MAT_DIELECTRIC, RT_MAT_GLOSSY, RT_MAT_EMISSIVE,
 RT_MAT_ISOTROPIC, RT_MAT_NORMAL_DEBUG) = range(1, 8)
(RT_TEX_CONST_COLOR, RT_TEX_CONST_FLOAT, RT_TEX_CHECKER, RT_TEX_CHECKER_SOLID, RT_TEX_LERP,
 RT_TEX_IMAGE, RT_TEX_NOISE_SOLID, RT_TEX_CHANNEL, RT_TEX_UV_DEBUG) = range(1, 10)


class RtNode(C.Structure):
    _fields_ = [("type", C.c_uint32), ("flags", C.c_uint32), ("material", C.c_int32),
                ("mesh", C.c_int32), ("transform", C.c_int32), ("first_child", C.c_uint32),
                ("n_children", C.c_uint32), ("_pad", C.c_uint32),
                ("bounds", C.c_double * 6), ("p", C.c_double * 12)]


class RtTransform(C.Structure):
    _fields_ = [("m", C.c_double * 16), ("inv", C.c_double * 16)]


class RtMesh(C.Structure):
    _fields_ = [("positions", C.POINTER(C.c_double)), ("normals", C.POINTER(C.c_double)),
                ("uvs", C.POINTER(C.c_double)), ("tri_pos", C.POINTER(C.c_uint32)),
                ("tri_nrm", C.POINTER(C.c_uint32)), ("tri_uv", C.POINTER(C.c_int32)),
                ("n_positions", C.c_uint32), ("n_normals", C.c_uint32), ("n_uvs", C.c_uint32),
                ("n_triangles", C.c_uint32), ("flags", C.c_uint32), ("_pad", C.c_uint32)]


class RtMaterial(C.Structure):
    _fields_ = [("type", C.c_uint32), ("tex_a", C.c_int32), ("tex_b", C.c_int32),
                ("tex_c", C.c_int32), ("ior", C.c_double)]


class RtTexture(C.Structure):
    _fields_ = [("type", C.c_uint32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32),
                ("channel", C.c_uint32), ("samples", C.c_uint32), ("v", C.c_double * 3),
                ("scale", C.c_double),
                ("texels", C.POINTER(C.c_float)), ("width", C.c_uint32), ("height", C.c_uint32),
                ("perlin_vec", C.POINTER(C.c_double)), ("perlin_perm", C.POINTER(C.c_uint32))]


class RtSceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_nodes", C.c_uint32),
                ("nodes", C.POINTER(RtNode)),
                ("n_child_indices", C.c_uint32), ("n_transforms", C.c_uint32),
                ("child_indices", C.POINTER(C.c_uint32)),
                ("transforms", C.POINTER(RtTransform)),
                ("n_meshes", C.c_uint32), ("n_materials", C.c_uint32),
                ("meshes", C.POINTER(RtMesh)), ("materials", C.POINTER(RtMaterial)),
                ("n_textures", C.c_uint32), ("world_root", C.c_uint32),
                ("textures", C.POINTER(RtTexture)),
                ("lights_root", C.c_uint32), ("flags", C.c_uint32)]


class RtCameraDesc(C.Structure):
    _fields_ = [("image_width", C.c_uint32), ("image_height", C.c_uint32),
                ("position", C.c_double * 3), ("first_pixel", C.c_double * 3),
                ("pixel_delta_u", C.c_double * 3), ("pixel_delta_v", C.c_double * 3),
                ("basis_u", C.c_double * 3), ("basis_v", C.c_double * 3),
                ("has_aperture", C.c_uint32), ("_pad", C.c_uint32),
                ("aperture_radius", C.c_double)]


class RtRenderParams(C.Structure):
    _fields_ = [("sqrt_spt", C.c_uint32), ("thread_count", C.c_uint32),
                ("max_depth", C.c_uint32), ("has_background", C.c_uint32),
                ("light_bias", C.c_double), ("background", C.c_double * 3),
                ("seed", C.c_uint64),
                ("band_rows", C.c_uint32), ("n_parts", C.c_uint32), ("part", C.c_uint32),
                ("precision", C.c_uint32), ("pipeline", C.c_uint32),
                ("collect_stats", C.c_uint32)]

    def copy(self) -> "RtRenderParams":
        out = RtRenderParams()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(RtRenderParams))
        return out

    @property
    def spp(self) -> int:
        return self.sqrt_spt * self.sqrt_spt * self.thread_count


class RtRenderStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("traversal_kernel_ms", C.c_double),
                ("n_launches", C.c_uint32), ("pipeline_used", C.c_uint32),
                ("samples", C.c_uint64), ("rays", C.c_uint64), ("mesh_rays", C.c_uint64),
                ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64),
                ("prim_tests", C.c_uint64), ("bytes_node", C.c_uint64),
                ("bytes_tri", C.c_uint64), ("bytes_attr", C.c_uint64),
                ("bytes_state", C.c_uint64),
                ("prims_kernel_ms", C.c_double), ("shade_kernel_ms", C.c_double),
                ("bytes_state_prims", C.c_uint64), ("bytes_state_shade", C.c_uint64),
                ("n_iterations", C.c_uint32), ("n_replica_groups", C.c_uint32),
                ("n_tail_compactions", C.c_uint32), ("_reserved", C.c_uint32)]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_}


class RtError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"rt status {status}: {message}")
        self.status = status


HOST_LIB_PATH = os.path.join(PKG_DIR, "librt_host.so")
DEVICE_LIB_PATH = os.environ.get("RT_DEVICE_LIB") or os.path.join(PKG_DIR, "librt_mi355.so")  # env override: A/B builds

_host_lib = None
_device_lib = None


def load_host_lib() -> C.CDLL:
    global _host_lib
    if _host_lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise FileNotFoundError(
                f"{HOST_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(HOST_LIB_PATH)
        lib.rth_load.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]
        lib.rth_load.restype = C.c_int
        lib.rth_destroy.argtypes = [C.c_void_p]
        lib.rth_destroy.restype = None
        lib.rth_scene.argtypes = [C.c_void_p]
        lib.rth_scene.restype = C.POINTER(RtSceneDesc)
        lib.rth_camera.argtypes = [C.c_void_p]
        lib.rth_camera.restype = C.POINTER(RtCameraDesc)
        lib.rth_params.argtypes = [C.c_void_p]
        lib.rth_params.restype = C.POINTER(RtRenderParams)
        lib.rth_gpus.argtypes = [C.c_void_p]
        lib.rth_gpus.restype = C.c_uint32
        lib.rth_band_rows.argtypes = [C.c_uint32, C.c_uint32]
        lib.rth_band_rows.restype = C.c_uint32
        lib.rth_samples_per_pixel.argtypes = [C.c_void_p]
        lib.rth_samples_per_pixel.restype = C.c_uint32
        lib.rth_log.argtypes = [C.c_void_p]
        lib.rth_log.restype = C.c_char_p
        lib.rth_make_camera.argtypes = [C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double,
                                        C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.POINTER(RtCameraDesc)]
        lib.rth_make_camera.restype = C.c_int
        lib.rth_tonemap_rgb8.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        lib.rth_tonemap_rgb8.restype = C.c_int
        lib.rth_save_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        lib.rth_save_png.restype = C.c_int
        lib.rth_load_image.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.rth_load_image.restype = C.c_int
        lib.rth_free_image.argtypes = [C.POINTER(C.c_float)]
        lib.rth_free_image.restype = None
        lib.rth_last_error.argtypes = []
        lib.rth_last_error.restype = C.c_char_p
        _host_lib = lib
    return _host_lib


def load_device_lib() -> C.CDLL:
    """Loads the HIP library.  Raises if it has not been built: there is no fallback."""
    global _device_lib
    if _device_lib is None:
        if not os.path.exists(DEVICE_LIB_PATH):
            raise FileNotFoundError(
                f"{DEVICE_LIB_PATH} is missing — the HIP render path was not built "
                "(run __graft_entry__.build()); refusing to fall back to any CPU path")
        # A process that also uses PyTorch must load PyTorch's HIP runtime FIRST: the wheel bundles its own
        # libamdhip64, and if the system copy (our dependency) is already loaded, torch then sees "No HIP GPUs".
        # Loaded in this order, both share torch's copy.  (Pure C / C++ callers are not affected.)
        if "torch" not in sys.modules and os.environ.get("RT_NO_TORCH_PRELOAD", "0") != "1":
            try:
                import torch  # noqa: F401
            except Exception:  # PyTorch absent or broken: the library works on its own
                pass
        lib = C.CDLL(DEVICE_LIB_PATH)
        lib.rt_device_count.argtypes = []
        lib.rt_device_count.restype = C.c_int
        lib.rt_scene_create.argtypes = [C.POINTER(RtSceneDesc), C.c_int, C.POINTER(C.c_void_p)]
        lib.rt_scene_create.restype = C.c_int
        lib.rt_scene_destroy.argtypes = [C.c_void_p]
        lib.rt_scene_destroy.restype = None
        lib.rt_owned_rows.argtypes = [C.c_uint32, C.POINTER(RtRenderParams)]
        lib.rt_owned_rows.restype = C.c_uint32
        lib.rt_render.argtypes = [C.c_void_p, C.POINTER(RtCameraDesc), C.POINTER(RtRenderParams), C.c_void_p]
        lib.rt_render.restype = C.c_int
        lib.rt_render_device.argtypes = [C.c_void_p, C.POINTER(RtCameraDesc), C.POINTER(RtRenderParams),
                                         C.c_void_p, C.c_void_p]
        lib.rt_render_device.restype = C.c_int
        lib.rt_get_stats.argtypes = [C.c_void_p, C.POINTER(RtRenderStats)]
        lib.rt_get_stats.restype = C.c_int
        lib.rt_debug_trace_sample.argtypes = [C.c_void_p, C.POINTER(RtCameraDesc), C.POINTER(RtRenderParams),
                                              C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_uint32]
        lib.rt_debug_trace_sample.restype = C.c_int
        lib.rt_scene_info.argtypes = [C.POINTER(RtSceneDesc), C.POINTER(C.c_uint32)]
        lib.rt_scene_info.restype = C.c_int
        if hasattr(lib, "rt_scene_set_tail_flag"):  # absent from older A/B builds loaded through RT_DEVICE_LIB
            lib.rt_scene_set_tail_flag.argtypes = [C.c_void_p, C.c_void_p]
            lib.rt_scene_set_tail_flag.restype = C.c_int
        if hasattr(lib, "rt_scene_mesh_stats"):  # absent from older A/B builds loaded through RT_DEVICE_LIB
            lib.rt_scene_mesh_stats.argtypes = [C.POINTER(RtSceneDesc), C.POINTER(C.c_uint64)]
            lib.rt_scene_mesh_stats.restype = C.c_int
        lib.rt_last_error.argtypes = []
        lib.rt_last_error.restype = C.c_char_p
        _device_lib = lib
    return _device_lib


RT_SCENE_INFO_ZERO_WEIGHT_STOP, RT_SCENE_INFO_TEX_INTERPRETER, RT_SCENE_INFO_VOLUMES = 1, 2, 4


def scene_info(desc) -> int:
    """rt_scene_info: the scene compiler's classification flags (host only, no device needed)."""
    lib = load_device_lib()
    flags = C.c_uint32()
    st = lib.rt_scene_info(desc, C.byref(flags))
    if st != RT_OK:
        raise RtError(st, lib.rt_last_error().decode())
    return flags.value


def scene_mesh_stats(desc) -> dict:
    """rt_scene_mesh_stats: triangle records / BVH node counts / depths (host only)."""
    lib = load_device_lib()
    out = (C.c_uint64 * 8)()
    st = lib.rt_scene_mesh_stats(desc, out)
    if st != RT_OK:
        raise RtError(st, lib.rt_last_error().decode())
    return dict(zip(("triangles", "bvh2_nodes", "bvh4_nodes", "bvh2_depth", "bvh4_stack", "ops", "rebuilt_groups", "rebuilt_prims"),
                    [int(x) for x in out]))


def scene_program(desc) -> tuple:
    """rt_scene_program: (ops as an (n, 4) int32 array of type / arg / skip / chain, info dict); host only."""
    import numpy as np
    lib = load_device_lib()
    lib.rt_scene_program.argtypes = [C.POINTER(RtSceneDesc), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    lib.rt_scene_program.restype = C.c_int
    n = C.c_uint32()
    info = (C.c_uint64 * 8)()
    st = lib.rt_scene_program(desc, None, 0, C.byref(n), info)
    if st != RT_OK:
        raise RtError(st, lib.rt_last_error().decode())
    ops = np.zeros((n.value, 4), dtype=np.int32)
    st = lib.rt_scene_program(desc, ops.ctypes.data_as(C.POINTER(C.c_int32)), n.value, C.byref(n), info)
    if st != RT_OK:
        raise RtError(st, lib.rt_last_error().decode())
    plan = int(info[6])
    return ops, {"mesh_ops": int(info[0]), "groups": int(info[1]), "group_nodes": int(info[2]), "group_stack": int(info[3]),
                 "lights": int(info[4]), "volumes": int(info[5]), "group_prims": int(info[7]),
                 "split": bool(plan & 1), "vol_prims": bool(plan & 2), "multi_mesh": bool(plan & 4), "group_bvh": bool(plan & 8)}


def owned_rows(height: int, params: RtRenderParams) -> list:
    """Rows of the frame that the partition in `params` assigns to this part (rt_owned_rows)."""
    if params.band_rows == 0 or params.n_parts <= 1:
        return list(range(height))
    return [y for y in range(height) if (y // params.band_rows) % params.n_parts == params.part]


class HostScene:
    """`(Camera, world, lights)` as loaded by the reference's main() (src/main.rs:26-59)."""

    def __init__(self, args: Sequence[str], cwd: Optional[str] = None):
        lib = load_host_lib()
        argv = [b"rtrace"] + [a.encode() for a in args]
        arr = (C.c_char_p * len(argv))(*argv)
        handle = C.c_void_p()
        old = os.getcwd()
        try:
            os.chdir(cwd or REPO_DIR)  # scene/asset paths are relative to the CWD, like the reference
            st = lib.rth_load(len(argv), arr, C.byref(handle))
        finally:
            os.chdir(old)
        if st != RT_OK:
            raise RtError(st, lib.rth_last_error().decode())
        self._lib = lib
        self._h = handle
        self.desc = lib.rth_scene(handle)          # POINTER(RtSceneDesc)
        self.camera = lib.rth_camera(handle).contents
        self.params = lib.rth_params(handle).contents.copy()
        self.gpus = lib.rth_gpus(handle)
        self.spp = lib.rth_samples_per_pixel(handle)
        self.log = lib.rth_log(handle).decode()

    @property
    def width(self) -> int:
        return self.camera.image_width

    @property
    def height(self) -> int:
        return self.camera.image_height

    def close(self):
        if self._h:
            self._lib.rth_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tonemap_rgb8(rgba: np.ndarray) -> np.ndarray:
    """ACES + sRGB + 8-bit quantisation of an (H, W, 4) f64 frame -> (H, W, 3) uint8."""
    lib = load_host_lib()
    rgba = np.ascontiguousarray(rgba, dtype=np.float64)
    h, w = rgba.shape[:2]
    out = np.empty((h, w, 3), dtype=np.uint8)
    st = lib.rth_tonemap_rgb8(rgba.ctypes.data, w, h, out.ctypes.data)
    if st != RT_OK:
        raise RtError(st, lib.rth_last_error().decode())
    return out


def save_png(path: str, rgba: np.ndarray) -> None:
    lib = load_host_lib()
    rgba = np.ascontiguousarray(rgba, dtype=np.float64)
    h, w = rgba.shape[:2]
    st = lib.rth_save_png(path.encode(), rgba.ctypes.data, w, h)
    if st != RT_OK:
        raise RtError(st, lib.rth_last_error().decode())


def load_image(path: str) -> np.ndarray:
    """Buffer::from_image (buffer.rs:30-48): PNG / baseline JPEG -> (h, w, 3) float32."""
    lib = load_host_lib()
    ptr = C.POINTER(C.c_float)()
    w, h = C.c_uint32(), C.c_uint32()
    st = lib.rth_load_image(path.encode(), C.byref(ptr), C.byref(w), C.byref(h))
    if st != RT_OK:
        raise RtError(st, lib.rth_last_error().decode())
    try:
        return np.ctypeslib.as_array(ptr, shape=(h.value, w.value, 3)).copy()
    finally:
        lib.rth_free_image(ptr)


class DeviceScene:
    """RtScene on one GPU: the drop-in for `camera.render(world, lights, &mut buf)`."""

    def __init__(self, desc, device: int = 0):
        lib = load_device_lib()
        handle = C.c_void_p()
        st = lib.rt_scene_create(desc, device, C.byref(handle))
        if st != RT_OK:
            raise RtError(st, lib.rt_last_error().decode())
        self._lib = lib
        self._h = handle
        self.device = device

    def render(self, camera: RtCameraDesc, params: RtRenderParams) -> np.ndarray:
        rows = self._lib.rt_owned_rows(camera.image_height, C.byref(params))
        out = np.empty((rows, camera.image_width, 4), dtype=np.float64)
        st = self._lib.rt_render(self._h, C.byref(camera), C.byref(params), out.ctypes.data)
        if st != RT_OK:
            raise RtError(st, self._lib.rt_last_error().decode())
        return out

    def render_device(self, camera: RtCameraDesc, params: RtRenderParams, d_out_ptr: int, stream: int = 0) -> None:
        st = self._lib.rt_render_device(self._h, C.byref(camera), C.byref(params),
                                        C.c_void_p(d_out_ptr), C.c_void_p(stream))
        if st != RT_OK:
            raise RtError(st, self._lib.rt_last_error().decode())

    def trace_sample(self, camera, params, tid, x, y, sx, sy, max_bounces=64):
        """Diagnostic: (rgb[3], trace[n, 17]) of one sample traced on the device."""
        rgb = (C.c_double * 3)()
        tr = (C.c_double * (17 * max_bounces))()
        n = self._lib.rt_debug_trace_sample(self._h, C.byref(camera), C.byref(params), tid, x, y, sx, sy,
                                            rgb, tr, max_bounces)
        if n < 0:
            raise RtError(n, self._lib.rt_last_error().decode())
        return np.array(list(rgb)), np.array(list(tr)).reshape(max_bounces, 17)[:min(n, max_bounces)]

    def stats(self) -> RtRenderStats:
        s = RtRenderStats()
        st = self._lib.rt_get_stats(self._h, C.byref(s))
        if st != RT_OK:
            raise RtError(st, self._lib.rt_last_error().decode())
        return s

    def close(self):
        if self._h:
            self._lib.rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FramePipeline:
    """A sequence of frames of one scene on one GPU with their tails overlapped (rt_scene_set_tail_flag, include/rt_mi355.h).

    The end of a render is a chain of small, latency-bound launches (the last, longest paths: 20 of the 44 iterations of a
    1/8-frame share, 10 % of its time) that no scheduling inside ONE frame can fill.  Here `depth` device scenes of the
    same description are driven by one host thread each, on one stream each; frame k starts when frame k-1 reports that
    it has entered its tail, so its full launches run underneath.  Every frame is the frame DeviceScene.render_device
    produces on its own (same kernels, same per-sample RNG keys): tests/test_gpu_parity.py::test_frame_pipeline_*.
    """

    def __init__(self, desc, device: int = 0, depth: int = 2, scenes=None):
        """`scenes`: ready-made device scenes instead of `depth` new ones (the CPU tests of the ordering logic pass stand-ins)."""
        self.scenes = list(scenes) if scenes is not None else [DeviceScene(desc, device) for _ in range(max(1, depth))]
        self.device = device

    @property
    def depth(self) -> int:
        return len(self.scenes)

    def render_frames(self, camera: RtCameraDesc, params_list, d_out_ptrs, streams):
        """Frame k: params_list[k] into device buffer d_out_ptrs[k]; streams: one raw stream handle per device scene (all
        different, none the NULL stream).  Blocks until every frame is rendered; returns one RtRenderStats per frame."""
        import threading
        import time
        n, depth = len(params_list), len(self.scenes)
        if len(d_out_ptrs) != n or len(streams) < depth:
            raise ValueError("render_frames: one output buffer per frame and one stream per device scene")
        flags = (C.c_int32 * max(n, 1))()  # flags[k]: frame k has entered its tail (or returned)
        log = os.environ.get("RT_PIPE_LOG", "0") == "1"
        t_origin = time.perf_counter()
        stats = [None] * n
        errors = []

        def worker(i):
            sc = self.scenes[i]
            lib = sc._lib
            k = i
            try:
                while k < n:
                    if k > 0:
                        while flags[k - 1] == 0 and not errors:
                            time.sleep(0.0002)
                    if errors:
                        break
                    lib.rt_scene_set_tail_flag(sc._h, C.addressof(flags) + 4 * k)
                    t_start = time.perf_counter()
                    sc.render_device(camera, params_list[k], d_out_ptrs[k], streams[i])
                    stats[k] = sc.stats()
                    if log:
                        sys.stderr.write(f"[frame pipeline] frame {k} on scene {i}: start {1e3 * (t_start - t_origin):.1f} ms, "
                                         f"end {1e3 * (time.perf_counter() - t_origin):.1f} ms, kernels {stats[k].kernel_ms:.1f} ms\n")
                    k += depth
            except Exception as e:  # noqa: BLE001 - reported to the caller below
                errors.append(e)
            finally:
                for j in range(k, n, depth):  # frames this thread will not render: nobody may wait for them
                    flags[j] = 1
                lib.rt_scene_set_tail_flag(sc._h, None)

        threads = [threading.Thread(target=worker, args=(i,), name=f"rt-frame-{i}") for i in range(min(depth, n))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return stats

    def close(self):
        for sc in self.scenes:
            sc.close()

