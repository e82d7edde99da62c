"""raytracer_amd — MI355X (gfx950) render core for the per-pixel path of
MitchelPaulin/RayTracer, plus a thin Python view of its host mirror.

The compute lives in libraytracer_amd.so (hand-written HIP, see csrc/); this
module only binds the C ABI declared in include/rt_core.h and
include/rt_host_c.h with ctypes.  There is no Python or CPU implementation of
the render path: without the library, or without a GPU at commit time, calls
raise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
LIB_PATH = os.environ.get("RT_LIB") or os.path.join(HERE, "libraytracer_amd.so")  # RT_LIB: A/B builds
OBJ_DIR = os.path.join(REPO, "assets", "obj")

SPHERE, PLANE, CUBE, CYLINDER, CONE = range(5)
GROUP = 7
SOLID, STRIPE, GRADIENT, RINGS, CHECKERED = range(5)
MODE_F64_PARITY, MODE_F32_FAST = 0, 1
MODE_F32_SEARCH = MODE_F32_FAST  # what it is: the f32-search variant (same frames, not faster on CDNA4)
PIPELINE_MEGAKERNEL, PIPELINE_WAVEFRONT, PIPELINE_AUTO = 0, 1, 2
# pipeline used when a render call does not name one
DEFAULT_PIPELINE = PIPELINE_WAVEFRONT
TILE = 32


class RtError(RuntimeError):
    pass


class Material(C.Structure):
    """rth_material (include/rt_host_c.h): draw/material.rs:6-15 + pattern."""
    _fields_ = [
        ("ambient", C.c_double), ("diffuse", C.c_double), ("specular", C.c_double),
        ("shininess", C.c_double), ("reflective", C.c_double), ("transparency", C.c_double),
        ("refractive_index", C.c_double),
        ("pattern_kind", C.c_int32), ("has_pattern_transform", C.c_int32),
        ("color_a", C.c_double * 3), ("color_b", C.c_double * 3),
        ("pattern_transform", C.c_double * 16),
    ]


def material(ambient=0.1, diffuse=0.9, specular=0.9, shininess=200.0, reflective=0.0,
             transparency=0.0, refractive_index=1.0, pattern=SOLID, a=(1.0, 1.0, 1.0),
             b=(0.0, 0.0, 0.0), pattern_transform=None):
    """Material::default_material() (draw/material.rs:18-29) with overrides."""
    m = Material()
    m.ambient, m.diffuse, m.specular, m.shininess = ambient, diffuse, specular, shininess
    m.reflective, m.transparency, m.refractive_index = reflective, transparency, refractive_index
    m.pattern_kind = pattern
    m.color_a[:] = a
    m.color_b[:] = b
    if pattern_transform is not None:
        m.has_pattern_transform = 1
        m.pattern_transform[:] = np.asarray(pattern_transform, dtype=np.float64).reshape(16)
    return m


class CameraABI(C.Structure):
    """rt_camera (include/rt_core.h)."""
    _fields_ = [("hsize", C.c_int32), ("vsize", C.c_int32), ("half_width", C.c_double),
                ("half_height", C.c_double), ("pixel_size", C.c_double), ("inv", C.c_double * 16)]


class Stats(C.Structure):
    """rt_stats (include/rt_core.h)."""
    _fields_ = [(n, C.c_uint64) for n in (
        "rays_primary", "rays_shadow", "rays_reflect", "rays_refract", "depth0_skipped",
        "pred_queries", "nodes_visited", "tri_tests", "prim_tests", "inst_entries", "shade_calls",
        "launches")] + [("kernel_ms", C.c_double), ("total_ms", C.c_double),
                        ("max_nodes_per_query", C.c_uint64), ("node_wave_steps", C.c_uint64),
                        ("tri_wave_steps", C.c_uint64), ("fast_candidates", C.c_uint64),
                        ("fast_exact", C.c_uint64)]

    def rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_reflect + self.rays_refract

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_[:12]}
        d["kernel_ms"] = float(self.kernel_ms)
        d["total_ms"] = float(self.total_ms)
        d["rays"] = int(self.rays())
        d["max_nodes_per_query"] = int(self.max_nodes_per_query)
        d["node_wave_steps"] = int(self.node_wave_steps)
        d["tri_wave_steps"] = int(self.tri_wave_steps)
        d["fast_candidates"] = int(self.fast_candidates)
        d["fast_exact"] = int(self.fast_exact)
        return d


class SceneInfo(C.Structure):
    """rt_scene_info (include/rt_core.h)."""
    _fields_ = [(n, C.c_int64) for n in (
        "n_groups", "n_primitives", "n_meshes", "n_triangles", "n_materials", "n_lights",
        "n_bvh_nodes", "device_bytes")] + [
        ("node_bytes", C.c_int32), ("tri_bytes", C.c_int32), ("device", C.c_int32),
        ("bvh_builder", C.c_int32), ("commit_ms", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def build(force=False):
    """Compile libraytracer_amd.so for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", HERE, "-s", "-j4"]
    if force:
        subprocess.check_call(["make", "-C", HERE, "-s", "clean"])
    subprocess.check_call(args)
    return LIB_PATH


_lib = None


def lib():
    """The loaded C ABI.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; "
                      f"g.build()'` (or `make -C raytracer_amd`) first")
    # PyTorch ships its own copy of the HIP runtime.  Load it first so the process holds ONE
    # runtime: device pointers and streams handed over by torch then belong to the runtime
    # the kernels are launched with (loading ours first leaves torch without visible GPUs).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    dp, u8p, vp = C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.c_void_p
    L.rt_last_error.restype = C.c_char_p
    L.rth_last_error.restype = C.c_char_p
    L.rt_device_count.restype = C.c_int
    L.rt_scene_create.restype = vp
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_scene_add_material.argtypes = [vp, vp]
    L.rt_scene_add_group.argtypes = [vp, vp]
    L.rt_scene_add_primitive.argtypes = [vp, vp]
    L.rt_scene_add_mesh.argtypes = [vp, vp]
    L.rt_scene_add_light.argtypes = [vp, dp, dp]
    L.rt_scene_commit.argtypes = [vp, C.c_int, C.c_int]
    L.rt_scene_get_info.argtypes = [vp, C.POINTER(SceneInfo)]
    L.rt_render.argtypes = [vp, C.POINTER(CameraABI), C.c_int, C.c_int, C.c_int, dp, u8p,
                            C.POINTER(Stats)]
    L.rt_render_device.argtypes = [vp, C.POINTER(CameraABI), C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, vp, vp, vp, C.c_int]
    L.rt_render_stats.argtypes = [vp, C.POINTER(Stats)]
    L.rt_render_device_noalloc.argtypes = [vp, C.POINTER(CameraABI), C.c_int, C.c_int, C.c_int, C.c_int,
                                           vp, vp, vp]
    L.rt_noalloc_max_tiles.argtypes = [C.c_int]
    L.rt_noalloc_max_tiles.restype = C.c_int64
    L.rt_shard_tiles.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.rt_shard_tiles.restype = C.c_int64
    L.rt_detile_u8.argtypes = [vp, C.c_int, C.c_int64, C.c_int, C.c_int, vp, vp]
    L.rt_trace_rays.argtypes = [vp, C.c_int64, dp, dp, C.c_int, dp]
    L.rth_world_new.restype = vp
    L.rth_world_free.argtypes = [vp]
    L.rth_load_example.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
    L.rth_add_light.argtypes = [vp, dp, dp]
    L.rth_add_shape.argtypes = [vp, C.c_int, C.c_int, dp, C.POINTER(Material), C.c_double,
                                C.c_double, C.c_int]
    L.rth_add_triangle.argtypes = [vp, C.c_int, dp, dp]
    L.rth_add_obj.argtypes = [vp, C.c_int, C.c_char_p, dp, C.POINTER(Material)]
    L.rth_group_child_count.argtypes = [vp, C.c_int]
    L.rth_set_camera.argtypes = [vp, C.c_int, C.c_int, C.c_double, dp]
    L.rth_get_camera.argtypes = [vp, C.POINTER(CameraABI)]
    L.rth_commit.argtypes = [vp, C.c_int, C.c_int]
    L.rth_commit_multi.argtypes = [vp, C.POINTER(C.c_int), C.c_int, C.c_int]
    L.rt_render_multi.argtypes = [vp, C.POINTER(CameraABI), C.c_int, C.c_int, C.c_int, u8p, vp, C.POINTER(Stats)]
    L.rt_render_multi_f64.argtypes = [vp, C.POINTER(CameraABI), C.c_int, C.c_int, C.c_int, dp, vp, u8p, vp,
                                      C.POINTER(Stats)]
    L.rt_detile_f64.argtypes = [vp, C.c_int, C.c_int64, C.c_int, C.c_int, vp, vp]
    L.rt_scene_device_count.argtypes = [vp]
    L.rth_scene.argtypes = [vp]
    L.rth_scene.restype = vp
    L.rth_render.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp, u8p, C.POINTER(Stats)]
    L.rth_write_ppm.argtypes = [C.c_char_p, C.c_int, C.c_int, u8p]
    L.rth_write_png.argtypes = [C.c_char_p, C.c_int, C.c_int, u8p]
    L.rth_mat_inverse.argtypes = [dp, C.c_int, dp]
    L.rth_mat_determinant.argtypes = [dp, C.c_int]
    L.rth_mat_determinant.restype = C.c_double
    L.rth_mat_mul.argtypes = [dp, dp, dp]
    L.rth_mat_make.argtypes = [C.c_int, dp, dp]
    L.rth_view_transform.argtypes = [dp, dp, dp, dp]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _vec(v, n=3):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
    if a.size != n:
        raise ValueError(f"expected {n} values")
    return a


def _mat(m):
    if m is None:
        return None, None
    a = np.ascontiguousarray(np.asarray(m, dtype=np.float64).reshape(16))
    return a, _dp(a)


def device_count():
    return lib().rt_device_count()


class World:
    """World + Camera of the reference (scene/world.rs:9-12, scene/camera.rs:11-19) held by the
    C++ host mirror; `commit()` flattens it onto a GPU, `render()` is scene::camera::render
    (camera.rs:83)."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.rth_world_new())
        self.committed = False

    def __del__(self):
        try:
            if self.h:
                self.L.rth_world_free(self.h)
                self.h = None
        except Exception:
            pass

    def _err(self, what):
        raise RtError(f"{what}: {self.L.rth_last_error().decode()}")

    @classmethod
    def example(cls, name, hsize=0, vsize=0, synthetic_tris=0, obj_dir=OBJ_DIR):
        """examples.rs: 'cover' | 'pawn' | 'tea set' | 'test_scene' | 'synthetic'."""
        w = cls()
        if w.L.rth_load_example(w.h, name.encode(), obj_dir.encode(), hsize, vsize,
                                synthetic_tris) != 0:
            w._err(f"example {name!r}")
        return w

    def add_light(self, intensity, position):
        i, p = _vec(intensity), _vec(position)
        return self.L.rth_add_light(self.h, _dp(i), _dp(p))

    def add_shape(self, kind, transform=None, mat=None, parent=-1, minimum=-np.inf,
                  maximum=np.inf, closed=False):
        keep, tp = _mat(transform)
        mp = C.byref(mat) if mat is not None else None
        h = self.L.rth_add_shape(self.h, parent, kind, tp, mp, minimum, maximum, int(closed))
        if h < 0:
            self._err("add_shape")
        return h

    def add_group(self, transform=None, mat=None, parent=-1):
        return self.add_shape(GROUP, transform, mat, parent)

    def add_triangle(self, p, n=None, parent=-1):
        pa = _vec(p, 9)
        na = _vec(n, 9) if n is not None else None
        h = self.L.rth_add_triangle(self.h, parent, _dp(pa), _dp(na) if na is not None else None)
        if h < 0:
            self._err("add_triangle")
        return h

    def add_obj(self, text, transform=None, mat=None, parent=-1):
        keep, tp = _mat(transform)
        mp = C.byref(mat) if mat is not None else None
        h = self.L.rth_add_obj(self.h, parent, text.encode(), tp, mp)
        if h < 0:
            self._err("add_obj")
        return h

    def group_child_count(self, handle):
        return self.L.rth_group_child_count(self.h, handle)

    def set_camera(self, hsize, vsize, fov, transform=None):
        keep, tp = _mat(transform)
        if self.L.rth_set_camera(self.h, hsize, vsize, fov, tp) != 0:
            self._err("set_camera")

    def camera_abi(self):
        cam = CameraABI()
        if self.L.rth_get_camera(self.h, C.byref(cam)) != 0:
            self._err("get_camera")
        return cam

    def commit(self, device=-1, bvh_builder=1):
        if self.L.rth_commit(self.h, device, bvh_builder) != 0:
            self._err("commit")
        self.committed = True
        return self

    def commit_multi(self, devices, bvh_builder=1):
        """rt_scene_commit_multi: the scene on several GPUs; render() then uses all of them."""
        devs = (C.c_int * len(devices))(*devices)
        if self.L.rth_commit_multi(self.h, devs, len(devices), bvh_builder) != 0:
            self._err("commit_multi")
        self.committed = True
        return self

    def render_multi_f64(self, d_f64=None, d_u8=None, want_host_f64=True, want_host_u8=True, depth=5,
                         mode=MODE_F64_PARITY, pipeline=None, want_stats=False, cam=None):
        """rt_render_multi_f64: returns (f64 HxWx3 or None, u8 HxWx3 or None, Stats or None); d_f64 /
        d_u8 = device pointers on the first GPU for the stitched images."""
        pipeline = DEFAULT_PIPELINE if pipeline is None else pipeline
        cam = cam or self.camera_abi()
        n = cam.hsize * cam.vsize * 3
        f64 = np.zeros(n) if want_host_f64 else None
        u8 = np.zeros(n, dtype=np.uint8) if want_host_u8 else None
        st = Stats() if want_stats else None
        rc = self.L.rt_render_multi_f64(self.scene, C.byref(cam), depth, mode, pipeline,
                                        _dp(f64) if want_host_f64 else None, C.c_void_p(d_f64 or 0),
                                        u8.ctypes.data_as(C.POINTER(C.c_uint8)) if want_host_u8 else None,
                                        C.c_void_p(d_u8 or 0), C.byref(st) if want_stats else None)
        if rc != 0:
            raise RtError(f"rt_render_multi_f64: {self.L.rt_last_error().decode()}")
        shape = (cam.vsize, cam.hsize, 3)
        return (f64.reshape(shape) if want_host_f64 else None), (u8.reshape(shape) if want_host_u8 else None), st

    def render_multi(self, d_u8=None, want_host=True, depth=5, mode=MODE_F64_PARITY, pipeline=None,
                     want_stats=False, cam=None):
        """rt_render_multi: returns (u8 HxWx3 or None, Stats or None); d_u8 = device pointer on the
        first GPU for the stitched image."""
        pipeline = DEFAULT_PIPELINE if pipeline is None else pipeline
        cam = cam or self.camera_abi()
        u8 = np.zeros(cam.hsize * cam.vsize * 3, dtype=np.uint8) if want_host else None
        st = Stats() if want_stats else None
        rc = self.L.rt_render_multi(self.scene, C.byref(cam), depth, mode, pipeline,
                                    u8.ctypes.data_as(C.POINTER(C.c_uint8)) if want_host else None,
                                    C.c_void_p(d_u8 or 0), C.byref(st) if want_stats else None)
        if rc != 0:
            raise RtError(f"rt_render_multi: {self.L.rt_last_error().decode()}")
        return (u8.reshape(cam.vsize, cam.hsize, 3) if want_host else None), st

    @property
    def scene(self):
        """The committed rt_scene* (include/rt_core.h)."""
        p = self.L.rth_scene(self.h)
        if not p:
            raise RtError("world is not committed")
        return C.c_void_p(p)

    def info(self):
        si = SceneInfo()
        if self.L.rt_scene_get_info(self.scene, C.byref(si)) != 0:
            raise RtError(self.L.rt_last_error().decode())
        return si.as_dict()

    def render(self, depth=5, mode=MODE_F64_PARITY, pipeline=None, want_f64=False,
               want_stats=False):
        """Returns (u8 HxWx3, f64 HxWx3 or None, Stats or None)."""
        pipeline = DEFAULT_PIPELINE if pipeline is None else pipeline
        cam = self.camera_abi()
        n = cam.hsize * cam.vsize * 3
        u8 = np.zeros(n, dtype=np.uint8)
        f64 = np.zeros(n) if want_f64 else None
        st = Stats() if want_stats else None
        rc = self.L.rth_render(self.h, depth, mode, pipeline, _dp(f64) if want_f64 else None,
                               u8.ctypes.data_as(C.POINTER(C.c_uint8)),
                               C.byref(st) if want_stats else None)
        if rc != 0:
            self._err("render")
        shape = (cam.vsize, cam.hsize, 3)
        return u8.reshape(shape), (f64.reshape(shape) if want_f64 else None), st

    def render_sequence(self, cameras, depth=5, mode=MODE_F64_PARITY, pipeline=None, inflight=4):
        """rth_render_sequence: one u8 HxWx3 image per camera (CameraABI), frames in flight."""
        pipeline = DEFAULT_PIPELINE if pipeline is None else pipeline
        cams = (CameraABI * len(cameras))(*cameras)
        sizes = [c.hsize * c.vsize * 3 for c in cameras]
        buf = np.zeros(sum(sizes), dtype=np.uint8)
        rc = self.L.rth_render_sequence(self.h, cams, len(cameras), depth, mode, pipeline, inflight,
                                        buf.ctypes.data_as(C.POINTER(C.c_uint8)))
        if rc != 0:
            self._err("render_sequence")
        out, off = [], 0
        for c, n in zip(cameras, sizes):
            out.append(buf[off:off + n].reshape(c.vsize, c.hsize, 3))
            off += n
        return out

    def sweep_counters(self):
        """rth_sweep_counters: (frames issued on an estimate, of those rendered again)."""
        a, b = C.c_int64(0), C.c_int64(0)
        if self.L.rth_sweep_counters(self.h, C.byref(a), C.byref(b)) != 0:
            self._err("sweep_counters")
        return a.value, b.value

    def render_device(self, d_u8=None, d_f64=None, stream=None, depth=5, mode=MODE_F64_PARITY,
                      pipeline=None, tile_first=0, tile_step=1, want_stats=False,
                      cam=None):
        """rt_render_device: device pointers (ints), asynchronous on `stream` (a hipStream_t)."""
        pipeline = DEFAULT_PIPELINE if pipeline is None else pipeline
        cam = cam or self.camera_abi()
        rc = self.L.rt_render_device(self.scene, C.byref(cam), depth, mode, pipeline, tile_first,
                                     tile_step, C.c_void_p(d_f64 or 0), C.c_void_p(d_u8 or 0),
                                     C.c_void_p(stream or 0), int(want_stats))
        if rc != 0:
            raise RtError(f"rt_render_device: {self.L.rt_last_error().decode()}")

    def render_device_noalloc(self, d_u8=None, d_f64=None, stream=None, depth=5, mode=MODE_F64_PARITY,
                              tile_first=0, tile_step=1, cam=None):
        """rt_render_device_noalloc: the wavefront frame out of the library's fixed device array
        (no allocation, no synchronisation); same outputs as render_device."""
        cam = cam or self.camera_abi()
        rc = self.L.rt_render_device_noalloc(self.scene, C.byref(cam), depth, mode, tile_first, tile_step,
                                             C.c_void_p(d_f64 or 0), C.c_void_p(d_u8 or 0),
                                             C.c_void_p(stream or 0))
        if rc != 0:
            raise RtError(f"rt_render_device_noalloc: {self.L.rt_last_error().decode()}")

    def stats(self):
        st = Stats()
        if self.L.rt_render_stats(self.scene, C.byref(st)) != 0:
            raise RtError(self.L.rt_last_error().decode())
        return st

    def trace_rays(self, origins, directions, depth=5):
        """rt_trace_rays: per ray a dict-like row of 32 doubles (see include/rt_core.h)."""
        o = np.ascontiguousarray(np.asarray(origins, dtype=np.float64).reshape(-1, 3))
        d = np.ascontiguousarray(np.asarray(directions, dtype=np.float64).reshape(-1, 3))
        out = np.zeros((o.shape[0], 32))
        rc = self.L.rt_trace_rays(self.scene, o.shape[0], _dp(o), _dp(d), depth, _dp(out))
        if rc != 0:
            raise RtError(f"rt_trace_rays: {self.L.rt_last_error().decode()}")
        return out


NOALLOC_MAX_TILES = 8192
NOALLOC_SCRATCH_BYTES = 64 << 20


def noalloc_max_tiles(depth):
    return lib().rt_noalloc_max_tiles(depth)


def shard_tiles(hsize, vsize, tile_first, tile_step):
    return lib().rt_shard_tiles(hsize, vsize, tile_first, tile_step)


def detile_u8(d_tiles, nshards, tiles_per_shard, hsize, vsize, d_image, stream=None):
    rc = lib().rt_detile_u8(C.c_void_p(d_tiles), nshards, tiles_per_shard, hsize, vsize,
                            C.c_void_p(d_image), C.c_void_p(stream or 0))
    if rc != 0:
        raise RtError(f"rt_detile_u8: {lib().rt_last_error().decode()}")


def write_png(path, rgb):
    """The quantised image as an 8-bit RGB PNG (the format of the reference's renders/)."""
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = a.shape
    if lib().rth_write_png(path.encode(), w, h, a.ctypes.data_as(C.POINTER(C.c_uint8))) != 0:
        raise RtError(f"cannot write {path}")


def write_ppm(path, rgb):
    """Canvas::write_to_ppm (draw/canvas.rs:29-44)."""
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = a.shape
    if lib().rth_write_ppm(path.encode(), w, h, a.ctypes.data_as(C.POINTER(C.c_uint8))) != 0:
        raise RtError(f"cannot write {path}")


# ---- math pre-computation (host side of math/matrix.rs) ---------------------------------
def mat_inverse(m):
    a = np.ascontiguousarray(np.asarray(m, dtype=np.float64))
    n = a.shape[0]
    if a.ndim != 2 or a.shape[1] != n:
        raise RtError("matrix must be square")
    flat = a.reshape(-1).copy()
    out = np.zeros(n * n)
    if lib().rth_mat_inverse(_dp(flat), n, _dp(out)) != 0:
        raise RtError(lib().rth_last_error().decode())
    return out.reshape(n, n)


def mat_determinant(m):
    a = np.ascontiguousarray(np.asarray(m, dtype=np.float64))
    flat = a.reshape(-1).copy()
    if a.ndim != 2 or a.shape[0] != a.shape[1] or not 2 <= a.shape[0] <= 4:
        raise RtError("matrix size must be 2, 3 or 4")
    return lib().rth_mat_determinant(_dp(flat), a.shape[0])


def mat_mul(a, b):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(16))
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float64).reshape(16))
    out = np.zeros(16)
    lib().rth_mat_mul(_dp(a), _dp(b), _dp(out))
    return out.reshape(4, 4)


def _make(kind, *p):
    pa = np.zeros(6)
    pa[:len(p)] = p
    out = np.zeros(16)
    lib().rth_mat_make(kind, _dp(pa), _dp(out))
    return out.reshape(4, 4)


def translation(x, y, z): return _make(0, x, y, z)
def scaling(x, y, z): return _make(1, x, y, z)
def rotation_x(r): return _make(2, r)
def rotation_y(r): return _make(3, r)
def rotation_z(r): return _make(4, r)
def shear(*p): return _make(5, *p)


def view_transform(frm, to, up):
    f, t, u = _vec(frm), _vec(to), _vec(up)
    out = np.zeros(16)
    lib().rth_view_transform(_dp(f), _dp(t), _dp(u), _dp(out))
    return out.reshape(4, 4)
