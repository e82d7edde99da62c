"""ctypes front-end of the CPU oracle (oracle/rt_oracle.cpp).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
OBJ_DIR = os.path.join(REPO, "assets", "obj")
LIB_PATH = os.path.join(HERE, "liboracle.so")

SPHERE, PLANE, CUBE, CYLINDER, CONE, TRIANGLE, SMOOTH_TRIANGLE, GROUP = range(8)
SOLID, STRIPE, GRADIENT, RINGS, CHECKERED = range(5)


class Material(C.Structure):
    """Mirror of draw/material.rs:6-43 plus the pattern of draw/patterns.rs."""
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


def build(force=False):
    src = os.path.join(HERE, "rt_oracle.cpp")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double)
    L.orc_scene_new.restype = C.c_void_p
    L.orc_scene_free.argtypes = [C.c_void_p]
    L.orc_scene_load_example.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
    L.orc_set_use_bvh.argtypes = [C.c_void_p, C.c_int]
    L.orc_add_light.argtypes = [C.c_void_p, dp, dp]
    L.orc_add_shape.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, C.POINTER(Material), C.c_double,
                                C.c_double, C.c_int]
    L.orc_add_triangle.argtypes = [C.c_void_p, C.c_int, dp, dp]
    L.orc_add_obj.argtypes = [C.c_void_p, C.c_int, C.c_char_p, dp, C.POINTER(Material)]
    L.orc_group_child_count.argtypes = [C.c_void_p, C.c_int]
    L.orc_set_camera.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, dp]
    L.orc_get_camera.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), dp, dp]
    L.orc_object_count.argtypes = [C.c_void_p]
    L.orc_leaf_count.argtypes = [C.c_void_p]
    L.orc_leaf_count.restype = C.c_int64
    L.orc_mat_inverse.argtypes = [dp, C.c_int, dp]
    L.orc_mat_determinant.argtypes = [dp, C.c_int]
    L.orc_mat_determinant.restype = C.c_double
    L.orc_mat_mul.argtypes = [dp, dp, dp]
    L.orc_mat_make.argtypes = [C.c_int, dp, dp]
    L.orc_view_transform.argtypes = [dp, dp, dp, dp]
    L.orc_clamp_u8.argtypes = [C.c_double]
    L.orc_clamp_u8.restype = C.c_uint8
    L.orc_pattern_color_at.argtypes = [C.POINTER(Material), dp, dp]
    L.orc_lighting.argtypes = [C.POINTER(Material), dp, dp, dp, dp, dp, C.c_int, dp, dp]
    L.orc_ray_for_pixel.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp]
    L.orc_intersect_world.argtypes = [C.c_void_p, dp, dp, C.c_int, dp, C.POINTER(C.c_int32)]
    L.orc_trace_info.argtypes = [C.c_void_p, dp, dp, dp]
    L.orc_color_at.argtypes = [C.c_void_p, dp, dp, C.c_int, dp]
    L.orc_is_shadowed.argtypes = [C.c_void_p, C.c_int, dp]
    L.orc_normal_at.argtypes = [C.c_void_p, C.c_int, dp, C.c_double, C.c_double, dp]
    L.orc_shape_id.argtypes = [C.c_void_p, C.c_int]
    L.orc_render_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp,
                                  C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)]
    L.orc_render_pixels.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32), C.c_int, C.c_int, dp,
                                    C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)]
    L.orc_write_ppm.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _vec(v, n=3):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
    assert a.size == n
    return a


def _mat(m):
    if m is None:
        return None, None
    a = np.ascontiguousarray(np.asarray(m, dtype=np.float64).reshape(16))
    return a, _dp(a)


COUNTER_NAMES = ("primary", "shadow", "reflect", "refract", "depth0", "tri_tests", "nan_t", "pad")


def default_threads():
    return min(os.cpu_count() or 1, 16)


class Scene:
    """A World + Camera pair held by the oracle (scene/world.rs:9-12, scene/camera.rs:11-19)."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_scene_new())

    def __del__(self):
        try:
            if self.h:
                self.L.orc_scene_free(self.h)
                self.h = None
        except Exception:
            pass

    @classmethod
    def example(cls, name, hsize=0, vsize=0, synthetic_tris=0, obj_dir=OBJ_DIR):
        s = cls()
        rc = s.L.orc_scene_load_example(s.h, name.encode(), obj_dir.encode(), hsize, vsize,
                                        synthetic_tris)
        if rc != 0:
            raise RuntimeError(f"oracle: cannot build example {name!r} (obj dir {obj_dir})")
        return s

    def use_bvh(self, on=True):
        self.L.orc_set_use_bvh(self.h, 1 if on else 0)
        return self

    def add_light(self, intensity, position):
        i, p = _vec(intensity), _vec(position)
        return self.L.orc_add_light(self.h, _dp(i), _dp(p))

    def add_shape(self, kind, transform=None, mat=None, parent=-1, minimum=-np.inf,
                  maximum=np.inf, closed=False):
        keep, tp = _mat(transform)
        mp = C.byref(mat) if mat is not None else None
        return self.L.orc_add_shape(self.h, parent, kind, tp, mp, minimum, maximum, int(closed))

    def add_group(self, transform=None, mat=None, parent=-1):
        return self.add_shape(GROUP, transform, mat, parent)

    def add_triangle(self, p, n=None, parent=-1):
        pa = _vec(p, 9)
        na = _vec(n, 9) if n is not None else None
        return self.L.orc_add_triangle(self.h, parent, _dp(pa), _dp(na) if na is not None else None)

    def add_obj(self, text, transform=None, mat=None, parent=-1):
        keep, tp = _mat(transform)
        mp = C.byref(mat) if mat is not None else None
        return self.L.orc_add_obj(self.h, parent, text.encode(), tp, mp)

    def group_child_count(self, handle):
        return self.L.orc_group_child_count(self.h, handle)

    def set_camera(self, hsize, vsize, fov, transform=None):
        keep, tp = _mat(transform)
        self.L.orc_set_camera(self.h, hsize, vsize, fov, tp)

    def camera(self):
        hs, vs = C.c_int(), C.c_int()
        consts = np.zeros(3)
        inv = np.zeros(16)
        self.L.orc_get_camera(self.h, C.byref(hs), C.byref(vs), _dp(consts), _dp(inv))
        return dict(hsize=hs.value, vsize=vs.value, half_width=consts[0], half_height=consts[1],
                    pixel_size=consts[2], inv=inv.reshape(4, 4))

    def ray_for_pixel(self, px, py):
        o, d = np.zeros(3), np.zeros(3)
        self.L.orc_ray_for_pixel(self.h, px, py, _dp(o), _dp(d))
        return o, d

    def intersect_world(self, origin, direction, cap=256):
        o, d = _vec(origin), _vec(direction)
        ts = np.zeros(cap)
        ids = np.zeros(cap, dtype=np.int32)
        n = self.L.orc_intersect_world(self.h, _dp(o), _dp(d), cap, _dp(ts),
                                       ids.ctypes.data_as(C.POINTER(C.c_int32)))
        n = min(n, cap)
        return ts[:n].copy(), ids[:n].copy()

    def trace_info(self, origin, direction):
        o, d = _vec(origin), _vec(direction)
        out = np.zeros(32)
        self.L.orc_trace_info(self.h, _dp(o), _dp(d), _dp(out))
        return dict(hit=bool(out[0]), t=out[1], shape_id=int(out[2]), u=out[3], v=out[4],
                    inside=bool(out[5]), n1=out[6], n2=out[7], point=out[8:11].copy(),
                    over_point=out[11:14].copy(), under_point=out[14:17].copy(),
                    eyev=out[17:20].copy(), normalv=out[20:23].copy(),
                    reflectv=out[23:26].copy(), schlick=out[26], count=int(out[27]))

    def color_at(self, origin, direction, depth=5):
        o, d = _vec(origin), _vec(direction)
        out = np.zeros(3)
        self.L.orc_color_at(self.h, _dp(o), _dp(d), depth, _dp(out))
        return out

    def is_shadowed(self, point, light=0):
        p = _vec(point)
        return bool(self.L.orc_is_shadowed(self.h, light, _dp(p)))

    def normal_at(self, handle, point, u=0.0, v=0.0):
        p = _vec(point)
        out = np.zeros(3)
        self.L.orc_normal_at(self.h, handle, _dp(p), u, v, _dp(out))
        return out

    def shape_id(self, handle):
        return self.L.orc_shape_id(self.h, handle)

    def render(self, threads=None, depth=5, rows=None, schedule=1, want_f64=True):
        """Render rows [y0,y1) (default: all).  Returns (f64 HxWx3 or None, u8 HxWx3, counters)."""
        cam = self.camera()
        y0, y1 = rows if rows is not None else (0, cam["vsize"])
        n = (y1 - y0) * cam["hsize"] * 3
        f64 = np.zeros(n) if want_f64 else None
        u8 = np.zeros(n, dtype=np.uint8)
        cnt = np.zeros(8, dtype=np.uint64)
        rc = self.L.orc_render_rows(self.h, y0, y1, threads or default_threads(), schedule, depth,
                                    _dp(f64) if want_f64 else None,
                                    u8.ctypes.data_as(C.POINTER(C.c_uint8)),
                                    cnt.ctypes.data_as(C.POINTER(C.c_uint64)))
        if rc != 0:
            raise RuntimeError("oracle render failed")
        shape = (y1 - y0, cam["hsize"], 3)
        return (f64.reshape(shape) if want_f64 else None, u8.reshape(shape),
                dict(zip(COUNTER_NAMES, (int(x) for x in cnt))))

    def render_pixels(self, xs, ys, threads=None, depth=5):
        xs = np.ascontiguousarray(xs, dtype=np.int32)
        ys = np.ascontiguousarray(ys, dtype=np.int32)
        n = xs.size
        f64 = np.zeros(n * 3)
        u8 = np.zeros(n * 3, dtype=np.uint8)
        cnt = np.zeros(8, dtype=np.uint64)
        rc = self.L.orc_render_pixels(self.h, n, xs.ctypes.data_as(C.POINTER(C.c_int32)),
                                      ys.ctypes.data_as(C.POINTER(C.c_int32)),
                                      threads or default_threads(), depth, _dp(f64),
                                      u8.ctypes.data_as(C.POINTER(C.c_uint8)),
                                      cnt.ctypes.data_as(C.POINTER(C.c_uint64)))
        if rc != 0:
            raise RuntimeError("oracle render_pixels failed")
        return f64.reshape(n, 3), u8.reshape(n, 3), dict(zip(COUNTER_NAMES, (int(x) for x in cnt)))


# ---- math helpers (known-answer tests) ---------------------------------------
def mat_inverse(m):
    a = np.ascontiguousarray(np.asarray(m, dtype=np.float64))
    n = a.shape[0]
    out = np.zeros(n * n)
    flat = a.reshape(-1).copy()
    lib().orc_mat_inverse(_dp(flat), n, _dp(out))
    return out.reshape(n, n)


def mat_determinant(m):
    a = np.ascontiguousarray(np.asarray(m, dtype=np.float64))
    flat = a.reshape(-1).copy()
    return lib().orc_mat_determinant(_dp(flat), a.shape[0])


def mat_mul(a, b):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(16))
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float64).reshape(16))
    out = np.zeros(16)
    lib().orc_mat_mul(_dp(a), _dp(b), _dp(out))
    return out.reshape(4, 4)


def _make(kind, *p):
    pa = np.zeros(6)
    pa[:len(p)] = p
    out = np.zeros(16)
    lib().orc_mat_make(kind, _dp(pa), _dp(out))
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
    lib().orc_view_transform(_dp(f), _dp(t), _dp(u), _dp(out))
    return out.reshape(4, 4)


def clamp_u8(v):
    return int(lib().orc_clamp_u8(float(v)))


def pattern_color_at(mat, p):
    pa = _vec(p)
    out = np.zeros(3)
    lib().orc_pattern_color_at(C.byref(mat), _dp(pa), _dp(out))
    return out


def lighting(mat, light_intensity, light_pos, pos, eyev, normalv, in_shadow, object_point=None):
    li, lp, p, e, n = (_vec(light_intensity), _vec(light_pos), _vec(pos), _vec(eyev),
                       _vec(normalv))
    op = _vec(object_point if object_point is not None else pos)
    out = np.zeros(3)
    lib().orc_lighting(C.byref(mat), _dp(li), _dp(lp), _dp(p), _dp(e), _dp(n), int(in_shadow),
                       _dp(op), _dp(out))
    return out


# ---- golden images -------------------------------------------------------------
def read_png_rgb8(path):
    """Minimal decoder for 8-bit non-interlaced RGB PNG files (the format of the
    reference's committed renders)."""
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, [], None
    while pos < len(b):
        ln, typ = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + ln]
        if typ == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", data)
            assert (depth, ctype, interlace) == (8, 2, 0), "only 8-bit RGB non-interlaced"
        elif typ == b"IDAT":
            idat.append(data)
        pos += 12 + ln
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8)
    stride = w * 3
    raw = raw.reshape(h, stride + 1)
    out = np.zeros((h, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    bpp = 3
    for y in range(h):
        ft = int(raw[y, 0])
        line = raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:
            cur = line.reshape(-1, bpp).cumsum(axis=0).reshape(-1) & 255
        else:
            cur = np.zeros(stride, dtype=np.int32)
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                up = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 3:
                    pred = (a + up) >> 1
                else:
                    p = a + up - c
                    pa, pb, pc = abs(p - a), abs(p - up), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (up if pb <= pc else c)
                cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, 3)


GOLDEN = {
    "cover": ("cover.png", 2000, 2000),
    "pawn": ("pawn.png", 1000, 1000),
    "tea set": ("teaset.png", 1750, 1000),
}


def golden_image(name):
    fn, w, h = GOLDEN[name]
    npz = os.path.join(REPO, "tests", "golden", fn.replace(".png", ".rgb.npz"))
    if os.path.exists(npz):
        return np.load(npz)["rgb"]
    img = read_png_rgb8(os.path.join(REPO, "tests", "golden", fn))
    assert img.shape == (h, w, 3)
    return img


def write_ppm(path, rgb):
    """P3 file in the reference's layout (canvas.rs:29-44) from an HxWx3 uint8 array."""
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = a.shape[0], a.shape[1]
    if lib().orc_write_ppm(path.encode(), w, h, a.ctypes.data_as(C.POINTER(C.c_uint8))) != 0:
        raise OSError(f"could not write {path}")
