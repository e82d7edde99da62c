"""Host side of the product (no GPU needed): the C ABI library loads and exports every
declared symbol, the host mirror pre-computes the same bits as the oracle, the OBJ loader
and the example scenes have the reference's shapes, and the core refuses to run without a
GPU instead of falling back to anything."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import scenes

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for hdr in ("rt_core.h", "rt_host_c.h"):
        text = open(os.path.join(REPO, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(rth?_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_exports_every_declared_symbol(rt):
    L = rt.lib()
    syms = declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/ but not exported"


def test_struct_sizes_match_header(rt):
    # rt_material: 7 doubles + 2 int32 + 3+3+16 doubles; rt_camera: 2 int32 + 3 + 16 doubles
    assert C.sizeof(rt.Material) == 8 * 7 + 8 + 8 * 22
    assert C.sizeof(rt.CameraABI) == 8 + 8 * 19
    assert C.sizeof(rt.Stats) == 8 * 12 + 16 + 8 * 5


def test_host_matrix_math_is_bitwise_the_oracles(rt, O):
    rng = np.random.default_rng(11)
    for _ in range(200):
        args = rng.uniform(-100, 100, 3)
        ang = rng.uniform(-7, 7)
        for name, a in (("translation", args), ("scaling", args), ("rotation_x", (ang,)),
                        ("rotation_y", (ang,)), ("rotation_z", (ang,))):
            assert np.array_equal(getattr(rt, name)(*a), getattr(O, name)(*a))
        m = rt.mat_mul(rt.translation(*args), rt.mat_mul(rt.rotation_y(ang), rt.scaling(*(np.abs(args) + 0.1))))
        assert np.array_equal(m, O.mat_mul(O.translation(*args), O.mat_mul(O.rotation_y(ang), O.scaling(*(np.abs(args) + 0.1)))))
        assert np.array_equal(rt.mat_inverse(m), O.mat_inverse(m))
        g = rng.uniform(-9, 9, (4, 4))
        assert np.array_equal(rt.mat_inverse(g), O.mat_inverse(g))
        assert rt.mat_determinant(g) == O.mat_determinant(g)
    f, t, u = rng.uniform(-5, 5, 3), rng.uniform(-5, 5, 3), rng.uniform(-1, 1, 3)
    assert np.array_equal(rt.view_transform(f, t, u), O.view_transform(f, t, u))
    with pytest.raises(rt.RtError):
        rt.mat_inverse(np.zeros((4, 4)))  # matrix.rs:158 asserts det != 0


@pytest.mark.parametrize("name", ["cover", "pawn", "tea set", "test_scene"])
def test_example_cameras_match_oracle(rt, O, name):
    for size in ((0, 0), (1920, 1080), (400, 400), (33, 57)):
        w = rt.World.example(name, *size)
        cam = w.camera_abi()
        oc = O.Scene.example(name, *size).camera()
        assert (cam.hsize, cam.vsize) == (oc["hsize"], oc["vsize"])
        assert cam.half_width == oc["half_width"] and cam.half_height == oc["half_height"]
        assert cam.pixel_size == oc["pixel_size"]
        assert np.array_equal(np.array(cam.inv[:]).reshape(4, 4), oc["inv"])


def test_obj_loader(rt):  # obj_parser.rs:107-132 and the asset triangle counts
    w = rt.World()
    g = w.add_obj("\n        v -1 1 0\n        v -1 0 0\n        v 1 0 0\n        v 1 1 0\n        f 1 2 3\n        f 1 3 4")
    assert w.group_child_count(g) == 2
    g = w.add_obj("\n        v -1 1 0\n        v -1 0 0\n        v 1 0 0\n        v 1 1 0\n        v 0 2 0\n        f 1 2 3 4 5")
    assert w.group_child_count(g) == 3
    # CRLF files, "v  x" double spaces, a/b/c and a//c tokens, tabs inside a token drop it
    g = w.add_obj("v  0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nvn 0 0 1\r\nf 1//1 2//1 3//1\r\nf 1/5/1 2/5/1 3/5/1\r\n# c\r\nvt 0 0\r\n")
    assert w.group_child_count(g) == 2
    for name, n in (("pawn-chess.obj", 31776), ("teapot.obj", 6320), ("teacup.obj", 8378)):
        w = rt.World()
        g = w.add_obj(open(os.path.join(rt.OBJ_DIR, name), newline="").read())
        assert w.group_child_count(g) == n
    with pytest.raises(rt.RtError):
        rt.World().add_obj("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1/2 2/2 3/2\n")  # the reference panics


def test_unknown_example_and_missing_assets(rt):
    with pytest.raises(rt.RtError):
        rt.World.example("no such scene")      # main.rs:50 "Unrecognized scene"
    with pytest.raises(rt.RtError):
        rt.World.example("pawn", obj_dir="/nonexistent")  # examples.rs:23 expect()


def test_core_rejects_bad_input(rt):
    L = rt.lib()
    s = L.rt_scene_create()
    try:
        assert L.rt_scene_add_material(s, None) < 0
        cam = rt.CameraABI()
        # render before commit -> state error
        assert L.rt_render(s, C.byref(cam), 5, 0, 0, None, None, None) == 2
        assert b"not committed" in L.rt_last_error()
        assert L.rt_shard_tiles(64, 64, 2, 2) == -1
        assert L.rt_shard_tiles(64, 64, 1, 3) == 1 and L.rt_shard_tiles(64, 64, 0, 3) == 2
        assert L.rt_shard_tiles(33, 17, 0, 1) == 2
    finally:
        L.rt_scene_destroy(C.c_void_p(s))


def test_projective_transform_is_refused(rt):
    w = rt.World()
    m = np.eye(4)
    m[3, 0] = 0.25
    w.add_shape(rt.SPHERE, m)
    w.set_camera(4, 4, 1.0)
    if rt.device_count() == 0:
        with pytest.raises(rt.RtError):
            w.commit()
    else:
        with pytest.raises(rt.RtError, match="affine"):
            w.commit()


def test_no_gpu_means_no_render(rt):
    """Without a GPU the product must fail loudly: there is no CPU path behind the ABI."""
    if rt.device_count() > 0:
        pytest.skip("a GPU is visible")
    w = rt.World.example("cover", 16, 16)
    with pytest.raises(rt.RtError, match="no HIP device|no CPU path"):
        w.commit()
    with pytest.raises(rt.RtError):
        w.render()


def test_product_does_not_reference_the_oracle():
    bad = []
    for root, _, files in os.walk(os.path.join(REPO, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                text = open(os.path.join(root, f), errors="ignore").read()
                if re.search(r"\boracle\b|liboracle|rt_oracle", text):
                    bad.append(os.path.join(root, f))
    assert not bad, bad


def test_ppm_writer_format(rt, tmp_path):  # canvas.rs:29-44
    rgb = np.arange(18, dtype=np.uint8).reshape(2, 3, 3)
    p = str(tmp_path / "c.ppm")
    rt.write_ppm(p, rgb)
    assert open(p).read() == "P3\n3 2\n255\n0 1 2 3 4 5 6 7 8 \n9 10 11 12 13 14 15 16 17 \n"


@pytest.mark.parametrize("size", [(1, 1), (7, 3), (300, 200), (1000, 90)])
def test_png_writer_round_trip(rt, O, tmp_path, size):
    """Canvas::write_to_png against the test suite's PNG decoder (the one that reads the
    reference's renders): sizes below, at and above one stored deflate block (65535 bytes)."""
    w, h = size
    rgb = np.random.default_rng(w * 1000 + h).integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    p = str(tmp_path / "c.png")
    rt.write_png(p, rgb)
    assert np.array_equal(O.read_png_rgb8(p), rgb)  # inflates the stream: Adler-32 checked by zlib
    import struct
    import zlib
    b, pos, kinds = open(p, "rb").read(), 8, []
    while pos < len(b):  # every chunk's CRC-32, which the decoder above does not look at
        ln, typ = struct.unpack(">I4s", b[pos:pos + 8])
        crc, = struct.unpack(">I", b[pos + 8 + ln:pos + 12 + ln])
        assert crc == zlib.crc32(b[pos + 4:pos + 8 + ln])
        kinds.append(typ)
        pos += 12 + ln
    assert kinds == [b"IHDR", b"IDAT", b"IEND"] and pos == len(b)
    with pytest.raises(rt.RtError):
        rt.write_png(str(tmp_path / "no_such_dir" / "c.png"), rgb)


# ---- every mirror of the C structs has the C compiler's layout ---------------------------------
RUST_SIZES = {"f64": 8, "i64": 8, "u64": 8, "i32": 4, "u32": 4, "u8": 1}


def rust_struct_sizes():
    """Size and alignment of every #[repr(C)] struct of the Rust shim, by the C layout rules."""
    import re
    src = open(os.path.join(REPO, "integration", "rust", "src", "lib.rs")).read()
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[[^\]]*\])*\s*pub struct (\w+)\s*\{(.*?)\n\}", src, re.S):
        off, align = 0, 1
        for f in re.finditer(r"pub \w+:\s*([^,\n]+),", m.group(2)):
            ty = f.group(1).strip()
            arr = re.match(r"\[(\w+);\s*(\d+)\]", ty)
            if arr:
                size, al = RUST_SIZES[arr.group(1)] * int(arr.group(2)), RUST_SIZES[arr.group(1)]
            elif ty.startswith("*"):
                size = al = 8
            else:
                size = al = RUST_SIZES[ty]
            off = (off + al - 1) // al * al + size
            align = max(align, al)
        out[m.group(1)] = (off + align - 1) // align * align
    return out


def test_every_mirror_has_the_c_layout(rt):
    """tests/c_abi/sizes.c (gcc -std=c11 on include/rt_core.h) gives the truth; the ctypes
    mirrors of the Python binding and the repr(C) mirrors of integration/rust must agree, struct
    by struct."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "tests", "c_abi"), "sizes"])
    c = dict(line.split() for line in subprocess.check_output(
        [os.path.join(REPO, "tests", "c_abi", "sizes")], text=True).splitlines())
    c = {k: int(v) for k, v in c.items()}
    assert set(c) == {"rt_material", "rt_group", "rt_primitive", "rt_mesh", "rt_camera", "rt_stats", "rt_scene_info"}
    rust = rust_struct_sizes()
    pairs = {"rt_material": "RtMaterial", "rt_group": "RtGroup", "rt_primitive": "RtPrimitive", "rt_mesh": "RtMesh",
             "rt_camera": "RtCamera", "rt_stats": "RtStats", "rt_scene_info": "RtSceneInfo"}
    assert set(rust) == set(pairs.values())  # every repr(C) struct of the shim is named here
    for cname, rname in pairs.items():
        assert rust[rname] == c[cname], (cname, rust[rname], c[cname])
    assert C.sizeof(rt.CameraABI) == c["rt_camera"]
    assert C.sizeof(rt.Stats) == c["rt_stats"]
    assert C.sizeof(rt.SceneInfo) == c["rt_scene_info"]


def test_rust_shim_binds_every_core_entry_point():
    """The extern block of the Rust shim names every function include/rt_core.h declares."""
    import re
    hdr = open(os.path.join(REPO, "include", "rt_core.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)))
    src = open(os.path.join(REPO, "integration", "rust", "src", "lib.rs")).read()
    bound = set(re.findall(r"pub fn (rt_[a-z0-9_]+)", src))
    assert declared and declared <= bound, sorted(declared - bound)


def test_plain_c_host_compiles():
    """tests/c_abi/cover.c includes include/rt_core.h from C11 with -Wall -Wextra -Werror -pedantic."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "tests", "c_abi")])
    assert os.path.exists(os.path.join(REPO, "tests", "c_abi", "cover"))
