"""Parity of the HIP render core (through the C ABI) with the CPU oracle and with the
reference's own committed renders.  All tests need an MI355X: `pytest -m gpu`.

Bars: after 8-bit quantisation >= 99.9 % of the pixels equal,
<= 1/255 on the rest.  The f64 path reaches more: geometry decisions are bit-identical, so
every pixel is equal; the f64 colours differ from the oracle's only through pow() in the
specular term (device libm vs glibc), bounded here by 1e-12.
"""
import math

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu

F64_TOL = 1e-12
S2 = math.sqrt(2.0)


@pytest.fixture(autouse=True, params=["wavefront", "megakernel"])
def pipeline(request, rt):
    """Every test runs once per kernel organisation; both must meet the same bar."""
    old = rt.DEFAULT_PIPELINE
    rt.DEFAULT_PIPELINE = {"wavefront": rt.PIPELINE_WAVEFRONT,
                           "megakernel": rt.PIPELINE_MEGAKERNEL}[request.param]
    yield request.param
    rt.DEFAULT_PIPELINE = old


def u8_report(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32)).max(axis=-1)
    return float((d == 0).mean()), int(d.max())


def build_both(rt, O, builder, *args, bvh_builder=0, **kw):
    w = rt.World()
    builder(w, rt.material, rt, *args, **kw)
    s = O.Scene()
    builder(s, O.material, rt, *args, **kw)  # same matrices (product's host math) on both sides
    w.commit(bvh_builder=bvh_builder)
    return w, s


# ---- the reference's own renders -------------------------------------------------------------
@pytest.mark.parametrize("name", ["cover", "pawn", "tea set"])
def test_default_size_render_equals_reference_png(rt, O, name):
    gold = O.golden_image(name)
    w = rt.World.example(name).commit()
    u8, _, _ = w.render()
    exact, worst = u8_report(u8, gold)
    assert exact >= 0.999 and worst <= 1, (exact, worst)
    assert exact == 1.0


# ---- the benchmark configurations against the oracle ---------------------------------------------
def test_cover_400_full_image_and_counters(rt, O):
    w = rt.World.example("cover", 400, 400).commit()
    u8, f64, st = w.render(want_f64=True, want_stats=True)
    of64, ou8, cnt = O.Scene.example("cover", 400, 400).render()
    assert np.array_equal(u8, ou8)
    assert np.abs(f64 - of64).max() <= F64_TOL
    assert (st.rays_primary, st.rays_shadow, st.rays_reflect, st.rays_refract, st.depth0_skipped) == (
        cnt["primary"], cnt["shadow"], cnt["reflect"], cnt["refract"], cnt["depth0"])


@pytest.mark.parametrize("name", ["pawn", "tea set"])
def test_1080p_full_image_vs_oracle(rt, O, name):
    """Every pixel of the 1920x1080 configs: oracle in culling mode (bit-identical to its
    brute-force mode, see test_oracle_kats.test_bvh_mode_is_bit_identical), plus a brute-force
    sample."""
    w = rt.World.example(name, 1920, 1080).commit()
    u8, f64, st = w.render(want_f64=True, want_stats=True)
    sc = O.Scene.example(name, 1920, 1080).use_bvh()
    of64, ou8, cnt = sc.render()
    assert np.array_equal(u8, ou8)
    assert np.abs(f64 - of64).max() <= F64_TOL
    assert (st.rays_primary, st.rays_shadow, st.rays_reflect, st.rays_refract, st.depth0_skipped) == (
        cnt["primary"], cnt["shadow"], cnt["reflect"], cnt["refract"], cnt["depth0"])
    brute = O.Scene.example(name, 1920, 1080)
    rng = np.random.default_rng(5)
    xs, ys = rng.integers(0, 1920, 600), rng.integers(0, 1080, 600)
    bf, bu, _ = brute.render_pixels(xs, ys)
    assert np.array_equal(bu, u8[ys, xs])
    assert np.abs(bf - f64[ys, xs]).max() <= F64_TOL


def test_test_scene_cylinders_cones_patterns(rt, O):
    """examples.rs:270-404 (unreachable from the reference CLI): cylinder, cone, Rings and
    Checkered with pattern transforms, two transparent shapes."""
    w = rt.World.example("test_scene", 480, 270).commit()
    u8, f64, _ = w.render(want_f64=True)
    of64, ou8, _ = O.Scene.example("test_scene", 480, 270).render()
    assert np.array_equal(u8, ou8)
    assert np.abs(f64 - of64).max() <= F64_TOL


def test_synthetic_scene_small(rt, O):
    """The stress scene at 20k triangles / 320x180: full image vs oracle culling mode and a
    brute-force sample."""
    w = rt.World.example("synthetic", 320, 180, synthetic_tris=20000).commit()
    u8, f64, _ = w.render(want_f64=True)
    sc = O.Scene.example("synthetic", 320, 180, synthetic_tris=20000).use_bvh()
    of64, ou8, _ = sc.render()
    assert np.array_equal(u8, ou8)
    assert np.abs(f64 - of64).max() <= F64_TOL
    brute = O.Scene.example("synthetic", 320, 180, synthetic_tris=20000)
    rng = np.random.default_rng(9)
    xs, ys = rng.integers(0, 320, 400), rng.integers(0, 180, 400)
    bf, bu, _ = brute.render_pixels(xs, ys)
    assert np.array_equal(bu, u8[ys, xs])


def test_synthetic_full_size(rt, O, pipeline):
    """The 1M-triangle stress config at its full 3840x2160: every pixel against the oracle's
    culling mode, a 1024-pixel sample against its brute-force mode (every ray against every
    triangle, like the reference).  The oracle work is shared by running once."""
    if pipeline != "wavefront":
        pytest.skip("both pipelines are rendered in the wavefront instance of this test")
    n = 1000000
    w = rt.World.example("synthetic", 3840, 2160, synthetic_tris=n).commit()
    a, _, sta = w.render(pipeline=rt.PIPELINE_WAVEFRONT, want_stats=True)
    b, _, stb = w.render(pipeline=rt.PIPELINE_MEGAKERNEL, want_stats=True)
    assert np.array_equal(a, b)
    sc = O.Scene.example("synthetic", 3840, 2160, synthetic_tris=n).use_bvh()
    _, ou8, cnt = sc.render()
    exact, worst = u8_report(a, ou8)
    assert exact == 1.0, (exact, worst)
    for st in (sta, stb):
        assert (st.rays_primary, st.rays_shadow, st.rays_reflect, st.rays_refract) == (
            cnt["primary"], cnt["shadow"], cnt["reflect"], cnt["refract"])
    brute = O.Scene.example("synthetic", 3840, 2160, synthetic_tris=n)
    rng = np.random.default_rng(5)
    xs, ys = rng.integers(0, 3840, 1024), rng.integers(0, 2160, 1024)
    _, bu, _ = brute.render_pixels(xs, ys)
    assert np.array_equal(bu, a[ys, xs])


def test_pawn_8k_sample(rt, O, pipeline):
    """The 7680x4320 pawn frame (33 M pixels): 150 000 random pixels and 8 full rows against
    the oracle, and the frame's ray count against 16x the per-pixel average of the sample."""
    w = rt.World.example("pawn", 7680, 4320).commit()
    u8, _, st = w.render(want_stats=True)
    sc = O.Scene.example("pawn", 7680, 4320).use_bvh()
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.integers(0, 7680, 150000), np.tile(np.arange(7680), 8)]).astype(np.int32)
    ys = np.concatenate([rng.integers(0, 4320, 150000), np.repeat(np.arange(8) * 540 + 270, 7680)]).astype(np.int32)
    _, ou8, cnt = sc.render_pixels(xs, ys)
    exact, worst = u8_report(u8[ys, xs], ou8)
    assert exact == 1.0, (exact, worst)
    assert st.rays_primary == 7680 * 4320


@pytest.mark.parametrize("name,size,ntris", [("pawn", (480, 270), 0), ("tea set", (480, 270), 0),
                                             ("synthetic", (320, 180), 50000)])
def test_device_built_bvh(rt, O, name, size, ntris):
    """bvh_builder=1 (Morton codes, radix sort, Karras tree and refit on the GPU): the
    hierarchy only culls, so the frame must be the same bits as with the host SAH build,
    and equal to the oracle."""
    a = rt.World.example(name, *size, synthetic_tris=ntris).commit(bvh_builder=0)
    b = rt.World.example(name, *size, synthetic_tris=ntris).commit(bvh_builder=1)
    assert b.info()["bvh_builder"] == 1 and b.info()["n_bvh_nodes"] > 0
    ua, fa, _ = a.render(want_f64=True)
    ub, fb, sb = b.render(want_f64=True, want_stats=True)
    assert np.array_equal(fa, fb)
    assert np.array_equal(ua, ub)
    sc = O.Scene.example(name, *size, synthetic_tris=ntris).use_bvh()
    _, ou8, cnt = sc.render()
    assert np.array_equal(ub, ou8)
    assert sb.rays_shadow == cnt["shadow"]


def test_device_built_bvh_with_and_without_rebuilt_subtrees(rt, monkeypatch):
    """The device builder rebuilds the radix tree's subtrees of up to 64 triangles by the
    surface-area heuristic (RT_LBVH_TREELETS=0 keeps them): another tree, fewer triangle tests,
    the same frame."""
    frames, tests = [], []
    for flag in ("1", "0"):
        monkeypatch.setenv("RT_LBVH_TREELETS", flag)
        w = rt.World.example("pawn", 320, 180).commit(bvh_builder=1)
        u8, f64, st = w.render(want_f64=True, want_stats=True)
        frames.append(f64)
        tests.append(st.tri_tests + 4 * st.nodes_visited)
    assert np.array_equal(frames[0], frames[1])
    assert tests[0] < tests[1]


def test_bvh_builder_by_mesh_size(rt, O):
    """bvh_builder=2: meshes of 100 000 triangles and more are built on the device, smaller ones
    on the host; the frame is the host build's."""
    n = 120000
    a = rt.World.example("synthetic", 240, 135, synthetic_tris=n).commit(bvh_builder=0)
    b = rt.World.example("synthetic", 240, 135, synthetic_tris=n).commit(bvh_builder=2)
    assert b.info()["bvh_builder"] == 2 and b.info()["n_bvh_nodes"] != a.info()["n_bvh_nodes"]
    ua, fa, _ = a.render(want_f64=True)
    ub, fb, _ = b.render(want_f64=True)
    assert np.array_equal(ua, ub) and np.array_equal(fa, fb)
    c = rt.World.example("pawn", 96, 54).commit(bvh_builder=2)  # 31 776 triangles: host build
    d = rt.World.example("pawn", 96, 54).commit(bvh_builder=0)
    assert c.info()["n_bvh_nodes"] == d.info()["n_bvh_nodes"]
    with pytest.raises(rt.RtError):
        rt.World.example("cover", 16, 16).commit(bvh_builder=3)


@pytest.mark.parametrize("name,size,ntris", [("pawn", (1920, 1080), 0), ("tea set", (1920, 1080), 0),
                                             ("cover", (400, 400), 0), ("synthetic", (960, 540), 200000)])
def test_f32_fast_mode_meets_the_stated_tolerance(rt, O, name, size, ntris, pipeline):
    """RT_MODE_F32_FAST searches triangles in f32 with error bounds and decides in f64 whatever
    the f32 values cannot (edges, the EPSILON cut, near-equal t): held to BOTH clauses of the
    stated tolerance against the oracle — >= 99.9 % of the pixels equal after quantisation and
    no channel off by more than 1/255 — and, since every decision ends in the f64 arithmetic,
    expected to be the parity mode's frame."""
    if pipeline != "wavefront":
        pytest.skip("the fast mode runs on the wavefront pipeline")
    w = rt.World.example(name, *size, synthetic_tris=ntris).commit()
    u8, f64, st = w.render(mode=rt.MODE_F32_FAST, want_f64=True, want_stats=True)
    pu8, pf64, pst = w.render(mode=rt.MODE_F64_PARITY, want_f64=True, want_stats=True)
    _, ou8, _ = O.Scene.example(name, *size, synthetic_tris=ntris).use_bvh().render()
    exact, worst = u8_report(u8, ou8)
    assert exact >= 0.999 and worst <= 1, (exact, worst)
    assert np.array_equal(u8, pu8) and np.array_equal(f64, pf64)
    assert (st.rays_shadow, st.rays_reflect, st.rays_refract) == (pst.rays_shadow, pst.rays_reflect, pst.rays_refract)
    with pytest.raises(rt.RtError):
        w.render(mode=rt.MODE_F32_FAST, pipeline=rt.PIPELINE_MEGAKERNEL)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_f32_fast_mode_random_scenes(rt, O, seed, pipeline):
    if pipeline != "wavefront":
        pytest.skip("the fast mode runs on the wavefront pipeline")
    w, s = build_both(rt, O, scenes.random_scene, seed)
    u8, f64, _ = w.render(mode=rt.MODE_F32_FAST, want_f64=True)
    of64, ou8, _ = s.render()
    exact, worst = u8_report(u8, ou8)
    assert exact >= 0.999 and worst <= 1, (exact, worst)
    assert np.abs(f64 - of64).max() <= 1e-11


def test_auto_pipeline_renders_the_same_frame(rt, O):
    """RT_PIPELINE_AUTO times both pipelines once per frame and keeps the faster; whatever it
    picks, the frame is the oracle's."""
    w = rt.World.example("tea set", 320, 180).commit()
    a, fa, _ = w.render(pipeline=rt.PIPELINE_AUTO, want_f64=True)
    b, fb, st = w.render(pipeline=rt.PIPELINE_AUTO, want_f64=True, want_stats=True)
    _, ou8, cnt = O.Scene.example("tea set", 320, 180).use_bvh().render()
    assert np.array_equal(a, ou8) and np.array_equal(b, ou8) and np.array_equal(fa, fb)
    assert st.rays_shadow == cnt["shadow"]


def test_frames_in_flight_on_several_streams(rt, pipeline):
    """Frames issued back to back on different streams overlap on the GPU (each stream has its
    own ray queues); every one of them must be the frame a lone render produces."""
    import torch
    w = rt.World.example("pawn", 640, 360).commit()
    ref, _, _ = w.render()
    streams = [torch.cuda.Stream() for _ in range(5)]  # more streams than buffer sets
    outs = [torch.zeros((360, 640, 3), dtype=torch.uint8, device="cuda") for _ in range(10)]
    for rep in range(2):
        for k, out in enumerate(outs):
            st = streams[k % len(streams)]
            w.render_device(d_u8=out.data_ptr(), stream=st.cuda_stream)
        torch.cuda.synchronize()
        for out in outs:
            assert np.array_equal(out.cpu().numpy(), ref)
            out.zero_()


def test_cli_writes_the_reference_ppm(rt, O, tmp_path, pipeline):
    """rt_cli (the reference's command line, main.rs:13-55, on the GPU core): the P3 file it
    writes equals the oracle's writer (canvas.rs:29-44) on the oracle's frame, byte for byte."""
    import os
    import subprocess
    if pipeline != "wavefront":
        pytest.skip("the command line has one pipeline")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "cli.ppm"
    r = subprocess.run([os.path.join(repo, "raytracer_amd", "rt_cli"), "-e", "pawn", "-t", "6", "--width", "160",
                        "--height", "90", "--obj-dir", os.path.join(repo, "assets", "obj"), "--out", str(out),
                        "--stats"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Rendering image" in r.stdout and "Mrays/s" in r.stdout
    _, ou8, _ = O.Scene.example("pawn", 160, 90).render()
    want = tmp_path / "oracle.ppm"
    O.write_ppm(str(want), ou8)
    assert out.read_bytes() == want.read_bytes()
    assert out.read_bytes().startswith(b"P3\n160 90\n255\n")
    # the same frame as a PNG (an output name ending in .png)
    png = tmp_path / "cli.png"
    r = subprocess.run([os.path.join(repo, "raytracer_amd", "rt_cli"), "-e", "pawn", "--width", "160", "--height", "90",
                        "--obj-dir", os.path.join(repo, "assets", "obj"), "--out", str(png)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(O.read_png_rgb8(str(png)), ou8)
    # a turntable of three frames over one committed scene: frame 0 is the example's own view
    r = subprocess.run([os.path.join(repo, "raytracer_amd", "rt_cli"), "-e", "pawn", "--width", "160", "--height", "90",
                        "--obj-dir", os.path.join(repo, "assets", "obj"), "--out", str(tmp_path / "turn.png"),
                        "--sweep", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    turn = [O.read_png_rgb8(str(tmp_path / f"turn_{k:03d}.png")) for k in range(3)]
    assert np.array_equal(turn[0], ou8)
    assert not np.array_equal(turn[1], ou8) and not np.array_equal(turn[2], turn[1])


def test_plain_c_host_renders_the_default_world(rt, O, tmp_path, pipeline):
    """tests/c_abi/cover.c — a C11 program built with gcc against include/rt_core.h — builds the
    reference's default world (world.rs:180-204) through rt_scene_add_*, renders it with
    rt_render; the frame it writes equals the oracle's."""
    import os
    import subprocess
    import zlib
    if pipeline != "wavefront":
        pytest.skip("the C host names its pipeline itself")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(repo, "tests", "c_abi", "cover")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(exe)])
    out = tmp_path / "cover.rgb"
    r = subprocess.run([exe, str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = np.frombuffer(out.read_bytes(), dtype=np.uint8).reshape(64, 96, 3)
    assert f"crc32={zlib.crc32(got.tobytes()):08x}" in r.stdout and "primitives=2" in r.stdout

    def b(T, M, X):
        scenes.populated_world(T, M, X)
        T.set_camera(96, 64, math.pi / 3, X.view_transform((0, 0, -5), (0, 0, 0), (0, 1, 0)))
    s = O.Scene()
    b(s, O.material, rt)
    _, ou8, cnt = s.render()
    assert np.array_equal(got, ou8)
    assert f"rays={cnt['primary'] + cnt['shadow'] + cnt['reflect'] + cnt['refract']} " in r.stdout


def test_camera_sweep_equals_single_frames(rt, O, pipeline):
    """render_sequence (frames in flight on streams of the library's own, pinned copies back):
    every frame equals the one rendered alone, for new cameras and for cameras seen before."""
    import ctypes as C
    import torch
    pipe = rt.PIPELINE_WAVEFRONT if pipeline == "wavefront" else rt.PIPELINE_MEGAKERNEL
    w = rt.World.example("pawn", 96, 54)
    base = w.camera_abi()
    w.commit()
    cams = []
    for k in range(7):  # the example's camera with its inverse turned about the y axis, two sizes
        a = 0.15 * k
        rot = np.array([[math.cos(a), 0, math.sin(a), 0], [0, 1, 0, 0], [-math.sin(a), 0, math.cos(a), 0],
                        [0, 0, 0, 1]], dtype=np.float64)
        c = rt.CameraABI()
        C.memmove(C.byref(c), C.byref(base), C.sizeof(c))
        inv = rot @ np.array(list(base.inv), dtype=np.float64).reshape(4, 4)
        c.inv[:] = inv.reshape(16)
        cams.append(c)
    alone = []
    for c in cams:
        buf = torch.zeros((c.vsize, c.hsize, 3), dtype=torch.uint8, device="cuda")
        w.render_device(d_u8=buf.data_ptr(), pipeline=pipe, cam=c)
        torch.cuda.synchronize()
        alone.append(buf.cpu().numpy())
    assert any(not np.array_equal(alone[0], a) for a in alone[1:])
    for inflight in (1, 3, 8):
        for rep in range(2):
            got = w.render_sequence(cams, pipeline=pipe, inflight=inflight)
            assert len(got) == len(cams)
            for g, a in zip(got, alone):
                assert np.array_equal(g, a)
    assert w.render_sequence([], pipeline=pipe) == []


def turned(rt, base, angle, hsize=None, vsize=None):
    """The camera `base` with its inverse transform turned about the y axis."""
    import ctypes as C
    rot = np.array([[math.cos(angle), 0, math.sin(angle), 0], [0, 1, 0, 0],
                    [-math.sin(angle), 0, math.cos(angle), 0], [0, 0, 0, 1]], dtype=np.float64)
    c = rt.CameraABI()
    C.memmove(C.byref(c), C.byref(base), C.sizeof(c))
    c.inv[:] = (rot @ np.array(list(base.inv), dtype=np.float64).reshape(4, 4)).reshape(16)
    return c


def test_sweep_of_new_cameras_is_issued_on_estimates(rt, O, pipeline, monkeypatch):
    """Cameras the scene has never rendered: after the first frame the sweep issues them without
    read-backs, sized from the latest finished frame.  Frames equal the ones rendered alone —
    also when an estimate does not hold (a frame of the empty sky followed by the glass pawn:
    its ray queues outgrow the estimate and the frame is rendered again)."""
    import ctypes as C
    import os
    import torch
    if pipeline != "wavefront":
        pytest.skip("estimates belong to the wavefront pipeline")
    # whole frames: a frame split over internal streams is always issued the exact way, and the
    # queue lengths it leaves behind are those of its parts
    monkeypatch.setenv("RT_FRAME_SPLIT", "1")
    w = rt.World.example("pawn", 640, 360)
    base = w.camera_abi()
    w.commit()
    assert w.sweep_counters() == (0, 0)
    sweep = [turned(rt, base, 0.01 * k) for k in range(1, 9)]
    got = w.render_sequence(sweep, inflight=4)
    issued, again = w.sweep_counters()
    assert issued >= 4 and again == 0   # everything after the first frames in flight
    # a frame that sees nothing (the camera turns its head straight up, away from floor and pawn)
    # becomes the latest finished frame; the estimate it gives the next new camera cannot hold
    buf = torch.zeros((360, 640, 3), dtype=torch.uint8, device="cuda")
    blank = None
    for pitch in (math.pi / 2, -math.pi / 2):
        c = rt.CameraABI()
        C.memmove(C.byref(c), C.byref(base), C.sizeof(c))
        rx = np.array([[1, 0, 0, 0], [0, math.cos(pitch), -math.sin(pitch), 0],
                       [0, math.sin(pitch), math.cos(pitch), 0], [0, 0, 0, 1]], dtype=np.float64)
        c.inv[:] = (np.array(list(base.inv), dtype=np.float64).reshape(4, 4) @ rx).reshape(16)
        w.render_device(d_u8=buf.data_ptr(), cam=c)
        torch.cuda.synchronize()
        if len(np.unique(buf.cpu().numpy())) == 1:
            blank = c
    assert blank is not None
    w.render_device(d_u8=buf.data_ptr(), cam=blank)  # seen before: moves its queue lengths to the front
    torch.cuda.synchronize()
    back = turned(rt, base, 0.005)
    got += w.render_sequence([back, back], inflight=1)
    issued2, again2 = w.sweep_counters()
    assert issued2 == issued + 1 and again2 == 1
    for c, g in zip(sweep + [back, back], got):
        w.render_device(d_u8=buf.data_ptr(), cam=c)
        torch.cuda.synchronize()
        assert np.array_equal(g, buf.cpu().numpy())
    assert len(np.unique(got[-1])) > 50 and not np.array_equal(got[-1], got[0])


def test_estimated_sweep_after_many_caller_streams(rt, pipeline, monkeypatch):
    """More streams than the scene keeps ray-queue sets for: twelve caller streams render first,
    then a sweep of new cameras with seven frames in flight (eight slots of the scene's own) has
    to take sets over.  A set whose frame was issued on an estimate must not lose its verdict:
    the sweep starts behind a frame of empty sky, so the first estimates cannot hold, and every
    frame must still equal the one rendered alone."""
    import ctypes as C
    import os
    import torch
    if pipeline != "wavefront":
        pytest.skip("estimates belong to the wavefront pipeline")
    # whole frames: a frame split over internal streams is always issued the exact way, and the
    # queue lengths it leaves behind are those of its parts
    monkeypatch.setenv("RT_FRAME_SPLIT", "1")
    w = rt.World.example("pawn", 320, 180)
    base = w.camera_abi()
    w.commit()
    buf = torch.zeros((180, 320, 3), dtype=torch.uint8, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(12)]
    for st in streams:
        w.render_device(d_u8=buf.data_ptr(), stream=st.cuda_stream)
        torch.cuda.synchronize()
    blank = None
    for pitch in (math.pi / 2, -math.pi / 2):
        c = rt.CameraABI()
        C.memmove(C.byref(c), C.byref(base), C.sizeof(c))
        rx = np.array([[1, 0, 0, 0], [0, math.cos(pitch), -math.sin(pitch), 0],
                       [0, math.sin(pitch), math.cos(pitch), 0], [0, 0, 0, 1]], dtype=np.float64)
        c.inv[:] = (np.array(list(base.inv), dtype=np.float64).reshape(4, 4) @ rx).reshape(16)
        w.render_device(d_u8=buf.data_ptr(), cam=c)  # the null stream: a thirteenth set
        torch.cuda.synchronize()
        if len(np.unique(buf.cpu().numpy())) == 1:
            blank = c
    assert blank is not None
    w.render_device(d_u8=buf.data_ptr(), cam=blank)
    torch.cuda.synchronize()
    sweep = [turned(rt, base, 0.02 * k) for k in range(1, 17)]
    got = w.render_sequence(sweep, inflight=7)
    issued, again = w.sweep_counters()
    assert issued >= 8 and again >= 1
    for c, g in zip(sweep, got):
        w.render_device(d_u8=buf.data_ptr(), cam=c)
        torch.cuda.synchronize()
        assert np.array_equal(g, buf.cpu().numpy())
    assert len(np.unique(got[0])) > 50


# ---- random and adversarial scenes -------------------------------------------------------------
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_scenes(rt, O, seed):
    w, s = build_both(rt, O, scenes.random_scene, seed)
    u8, f64, _ = w.render(want_f64=True)
    of64, ou8, cnt = s.render()
    assert cnt["nan_t"] == 0
    assert np.array_equal(u8, ou8)
    assert np.abs(f64 - of64).max() <= 1e-11


@pytest.mark.parametrize("n_tris", [1, 2, 3, 13, 16, 17, 63, 64, 65, 129, 400])
@pytest.mark.parametrize("bvh_builder", [0, 1])
def test_mesh_sizes_around_the_leaf_limits(rt, O, n_tris, bvh_builder):
    """Meshes of one triangle, of exactly one leaf, of one leaf plus one, ... with both BVH
    builders: single-leaf roots, empty child slots and the largest leaves."""
    w, s = build_both(rt, O, scenes.random_scene, 40 + n_tris, n_prims=3, n_tris=n_tris, hsize=72, vsize=48,
                      bvh_builder=bvh_builder)
    u8, f64, _ = w.render(want_f64=True)
    of64, ou8, cnt = s.render()
    assert np.array_equal(u8, ou8)
    assert np.abs(f64 - of64).max() <= 1e-11


@pytest.mark.parametrize("bvh_builder", [0, 1])
def test_worst_case_trees(rt, O, bvh_builder):
    """Hundreds of coincident triangles and a geometric row of shrinking ones: the deepest
    trees either builder makes must fit the traversal stack and render the reference's bits."""
    w, s = build_both(rt, O, scenes.deep_tree_scene, bvh_builder=bvh_builder)
    u8, f64, _ = w.render(want_f64=True)
    of64, ou8, _ = s.render()
    assert np.array_equal(u8, ou8)
    assert np.abs(f64 - of64).max() <= 1e-11


def test_equal_t_ties_resolved_by_leaf_order(rt, O):
    w, s = build_both(rt, O, scenes.tie_scene)
    u8, f64, _ = w.render(want_f64=True)
    of64, ou8, _ = s.render()
    assert np.array_equal(u8, ou8)
    assert np.abs(f64 - of64).max() <= F64_TOL
    # the first-added red cube wins over the coincident green one
    assert (u8[..., 0] > 100).any() and not ((u8[..., 1] > 200) & (u8[..., 0] < 50) & (u8[..., 2] < 50)).any()


# ---- single-ray known answers of the reference's tests, through rt_trace_rays ---------------
def trace(w, o, d, depth=5):
    r = w.trace_rays([o], [d], depth)[0]
    return dict(hit=bool(r[0]), t=r[1], order=int(r[2]), u=r[3], v=r[4], inside=bool(r[5]), n1=r[6],
                n2=r[7], point=r[8:11], over=r[11:14], under=r[14:17], eyev=r[17:20], normalv=r[20:23],
                reflectv=r[23:26], schlick=r[26], color=r[27:30], shadowed=bool(r[30]))


def close(a, b, eps=1e-5):
    return np.all(np.abs(np.asarray(a, dtype=float) - np.asarray(b, dtype=float)) < eps)


def test_kat_default_world(rt, O):  # world.rs:206-269, 317-354
    w, s = build_both(rt, O, scenes.populated_world)
    r = trace(w, (0, 0, -5), (0, 0, 1))
    assert r["hit"] and abs(r["t"] - 4.0) < 1e-5 and r["order"] == 0
    assert close(r["color"], (0.38066, 0.47583, 0.2855))
    assert not trace(w, (0, 0, -5), (0, 1, 0))["hit"]
    assert close(r["point"], (0, 0, -1)) and close(r["eyev"], (0, 0, -1)) and close(r["normalv"], (0, 0, -1))
    # against the oracle bit for bit (no specular pow on this path except the lit sphere)
    oi = s.trace_info((0, 0, -5), (0, 0, 1))
    for k, ok in (("point", "point"), ("over", "over_point"), ("under", "under_point"),
                  ("normalv", "normalv"), ("reflectv", "reflectv")):
        assert np.array_equal(r[k], oi[ok]), k
    assert r["t"] == oi["t"]


def test_kat_reflective_plane(rt, O):  # world.rs:339-354
    def b(T, M, X):
        scenes.populated_world(T, M, X)
        T.add_shape(scenes.PLANE, X.translation(0, -1, 0), M(reflective=0.5))
    w, s = build_both(rt, O, b)
    r = trace(w, (0, 0, -3), (0, -S2 / 2, S2 / 2))
    assert close(r["color"], (0.8767572837020907, 0.924340334075874, 0.8291742333283075), 1e-9)
    assert close(r["reflectv"], (0, S2 / 2, S2 / 2))


def test_kat_n1_n2_nested_glass(rt, O):  # intersect.rs:378-406
    w, s = build_both(rt, O, scenes.glass_spheres)
    expected = [(1.0, 1.5), (1.5, 2.0), (2.0, 2.5), (2.5, 2.5), (2.5, 1.5), (1.5, 1.0)]
    for z0, (n1, n2) in zip((-4.0, -1.5, -1.0, 0.5, 1.0, 1.5), expected):
        r = trace(w, (0, 0, z0), (0, 0, 1))
        assert r["n1"] == n1 and r["n2"] == n2, (z0, r["n1"], r["n2"])


def test_kat_transparent_floor_and_schlick(rt, O):  # world.rs:386-441
    def floor(reflective):
        def b(T, M, X):
            scenes.populated_world(T, M, X)
            T.add_shape(scenes.PLANE, X.translation(0, -1, 0),
                        M(transparency=0.5, refractive_index=1.5, reflective=reflective))
            T.add_shape(scenes.SPHERE, X.translation(0, -3.5, -0.5), M(a=(1, 0, 0), ambient=0.5))
        return b
    w, _ = build_both(rt, O, floor(0.0))
    assert close(trace(w, (0, 0, -3), (0, -S2 / 2, S2 / 2))["color"], (0.93642, 0.68642, 0.68642))
    w, _ = build_both(rt, O, floor(0.5))
    assert close(trace(w, (0, 0, -3), (0, -S2 / 2, S2 / 2))["color"], (0.93391, 0.69643, 0.69243))

    def glass(T, M, X):
        T.add_shape(scenes.SPHERE, None, M(transparency=1.0, refractive_index=1.5, reflective=0.5, a=(0, 0, 0)))
        T.add_light((1, 1, 1), (-10, 10, -10))
    w, s = build_both(rt, O, glass)
    r = trace(w, (0, 0.99, -2.0), (0, 0, 1))
    assert abs(r["schlick"] - 0.4888143830387389) < 1e-5
    assert r["schlick"] == s.trace_info((0, 0.99, -2.0), (0, 0, 1))["schlick"]


def test_kat_shadows(rt, O):  # world.rs:272-297: the four is_shadowed cases
    def b(T, M, X):
        scenes.populated_world(T, M, X)
        # a big far plane only to give the probe rays something to hit near the test points
    w, s = build_both(rt, O, b)
    # probe: rays whose first hit's over_point is then tested towards light 0
    for o, d in (((0, 0, -5), (0, 0, 1)), ((0, 0, 5), (0, 0, -1)), ((5, 0.2, 0), (-1, 0, 0))):
        r = trace(w, o, d)
        oi = s.trace_info(o, d)
        assert r["shadowed"] == s.is_shadowed(oi["over_point"])


def test_kat_nested_group_normal(rt, O):  # group.rs:186-219
    w, s = build_both(rt, O, scenes.nested_groups)
    target = np.array([1.7321, 1.1547, -5.5774])
    o = target * 3.0
    d = (target - o) / np.linalg.norm(target - o)
    r, oi = trace(w, o, d), s.trace_info(o, d)
    assert r["hit"] and np.array_equal(r["normalv"], oi["normalv"]) and r["t"] == oi["t"]
    assert close(r["normalv"], (0.28570368184140726, 0.42854315178114105, -0.8571605294481017), 2e-4)


def test_kat_smooth_triangle(rt, O):  # smooth_triangle.rs:175-200
    def b(T, M, X):
        g = T.add_group()
        T.add_triangle([0, 1, 0, -1, 0, 0, 1, 0, 0], [0, 1, 0, -1, 0, 0, 1, 0, 0], parent=g)
        T.add_light((1, 1, 1), (-10, 10, -10))
    w, s = build_both(rt, O, b)
    r = trace(w, (-0.2, 0.3, -2.0), (0, 0, 1))
    assert abs(r["u"] - 0.45) < 1e-5 and abs(r["v"] - 0.25) < 1e-5
    assert close(r["normalv"], (-0.5547, 0.83205, 0.0))


def test_many_random_rays_bitwise(rt, O):
    """20k random rays through the pawn scene: hit t, u, v, n1/n2, normal and the offset points
    must equal the oracle's bit for bit (everything up to the specular pow)."""
    w = rt.World.example("pawn", 64, 64).commit()
    s = O.Scene.example("pawn", 64, 64).use_bvh()
    rng = np.random.default_rng(77)
    n = 4000
    o = np.stack([rng.uniform(-3, 3, n), rng.uniform(0.2, 5, n), rng.uniform(-5, 3, n)], axis=1)
    tgt = np.stack([rng.uniform(-0.8, 0.8, n), rng.uniform(0, 2.5, n), rng.uniform(-0.8, 0.8, n)], axis=1)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out = w.trace_rays(o, d)
    hits = 0
    for i in range(n):
        oi = s.trace_info(o[i], d[i])
        assert bool(out[i, 0]) == oi["hit"]
        if not oi["hit"]:
            continue
        hits += 1
        assert out[i, 1] == oi["t"] and out[i, 3] == oi["u"] and out[i, 4] == oi["v"]
        assert np.array_equal(out[i, 20:23], oi["normalv"])
        assert np.array_equal(out[i, 11:14], oi["over_point"])
        if int(out[i, 2]) < 31776:  # glass triangle: the device evaluates n1/n2 (plane: unused)
            assert out[i, 6] == oi["n1"] and out[i, 7] == oi["n2"]
    assert hits > n // 2


def test_kat_cube_near_parallel_ray_is_not_culled(rt, O):
    """cube.rs:54-60 treats |direction| < EPSILON on an axis as parallel and ignores the drift
    along it: a ray that starts just inside the x slab and leaves it by 9e-4 before it reaches
    the cube is still a hit in the reference.  The world-space box in front of the cube test must
    not cull it."""
    def b(T, M, X):
        T.add_shape(scenes.CUBE, None, M(a=(1, 0, 0)))
        T.add_light((1, 1, 1), (-10, 10, -10))
        T.set_camera(8, 8, 1.0, X.view_transform((0, 0, -5), (0, 0, 0), (0, 1, 0)))
    w, s = build_both(rt, O, b)
    for o, d in (((1 - 1e-9, -100.0, 0.0), (9e-6, 1.0, 0.0)), ((0.0, 1 - 1e-9, -50.0), (0.0, 9e-6, 1.0)),
                 ((-(1 - 1e-9), 0.3, 40.0), (-9e-6, 0.0, -1.0))):
        r, oi = trace(w, o, d), s.trace_info(o, d)
        assert oi["hit"] and r["hit"], (o, d)
        assert r["t"] == oi["t"] and np.array_equal(r["normalv"], oi["normalv"])
        assert r["shadowed"] == s.is_shadowed(oi["over_point"])


def test_host_wrappers_refuse_bad_calls(rt):
    with pytest.raises(rt.RtError):
        rt.mat_inverse(np.eye(5))
    with pytest.raises(rt.RtError):
        rt.mat_determinant(np.eye(1))
    w = rt.World.example("cover", 16, 16).commit()
    with pytest.raises(rt.RtError):
        w.add_shape(scenes.SPHERE)
    with pytest.raises(rt.RtError):
        w.add_triangle([0, 1, 0, -1, 0, 0, 1, 0, 0])
    assert w.add_light((1, 1, 1), (0, 0, 0)) < 0


# ---- edge cases --------------------------------------------------------------------------------
def test_empty_world_and_no_lights(rt, O):
    w = rt.World()
    w.set_camera(33, 17, 1.0)
    w.commit()
    u8, f64, st = w.render(want_f64=True, want_stats=True)
    assert u8.shape == (17, 33, 3) and not u8.any() and not f64.any()
    assert st.rays_primary == 33 * 17 and st.rays_shadow == 0

    def b(T, M, X):  # objects but no light: every pixel black, no shadow rays
        T.add_shape(scenes.SPHERE, None, M(reflective=0.5))
        T.set_camera(16, 16, 1.0, X.view_transform((0, 0, -5), (0, 0, 0), (0, 1, 0)))
    w, s = build_both(rt, O, b)
    u8, _, _ = w.render()
    assert np.array_equal(u8, s.render()[1])


@pytest.mark.parametrize("size", [(1, 1), (7, 3), (31, 33), (65, 64), (100, 1)])
def test_ragged_image_sizes(rt, O, size):
    w = rt.World.example("cover", *size).commit()
    u8, f64, _ = w.render(want_f64=True)
    of64, ou8, _ = O.Scene.example("cover", *size).render()
    assert np.array_equal(u8, ou8) and np.abs(f64 - of64).max() <= F64_TOL


@pytest.mark.parametrize("depth", [0, 1, 2, 3, 8])
def test_recursion_depths(rt, O, depth):
    w = rt.World.example("cover", 96, 96).commit()
    u8, _, st = w.render(depth=depth, want_stats=True)
    _, ou8, cnt = O.Scene.example("cover", 96, 96).render(depth=depth)
    assert np.array_equal(u8, ou8)
    if depth > 0:
        assert st.rays() == cnt["primary"] + cnt["shadow"] + cnt["reflect"] + cnt["refract"]


def test_repeated_renders_are_deterministic(rt):
    w = rt.World.example("pawn", 320, 240).commit()
    a = w.render(want_f64=True)
    b = w.render(want_f64=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ---- tile shards (the multi-GPU decomposition) on one GPU ------------------------------------
@pytest.mark.parametrize("nshards", [2, 3, 8])
def test_tile_shards_stitch_to_the_full_image(rt, nshards):
    import torch
    name, hs, vs = "pawn", 250, 141  # ragged: partial tiles on both edges
    w = rt.World.example(name, hs, vs).commit()
    full, _, _ = w.render()
    per = max(rt.shard_tiles(hs, vs, s, nshards) for s in range(nshards))
    tiles = torch.zeros((nshards, per * rt.TILE * rt.TILE * 3), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for s in range(nshards):
        w.render_device(d_u8=tiles[s].data_ptr(), stream=stream, tile_first=s, tile_step=nshards)
    img = torch.zeros((vs, hs, 3), dtype=torch.uint8, device="cuda")
    rt.detile_u8(tiles.data_ptr(), nshards, per, hs, vs, img.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(img.cpu().numpy(), full)


def test_pawn_8k_tile_shards(rt, O, pipeline):
    """The 7680x4320 pawn frame as the multi-GPU decomposition cuts it (240 x 135 tiles dealt
    round-robin to 8 and to 2 shards), every shard rendered on this one GPU, stitched with
    rt_detile_u8: byte-equal to the unsharded frame and equal to the oracle on a pixel sample."""
    import torch
    if pipeline != "wavefront":
        pytest.skip("the shards of the 8K frame are rendered with the pipeline the bench uses")
    hs, vs = 7680, 4320
    w = rt.World.example("pawn", hs, vs).commit()
    stream = torch.cuda.current_stream().cuda_stream
    full = torch.zeros((vs, hs, 3), dtype=torch.uint8, device="cuda")
    w.render_device(d_u8=full.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert len(torch.unique(full[::16, ::16])) > 50
    for nshards in (8, 2):
        per = max(rt.shard_tiles(hs, vs, s, nshards) for s in range(nshards))
        assert sum(rt.shard_tiles(hs, vs, s, nshards) for s in range(nshards)) == 240 * 135
        tiles = torch.zeros((nshards, per * rt.TILE * rt.TILE * 3), dtype=torch.uint8, device="cuda")
        for s in range(nshards):
            w.render_device(d_u8=tiles[s].data_ptr(), stream=stream, tile_first=s, tile_step=nshards)
        img = torch.zeros((vs, hs, 3), dtype=torch.uint8, device="cuda")
        rt.detile_u8(tiles.data_ptr(), nshards, per, hs, vs, img.data_ptr(), stream)
        torch.cuda.synchronize()
        assert torch.equal(img, full), nshards
        del tiles, img
    # the library-owned path (rt_scene_commit_multi / rt_render_multi: tiles, RCCL exchange, de-tile)
    wm = rt.World.example("pawn", hs, vs).commit_multi([0])
    img = torch.zeros((vs, hs, 3), dtype=torch.uint8, device="cuda")
    wm.render_multi(d_u8=img.data_ptr(), want_host=False)
    assert torch.equal(img, full)
    del img, wm
    sc = O.Scene.example("pawn", hs, vs).use_bvh()
    rng = np.random.default_rng(12)
    xs, ys = rng.integers(0, hs, 30000).astype(np.int32), rng.integers(0, vs, 30000).astype(np.int32)
    _, ou8, _ = sc.render_pixels(xs, ys)
    assert np.array_equal(full.cpu().numpy()[ys, xs], ou8)


def test_render_multi_on_one_gpu_equals_render(rt, O, pipeline, tmp_path):
    """rt_scene_commit_multi / rt_render_multi with one GPU run every step of the multi-GPU
    frame — compact tile buffer, the RCCL exchange (the shard is sent to itself), the de-tile
    kernel — and must deliver the image rt_render delivers, byte for byte; so must the command
    line with --gpus 1."""
    import os
    import subprocess
    import torch
    pipe = rt.PIPELINE_WAVEFRONT if pipeline == "wavefront" else rt.PIPELINE_MEGAKERNEL
    hs, vs = 250, 141  # ragged: partial tiles on both edges
    ref, _, rst = rt.World.example("pawn", hs, vs).commit().render(pipeline=pipe, want_stats=True)
    w = rt.World.example("pawn", hs, vs).commit_multi([0])
    assert w.L.rt_scene_device_count(w.scene) == 1
    for _ in range(3):  # unseen, planned, planned
        got, st = w.render_multi(pipeline=pipe, want_stats=True)
        assert np.array_equal(got, ref)
        assert st.rays() == rst.rays() and st.rays_shadow == rst.rays_shadow
    dev = torch.zeros((vs, hs, 3), dtype=torch.uint8, device="cuda")
    got, _ = w.render_multi(d_u8=dev.data_ptr(), want_host=False, pipeline=pipe)
    assert got is None and np.array_equal(dev.cpu().numpy(), ref)
    u8, _, _ = w.render(pipeline=pipe)  # the host mirror's render() on a multi scene
    assert np.array_equal(u8, ref)
    with pytest.raises(rt.RtError):
        rt.World.example("cover", 32, 32).commit_multi([0, 0])
    with pytest.raises(rt.RtError):
        rt.World.example("cover", 32, 32).commit_multi([99])
    if pipeline == "wavefront":
        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        out = tmp_path / "multi.png"
        r = subprocess.run([os.path.join(repo, "raytracer_amd", "rt_cli"), "-e", "pawn", "--width", str(hs), "--height",
                            str(vs), "--gpus", "1", "--obj-dir", os.path.join(repo, "assets", "obj"), "--out", str(out)],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert "on 1 GPU(s)" in r.stdout
        assert np.array_equal(O.read_png_rgb8(str(out)), ref)


@pytest.mark.parametrize("split", [2, 3])
def test_frame_split_over_internal_streams(rt, split, monkeypatch):
    """RT_FRAME_SPLIT renders a wavefront frame as interleaved sub-shards on the scene's own
    streams: the full image and a shard's compact tile buffer must come out bit-identical."""
    import torch
    name, hs, vs = "pawn", 1000, 562  # ragged; 32 x 18 tiles
    ref = rt.World.example(name, hs, vs).commit()
    full, _, _ = ref.render(pipeline=rt.PIPELINE_WAVEFRONT)
    monkeypatch.setenv("RT_FRAME_SPLIT", str(split))
    w = rt.World.example(name, hs, vs).commit()
    for _ in range(2):  # the second frame is issued from the remembered queue lengths
        got, _, _ = w.render(pipeline=rt.PIPELINE_WAVEFRONT)
        assert np.array_equal(got, full)
    nshards = 2
    per = max(rt.shard_tiles(hs, vs, s, nshards) for s in range(nshards))
    tiles = torch.zeros((nshards, per * rt.TILE * rt.TILE * 3), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for s in range(nshards):
        w.render_device(d_u8=tiles[s].data_ptr(), stream=stream, tile_first=s, tile_step=nshards,
                        pipeline=rt.PIPELINE_WAVEFRONT)
    img = torch.zeros((vs, hs, 3), dtype=torch.uint8, device="cuda")
    rt.detile_u8(tiles.data_ptr(), nshards, per, hs, vs, img.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(img.cpu().numpy(), full)
