"""Every product path that is not the default one, against the CPU oracle: the tuning knobs the
library reads from the environment when a scene is committed (coherence sort of the ray queues,
per-lane leaf tests, leaf size limits, the builders' cost models), the automatic frame split
that engages after four frames on one stream, and the single-GPU entry points on a scene that
was committed for several GPUs.  All need an MI355X: `pytest -m gpu`.

Bar: the same as tests/test_gpu_parity.py — the quantised frame equal on every pixel, f64
colours within 1e-12 (the specular pow), ray counts equal.
"""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu

F64_TOL = 1e-12

# (environment, bvh_builder): 0 = host SAH tree, 1 = device-built tree
KNOBS = [
    ({"RT_SORT_RAYS": "0"}, 0),
    ({"RT_SORT_RAYS": "0"}, 1),
    ({"RT_SORT_RAYS": "1", "RT_SORT_MIN_LEVEL": "2"}, 0),
    ({"RT_SORT_RAYS": "1"}, 0),
    ({"RT_SORT_RAYS": "1", "RT_SORT_MIN_LEVEL": "1"}, 0),
    ({"RT_SORT_RAYS": "1", "RT_SORT_MIN_LEVEL": "1"}, 1),
    ({"RT_SHARE_LEAVES": "0"}, 0),
    ({"RT_MAX_LEAF": "1"}, 0),
    ({"RT_MAX_LEAF": "15"}, 0),
    ({"RT_MAX_LEAF": "1"}, 1),
    ({"RT_MAX_LEAF": "15"}, 1),
    ({"RT_SAH_BOTTOMUP": "0"}, 0),
    ({"RT_SAH_TRAV": "1"}, 0),
    ({"RT_SAH_SWEEP": "100000000"}, 0),
    ({"RT_LBVH_TRAV": "1"}, 1),
    ({"RT_LBVH_TREELETS": "0"}, 1),
    ({"RT_LBVH_BIG_TREELETS": "0"}, 1),
    ({"RT_FRAME_SPLIT": "2"}, 0),
    ({"RT_SHADOW_PASS": "0"}, 0),
    ({"RT_SHADOW_PASS": "1"}, 0),
    ({"RT_SHADOW_PASS": "1", "RT_SORT_RAYS": "1", "RT_SORT_MIN_LEVEL": "1"}, 0),
]


def knob_id(k):
    return ",".join(f"{a}={b}" for a, b in k[0].items()) + ("/lbvh" if k[1] else "/sah")


@pytest.fixture(scope="module")
def oracle_frames(O, rt):
    """The oracle's frames of the two scenes every knob is checked on (made once)."""
    out = {}
    sc = O.Scene.example("pawn", 320, 180).use_bvh()
    out["pawn"] = sc.render()
    s = O.Scene()
    scenes.random_scene(s, O.material, rt, 11, n_prims=8, n_tris=700, hsize=160, vsize=96)
    out["random"] = s.render()
    return out


def check(w, rt, want, pipeline):
    of64, ou8, cnt = want
    u8, f64, st = w.render(want_f64=True, want_stats=True, pipeline=pipeline)
    assert np.array_equal(u8, ou8)
    assert np.abs(f64 - of64).max() <= F64_TOL
    assert (st.rays_primary, st.rays_shadow, st.rays_reflect, st.rays_refract, st.depth0_skipped) == (
        cnt["primary"], cnt["shadow"], cnt["reflect"], cnt["refract"], cnt["depth0"])
    # the frame again, twice: issued from the remembered queue lengths, without read-backs, its rays
    # placed in key order as they are emitted (the first render's sort passes left the places)
    for _ in range(2):
        u8b, f64b, stb = w.render(pipeline=pipeline, want_f64=True, want_stats=True)
        assert np.array_equal(u8b, ou8) and np.array_equal(f64b, f64)
        assert stb.rays() == st.rays() and stb.rays_shadow == st.rays_shadow


@pytest.mark.parametrize("knob", KNOBS, ids=knob_id)
def test_tuning_knobs_do_not_change_the_frame(rt, O, oracle_frames, knob, monkeypatch):
    env, builder = knob
    for k, v in env.items():
        monkeypatch.setenv(k, v)  # read by rt_scene_commit
    w = rt.World.example("pawn", 320, 180).commit(bvh_builder=builder)
    check(w, rt, oracle_frames["pawn"], rt.PIPELINE_WAVEFRONT)
    w2 = rt.World()
    scenes.random_scene(w2, rt.material, rt, 11, n_prims=8, n_tris=700, hsize=160, vsize=96)
    w2.commit(bvh_builder=builder)
    check(w2, rt, oracle_frames["random"], rt.PIPELINE_WAVEFRONT)


def test_sorted_queues_on_a_large_frame(rt, O, monkeypatch):
    """The coherence sort where its queues are long (pawn 1920x1080, levels 1-4 sorted): the
    frame of the unsorted pipeline, byte for byte, and the oracle's on a pixel sample."""
    ref, _, rst = rt.World.example("pawn", 1920, 1080).commit().render(pipeline=rt.PIPELINE_WAVEFRONT,
                                                                         want_stats=True)
    monkeypatch.setenv("RT_SORT_RAYS", "1")
    monkeypatch.setenv("RT_SORT_MIN_LEVEL", "1")
    w = rt.World.example("pawn", 1920, 1080).commit()
    for _ in range(2):
        u8, _, st = w.render(pipeline=rt.PIPELINE_WAVEFRONT, want_stats=True)
        assert np.array_equal(u8, ref)
        assert st.rays() == rst.rays()
    rng = np.random.default_rng(3)
    xs, ys = rng.integers(0, 1920, 3000).astype(np.int32), rng.integers(0, 1080, 3000).astype(np.int32)
    _, ou8, _ = O.Scene.example("pawn", 1920, 1080).use_bvh().render_pixels(xs, ys)
    assert np.array_equal(u8[ys, xs], ou8)


def test_automatic_frame_split_after_frames_on_one_stream(rt, O, capfd, monkeypatch):
    """A host that renders one image at a time: after four wavefront frames on one stream the
    library renders a frame with much recursion as three interleaved parts on streams of its
    own.  Six frames on one stream; the sixth (split, and issued from remembered queue lengths)
    must be the first one's bytes and the oracle's."""
    import torch
    monkeypatch.setenv("RT_TRACE_SPLIT", "1")
    hs, vs = 640, 360  # 20 x 12 = 240 tiles: enough for three parts of at least 64 tiles
    w = rt.World.example("pawn", hs, vs).commit()
    stream = torch.cuda.current_stream().cuda_stream
    imgs = [torch.zeros((vs, hs, 3), dtype=torch.uint8, device="cuda") for _ in range(6)]
    for k in range(6):
        w.render_device(d_u8=imgs[k].data_ptr(), stream=stream, pipeline=rt.PIPELINE_WAVEFRONT)
        torch.cuda.synchronize()
    err = capfd.readouterr().err
    assert "rt: split 1" in err and "rt: split 3" in err, err  # both kinds of frame were issued
    assert err.strip().splitlines()[-1] == "rt: split 3"
    _, ou8, _ = O.Scene.example("pawn", hs, vs).use_bvh().render()
    for k in range(6):
        assert np.array_equal(imgs[k].cpu().numpy(), ou8), k


def test_single_gpu_entry_points_on_a_multi_committed_scene(rt, O):
    """A scene committed with rt_scene_commit_multi keeps serving rt_render, rt_render_device and
    rt_render_device_noalloc, row-major like any other scene, also when the image's sides are
    no multiples of the 32-pixel tile (a compact tile layout would overrun a 100x100 image)."""
    import torch
    hs, vs = 100, 100
    ref, _, _ = rt.World.example("cover", hs, vs).commit().render()
    _, ou8, _ = O.Scene.example("cover", hs, vs).render()
    assert np.array_equal(ref, ou8)
    w = rt.World.example("cover", hs, vs).commit_multi([0])
    import ctypes as C
    cam = w.camera_abi()
    for pipe in (rt.PIPELINE_WAVEFRONT, rt.PIPELINE_MEGAKERNEL, rt.PIPELINE_AUTO):
        u8 = np.zeros(hs * vs * 3 + 4096, dtype=np.uint8)
        f64 = np.zeros(hs * vs * 3)
        rc = w.L.rt_render(w.scene, C.byref(cam), 5, rt.MODE_F64_PARITY, pipe,
                           f64.ctypes.data_as(C.POINTER(C.c_double)), u8.ctypes.data_as(C.POINTER(C.c_uint8)), None)
        assert rc == 0, w.L.rt_last_error()
        assert np.array_equal(u8[:hs * vs * 3].reshape(vs, hs, 3), ref), pipe
        assert not u8[hs * vs * 3:].any()
    guard = 4096
    buf = torch.full((hs * vs * 3 + guard,), 0xAB, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    w.render_device(d_u8=buf.data_ptr(), stream=stream, pipeline=rt.PIPELINE_WAVEFRONT)
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    assert np.array_equal(got[:hs * vs * 3].reshape(vs, hs, 3), ref)
    assert (got[hs * vs * 3:] == 0xAB).all()  # nothing written past the image
    buf.fill_(0xAB)
    w.render_device_noalloc(d_u8=buf.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    assert np.array_equal(got[:hs * vs * 3].reshape(vs, hs, 3), ref)
    assert (got[hs * vs * 3:] == 0xAB).all()
    got, _ = w.render_multi()  # and the multi-GPU frame itself is still right afterwards
    assert np.array_equal(got, ref)


def test_render_multi_f64_equals_render(rt, O):
    """rt_render_multi_f64 on one GPU (f64 and u8 shards through the RCCL exchange and the
    de-tile kernels): the f64 colours and the quantised frame rt_render delivers, bit for bit,
    on a ragged image; host and device outputs."""
    import torch
    hs, vs = 250, 141
    ref_u8, ref_f64, _ = rt.World.example("pawn", hs, vs).commit().render(want_f64=True)
    w = rt.World.example("pawn", hs, vs).commit_multi([0])
    for _ in range(2):  # unseen, then planned
        f64, u8, st = w.render_multi_f64(want_stats=True)
        assert np.array_equal(f64, ref_f64) and np.array_equal(u8, ref_u8)
    f64, u8, _ = w.render_multi_f64(want_host_u8=False)  # f64 alone
    assert u8 is None and np.array_equal(f64, ref_f64)
    dev = torch.zeros((vs, hs, 3), dtype=torch.float64, device="cuda")
    w.render_multi_f64(d_f64=dev.data_ptr(), want_host_f64=False, want_host_u8=False)
    assert np.array_equal(dev.cpu().numpy(), ref_f64)
    of64, ou8, _ = O.Scene.example("pawn", hs, vs).use_bvh().render()
    assert np.array_equal(ref_u8, ou8) and np.abs(ref_f64 - of64).max() <= F64_TOL
    u8, f64, _ = w.render(want_f64=True)  # the host mirror's render() on a multi scene
    assert np.array_equal(u8, ref_u8) and np.array_equal(f64, ref_f64)


@pytest.mark.parametrize("n_tris", [65, 700, 8191, 8192, 8193, 20000])
def test_device_builder_around_the_large_subtree_limits(rt, O, n_tris):
    """The device builder splits every maximal subtree of at most 8192 triangles again by the
    binned surface-area heuristic, down to the 64-triangle subtrees that are rebuilt exactly:
    meshes just inside, at and beyond both limits against the oracle (its culling mode, which is
    bit-identical to brute force), and the same frame from the host-built tree."""
    w = rt.World()
    scenes.random_scene(w, rt.material, rt, 70 + n_tris, n_prims=3, n_tris=n_tris, hsize=72, vsize=48)
    w.commit(bvh_builder=1)
    s = O.Scene()
    scenes.random_scene(s, O.material, rt, 70 + n_tris, n_prims=3, n_tris=n_tris, hsize=72, vsize=48)
    s.use_bvh()
    u8, f64, _ = w.render(want_f64=True)
    of64, ou8, _ = s.render()
    assert np.array_equal(u8, ou8)
    assert np.abs(f64 - of64).max() <= 1e-11
    h = rt.World()
    scenes.random_scene(h, rt.material, rt, 70 + n_tris, n_prims=3, n_tris=n_tris, hsize=72, vsize=48)
    h.commit(bvh_builder=0)
    hu8, hf64, _ = h.render(want_f64=True)
    assert np.array_equal(hu8, u8) and np.array_equal(hf64, f64)


def test_device_builder_on_thousands_of_coincident_triangles(rt, O):
    """9000 copies of one triangle (no split separates them: every centroid falls into one bin, the
    large-subtree rebuild halves the ranges as they stand) and a geometric row of 3000 shrinking
    ones, built on the device."""
    w = rt.World()
    scenes.deep_tree_scene(w, rt.material, rt, n_same=9000, n_line=3000)
    w.commit(bvh_builder=1)
    s = O.Scene()
    scenes.deep_tree_scene(s, O.material, rt, n_same=9000, n_line=3000)
    s.use_bvh()
    u8, f64, _ = w.render(want_f64=True)
    of64, ou8, _ = s.render()
    assert np.array_equal(u8, ou8)
    assert np.abs(f64 - of64).max() <= 1e-11


def test_host_and_device_trees_render_the_same_1080p_frame(rt):
    """The default tree is the device-built one; the host SAH build must give the same bytes at
    the benchmark's size."""
    a, _, sa = rt.World.example("pawn", 1920, 1080).commit(bvh_builder=0).render(want_stats=True)
    b, _, sb = rt.World.example("pawn", 1920, 1080).commit().render(want_stats=True)
    assert np.array_equal(a, b) and sa.rays() == sb.rays()


def test_multi_entry_points_refuse_bad_calls(rt):
    """rt_render_multi / rt_render_multi_f64: a scene committed for one GPU the ordinary way, no
    output at all, a camera of no size — each refused with a status and a message, none fatal."""
    import ctypes as C
    single = rt.World.example("cover", 64, 48).commit()
    with pytest.raises(rt.RtError, match="rt_scene_commit_multi"):
        single.render_multi()
    w = rt.World.example("cover", 64, 48).commit_multi([0])
    cam = w.camera_abi()
    rc = w.L.rt_render_multi_f64(w.scene, C.byref(cam), 5, rt.MODE_F64_PARITY, rt.PIPELINE_WAVEFRONT, None, None, None,
                                 None, None)
    assert rc != 0 and b"no output buffer" in w.L.rt_last_error()
    rc = w.L.rt_render_multi(w.scene, C.byref(cam), 5, rt.MODE_F64_PARITY, rt.PIPELINE_WAVEFRONT, None, None, None)
    assert rc != 0 and b"no output buffer" in w.L.rt_last_error()
    bad = rt.CameraABI()
    C.memmove(C.byref(bad), C.byref(cam), C.sizeof(cam))
    bad.hsize = 0
    u8 = np.zeros(64 * 48 * 3, dtype=np.uint8)
    rc = w.L.rt_render_multi(w.scene, C.byref(bad), 5, rt.MODE_F64_PARITY, rt.PIPELINE_WAVEFRONT,
                             u8.ctypes.data_as(C.POINTER(C.c_uint8)), None, None)
    assert rc != 0
    got, _ = w.render_multi()  # and the scene still renders
    ref, _, _ = single.render()
    assert np.array_equal(got, ref)


def test_two_rank_bench_rehearsal_gives_the_one_gpu_frame(rt):
    """bench.py as the driver launches it for N = 2 (one process per rank under
    torch.distributed.run: tile shards, gather to rank 0, de-tile), rehearsed on this one GPU
    with gloo instead of RCCL: the frame's CRC must be the one-GPU run's."""
    import json
    import os
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--workload", "pawn1080", "--steps", "2", "--warmup", "1", "--sustain-s", "0", "--no-cpu-baseline"]
    env = dict(os.environ)
    one = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1"] + common,
                         capture_output=True, text=True, timeout=300, env=env, cwd=repo)
    assert one.returncode == 0, one.stderr[-2000:]
    line1 = json.loads(one.stdout.strip().splitlines()[-1])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env["RT_BENCH_REHEARSAL_ONE_GPU"] = "1"
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(repo, "bench.py"), "--gpus", "2"] +
                         # with the sustained loop the driver's default flags run: its length must be the
                         # same on every rank (one exchange per frame)
                         [a if a != "0" else "0.3" for a in common],
                         capture_output=True, text=True, timeout=300, env=env, cwd=repo)
    assert two.returncode == 0, two.stderr[-2000:]
    line2 = json.loads([l for l in two.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line2["n_gpus"] == 2 and line1["n_gpus"] == 1
    assert line2["config"]["frame_crc32"] == line1["config"]["frame_crc32"]
    assert line2["config"]["rays_per_frame"] == line1["config"]["rays_per_frame"]
    assert line2["sustained"]["steps"] >= 2 and line2["sustained"]["seconds"] > 0
