"""Known-answer tests of the reference (its in-file #[test] blocks) restated against
the CPU oracle, tolerance EPSILON = 1e-5 as in the reference (math/utils.rs:1-4)."""
import math

import numpy as np

import scenes

EPS = 1e-5
S2 = math.sqrt(2.0)


def close(a, b, eps=EPS):
    return np.all(np.abs(np.asarray(a, dtype=float) - np.asarray(b, dtype=float)) < eps)


# ---- math/matrix.rs ---------------------------------------------------------------------
def test_determinants(O):  # matrix.rs:343-372
    assert O.mat_determinant([[1, 5], [-3, 2]]) == 17.0
    assert O.mat_determinant([[1, 2, 6], [-5, 8, -4], [2, 6, 4]]) == -196.0
    assert O.mat_determinant([[-2, -8, 3, 5], [-3, 1, 7, 3], [1, 2, -9, 6], [-6, 7, 7, -9]]) == -4071.0


def test_inverse_4x4(O):  # matrix.rs:435-478
    A = [[8, -5, 9, 2], [7, 5, 6, 1], [-6, 0, 9, 6], [-3, 0, -9, -4]]
    inv = [[-0.15385, -0.15385, -0.28205, -0.53846], [-0.07692, 0.12308, 0.02564, 0.03077],
           [0.35897, 0.35897, 0.43590, 0.92308], [-0.69231, -0.69231, -0.76923, -1.92308]]
    assert close(O.mat_inverse(A), inv)
    A = [[9, 3, 0, 9], [-5, -2, -6, -3], [-4, 9, 6, 4], [-7, 6, 6, 2]]
    inv = [[-0.04074, -0.07778, 0.14444, -0.22222], [-0.07778, 0.03333, 0.36667, -0.33333],
           [-0.02901, -0.14630, -0.10926, 0.12963], [0.17778, 0.06667, -0.26667, 0.33333]]
    assert close(O.mat_inverse(A), inv)


def test_inverse_sanity(O):  # matrix.rs:481-500
    A = np.array([[3, -9, 7, 3], [3, -8, 2, -9], [-4, 4, 4, 1], [-6, 5, -1, 1]], dtype=float)
    B = np.array([[8, 2, 2, 2], [3, -1, 7, 0], [7, 0, 5, 4], [6, -2, 0, 5]], dtype=float)
    C = O.mat_mul(A, B)
    assert close(O.mat_mul(C, O.mat_inverse(B)), A)


def test_affine_inverse_keeps_last_row(O):
    """The device code relies on it: the cofactor inverse of an affine matrix has the exact
    last row 0,0,0,1 (so w stays exactly 0 or 1 through every transform)."""
    rng = np.random.default_rng(7)
    for _ in range(200):
        m = O.mat_mul(O.translation(*rng.uniform(-500, 500, 3)),
                      O.mat_mul(O.rotation_y(rng.uniform(-4, 4)),
                                O.mat_mul(O.rotation_x(rng.uniform(-4, 4)),
                                          O.scaling(*rng.uniform(0.01, 50, 3)))))
        inv = O.mat_inverse(m)
        assert np.array_equal(inv[3], [0.0, 0.0, 0.0, 1.0])


def test_transform_constructors(O):  # matrix.rs:503-629
    p = np.array([-3.0, 4.0, 5.0, 1.0])
    assert close(O.translation(5, -3, 2) @ p, [2, 1, 7, 1])
    assert close(O.mat_inverse(O.translation(5, -3, 2)) @ p, [-8, 7, 3, 1])
    assert close(O.scaling(2, 3, 4) @ np.array([-4, 6, 8, 1.0]), [-8, 18, 32, 1])
    q = np.array([0, 1, 0, 1.0])
    assert close(O.rotation_x(math.pi / 4) @ q, [0, S2 / 2, S2 / 2, 1])
    assert close(O.rotation_x(math.pi / 2) @ q, [0, 0, 1, 1])
    assert close(O.rotation_y(math.pi / 2) @ np.array([0, 0, 1, 1.0]), [1, 0, 0, 1])
    assert close(O.rotation_z(math.pi / 2) @ q, [-1, 0, 0, 1])
    assert close(O.shear(0, 0, 0, 0, 0, 1) @ np.array([2, 3, 4, 1.0]), [2, 3, 7, 1])
    T = O.mat_mul(O.translation(10, 5, 7), O.mat_mul(O.scaling(5, 5, 5), O.rotation_x(math.pi / 2)))
    assert close(T @ np.array([1, 0, 1, 1.0]), [15, 0, 7, 1])


# ---- scene/camera.rs ----------------------------------------------------------------------
def test_camera(O):  # camera.rs:220-242
    s = O.Scene()
    s.set_camera(200, 125, math.pi / 2)
    assert abs(s.camera()["pixel_size"] - 0.01) < EPS
    s.set_camera(125, 200, math.pi / 2)
    assert abs(s.camera()["pixel_size"] - 0.01) < EPS
    t = O.mat_mul(O.rotation_y(math.pi / 4), O.translation(0, -2, 5))
    s.set_camera(201, 101, math.pi / 2, t)
    o, d = s.ray_for_pixel(100, 50)
    assert close(o, [0, 2, -5]) and close(d, [S2 / 2, 0, -S2 / 2])


def test_view_transform(O):  # camera.rs:245-295
    assert close(O.view_transform((0, 0, 0), (0, 0, -1), (0, 1, 0)), np.eye(4))
    assert close(O.view_transform((0, 0, 0), (0, 0, 1), (0, 1, 0)), O.scaling(-1, 1, -1))
    assert close(O.view_transform((0, 0, 8), (0, 0, 0), (0, 1, 0)), O.translation(0, 0, -8))
    m = O.view_transform((1, 3, 2), (4, -2, 8), (1, 1, 0))
    assert close(m, [[-0.50709, 0.50709, 0.67612, -2.36643], [0.76772, 0.60609, 0.12122, -2.82843],
                     [-0.35857, 0.59761, -0.71714, 0.0], [0, 0, 0, 1]])


# ---- draw ------------------------------------------------------------------------------------
def test_clamp(O):  # color.rs:148-155, 163-168
    assert O.clamp_u8(0.5) == 127 and O.clamp_u8(1.0) == 255 and O.clamp_u8(0.0) == 0
    assert O.clamp_u8(-0.3) == 0 and O.clamp_u8(7.5) == 255 and O.clamp_u8(float("nan")) == 0
    assert O.clamp_u8(0.999999) == 254


def test_patterns(O):  # patterns.rs:113-119, 179-197, 276-283
    w, b = (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)
    st = O.material(pattern=O.STRIPE, a=w, b=b)
    for x, exp in ((0.0, w), (0.9, w), (1.0, b), (-0.1, b), (-1.0, b), (-1.1, w)):
        assert close(O.pattern_color_at(st, (x, 0, 0)), exp)
    gr = O.material(pattern=O.GRADIENT, a=w, b=b)
    for x, v in ((0.0, 1.0), (0.25, 0.75), (0.5, 0.5), (0.75, 0.25)):
        assert close(O.pattern_color_at(gr, (x, 0, 0)), (v, v, v))
    ck = O.material(pattern=O.CHECKERED, a=w, b=b)
    for p, exp in (((0, 0, 0), w), ((0.99, 0, 0), w), ((1.01, 0, 0), b), ((0, 1.01, 0), b),
                   ((0, 0, 1.01), b), ((-0.5, 0, 0), b), ((-0.5, -0.5, 0), w)):
        assert close(O.pattern_color_at(ck, p), exp)
    rg = O.material(pattern=O.RINGS, a=w, b=b)
    for p, exp in (((0, 0, 0), w), ((1, 0, 0), b), ((0, 0, 1), b), ((0.708, 0, 0.708), b)):
        assert close(O.pattern_color_at(rg, p), exp)


def test_lighting(O):  # light.rs:96-141
    m = O.material()
    n = (0, 0, -1)
    assert close(O.lighting(m, (1, 1, 1), (0, 0, -10), (0, 0, 0), (0, 0, -1), n, False), (1.9,) * 3)
    assert close(O.lighting(m, (1, 1, 1), (0, 0, -10), (0, 0, 0), (0, S2 / 2, -S2 / 2), n, False), (1.0,) * 3)
    assert close(O.lighting(m, (1, 1, 1), (0, 10, -10), (0, 0, 0), (0, 0, -1), n, False), (0.7364,) * 3)
    assert close(O.lighting(m, (1, 1, 1), (0, 0, -10), (0, 0, 0), (0, 0, -1), n, True), (0.1,) * 3)


# ---- shapes -------------------------------------------------------------------------------------
def one_shape(O, kind, transform=None, **kw):
    s = O.Scene()
    h = s.add_shape(kind, transform, None, **kw)
    return s, h


def norm(v):
    v = np.asarray(v, dtype=float)
    return v / math.sqrt((v * v).sum())


def test_sphere(O):  # sphere.rs:127-209
    s, h = one_shape(O, O.SPHERE)
    ts, _ = s.intersect_world((0, 0, -5), (0, 0, 1))
    assert close(ts, [4, 6])
    ts, _ = s.intersect_world((0, 1, -5), (0, 0, 1))
    assert close(ts, [5, 5])
    assert len(s.intersect_world((0, 2, -5), (0, 0, 1))[0]) == 0
    assert close(s.intersect_world((0, 0, 0), (0, 0, 1))[0], [-1, 1])
    assert close(s.intersect_world((0, 0, 5), (0, 0, 1))[0], [-6, -4])
    s, h = one_shape(O, O.SPHERE, O.scaling(2, 2, 2))
    assert close(s.intersect_world((0, 0, -5), (0, 0, 1))[0], [3, 7])
    s, h = one_shape(O, O.SPHERE, O.translation(5, 0, 0))
    assert len(s.intersect_world((0, 0, -5), (0, 0, 1))[0]) == 0
    # intersect.rs:253-280 normals
    s, h = one_shape(O, O.SPHERE, O.translation(0, 1, 0))
    assert close(s.normal_at(h, (0, 1.70711, -S2 / 2)), (0, S2 / 2, -S2 / 2))
    s, h = one_shape(O, O.SPHERE, O.mat_mul(O.scaling(1, 0.5, 1), O.rotation_z(math.pi / 5)))
    assert close(s.normal_at(h, (0, S2 / 2, -S2 / 2)), (0, 0.97014, -0.24254))


def test_plane(O):  # plane.rs:105-154
    s, h = one_shape(O, O.PLANE)
    assert len(s.intersect_world((0, 10, 0), (0, 0, 1))[0]) == 0
    assert len(s.intersect_world((0, 0, 0), (0, 0, 1))[0]) == 0
    assert close(s.intersect_world((0, 1, 0), (0, -1, 0))[0], [1])
    assert close(s.intersect_world((0, -1, 0), (0, 1, 0))[0], [1])
    for p in ((0, 0, 0), (10, 0, -10), (-5, 0, 150)):
        assert close(s.normal_at(h, p), (0, 1, 0))


def test_cube(O):  # cube.rs:143-230
    s, h = one_shape(O, O.CUBE)
    rays = [((5, .5, 0), (-1, 0, 0)), ((-5, .5, 0), (1, 0, 0)), ((.5, 5, 0), (0, -1, 0)),
            ((.5, -5, 0), (0, 1, 0)), ((.5, 0, 5), (0, 0, -1)), ((.5, 0, -5), (0, 0, 1))]
    for o, d in rays:
        assert close(s.intersect_world(o, d)[0], [4, 6])
    assert close(s.intersect_world((0, .5, 0), (0, 0, 1))[0], [-1, 1])
    miss = [((-2, 0, 0), (.2673, .5345, .8018)), ((0, -2, 0), (.8018, .2673, .5345)),
            ((0, 0, -2), (.5345, .8018, .2673)), ((2, 0, 2), (0, 0, -1)), ((0, 2, 2), (0, -1, 0)),
            ((2, 2, 0), (-1, 0, 0))]
    for o, d in miss:
        assert len(s.intersect_world(o, d)[0]) == 0
    pts = [((1, .5, -.8), (1, 0, 0)), ((-1, -.2, .9), (-1, 0, 0)), ((-.4, 1, -.1), (0, 1, 0)),
           ((.3, -1, -.7), (0, -1, 0)), ((-.6, .3, 1), (0, 0, 1)), ((.4, .4, -1), (0, 0, -1)),
           ((1, 1, 1), (1, 0, 0)), ((-1, -1, -1), (-1, 0, 0))]
    for p, n in pts:
        assert close(s.normal_at(h, p), n)


def test_cylinder(O):  # cylinder.rs:183-298
    s, h = one_shape(O, O.CYLINDER)
    for o, d in (((1, 0, 0), (0, 1, 0)), ((0, 0, 0), (0, 1, 0)), ((0, 0, -5), (1, 1, 1))):
        assert len(s.intersect_world(o, norm(d))[0]) == 0
    for o, d, ts in (((1, 0, -5), (0, 0, 1), (5, 5)), ((0, 0, -5), (0, 0, 1), (4, 6)),
                     ((.5, 0, -5), (.1, 1, 1), (6.80798, 7.08872))):
        assert close(s.intersect_world(o, norm(d))[0], ts)
    s, h = one_shape(O, O.CYLINDER, minimum=1.0, maximum=2.0)
    for o, d in (((0, 1.5, 0), (.1, 1, 0)), ((0, 3, -5), (0, 0, 1)), ((0, 0, -5), (0, 0, 1)),
                 ((0, 2, -5), (0, 0, 1)), ((0, 1, -5), (0, 0, 1))):
        assert len(s.intersect_world(o, norm(d))[0]) == 0
    assert len(s.intersect_world((0, 1.5, -2), (0, 0, 1))[0]) == 2
    s, h = one_shape(O, O.CYLINDER, minimum=1.0, maximum=2.0, closed=True)
    for o, d in (((0, 3, 0), (0, -1, 0)), ((0, 3, -2), (0, -1, 2)), ((0, 4, -2), (0, -1, 1)),
                 ((0, 0, -2), (0, 1, 2)), ((0, -1, -2), (0, 1, 1))):
        assert len(s.intersect_world(o, norm(d))[0]) == 2


def test_cone(O):  # cone.rs:180-265
    s, h = one_shape(O, O.CONE)
    for o, d, ts in (((0, 0, -5), (0, 0, 1), (5, 5)), ((0, 0, -5), (1, 1, 1), (8.66025, 8.66025)),
                     ((1, 1, -5), (-.5, -1, 1), (4.55006, 49.449944))):
        assert close(s.intersect_world(o, norm(d))[0], ts)
    ts = s.intersect_world((0, 0, -1), norm((0, 1, 1)))[0]
    assert len(ts) == 1 and abs(ts[0] - 0.35355) < EPS
    s, h = one_shape(O, O.CONE, minimum=-0.5, maximum=0.5, closed=True)
    for o, d, n in (((0, 0, -5), (0, 1, 0), 0), ((0, 0, -.25), (0, 1, 1), 2), ((0, 0, -.25), (0, 1, 0), 4)):
        assert len(s.intersect_world(o, norm(d))[0]) == n


def smooth_test_triangle(O):
    s = O.Scene()
    g = s.add_group()
    s.add_triangle([0, 1, 0, -1, 0, 0, 1, 0, 0], [0, 1, 0, -1, 0, 0, 1, 0, 0], parent=g)
    return s


def test_triangles(O):  # triangle.rs:142-223, smooth_triangle.rs:175-200
    s = O.Scene()
    g = s.add_group()
    s.add_triangle([0, 1, 0, -1, 0, 0, 1, 0, 0], None, parent=g)
    assert len(s.intersect_world((0, -1, -2), (0, 1, 0))[0]) == 0   # parallel
    assert len(s.intersect_world((1, 1, -2), (0, 0, 1))[0]) == 0    # p1-p3 edge
    assert len(s.intersect_world((-1, 1, -2), (0, 0, 1))[0]) == 0   # p1-p2 edge
    assert len(s.intersect_world((0, -1, -2), (0, 0, 1))[0]) == 0   # p2-p3 edge
    assert close(s.intersect_world((0, .5, -2), (0, 0, 1))[0], [2])
    info = s.trace_info((0, .5, -2), (0, 0, 1))
    assert close(info["normalv"], (0, 0, -1))
    s = smooth_test_triangle(O)
    info = s.trace_info((-0.2, 0.3, -2.0), (0, 0, 1))
    assert abs(info["u"] - 0.45) < EPS and abs(info["v"] - 0.25) < EPS
    assert close(info["normalv"], (-0.5547, 0.83205, 0.0))


def test_obj_parser(O):  # obj_parser.rs:107-132
    s = O.Scene()
    g = s.add_obj("\n        v -1 1 0\n        v -1 0 0\n        v 1 0 0\n        v 1 1 0\n        f 1 2 3\n        f 1 3 4")
    assert s.group_child_count(g) == 2
    g = s.add_obj("\n        v -1 1 0\n        v -1 0 0\n        v 1 0 0\n        v 1 1 0\n        v 0 2 0\n        f 1 2 3 4 5")
    assert s.group_child_count(g) == 3


def test_example_triangle_counts(O):
    s = O.Scene.example("pawn")
    assert s.L.orc_leaf_count(s.h) == 31776 + 1
    s = O.Scene.example("tea set")
    assert s.L.orc_leaf_count(s.h) == 6320 + 2 * 8378 + 1


# ---- groups ---------------------------------------------------------------------------------------
def test_groups(O):  # group.rs:147-219
    s = O.Scene()
    s.add_group()
    assert len(s.intersect_world((0, 0, 0), (0, 0, 1))[0]) == 0
    s = O.Scene()
    g = s.add_group()
    s1 = s.add_shape(O.SPHERE, None, parent=g)
    s2 = s.add_shape(O.SPHERE, O.translation(0, 0, -3), parent=g)
    s.add_shape(O.SPHERE, O.translation(5, 0, 0), parent=g)
    ts, ids = s.intersect_world((0, 0, -5), (0, 0, 1))
    assert list(ids) == [s.shape_id(s2), s.shape_id(s2), s.shape_id(s1), s.shape_id(s1)]
    s = O.Scene()
    g = s.add_group(O.scaling(2, 2, 2))
    s.add_shape(O.SPHERE, O.translation(5, 0, 0), parent=g)
    assert len(s.intersect_world((10, 0, -10), (0, 0, 1))[0]) == 2
    s = O.Scene()
    g1, g2, sp = scenes.nested_groups(s, O.material, O)
    n = s.normal_at(sp, (1.7321, 1.1547, -5.5774))
    assert close(n, (0.28570368184140726, 0.42854315178114105, -0.8571605294481017))


# ---- world / shading --------------------------------------------------------------------------------
def populated(O):
    s = O.Scene()
    scenes.populated_world(s, O.material, O)
    return s


def test_world_intersections_and_shading(O):  # world.rs:206-269
    s = populated(O)
    assert close(s.intersect_world((0, 0, -5), (0, 0, 1))[0], [4, 4.5, 5.5, 6])
    assert close(s.color_at((0, 0, -5), (0, 0, 1)), (0.38066, 0.47583, 0.2855))
    assert close(s.color_at((0, 0, -5), (0, 1, 0)), (0, 0, 0))
    s.set_camera(11, 11, math.pi / 2, O.view_transform((0, 0, -5), (0, 0, 0), (0, 1, 0)))
    f64, u8, _ = s.render(threads=1)
    assert close(f64[5, 5], (0.38066, 0.47583, 0.2855))
    s = O.Scene()
    s.add_light((1, 1, 1), (-10, 10, -10))
    s.add_shape(O.SPHERE, None, O.material(ambient=1.0, a=(0.1, 0.2, 0.3)))
    s.add_shape(O.SPHERE, None, O.material(ambient=1.0))
    # world.rs:255-269: both unit spheres coincide, so every t ties; the first object in the
    # list must win (stable sort + strict '<' in hit())
    assert close(s.color_at((0, 0, 0.75), (0, 0, -1)), (0.1, 0.2, 0.3))


def test_shadows(O):  # world.rs:272-297
    s = populated(O)
    assert not s.is_shadowed((0, 10, 0))
    assert s.is_shadowed((10, -10, 10))
    assert not s.is_shadowed((-20, 20, -20))
    assert not s.is_shadowed((-2, 2, -2))


def test_reflection(O):  # world.rs:317-354
    s = populated(O)
    s.add_shape(O.PLANE, O.translation(0, -1, 0), O.material(reflective=0.5))
    c = s.color_at((0, 0, -3), (0, -S2 / 2, S2 / 2))
    assert close(c, (0.8767572837020907, 0.924340334075874, 0.8291742333283075))


def test_prepare_computations(O):  # intersect.rs:325-375, 417-424
    s = O.Scene()
    s.add_shape(O.SPHERE)
    i = s.trace_info((0, 0, -5), (0, 0, 1))
    assert close(i["point"], (0, 0, -1)) and close(i["eyev"], (0, 0, -1)) and close(i["normalv"], (0, 0, -1))
    assert not i["inside"]
    i = s.trace_info((0, 0, 0), (0, 0, 1))
    assert close(i["point"], (0, 0, 1)) and close(i["normalv"], (0, 0, -1)) and i["inside"]
    s = O.Scene()
    s.add_shape(O.SPHERE, O.translation(0, 0, 1), O.material(transparency=1.0, refractive_index=1.5))
    i = s.trace_info((0, 0, -5), (0, 0, 1))
    assert i["over_point"][2] < -EPS / 2 and i["point"][2] > i["over_point"][2]
    assert i["under_point"][2] > EPS / 2 and i["point"][2] < i["under_point"][2]
    s = O.Scene()
    s.add_shape(O.PLANE)
    i = s.trace_info((0, 1, -1), (0, -S2 / 2, S2 / 2))
    assert close(i["reflectv"], (0, S2 / 2, S2 / 2))


def test_n1_n2_nested_glass(O):  # intersect.rs:378-406
    s = O.Scene()
    scenes.glass_spheres(s, O.material, O)
    expected = [(1.0, 1.5), (1.5, 2.0), (2.0, 2.5), (2.5, 2.5), (2.5, 1.5), (1.5, 1.0)]
    # one ray per entry of the sorted list: start just in front of each intersection
    for z0, (n1, n2) in zip((-4.0, -1.5, -1.0, 0.5, 1.0, 1.5), expected):
        i = s.trace_info((0, 0, z0), (0, 0, 1))
        assert abs(i["n1"] - n1) < EPS and abs(i["n2"] - n2) < EPS, (z0, i["n1"], i["n2"])


def test_refraction(O):  # world.rs:367-441
    s = O.Scene()
    s.add_light((1, 1, 1), (-10, 10, -10))
    s.add_shape(O.SPHERE, None, O.material(transparency=1.0, refractive_index=1.5))
    s.add_shape(O.SPHERE, O.scaling(.5, .5, .5), O.material(a=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2))
    i = s.trace_info((0, 0, S2 / 2), (0, 1, 0))   # from inside: total internal reflection
    assert i["inside"] and i["n1"] == 1.5 and i["n2"] == 1.0

    def floor_world(reflective):
        s = populated(O)
        s.add_shape(O.PLANE, O.translation(0, -1, 0),
                    O.material(transparency=0.5, refractive_index=1.5, reflective=reflective))
        s.add_shape(O.SPHERE, O.translation(0, -3.5, -0.5), O.material(a=(1, 0, 0), ambient=0.5))
        return s
    assert close(floor_world(0.0).color_at((0, 0, -3), (0, -S2 / 2, S2 / 2)), (0.93642, 0.68642, 0.68642))
    assert close(floor_world(0.5).color_at((0, 0, -3), (0, -S2 / 2, S2 / 2)), (0.93391, 0.69643, 0.69243))


def test_schlick(O):  # world.rs:410-416
    s = O.Scene()
    s.add_shape(O.SPHERE, None, O.material(transparency=1.0, refractive_index=1.5, a=(0, 0, 0)))
    i = s.trace_info((0, 0.99, -2.0), (0, 0, 1))
    assert abs(i["schlick"] - 0.4888143830387389) < EPS


def test_bvh_mode_is_bit_identical(O):
    """The oracle's culling mode must not change a single bit (sampled rays, pawn + tea set)."""
    rng = np.random.default_rng(3)
    for name in ("pawn", "tea set"):
        a = O.Scene.example(name, 160, 120)
        b = O.Scene.example(name, 160, 120).use_bvh()
        xs, ys = rng.integers(0, 160, 300), rng.integers(0, 120, 300)
        fa, ua, ca = a.render_pixels(xs, ys)
        fb, ub, cb = b.render_pixels(xs, ys)
        assert np.array_equal(fa, fb)
        for k in ("primary", "shadow", "reflect", "refract", "depth0"):
            assert ca[k] == cb[k]
        assert cb["tri_tests"] < ca["tri_tests"] / 20
