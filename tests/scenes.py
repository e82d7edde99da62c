"""Scene definitions shared by the oracle tests and the GPU parity tests.

Every builder takes a target `T` that is either `oracle.oracle.Scene` (the CPU
checker) or `raytracer_amd.World` (the product's host mirror): both expose the
same construction methods (add_light / add_shape / add_group / add_triangle /
add_obj / set_camera), so one definition feeds both sides with identical
inputs.  `M` is the matching `material(...)` factory.
"""
import math

import numpy as np

SPHERE, PLANE, CUBE, CYLINDER, CONE = range(5)
SOLID, STRIPE, GRADIENT, RINGS, CHECKERED = range(5)


def mats(module):
    """(material factory, matrix helpers) of the oracle module or of raytracer_amd."""
    return module.material


def populated_world(T, M, X):
    """populated_world() of the reference's tests (scene/world.rs:180-204)."""
    T.add_light((1.0, 1.0, 1.0), (-10.0, 10.0, -10.0))
    m = M(a=(0.8, 1.0, 0.6), diffuse=0.7, specular=0.2)
    s1 = T.add_shape(SPHERE, None, m)
    s2 = T.add_shape(SPHERE, X.scaling(0.5, 0.5, 0.5), m)
    return s1, s2


def glass_spheres(T, M, X):
    """Three nested glass spheres of intersect.rs:378-406."""
    def glass(ri):
        return M(transparency=1.0, refractive_index=ri, a=(0.0, 0.0, 0.0))
    T.add_shape(SPHERE, X.scaling(2.0, 2.0, 2.0), glass(1.5))
    T.add_shape(SPHERE, X.translation(0.0, 0.0, -0.25), glass(2.0))
    T.add_shape(SPHERE, X.translation(0.0, 0.0, 0.25), glass(2.5))
    T.add_light((1.0, 1.0, 1.0), (-10.0, 10.0, -10.0))


def nested_groups(T, M, X):
    """g1(rotation_y(pi/2)) > g2(scaling(1,2,3)) > sphere(translation(5,0,0)), group.rs:186-219."""
    g1 = T.add_group(X.rotation_y(math.pi / 2.0))
    g2 = T.add_group(X.scaling(1.0, 2.0, 3.0), parent=g1)
    s = T.add_shape(SPHERE, X.translation(5.0, 0.0, 0.0), parent=g2)
    T.add_light((1.0, 1.0, 1.0), (-10.0, 10.0, -10.0))
    return g1, g2, s


def random_scene(T, M, X, seed, n_prims=10, n_tris=60, hsize=96, vsize=64):
    """Random primitives of every kind (random affine transforms, patterns, reflective /
    transparent materials), a transformed group of random triangles (flat and smooth) with
    a nested sub-group, two lights and a camera."""
    rng = np.random.default_rng(seed)

    def rand_xf(scale=1.0, spread=3.0):
        t = X.translation(*(rng.uniform(-spread, spread, 3)))
        r = X.mat_mul(X.rotation_x(rng.uniform(-3, 3)), X.rotation_y(rng.uniform(-3, 3)))
        s = X.scaling(*(rng.uniform(0.4, 1.6, 3) * scale))
        return X.mat_mul(t, X.mat_mul(r, s))

    def rand_mat():
        kind = int(rng.integers(0, 5))
        kw = dict(pattern=kind, a=tuple(rng.uniform(0, 1, 3)), b=tuple(rng.uniform(0, 1, 3)),
                  ambient=rng.uniform(0.05, 0.4), diffuse=rng.uniform(0.2, 0.9),
                  specular=rng.uniform(0, 1), shininess=rng.uniform(5, 300))
        if kind != SOLID and rng.random() < 0.7:
            kw["pattern_transform"] = rand_xf(0.5, 1.0)
        u = rng.random()
        if u < 0.3:
            kw["reflective"] = rng.uniform(0.1, 0.9)
        elif u < 0.55:
            kw["transparency"] = rng.uniform(0.3, 1.0)
            kw["refractive_index"] = rng.uniform(1.0, 2.5)
            kw["reflective"] = rng.uniform(0.0, 0.9) if rng.random() < 0.7 else 0.0
        return M(**kw)

    T.add_light((1.0, 1.0, 1.0), (-8.0, 9.0, -10.0))
    T.add_light((0.3, 0.25, 0.2), (7.0, 3.0, -6.0))
    T.add_shape(PLANE, X.translation(0.0, -2.5, 0.0), rand_mat())
    for i in range(n_prims):
        kind = int(rng.integers(0, 5))
        kw = {}
        if kind in (CYLINDER, CONE):
            lo = rng.uniform(-1.5, 0.0)
            kw = dict(minimum=lo, maximum=lo + rng.uniform(0.5, 2.0), closed=bool(rng.random() < 0.7))
        T.add_shape(kind, rand_xf(), rand_mat(), **kw)
    # a glass shell around part of the scene, so rays start / travel inside containers
    T.add_shape(SPHERE, X.mat_mul(X.translation(0.5, 0.0, 0.0), X.scaling(2.5, 2.5, 2.5)),
                M(transparency=0.9, refractive_index=1.3, reflective=0.2, diffuse=0.1, ambient=0.05))
    g = T.add_group(rand_xf(1.0, 1.5), rand_mat())
    for i in range(n_tris):
        c = rng.uniform(-2, 2, 3)
        p = np.concatenate([c + rng.uniform(-0.8, 0.8, 3) for _ in range(3)])
        if rng.random() < 0.5:
            n = rng.normal(size=9)
            T.add_triangle(p, n, parent=g)
        else:
            T.add_triangle(p, None, parent=g)
    sub = T.add_group(rand_xf(0.7, 1.0), parent=g)
    T.add_shape(CUBE, rand_xf(0.5, 1.0), parent=sub)
    for i in range(n_tris // 3):
        c = rng.uniform(-1.5, 1.5, 3)
        p = np.concatenate([c + rng.uniform(-0.6, 0.6, 3) for _ in range(3)])
        T.add_triangle(p, None, parent=sub)
    T.add_shape(SPHERE, rand_xf(0.6, 1.0), parent=g)
    T.set_camera(hsize, vsize, math.pi / 3.0,
                 X.view_transform((0.0, 1.5, -9.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)))


def tie_scene(T, M, X):
    """Coincident geometry: equal-t intersections whose winner is decided by leaf order
    (stable sorts of group.rs:69 / world.rs:29 and the strict '<' of intersect.rs:133)."""
    T.add_light((1.0, 1.0, 1.0), (-5.0, 8.0, -10.0))
    red = M(a=(1.0, 0.0, 0.0), ambient=0.5)
    green = M(a=(0.0, 1.0, 0.0), ambient=0.5)
    blue = M(a=(0.0, 0.0, 1.0), ambient=0.5, transparency=0.6, refractive_index=1.4)
    # two identical cubes: the first added must win every pixel
    T.add_shape(CUBE, X.translation(-2.0, 0.0, 0.0), red)
    T.add_shape(CUBE, X.translation(-2.0, 0.0, 0.0), green)
    # coincident triangles in two groups, then inside one group (24 copies -> BVH leaves of >1)
    quad = [(-1, -1, 0), (1, -1, 0), (1, 1, 0), (-1, 1, 0)]
    g1 = T.add_group(X.translation(2.0, 0.0, 0.0), blue)
    g2 = T.add_group(X.translation(2.0, 0.0, 0.0), green)
    for g in (g1, g2):
        for k in range(12):
            T.add_triangle(np.array(quad[0] + quad[1] + quad[2], dtype=float), None, parent=g)
            T.add_triangle(np.array(quad[0] + quad[2] + quad[3], dtype=float), None, parent=g)
    T.add_shape(PLANE, X.translation(0.0, -1.0, 0.0), M(pattern=CHECKERED, a=(0, 0, 0), b=(1, 1, 1)))
    T.set_camera(80, 48, math.pi / 3.0,
                 X.view_transform((0.0, 1.0, -7.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)))


def deep_tree_scene(T, M, X, n_same=600, n_line=400):
    """Meshes that drive a BVH builder into its worst shapes: many copies of one triangle
    (no split separates them: the SAH builder falls back to median splits, Morton codes are
    all equal) and triangles whose sizes and gaps shrink geometrically along a line (every
    split peels one triangle off: a tree as deep as the builder allows)."""
    T.add_light((1.0, 1.0, 1.0), (-6.0, 8.0, -10.0))
    T.add_shape(PLANE, X.translation(0.0, -1.5, 0.0), M(pattern=CHECKERED, a=(0.9, 0.9, 0.9), b=(0.2, 0.2, 0.2)))
    g = T.add_group(X.translation(-1.5, 0.0, 0.0), M(a=(0.8, 0.3, 0.2), transparency=0.5, refractive_index=1.3))
    tri = np.array([-1.0, -1.0, 0.0, 1.0, -1.0, 0.0, 0.0, 1.0, 0.0])
    for i in range(n_same):
        T.add_triangle(tri, None, parent=g)
    h = T.add_group(X.translation(0.5, -1.0, 0.0), M(a=(0.2, 0.5, 0.9), reflective=0.3))
    x = 0.0
    size = 1.0
    for i in range(n_line):
        p = np.array([x, 0.0, 0.0, x + size, 0.0, 0.0, x, 2.0 * size + 0.05, 0.3])
        T.add_triangle(p, None, parent=h)
        x += 1.2 * size
        size *= 0.93
    T.set_camera(120, 80, math.pi / 3.0,
                 X.view_transform((0.5, 1.0, -7.0), (0.5, 0.0, 0.0), (0.0, 1.0, 0.0)))
