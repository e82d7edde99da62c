// rt_device.h — device-side scene layout and the per-ray device functions.
//
// Everything here is f64 and evaluates each expression in the association the
// reference uses (citations are file:line under the reference's src/), so that
// every comparison the reference makes (EPSILON cuts, edge tests, t > 0, shadow
// distance) sees the same bits.  Build with -ffp-contract=off.
//
// What differs from the reference is the search: a ray does not visit every
// triangle (shapes/group.rs:64-71) but walks a per-mesh box hierarchy that only
// discards triangles whose padded box the ray misses, and the sorted
// intersection list (scene/world.rs:22-32) is never materialised — the three
// consumers of that list are answered by three queries:
//   hit()                  -> closest positive t, ties by leaf order  (trace_closest)
//   is_shadowed()          -> any 0 < t < distance                    (trace_any)
//   refraction containers  -> last "open" shape before the hit        (containers_n1_n2)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_layout.h"
#include "rt_pow.h"

namespace rtd {

constexpr double EPS = 0.00001;  // math/utils.rs:1
#define RT_INF (__builtin_huge_val())

// ---------------------------------------------------------------------------------------
struct V3 {
  double x, y, z;
};
struct Ray {
  V3 o, d;
};
struct C3 {
  double r, g, b;
};

__device__ __forceinline__ V3 v3(double x, double y, double z) { return {x, y, z}; }
__device__ __forceinline__ V3 vadd(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 vmul(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 vneg(V3 a) { return {a.x * -1.0, a.y * -1.0, a.z * -1.0}; }
// tuples.rs:153-156: (x*ox + y*oy) + z*oz
__device__ __forceinline__ double vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// tuples.rs:158-165
__device__ __forceinline__ V3 vcross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// tuples.rs:141-151: sqrt of the sum, then three divisions
__device__ __forceinline__ double vmag(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 vnormalize(V3 a) {
  double m = vmag(a);
  return {a.x / m, a.y / m, a.z / m};
}
// tuples.rs:170-174: v - ((n*2.0) * dot(v,n))
__device__ __forceinline__ V3 vreflect(V3 v, V3 n) {
  double d = vdot(v, n);
  V3 n2 = vmul(n, 2.0);
  return vsub(v, vmul(n2, d));
}
__device__ __forceinline__ bool feq(double a, double b) { return fabs(a - b) < EPS; }

// matrix.rs:55-81 for an affine matrix: ((m0*x + m1*y) + m2*z) + m3*w with w = 1 (point)
__device__ __forceinline__ V3 xf_point(const double* m, V3 p) {
  return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
          m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
}
// ... and with w = 0 (vector): the last product is a zero and leaves the sum unchanged
__device__ __forceinline__ V3 xf_vector(const double* m, V3 v) {
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
          m[8] * v.x + m[9] * v.y + m[10] * v.z};
}
// inverse_transform_transpose * normal (intersect.rs:81): rows of the transpose are the
// columns of inv; the w row/column only feeds the w that the reference zeroes afterwards.
__device__ __forceinline__ V3 xf_normal(const double* m, V3 n) {
  return {m[0] * n.x + m[4] * n.y + m[8] * n.z, m[1] * n.x + m[5] * n.y + m[9] * n.z,
          m[2] * n.x + m[6] * n.y + m[10] * n.z};
}
__device__ __forceinline__ Ray xf_ray(const double* m, const Ray& r) {  // ray.rs:32-37
  return {xf_point(m, r.o), xf_vector(m, r.d)};
}
__device__ __forceinline__ V3 ray_at(const Ray& r, double t) {  // ray.rs:16-18
  return vadd(r.o, vmul(r.d, t));
}

// Ray into the object space of group g: the reference transforms at every level on the way
// down (intersect.rs:57-61 at the group, again at each nested group).
__device__ __forceinline__ Ray ray_to_group(const DScene& S, int g, Ray r) {
  const DGroup& G = S.groups[g];
  for (int k = 0; k < G.nchain; k++) r = xf_ray(S.groups[G.chain[k]].inv, r);
  return r;
}
__device__ __forceinline__ V3 point_to_group(const DScene& S, int g, V3 p) {  // intersect.rs:67-77
  const DGroup& G = S.groups[g];
  for (int k = 0; k < G.nchain; k++) p = xf_point(S.groups[G.chain[k]].inv, p);
  return p;
}
// intersect.rs:79-92 from the parent upwards: transform, drop w, normalise, per level
__device__ __forceinline__ V3 normal_up_groups(const DScene& S, int g, V3 n) {
  while (g >= 0) {
    const DGroup& G = S.groups[g];
    n = vnormalize(xf_normal(G.inv, n));
    g = G.parent;
  }
  return n;
}

#ifdef RT_TRI_PLANE_STATS
// debug build: per query kind (closest, any): triangle tests, plane met behind the origin, plane met
// beyond the interval, tests that hit
__device__ unsigned long long g_tri_plane[8];
#endif
// ---------------------------------------------------------------------------------------
// ttriangle.rs:9-34
__device__ __forceinline__ bool moller_trumbore(const DTri& T, const Ray& r, double& t, double& u,
                                                double& v) {
  V3 e1 = v3(T.e1[0], T.e1[1], T.e1[2]);
  V3 e2 = v3(T.e2[0], T.e2[1], T.e2[2]);
  V3 dce2 = vcross(r.d, e2);
  double det = vdot(e1, dce2);
  if (fabs(det) < EPS) return false;
  double f = 1.0 / det;
  V3 p1o = vsub(r.o, v3(T.p1[0], T.p1[1], T.p1[2]));
  u = f * vdot(p1o, dce2);
  if (!(u >= 0.0 && u <= 1.0)) return false;
  V3 oce1 = vcross(p1o, e1);
  v = f * vdot(r.d, oce1);
  if (v < 0.0 || (u + v) > 1.0) return false;
  t = f * vdot(e2, oce1);
  return true;
}

// The same test in f32 (fast arithmetic mode) with its own error bounds, so that it never
// decides what it cannot know.  Inputs are the f64 operands rounded to f32 (relative error
// d = 2^-24 each); the ray origin is re-based into the mesh's box, so every coordinate is of the
// mesh's magnitude: |o_i| + |p1_i| <= Pm.  The offset p = o - p1 is formed from two-float values
// (f32 + residual) of both, so it carries 2d |p_i| + 2^-31 Pm per component, not the 2d Pm of
// plain f32 operands: a ray that starts 1e-5 off the surface it left (intersect.rs:184-185)
// still sees on which side of that surface it is.  Below, Pm stands for that reduced term,
// pm + Pm/64.  With the L1 norms dm = |dir|_1, e1m, e2m, pm = |p|_1, first-order bounds:
//   dir x e2:   4d dm e2m per component         det = e1 . (dir x e2):  8d e1m dm e2m
//   p . (dir x e2)   (u):  d dm e2m (4 Pm + 10 pm)
//   p x e1:     d e1m (2 Pm + 4 pm) per component
//   dir . (p x e1)   (v):  d dm e1m (2 Pm + 7 pm)
//   e2 . (p x e1)    (t):  d e1m e2m (2 Pm + 8 pm)
// and a quotient N/det is off by at most (err N + |N/det| err det) / |det|.  Every bound is
// taken twice over.  The test answers
//   FT_REJECT  the f64 test (ttriangle.rs:9-34) misses for certain,
//   FT_SOLID   it hits for certain, at t within `et` of the f32 value,
//   FT_UNSURE  anything else (an edge, the EPSILON cut on the determinant, a grazing ray):
//              the caller evaluates the triangle in f64.
struct RayF {
  float ox, oy, oz, dx, dy, dz;
  float oxl, oyl, ozl;  // what f32 lost of the origin
};
enum { FT_REJECT = 0, FT_SOLID = 1, FT_UNSURE = 2 };
__device__ __forceinline__ int moller_trumbore_f32(const DTriF& T, const RayF& r, float Pm, float dm, float& t,
                                                   float& et) {
  constexpr float D2 = 2.02f * 5.9604644775390625e-08f;  // twice 2^-24, and the quotient's own rounding
  t = 0.0f;
  et = __builtin_huge_valf();
  const float cx = r.dy * T.e2[2] - r.dz * T.e2[1];
  const float cy = r.dz * T.e2[0] - r.dx * T.e2[2];
  const float cz = r.dx * T.e2[1] - r.dy * T.e2[0];
  const float det = T.e1[0] * cx + T.e1[1] * cy + T.e1[2] * cz;
  const float e1m = fabsf(T.e1[0]) + fabsf(T.e1[1]) + fabsf(T.e1[2]);
  const float e2m = fabsf(T.e2[0]) + fabsf(T.e2[1]) + fabsf(T.e2[2]);
  const float errdet = 8.0f * D2 * e1m * dm * e2m;
  const float den = fabsf(det);
  if (den < (float)EPS - errdet) return FT_REJECT;  // |det| < EPSILON in f64 too
  if (!(den > 8.0f * errdet)) return FT_UNSURE;     // ill-conditioned (or NaN): nothing below can be trusted
  const bool det_sure = den > (float)EPS + errdet;
  const float f = 1.0f / det, fa = fabsf(f);
  const float px = (r.ox - T.p1[0]) + (r.oxl - __uint_as_float(T.p1_lo_xy << 16));
  const float py = (r.oy - T.p1[1]) + (r.oyl - __uint_as_float(T.p1_lo_xy & 0xffff0000u));
  const float pz = (r.oz - T.p1[2]) + (r.ozl - __uint_as_float(T.p1_lo_z << 16));
  const float pm = fabsf(px) + fabsf(py) + fabsf(pz);
  Pm = pm + Pm * (1.0f / 64.0f);
  const float u = f * (px * cx + py * cy + pz * cz);
  const float eu = (D2 * dm * e2m * (4.0f * Pm + 10.0f * pm) + fmaxf(1.0f, fabsf(u)) * errdet) * fa;
  if (u < -eu || u > 1.0f + eu) return FT_REJECT;
  const float qx = py * T.e1[2] - pz * T.e1[1];
  const float qy = pz * T.e1[0] - px * T.e1[2];
  const float qz = px * T.e1[1] - py * T.e1[0];
  const float v = f * (r.dx * qx + r.dy * qy + r.dz * qz);
  const float ev = (D2 * dm * e1m * (2.0f * Pm + 7.0f * pm) + fmaxf(1.0f, fabsf(v)) * errdet) * fa;
  if (v < -ev || (u + v) > 1.0f + (eu + ev)) return FT_REJECT;
  t = f * (T.e2[0] * qx + T.e2[1] * qy + T.e2[2] * qz);
  et = (D2 * e1m * e2m * (2.0f * Pm + 8.0f * pm) + fabsf(t) * errdet) * fa;
  const bool solid = det_sure && u > eu && u < 1.0f - eu && v > ev && (u + v) < 1.0f - (eu + ev) && (et < 1e30f);
  return solid ? FT_SOLID : FT_UNSURE;
}

// f64::max / f64::min ignore a NaN operand (cube.rs:74-75, 85-92); so do fmax/fmin.
__device__ __forceinline__ void cube_axis(double origin, double direction, double& tmin,
                                          double& tmax) {  // cube.rs:48-66
  double a = -1.0 - origin;
  double b = 1.0 - origin;
  if (fabs(direction) >= EPS) {
    tmin = a / direction;
    tmax = b / direction;
  } else {
    tmin = a * RT_INF;
    tmax = b * RT_INF;
  }
  if (tmin > tmax) {
    double s = tmin;
    tmin = tmax;
    tmax = s;
  }
}

// local_intersect of the closed-form shapes; ts[] in the reference's emission order.
__device__ inline int prim_local_intersect(const DPrim& P, const Ray& r, double ts[4]) {
  int n = 0;
  switch (P.kind) {
    case 0: {  // sphere.rs:65-84 (origin - point(0,0,0) is the origin itself)
      V3 str = r.o;
      double a = vdot(r.d, r.d);
      double b = 2.0 * vdot(str, r.d);
      double c = vdot(str, str) - 1.0;
      double disc = b * b - 4.0 * a * c;
      if (disc < 0.0) return 0;
      double sq = sqrt(disc);
      ts[0] = (-b - sq) / (2.0 * a);
      ts[1] = (-b + sq) / (2.0 * a);
      return 2;
    }
    case 1: {  // plane.rs:49-58
      if (fabs(r.d.y) < EPS) return 0;
      ts[0] = -r.o.y / r.d.y;
      return 1;
    }
    case 2: {  // cube.rs:69-82
      double x0, x1, y0, y1, z0, z1;
      cube_axis(r.o.x, r.d.x, x0, x1);
      cube_axis(r.o.y, r.d.y, y0, y1);
      cube_axis(r.o.z, r.d.z, z0, z1);
      double tmin = fmax(fmax(x0, y0), z0);
      double tmax = fmin(fmin(x1, y1), z1);
      if (tmin > tmax) return 0;
      ts[0] = tmin;
      ts[1] = tmax;
      return 2;
    }
    case 3: {  // cylinder.rs:78-119
      double a = r.d.x * r.d.x + r.d.z * r.d.z;
      if (!(fabs(a) < EPS)) {
        double b = 2.0 * r.o.x * r.d.x + 2.0 * r.o.z * r.d.z;
        double c = r.o.x * r.o.x + r.o.z * r.o.z - 1.0;
        double disc = b * b - 4.0 * a * c;
        if (disc < 0.0) return 0;
        double sq = sqrt(disc);
        double t0 = (-b - sq) / (2.0 * a);
        double t1 = (-b + sq) / (2.0 * a);
        if (t0 > t1) {
          double s = t0;
          t0 = t1;
          t1 = s;
        }
        double y0 = r.o.y + t0 * r.d.y;
        if (P.minimum < y0 && y0 < P.maximum) ts[n++] = t0;
        double y1 = r.o.y + t1 * r.d.y;
        if (P.minimum < y1 && y1 < P.maximum) ts[n++] = t1;
      }
      // cylinder.rs:53-74, 166-171
      if (P.closed && !(fabs(r.d.y) < EPS)) {
        double t0 = (P.minimum - r.o.y) / r.d.y;
        double x = r.o.x + t0 * r.d.x, z = r.o.z + t0 * r.d.z;
        if ((x * x + z * z) <= 1.0) ts[n++] = t0;
        double t1 = (P.maximum - r.o.y) / r.d.y;
        x = r.o.x + t1 * r.d.x;
        z = r.o.z + t1 * r.d.z;
        if ((x * x + z * z) <= 1.0) ts[n++] = t1;
      }
      return n;
    }
    case 4: {  // cone.rs:85-122
      double a = r.d.x * r.d.x - r.d.y * r.d.y + r.d.z * r.d.z;
      double b = 2.0 * r.d.x * r.o.x - 2.0 * r.d.y * r.o.y + 2.0 * r.d.z * r.o.z;
      double c = r.o.x * r.o.x - r.o.y * r.o.y + r.o.z * r.o.z;
      if (fabs(a) <= EPS && fabs(b) > EPS) ts[n++] = -c / (2.0 * b);
      if (fabs(a) > EPS) {
        double disc = b * b - 4.0 * a * c;
        if (disc >= 0.0) {
          double sq = sqrt(disc);
          double t0 = (-b - sq) / (2.0 * a);
          double t1 = (-b + sq) / (2.0 * a);
          if (t0 > t1) {
            double s = t0;
            t0 = t1;
            t1 = s;
          }
          double y0 = r.o.y + t0 * r.d.y;
          if (P.minimum < y0 && y0 < P.maximum) ts[n++] = t0;
          double y1 = r.o.y + t1 * r.d.y;
          if (P.minimum < y1 && y1 < P.maximum) ts[n++] = t1;
        }
      }
      // cone.rs:53-82: caps with radius = the y bound
      if (P.closed && !(fabs(r.d.y) < EPS)) {
        double t0 = (P.minimum - r.o.y) / r.d.y;
        double x = r.o.x + t0 * r.d.x, z = r.o.z + t0 * r.d.z;
        if ((x * x + z * z) <= P.minimum * P.minimum) ts[n++] = t0;
        double t1 = (P.maximum - r.o.y) / r.d.y;
        x = r.o.x + t1 * r.d.x;
        z = r.o.z + t1 * r.d.z;
        if ((x * x + z * z) <= P.maximum * P.maximum) ts[n++] = t1;
      }
      return n;
    }
  }
  return 0;
}

__device__ inline V3 prim_local_normal(const DPrim& P, V3 p) {
  switch (P.kind) {
    case 0:
      return p;  // sphere.rs:57-60
    case 1:
      return v3(0.0, 1.0, 0.0);  // plane.rs:60-62
    case 2: {                    // cube.rs:84-101: the component itself, not +-1
      double ax = fabs(p.x), ay = fabs(p.y), az = fabs(p.z);
      double maxc = fmax(fmax(ax, ay), az);
      if (maxc == ax) return v3(p.x, 0.0, 0.0);
      if (maxc == ay) return v3(0.0, p.y, 0.0);
      return v3(0.0, 0.0, p.z);
    }
    case 3: {  // cylinder.rs:121-131
      double dist = p.x * p.x + p.z * p.z;
      if (dist < 1.0 && p.y >= P.maximum - EPS) return v3(0.0, 1.0, 0.0);
      if (dist < 1.0 && p.y <= P.minimum + EPS) return v3(0.0, -1.0, 0.0);
      return v3(p.x, 0.0, p.z);
    }
    default: {  // cone.rs:124-138 (keeps the reference's `dist < 1.0` cap test)
      double dist = p.x * p.x + p.z * p.z;
      if (dist < 1.0 && p.y >= P.maximum - EPS) return v3(0.0, 1.0, 0.0);
      if (dist < 1.0 && p.y <= P.minimum + EPS) return v3(0.0, -1.0, 0.0);
      double y = sqrt(dist);
      if (p.y > 0.0) y *= -1.0;
      return v3(p.x, y, p.z);
    }
  }
}

__device__ __forceinline__ Ray ray_to_prim(const DScene& S, const DPrim& P, const Ray& r) {
  Ray q = r;
  if (P.parent >= 0) q = ray_to_group(S, P.parent, q);
  return xf_ray(P.inv, q);
}

// World-space box test in front of the closed-form shapes: a shape whose padded box the
// ray misses inside (tlo, thi) cannot contribute an intersection there.  The boxes are
// padded by 1e-6 of their size, orders of magnitude above the rounding of this f64 slab
// test; NaN (0 * inf on a ray lying in a face plane) is dropped by fmin/fmax, i.e. passes.
struct BoxRay {
  double o[3], inv[3];
};
__device__ __forceinline__ BoxRay make_box_ray(const Ray& r) {
  BoxRay b;
  b.o[0] = r.o.x; b.o[1] = r.o.y; b.o[2] = r.o.z;
  b.inv[0] = 1.0 / r.d.x; b.inv[1] = 1.0 / r.d.y; b.inv[2] = 1.0 / r.d.z;
  return b;
}
__device__ __forceinline__ bool box_may_hit(const DPrimBox& B, const BoxRay& r, double tlo, double thi) {
  double tin = tlo, tout = thi;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    double t0 = (B.lo[a] - r.o[a]) * r.inv[a], t1 = (B.hi[a] - r.o[a]) * r.inv[a];
    tin = fmax(tin, fmin(t0, t1));
    tout = fmin(tout, fmax(t0, t1));
  }
  return !(tin > tout + 1e-9 * (fabs(tin) + fabs(tout)));
}

// A cube whose box the ray misses may still be hit by the reference: check_axis (cube.rs:54-60)
// treats an object-space direction component below EPSILON as parallel and ignores the drift
// along that axis, so a near-parallel ray that geometrically leaves the slab by up to
// EPSILON * t is still given (tmin, tmax).  Such rays are never culled.
__device__ __forceinline__ bool cube_cull_unsafe(const DScene& S, const DPrim& P, const Ray& r) {
  V3 d = r.d;
  if (P.parent >= 0) {
    const DGroup& G = S.groups[P.parent];
    for (int k = 0; k < G.nchain; k++) d = xf_vector(S.groups[G.chain[k]].inv, d);
  }
  d = xf_vector(P.inv, d);
  return fabs(d.x) < EPS || fabs(d.y) < EPS || fabs(d.z) < EPS;
}
#ifndef RT_FAST_POW
#define RT_FAST_POW 1
#endif
#ifndef RT_WORLD_BOUNDS
#define RT_WORLD_BOUNDS 1
#endif
// the same test against a mesh's world-space bounds (DMesh::wlo / whi; infinite ones always pass)
__device__ __forceinline__ bool bounds_may_hit(const double* lo, const double* hi, const BoxRay& r, double tlo,
                                               double thi) {
  double tin = tlo, tout = thi;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    double t0 = (lo[a] - r.o[a]) * r.inv[a], t1 = (hi[a] - r.o[a]) * r.inv[a];
    tin = fmax(tin, fmin(t0, t1));
    tout = fmin(tout, fmax(t0, t1));
  }
  return !(tin > tout + 1e-9 * (fabs(tin) + fabs(tout)));
}

// true when primitive p can be skipped for intersections inside (tlo, thi)
__device__ __forceinline__ bool prim_culled(const DScene& S, int p, const Ray& ray, const BoxRay& br, double tlo,
                                            double thi) {
  if (box_may_hit(S.prim_boxes[p], br, tlo, thi)) return false;
  const DPrim& P = S.prims[p];
  return !(P.kind == 2 && cube_cull_unsafe(S, P, ray));
}

// ---------------------------------------------------------------------------------------
// BVH traversal.  The hierarchy only culls; what a triangle test returns is the f64
// arithmetic of moller_trumbore() above, bit for bit.  Culling itself runs in f32 against
// 64-byte four-wide nodes (the four children's boxes as 16-bit steps of a grid over the mesh's
// padded bounds, rounded outward from padded f64 boxes, and their four child indices / leaf
// codes) and is made conservative analytically:
//   * the ray is re-based to where it enters the mesh's root box (f64), so every f32
//     operand is of the mesh's own magnitude S = max |coordinate|;
//   * a direction component smaller than 1e-10 of the largest is replaced by that bound:
//     inside the mesh the culling ray then strays from the true one by < 4e-10 S, far below
//     the 1e-7 S padding of every box, and 1/d stays finite;
//   * a box plane lo + q*cell is met at t = q * (cell/d) + (lo - o)/d: one fma with
//     a = f32(cell/d) and c pre-biased by E = 6 * 2^-24 * (S + |o|) * |1/d|, which bounds the
//     rounding of a (times q <= 65535, i.e. at most the mesh's extent) and of the fma; c is
//     rounded away from the interval (down for near planes, up for far planes);
//   * the query interval (0 or the running predecessor below, the closest hit or the
//     shadow distance or the trigger above) is converted with outward rounding.
// A child is skipped only if its (enlarged) interval misses the (enlarged) query interval.
struct NodeR {
  uint32_t q[12];
  int32_t c[4];
};
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ NodeR load_node(const DNode4* p) {
  const u32x4* w = reinterpret_cast<const u32x4*>(p);
  const u32x4 a = w[0], b = w[1], d = w[2], e = w[3];
  NodeR r;
  r.q[0] = a.x; r.q[1] = a.y; r.q[2] = a.z; r.q[3] = a.w;
  r.q[4] = b.x; r.q[5] = b.y; r.q[6] = b.z; r.q[7] = b.w;
  r.q[8] = d.x; r.q[9] = d.y; r.q[10] = d.z; r.q[11] = d.w;
  r.c[0] = (int32_t)e.x; r.c[1] = (int32_t)e.y; r.c[2] = (int32_t)e.z; r.c[3] = (int32_t)e.w;
  return r;
}
// One triangle record as five 16-byte loads.  pin() makes the values live at one point, so
// the compiler issues all the loads of a batch back to back and waits once, instead of
// fetching p1 in a second, dependent round trip after the determinant test.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct TriRaw {
  u32x4 a, b, c, d, e;
};
__device__ __forceinline__ TriRaw load_tri_raw(const DTri* p) {
  const u32x4* q = reinterpret_cast<const u32x4*>(p);
  TriRaw r;
  r.a = q[0]; r.b = q[1]; r.c = q[2]; r.d = q[3]; r.e = q[4];
  return r;
}
__device__ __forceinline__ void pin(TriRaw& r) {
  asm volatile("" : "+v"(r.a), "+v"(r.b), "+v"(r.c), "+v"(r.d), "+v"(r.e));
}
__device__ __forceinline__ void pin(TriRaw& r, TriRaw& s) {
  asm volatile(""
               : "+v"(r.a), "+v"(r.b), "+v"(r.c), "+v"(r.d), "+v"(r.e), "+v"(s.a), "+v"(s.b),
                 "+v"(s.c), "+v"(s.d), "+v"(s.e));
}
struct TriRawF {
  u32x4 a, b, c;
};
__device__ __forceinline__ TriRawF load_tri_raw(const DTriF* p) {
  const u32x4* q = reinterpret_cast<const u32x4*>(p);
  TriRawF r;
  r.a = q[0]; r.b = q[1]; r.c = q[2];
  return r;
}
__device__ __forceinline__ void pin(TriRawF& r, TriRawF& s) {
  asm volatile("" : "+v"(r.a), "+v"(r.b), "+v"(r.c), "+v"(s.a), "+v"(s.b), "+v"(s.c));
}
__device__ __forceinline__ DTriF decode_tri(const TriRawF& r) {
  DTriF t;
  t.p1[0] = __uint_as_float(r.a.x); t.p1[1] = __uint_as_float(r.a.y); t.p1[2] = __uint_as_float(r.a.z);
  t.e1[0] = __uint_as_float(r.a.w); t.e1[1] = __uint_as_float(r.b.x); t.e1[2] = __uint_as_float(r.b.y);
  t.e2[0] = __uint_as_float(r.b.z); t.e2[1] = __uint_as_float(r.b.w); t.e2[2] = __uint_as_float(r.c.x);
  t.orig = r.c.y;
  t.p1_lo_xy = r.c.z;
  t.p1_lo_z = r.c.w;
  return t;
}
__device__ __forceinline__ DTri decode_tri(const TriRaw& r) {
  DTri t;
  t.p1[0] = __hiloint2double((int)r.a.y, (int)r.a.x);
  t.p1[1] = __hiloint2double((int)r.a.w, (int)r.a.z);
  t.p1[2] = __hiloint2double((int)r.b.y, (int)r.b.x);
  t.e1[0] = __hiloint2double((int)r.b.w, (int)r.b.z);
  t.e1[1] = __hiloint2double((int)r.c.y, (int)r.c.x);
  t.e1[2] = __hiloint2double((int)r.c.w, (int)r.c.z);
  t.e2[0] = __hiloint2double((int)r.d.y, (int)r.d.x);
  t.e2[1] = __hiloint2double((int)r.d.w, (int)r.d.z);
  t.e2[2] = __hiloint2double((int)r.e.y, (int)r.e.x);
  t.orig = r.e.z;
  t.smooth = r.e.w;
  return t;
}

__device__ __forceinline__ float f32_down(double x) {  // largest float <= x
  float f = (float)x;
  if ((double)f > x) f = __uint_as_float(__float_as_uint(f) + ((f > 0.0f) ? -1 : 1));
  if (f == 0.0f && x < 0.0) f = -1.401298464e-45f;
  return f;
}
__device__ __forceinline__ float f32_up(double x) { return -f32_down(-x); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
struct CullRay {
  float a[3];    // cell / d of the culling ray
  f32x2 nf[3];   // .x: near-plane offset, biased down; .y: far-plane offset, biased up
  int sh[3];     // rotation that brings the near plane's step to the low half of a node word: 16 if d < 0 else 0
  double t_base;
  bool miss;  // the ray does not touch the mesh's root box at all
};

// want_negative: the query reaches behind the origin.
__device__ inline CullRay make_cull_ray(const DMesh& M, const Ray& lr, bool want_negative) {
  CullRay c;
  const double o[3] = {lr.o.x, lr.o.y, lr.o.z};
  const double d[3] = {lr.d.x, lr.d.y, lr.d.z};
  // where the true ray crosses the root box (f64; NaN from 0*inf is ignored by fmin/fmax)
  double tin = -RT_INF, tout = RT_INF;
  double invd[3];
  for (int a = 0; a < 3; a++) {
    double inv = 1.0 / d[a];
    invd[a] = inv;
    double t0 = (M.lo[a] - o[a]) * inv, t1 = (M.hi[a] - o[a]) * inv;
    tin = fmax(tin, fmin(t0, t1));
    tout = fmin(tout, fmax(t0, t1));
  }
  c.miss = !(tin <= tout) || (!want_negative && tout < 0.0);
  double tb = tin;
  if (!want_negative && !(tb > 0.0)) tb = 0.0;
  if (!(tb > -1e300 && tb < 1e300)) tb = 0.0;
  c.t_base = tb;
  double dmax = fmax(fmax(fabs(d[0]), fabs(d[1])), fabs(d[2]));
  bool degenerate = !(dmax > 0.0 && dmax < 1e300);
  for (int a = 0; a < 3; a++) {
    double oa = o[a] + d[a] * tb;  // re-based origin
    double da = d[a];
    double dmin = dmax * 1e-10;
    double inv = invd[a];
    if (!(fabs(da) >= dmin)) {  // rare: only then is a second division needed
      da = (da < 0.0) ? -dmin : dmin;
      inv = 1.0 / da;
    }
    double oid = (M.lo[a] - oa) * inv;
    double E = (6.0 * 5.9604644775390625e-08) * (M.scale + fabs(oa)) * fabs(inv);
    c.sh[a] = (da < 0.0) ? 16 : 0;
    if (degenerate || !(fabs(oid) < 1e30) || !(fabs(inv) < 1e30)) {
      c.a[a] = 0.0f;  // no culling on this axis
      c.nf[a].x = -__builtin_huge_valf();
      c.nf[a].y = __builtin_huge_valf();
    } else {
      c.a[a] = (float)(M.cell[a] * inv);
      c.nf[a].x = f32_down(oid - E);
      c.nf[a].y = f32_up(oid + E);
    }
  }
  return c;
}

// Enlarged parametric interval (in t - t_base) of one child box (three node words).  Per
// axis: one rotation puts the near step in the low half, two half-word conversions, and one
// packed fma for both planes (each half rounds exactly as a scalar fma would).
__device__ __forceinline__ void slab32(const uint32_t* q, const CullRay& c, float& tn, float& tf) {
  f32x2 t[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const uint32_t r = __builtin_amdgcn_alignbit(q[a], q[a], (uint32_t)c.sh[a]);
    f32x2 x, m;
    x.x = (float)(r & 0xffffu);
    x.y = (float)(r >> 16);
    m.x = c.a[a];
    m.y = c.a[a];
    t[a] = __builtin_elementwise_fma(x, m, c.nf[a]);
  }
  tn = fmaxf(fmaxf(t[0].x, t[1].x), t[2].x);
  tf = fminf(fminf(t[0].y, t[1].y), t[2].y);
}

// Orders this lane's LDS accesses against those of the other lanes of its wave.  A wave's LDS
// instructions already execute in order; this tells the compiler, which reasons per lane and
// may otherwise forward a lane's own earlier store to its load of a slot that another lane
// has written in between.  Generates no instructions.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Inclusive prefix sum over the 64 lanes of a wave, all of which must be active: four shifted
// adds inside each row of 16 lanes, then the row totals carried over with the two row
// broadcasts (six cross-lane VALU operations, no LDS).
__device__ __forceinline__ int wave_inclusive_sum(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1, 0 shifted in
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return v;
}

// true on the first active lane of the wave (statistics: counts an event once per wave)
__device__ __forceinline__ bool wave_leader() {
  unsigned long long m = __ballot(1);
  return (int)__lane_id() == __ffsll((long long)m) - 1;
}

struct HitRec {
  double t, u, v;
  uint32_t order;
  int32_t prim;  // >= 0: primitive index; -1: triangle
  int32_t mesh;
  int32_t tri;  // global record index in the triangle table
};

// Traversal stack of one lane of a 64-lane block: the first LDS_STACK entries live in LDS
// (entry i of lane l at word i*64 + l: the lanes of a wave touch distinct banks whatever
// their depths), deeper entries spill to a private array that well-built trees never reach.
// The LDS pointer carries its address space so pushes and pops are plain ds_write/ds_read.
constexpr int STACK_MAX = 96;
#ifndef RT_LDS_STACK
#define RT_LDS_STACK 24
#endif
constexpr int LDS_STACK = RT_LDS_STACK;
// The builders put the TOP_NODES nodes nearest each root first, breadth first, so the part
// of a tree every ray walks sits in a few adjacent cache lines.
constexpr int TOP_NODES = 128;
constexpr int COOP_SLOTS = 64 * LEAF_MAX + 16;  // + the overhang of the last leaf's writes
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
struct TravStack {
  lds_int* lds;     // this lane's column
  int* spill;       // STACK_MAX - LDS_STACK private words
  lds_u32x4* rays;                                          // cooperative leaf tests: lane l's ray at rays[3*l..]
  __attribute__((address_space(3))) unsigned char* owners;  // ... and the slot -> lane table
  __device__ __forceinline__ void put(int i, int v) {
    if (i < LDS_STACK)
      lds[i * 64] = v;
    else
      spill[i - LDS_STACK] = v;
  }
  __device__ __forceinline__ int get(int i) const {
    return i < LDS_STACK ? lds[i * 64] : spill[i - LDS_STACK];
  }
};
// In a kernel of 64-thread blocks: RT_TRAV_STACK(st) declares the storage and the handle.
#define RT_TRAV_STACK(name, scene)                                              \
  __shared__ int name##_lds[rtd::LDS_STACK * 64];                               \
  int name##_spill[rtd::STACK_MAX - rtd::LDS_STACK];                            \
  rtd::TravStack name;                                                          \
  name.lds = (rtd::lds_int*)(name##_lds) + threadIdx.x;                         \
  name.spill = name##_spill;                                                    \
  name.rays = nullptr;                                                          \
  name.owners = nullptr
// ... plus the areas the cooperative leaf test shares between the lanes of the wave.
#define RT_TRAV_STACK_COOP(name, scene)                                         \
  RT_TRAV_STACK(name, scene);                                                   \
  __shared__ rtd::u32x4 name##_rays[3 * 64];                                    \
  __shared__ unsigned char name##_owners[rtd::COOP_SLOTS];                      \
  name.rays = (rtd::lds_u32x4*)(name##_rays);                                   \
  name.owners = (__attribute__((address_space(3))) unsigned char*)(name##_owners)

// Sort key of one entry of the reference's intersection list: ascending t, ties by leaf
// order, then by emission order inside one shape (stable sorts, group.rs:69, world.rs:29).
struct Key {
  double t;
  uint32_t order;
  int32_t emit;
};
__device__ __forceinline__ bool key_less(const Key& a, const Key& b) {
  if (a.t < b.t) return true;
  if (a.t > b.t) return false;
  if (a.order != b.order) return a.order < b.order;
  return a.emit < b.emit;
}

enum { Q_CLOSEST = 0, Q_ANY = 1, Q_PRED = 2 };
// The descent is left when fewer than RT_BREAK_NUM / RT_BREAK_DEN of the lanes still walking a
// mesh are descending (the others hold a leaf).  Measured on MI355X, 1/8, 1/4, 1/2, 3/4: 1 M
// triangles 10.7 / 9.8 / 8.94 / 8.90 ms, pawn 2.60 / 2.48 / 2.45 / 2.42, tea set 0.575 / 0.573 /
// 0.560 / 0.574.
#ifndef RT_BREAK_NUM
#define RT_BREAK_NUM 1
#endif
#ifndef RT_BREAK_DEN
#define RT_BREAK_DEN 2
#endif


// State of one query over the meshes.
//   Q_CLOSEST: smallest t > 0, ties by leaf order  (hi = best t so far)
//   Q_ANY:     is there 0 < t < hi
//   Q_PRED:    largest key below K over the whole line (t may be negative)
struct Query {
  double hi;          // upper end of the t interval of interest
  uint32_t hi_order;  // Q_CLOSEST: leaf order of the best hit; Q_PRED: K.order
  Key best;           // Q_PRED: best predecessor so far
  bool found;
  int32_t mesh, tri;
  double u, v;
  // fast mode only: how far the f32 value in `hi` (Q_CLOSEST) / `best.t` (Q_PRED) may be from the
  // f64 one (0: it is the f64 value), and the record `best` belongs to
  double tau;
  int32_t best_rec;
};

// Accepts one candidate intersection (t, u, v of triangle `rec`) into the query.
template <int MODE>
__device__ __forceinline__ bool accept_hit(double t, double u, double v, uint32_t ord, int m, int rec,
                                           Query& q) {
  if (MODE == Q_CLOSEST) {
    if (t > 0.0 && (t < q.hi || (t == q.hi && ord < q.hi_order))) {
      q.hi = t;
      q.hi_order = ord;
      q.mesh = m;
      q.tri = rec;
      q.u = u;
      q.v = v;
      q.found = true;
    }
  } else if (MODE == Q_ANY) {
    if (t > 0.0 && t < q.hi) return true;
  } else {
    // q.tri: the record of the hit whose predecessor is sought.  In f64 its own entry equals
    // the trigger key and is never below it; an f32 t of the same triangle could be.
    if (rec == q.tri) return false;
    Key c{t, ord, 0};
    Key K{q.hi, q.hi_order, 0};
    if (key_less(c, K) && (!q.found || key_less(q.best, c))) {
      q.best = c;
      q.mesh = m;
      q.found = true;
    }
  }
  return false;
}
// The same decision as accept_hit for a candidate that may not exist (`valid`), written as
// selects: inside the wave-uniform collection loop of the cooperative leaf test a branch per
// candidate costs more than the nine conditional moves.
template <int MODE>
__device__ __forceinline__ bool accept_hit_select(bool valid, double t, double u, double v, uint32_t ord, int m,
                                                  int rec, Query& q) {
  if (MODE == Q_CLOSEST) {
    const bool take = valid & (t > 0.0) & ((t < q.hi) | ((t == q.hi) & (ord < q.hi_order)));
    q.hi = take ? t : q.hi;
    q.hi_order = take ? ord : q.hi_order;
    q.mesh = take ? m : q.mesh;
    q.tri = take ? rec : q.tri;
    q.found = q.found | take;  // (q.u, q.v: filled in by trace_closest_coop when the walk is over)
    return false;
  } else if (MODE == Q_ANY) {
    return valid & (t > 0.0) & (t < q.hi);
  } else {
    const bool below_k = (t < q.hi) | ((t == q.hi) & (ord < q.hi_order));
    const bool above_best = !q.found | (q.best.t < t) | ((q.best.t == t) & (q.best.order < ord));
    const bool take = valid & (rec != q.tri) & below_k & above_best;
    q.best.t = take ? t : q.best.t;
    q.best.order = take ? ord : q.best.order;
    q.best.emit = take ? 0 : q.best.emit;
    q.mesh = take ? m : q.mesh;
    q.found = q.found | take;
    return false;
  }
}
// Fast mode: takes one candidate of the f32 search into the query.  `t` is the f32 value (on the
// ray's own parameter), `et` its error bound, `cls` FT_SOLID or FT_UNSURE; `rec` the triangle's
// record in the table, `lr` the ray in the mesh's space.  Whatever the f32 values cannot decide
// for certain — an unsure triangle, t within its bound of 0 or of the interval's end, two
// candidates closer together than their bounds — is decided by moller_trumbore() in f64 (rare:
// a divergent branch), so the query ends with the triangle the f64 search would have found.
// A best hit taken on an f32 value has q.tau > 0 until it is evaluated in f64 (at the latest
// when the walk of its mesh ends).
template <int MODE, bool STATS>
__device__ inline bool accept_fast(const DScene& S, const Ray& lr, bool valid, int cls, double t, double et,
                                   uint32_t ord, int m, int rec, Query& q, Counters& cnt) {
  if (!valid) return false;
  if (STATS) cnt.fast_candidates++;
  bool exact = cls != FT_SOLID;
  if (MODE == Q_CLOSEST) {
    if (!exact) {
      if (t < -et) return false;  // behind the origin for certain
      const double band = q.tau + et;
      if (!(t > et)) {
        exact = true;  // the sign of t is open
      } else if (t < q.hi - band) {
        q.hi = t;
        q.tau = et;
        q.hi_order = ord;
        q.mesh = m;
        q.tri = rec;
        q.found = true;
        return false;
      } else if (t > q.hi + band) {
        return false;
      } else {
        exact = true;  // too close to the best so far to be ordered in f32
      }
    }
    double te, ue, ve;
    if (STATS) cnt.fast_exact++;
    if (!moller_trumbore(S.tris[rec], lr, te, ue, ve) || !(te > 0.0)) return false;
    if (q.found && q.tau > 0.0) {  // the best so far in f64 as well (it is a triangle of this mesh)
      double tb, ub, vb;
      if (moller_trumbore(S.tris[q.tri], lr, tb, ub, vb)) {
        q.hi = tb;
        q.u = ub;
        q.v = vb;
        q.tau = 0.0;
      }
    }
    if (te < q.hi || (te == q.hi && ord < q.hi_order)) {
      q.hi = te;
      q.tau = 0.0;
      q.hi_order = ord;
      q.mesh = m;
      q.tri = rec;
      q.u = ue;
      q.v = ve;
      q.found = true;
    }
    return false;
  } else if (MODE == Q_ANY) {
    if (!exact) {
      if (t > et && t < q.hi - et) return true;
      if (t < -et || t > q.hi + et) return false;
    }
    double te, ue, ve;
    if (STATS) cnt.fast_exact++;
    return moller_trumbore(S.tris[rec], lr, te, ue, ve) && te > 0.0 && te < q.hi;
  } else {
    if (rec == q.tri) return false;  // the hit whose predecessor is sought
    if (!exact) {
      if (t > q.hi + et) return false;  // above the trigger for certain
      const double band = q.tau + et;
      if (!(t < q.hi - et)) {
        exact = true;  // which side of the trigger?
      } else if (!q.found || t > q.best.t + band) {
        q.best = Key{t, ord, 0};
        q.tau = et;
        q.best_rec = rec;
        q.mesh = m;
        q.found = true;
        return false;
      } else if (t < q.best.t - band) {
        return false;
      } else {
        exact = true;
      }
    }
    double te, ue, ve;
    if (STATS) cnt.fast_exact++;
    if (!moller_trumbore(S.tris[rec], lr, te, ue, ve)) return false;
    const Key c{te, ord, 0}, K{q.hi, q.hi_order, 0};
    if (!key_less(c, K)) return false;
    if (q.found && q.tau > 0.0) {
      double tb, ub, vb;
      if (moller_trumbore(S.tris[q.best_rec], lr, tb, ub, vb)) {
        q.best.t = tb;
        q.tau = 0.0;
      }
    }
    if (!q.found || key_less(q.best, c)) {
      q.best = c;
      q.tau = 0.0;
      q.best_rec = rec;
      q.mesh = m;
      q.found = true;
    }
    return false;
  }
}
// Fast mode: the walk of mesh `m` is over; a best hit of it still held as an f32 value gets its
// f64 value (what the parity mode computes for that triangle).
template <int MODE>
__device__ inline void settle_fast(const DScene& S, const Ray& lr, int m, Query& q) {
  if (MODE == Q_ANY || !q.found || !(q.tau > 0.0) || q.mesh != m) return;
  double tb, ub, vb;
  const int rec = (MODE == Q_CLOSEST) ? q.tri : q.best_rec;
  if (moller_trumbore(S.tris[rec], lr, tb, ub, vb)) {
    if (MODE == Q_CLOSEST) {
      q.hi = tb;
      q.u = ub;
      q.v = vb;
    } else {
      q.best.t = tb;
    }
  }
  q.tau = 0.0;
}

template <int MODE, bool STATS>
__device__ __forceinline__ bool test_tri(const DTri& T, uint32_t order_base, int m, int rec,
                                         const Ray& lr, Query& q, Counters& cnt) {
  double t, u, v;
  if (STATS) {
    if (wave_leader()) cnt.tri_wave_steps++;
    cnt.tri_tests++;
  }
  if (!moller_trumbore(T, lr, t, u, v)) return false;
  return accept_hit<MODE>(t, u, v, order_base + T.orig, m, rec, q);
}
// The triangles of one leaf, two at a time: both records are in flight together.  Which
// triangle wins does not depend on the test order (the comparisons carry the leaf order).
template <int MODE, bool STATS>
__device__ __forceinline__ bool test_leaf(uint32_t order_base, int tri_base, int m, const DTri* tris,
                                          int first, int count, const Ray& lr, Query& q, Counters& cnt) {
  for (int k = 0; k < count; k += 2) {
    const bool two = k + 1 < count;
    auto r0 = load_tri_raw(tris + first + k);
    auto r1 = load_tri_raw(tris + first + (two ? k + 1 : k));
    pin(r0, r1);
    if (test_tri<MODE, STATS>(decode_tri(r0), order_base, m, tri_base + first + k, lr, q, cnt))
      return true;
    if (two && test_tri<MODE, STATS>(decode_tri(r1), order_base, m, tri_base + first + k + 1, lr, q, cnt))
      return true;
  }
  return false;
}

// One step down a four-wide node: tests the four child boxes against the query interval
// [qlo, qhi], returns the child to visit next (inner node index >= 0 or leaf code <= -2) and
// pushes the other children that were hit so that the best of them is popped first.  With no
// child hit the next entry comes off the stack (-1 when it is empty).
//   Q_CLOSEST / Q_ANY: nearest entry point first; Q_PRED: the child reaching furthest up first.
// The order only affects how soon the interval tightens, never the result: the comparisons
// of accept_hit carry the leaf order.
template <int MODE>
__device__ __forceinline__ int node_step(const NodeR& N, const CullRay& c, float qlo, float qhi, int& sp,
                                         TravStack& st) {
  float key[4];
  int code[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    float tn, tf;
    slab32(N.q + 3 * k, c, tn, tf);
    // an empty slot (code -1, inverted box) needs no test of its own: whatever its key, its
    // code stays -1 and is neither pushed nor visited
    const bool h = fmaxf(tn, qlo) <= fminf(tf, qhi);
    key[k] = h ? ((MODE == Q_PRED) ? -tf : tn) : __builtin_huge_valf();
    code[k] = h ? N.c[k] : -1;
  }
  // sorting network on (key, code): misses (key = +inf, code = -1) end up last
#define RT_CSWAP(i, j)                                  \
  {                                                     \
    const bool sw = key[j] < key[i];                    \
    const float ka = sw ? key[j] : key[i];              \
    const float kb = sw ? key[i] : key[j];              \
    const int ca = sw ? code[j] : code[i];              \
    const int cb = sw ? code[i] : code[j];              \
    key[i] = ka; key[j] = kb; code[i] = ca; code[j] = cb; \
  }
  // Shadow queries stop at the first hit wherever it is: visiting order buys them little, so
  // they take the hit children as they come (measured: 2 % of the frame on MI355X).
  if (MODE != Q_ANY) {
    RT_CSWAP(0, 1) RT_CSWAP(2, 3) RT_CSWAP(0, 2) RT_CSWAP(1, 3) RT_CSWAP(1, 2)
  }
#undef RT_CSWAP
  // every hit child is pushed or returned whatever its position, so an unordered key
  // (NaN from a degenerate axis) costs order, not correctness
  // the scene commit refuses trees whose walk could need more than STACK_MAX entries
  if (code[3] != -1) st.put(sp++, code[3]);
  if (code[2] != -1) st.put(sp++, code[2]);
  if (code[1] != -1) st.put(sp++, code[1]);
  if (code[0] != -1) return code[0];
  return sp ? st.get(--sp) : -1;
}

// Walks every mesh.  Returns true as soon as a Q_ANY query is satisfied.
template <int MODE, bool STATS>
__device__ inline bool traverse_meshes(const DScene& S, const Ray& ray, Query& q, Counters& cnt,
                                       TravStack& st) {
  for (int m = 0; m < S.n_meshes; m++) {
    const DMesh& M = S.meshes[m];
    Ray lr = ray_to_group(S, M.group, ray);
    if (STATS) cnt.inst_entries++;
    CullRay c = make_cull_ray(M, lr, MODE == Q_PRED);
    if (c.miss) continue;
    const int tri_base = M.tri_base;
    const uint32_t order_base = M.order_base;
    const DTri* tris = S.tris + tri_base;
    // query interval in t - t_base, rounded outward
    float qlo, qhi;
    if (MODE == Q_PRED) {
      qlo = q.found ? f32_down(q.best.t - c.t_base) : -__builtin_huge_valf();
      qhi = f32_up(q.hi - c.t_base);
    } else {
      qlo = f32_down(0.0 - c.t_base);
      qhi = (q.hi < 1e300) ? f32_up(q.hi - c.t_base) : __builtin_huge_valf();
    }
    // "while-while" traversal: the lanes descend inner nodes together, then test the
    // triangles of the leaves they hold together, so that node fetches and triangle tests
    // each run with as many active lanes as possible.  Stack and `cur` hold inner node
    // indices (>= 0) or leaf codes (<= -2); -1 means nothing left.
    int sp = 0;
    int cur = M.root;
    unsigned long long visited = 0;
    while (cur != -1) {
      const int alive = __popcll(__ballot(true));  // lanes of this wave still walking this mesh
      while (cur >= 0) {
        const NodeR N = load_node(S.nodes + cur);
        if (STATS) {
          if (wave_leader()) cnt.node_wave_steps++;
          cnt.nodes_visited++;
          visited++;
          if (visited > cnt.max_nodes) cnt.max_nodes = visited;
        }
        cur = node_step<MODE>(N, c, qlo, qhi, sp, st);
        // Once fewer than half of the wave's walking lanes are still descending, the rest
        // wait with a leaf: test those now and let the stragglers descend on their next turn.
        // Measured on MI355X (pawn 1080p): waiting for every lane 8.5 ms, this rule 7.05 ms;
        // thresholds of 1/8, 1/4, 3/4 and "any lane has a leaf" were all slower.
        if (__popcll(__ballot(cur >= 0)) * RT_BREAK_DEN < alive * RT_BREAK_NUM) break;
      }
      if (cur <= -2) {
        int code = -2 - cur;
        const bool done = test_leaf<MODE, STATS>(order_base, tri_base, m, tris, code >> LEAF_BITS, code & LEAF_MAX, lr, q, cnt);
        if (done) return true;
        // a leaf may have tightened the interval
        if (MODE == Q_CLOSEST) {
          if (q.hi < 1e300) qhi = f32_up(q.hi - c.t_base);
        } else if (MODE == Q_PRED) {
          if (q.found) qlo = f32_down(q.best.t - c.t_base);
        }
        cur = sp ? st.get(--sp) : -1;
      }
    }
  }
  return false;
}

// ---------------------------------------------------------------------------------------
// Cooperative traversal: the same walk as traverse_meshes, called by ALL 64 lanes of a wave
// in uniform control flow (`enabled` says whether this lane has a query).  Inner nodes are
// walked per lane as before.  The leaf phase is shared: the (lane, triangle) pairs of every
// leaf held in the wave are dealt out over all 64 lanes, so a phase in which 10 lanes hold
// leaves of 1-6 triangles costs one 64-wide test instead of three 10-wide pair tests.  The
// owner of a leaf takes the results (t, u, v) of its triangles from the workers' registers;
// they are the same arithmetic on the same operands as the per-lane path computes.
// `wr`: the world-space ray prepared for box tests (make_box_ray; read for enabled lanes only).
template <int MODE, bool STATS, bool FAST>
__device__ inline bool traverse_meshes_coop(const DScene& S, const Ray& ray, const BoxRay& wr, Query& q,
                                            bool enabled, Counters& cnt, TravStack& st) {
  const int lane = threadIdx.x & 63;
  bool satisfied = false;  // Q_ANY: this lane has its answer
  if (__ballot(enabled) == 0ull) return false;  // no lane of the wave has a query (wave-uniform)
  // A query whose interval misses a mesh's bounds in world space has nothing to find in it, and
  // a wave without any other query skips the mesh before a ray is taken into its space.  The
  // tests are made up front, on the interval the query starts with, and kept as one bit per
  // mesh (the first 64; later ones are always walked): the world-space ray is then done with.
  unsigned long long may = ~0ull;
  if (RT_WORLD_BOUNDS && enabled) {
    const double below = (MODE == Q_PRED) ? (q.found ? q.best.t - q.tau : -RT_INF) : 0.0;
    const double above = (MODE == Q_CLOSEST) ? q.hi + q.tau : q.hi;
    may = 0ull;
    for (int m = 0; m < S.n_meshes && m < 64; m++)
      if (bounds_may_hit(S.meshes[m].wlo, S.meshes[m].whi, wr, below, above)) may |= 1ull << m;
  }
  for (int m = 0; m < S.n_meshes; m++) {
    const DMesh& M = S.meshes[m];
    const bool wanted = enabled && !satisfied && (m >= 64 || ((may >> m) & 1ull));
    if (__ballot(wanted) == 0ull) continue;  // wave-uniform
    Ray lr = ray;
    CullRay c;
    c.miss = true;
    c.t_base = 0.0;
    for (int a = 0; a < 3; a++) {
      c.a[a] = 0.f; c.nf[a].x = 0.f; c.nf[a].y = 0.f; c.sh[a] = 0;
    }
    const bool walking0 = wanted;
    if (walking0) {
      lr = ray_to_group(S, M.group, ray);
      if (STATS) cnt.inst_entries++;
      c = make_cull_ray(M, lr, MODE == Q_PRED);
    }
    const int tri_base = M.tri_base;
    const uint32_t order_base = M.order_base;
    const DTri* tris = S.tris + tri_base;
    const DTriF* trisf = S.trisf + tri_base;
    // every lane publishes its object-space ray for the workers (fast mode: with the origin moved
    // to where the ray enters the mesh's box, so that the f32 test works on coordinates of the
    // mesh's own magnitude; its t then counts from c.t_base)
    {
      V3 po = lr.o;
      if (FAST) po = ray_at(lr, c.t_base);
      u32x4 w0, w1, w2;
      w0.x = (unsigned)__double2loint(po.x); w0.y = (unsigned)__double2hiint(po.x);
      w0.z = (unsigned)__double2loint(po.y); w0.w = (unsigned)__double2hiint(po.y);
      w1.x = (unsigned)__double2loint(po.z); w1.y = (unsigned)__double2hiint(po.z);
      w1.z = (unsigned)__double2loint(lr.d.x); w1.w = (unsigned)__double2hiint(lr.d.x);
      w2.x = (unsigned)__double2loint(lr.d.y); w2.y = (unsigned)__double2hiint(lr.d.y);
      w2.z = (unsigned)__double2loint(lr.d.z); w2.w = (unsigned)__double2hiint(lr.d.z);
      st.rays[3 * lane + 0] = w0;
      st.rays[3 * lane + 1] = w1;
      st.rays[3 * lane + 2] = w2;
      wave_lds_fence();
    }
    float qlo = 0.f, qhi = 0.f;
    if (MODE == Q_PRED) {
      qlo = q.found ? f32_down(q.best.t - c.t_base) : -__builtin_huge_valf();
      qhi = f32_up(q.hi - c.t_base);
    } else {
      qlo = f32_down(0.0 - c.t_base);
      qhi = (q.hi < 1e300) ? f32_up(q.hi - c.t_base) : __builtin_huge_valf();
    }
    const float mesh_scale = (float)M.scale * 1.0001f;
    int sp = 0;
    int cur = (walking0 && !c.miss) ? M.root : -1;
    unsigned long long visited = 0;
    while (__ballot(cur != -1) != 0ull) {  // wave-uniform
      const int alive = __popcll(__ballot(cur != -1));
      while (cur >= 0) {
        const NodeR N = load_node(S.nodes + cur);
        if (STATS) {
          if (wave_leader()) cnt.node_wave_steps++;
          cnt.nodes_visited++;
          visited++;
          if (visited > cnt.max_nodes) cnt.max_nodes = visited;
        }
        cur = node_step<MODE>(N, c, qlo, qhi, sp, st);
        if (__popcll(__ballot(cur >= 0)) * RT_BREAK_DEN < alive * RT_BREAK_NUM) break;
      }
      // ---- shared leaf phase (all 64 lanes)
      const bool has_leaf = cur <= -2;
      const int code = has_leaf ? -2 - cur : 0;
      const int lfirst = code >> LEAF_BITS;
      const int lcount = has_leaf ? (code & LEAF_MAX) : 0;
      // prefix sum of the counts over the wave
      const int upto = wave_inclusive_sum(lcount);
      const int off = upto - lcount;
      const int total = __builtin_amdgcn_readlane(upto, 63);
      bool hit_any = false;
      // Slot -> owner table without a branch per slot: every lane with a leaf writes its id
      // to LEAF_MAX slots from its first one, highest slot first.  What it writes beyond its
      // own leaf lands on later lanes' slots and is overwritten by them afterwards: a slot's
      // final writer is the lane with the latest start at or below it — its owner.  (Two
      // lanes never write one slot in the same step: their starts differ.)
      if (lcount > 0) {
#pragma unroll
        for (int j = LEAF_MAX - 1; j >= 0; j--) {
          st.owners[off + j] = (unsigned char)lane;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // keep the stores in this order
        }
      }
      wave_lds_fence();
      for (int base = 0; base < total; base += 64) {  // wave-uniform
        const int s = base + lane;
        const bool work = s < total;
        const int o = work ? (int)st.owners[s] : lane;
        const int o_off = __shfl(off, o, 64);
        const int o_first = __shfl(lfirst, o, 64);
        bool ok = false;
        double wt = 0.0, wu = 0.0, wv = 0.0;  // the worker's result
        float wft = 0.f, wfe = 0.f;           // ... in fast mode: f32 t (from c.t_base) and its error bound
        uint32_t worig = 0;                   // (fast mode: the test's verdict in the two top bits)
        if (work) {
          const int rec = o_first + (s - o_off);
          if (STATS) {
            if (wave_leader()) cnt.tri_wave_steps++;
            cnt.tri_tests++;
          }
          const u32x4 w0 = st.rays[3 * o + 0], w1 = st.rays[3 * o + 1], w2 = st.rays[3 * o + 2];
          if (FAST) {
            RayF rf;
            const double ox = __hiloint2double((int)w0.y, (int)w0.x), oy = __hiloint2double((int)w0.w, (int)w0.z),
                         oz = __hiloint2double((int)w1.y, (int)w1.x);
            rf.ox = (float)ox;
            rf.oy = (float)oy;
            rf.oz = (float)oz;
            rf.oxl = (float)(ox - (double)rf.ox);
            rf.oyl = (float)(oy - (double)rf.oy);
            rf.ozl = (float)(oz - (double)rf.oz);
            rf.dx = (float)__hiloint2double((int)w1.w, (int)w1.z);
            rf.dy = (float)__hiloint2double((int)w2.y, (int)w2.x);
            rf.dz = (float)__hiloint2double((int)w2.w, (int)w2.z);
            const DTriF T = decode_tri(load_tri_raw(trisf + rec));
            const float Pm = (fmaxf(fmaxf(fabsf(rf.ox), fabsf(rf.oy)), fabsf(rf.oz)) + mesh_scale) * 1.0001f;
            const float dm = (fabsf(rf.dx) + fabsf(rf.dy) + fabsf(rf.dz)) * 1.0001f;
            const int cls = moller_trumbore_f32(T, rf, Pm, dm, wft, wfe);
            ok = cls != FT_REJECT;
            worig = T.orig | ((uint32_t)cls << 30);
          } else {
            Ray r;
            r.o = v3(__hiloint2double((int)w0.y, (int)w0.x), __hiloint2double((int)w0.w, (int)w0.z),
                     __hiloint2double((int)w1.y, (int)w1.x));
            r.d = v3(__hiloint2double((int)w1.w, (int)w1.z), __hiloint2double((int)w2.y, (int)w2.x),
                     __hiloint2double((int)w2.w, (int)w2.z));
            TriRaw raw = load_tri_raw(tris + rec);
            pin(raw);
            const DTri T = decode_tri(raw);
            ok = moller_trumbore(T, r, wt, wu, wv) && (MODE == Q_PRED || wt > 0.0);
            worig = T.orig;
#ifdef RT_TRI_PLANE_STATS
            if (MODE != Q_PRED) {  // where does the ray meet the triangle's plane: behind, in the interval, beyond?
              const V3 N = vcross(v3(T.e1[0], T.e1[1], T.e1[2]), v3(T.e2[0], T.e2[1], T.e2[2]));
              const double a = vdot(N, vsub(r.o, v3(T.p1[0], T.p1[1], T.p1[2]))), b = vdot(N, r.d);
              const double tp = -a / b;
              const double ohi = __shfl(q.hi, o, 64);
              const int base_ = MODE == Q_CLOSEST ? 0 : 4;
              atomicAdd(&g_tri_plane[base_ + 0], 1ull);
              if (!(tp > 0.0)) atomicAdd(&g_tri_plane[base_ + 1], 1ull);
              else if (tp > ohi) atomicAdd(&g_tri_plane[base_ + 2], 1ull);
              if (ok) atomicAdd(&g_tri_plane[base_ + 3], 1ull);
            }
#endif
          }
        }
        unsigned long long mask = __ballot(ok);
        // Owners collect the results of their slots in this chunk, one per turn, straight
        // from the workers' registers: the values are the worker's, i.e. those of the same
        // arithmetic on the same operands as the per-lane path.
        const int lo = max(off, base) - base, hi = min(off + lcount, base + 64) - base;  // my slots here
        unsigned long long mine = (lo < hi) ? (mask >> lo) & ((1ull << (hi - lo)) - 1ull) : 0ull;
        while (__ballot(mine != 0ull) != 0ull) {  // wave-uniform
          const int k = mine ? __ffsll((long long)mine) - 1 : 0;
          const int src = mine ? lo + k : lane;  // worker lane holding my next candidate
          const uint32_t orig = (uint32_t)__shfl((int)worig, src, 64);
          const int rec = lfirst + (base + lo + k - off);
          if (FAST) {
            const double t = (double)__shfl(wft, src, 64) + c.t_base, et = (double)__shfl(wfe, src, 64);
            hit_any |= accept_fast<MODE, STATS>(S, lr, mine != 0ull, (int)(orig >> 30), t, et,
                                                order_base + (orig & 0x3fffffffu), m, tri_base + rec, q, cnt);
          } else {
            // (u and v stay with the worker: the closest-hit query evaluates its final triangle
            // once more when the walk is over, see trace_closest_coop)
            const double t = __shfl(wt, src, 64);
            hit_any |= accept_hit_select<MODE>(mine != 0ull, t, 0.0, 0.0, order_base + orig, m, tri_base + rec, q);
          }
          mine &= mine - 1ull;  // 0 stays 0
        }
      }
      // ---- owners move on
      if (has_leaf) {
        if (MODE == Q_ANY && hit_any) {
          satisfied = true;
          cur = -1;
        } else {
          // (q.tau: fast mode's bound on an f32 value; 0 otherwise)
          if (MODE == Q_CLOSEST) {
            if (q.hi < 1e300) qhi = f32_up((q.hi + q.tau) - c.t_base);
          } else if (MODE == Q_PRED) {
            if (q.found) qlo = f32_down((q.best.t - q.tau) - c.t_base);
          }
          cur = sp ? st.get(--sp) : -1;
        }
      }
    }
    if (FAST && walking0) settle_fast<MODE>(S, lr, m, q);
  }
  return satisfied;
}

// hit() of intersect.rs:125-139 over the list of world.rs:22-32: smallest t > 0, the
// earliest leaf among equal t.
template <bool STATS>
__device__ inline bool trace_closest(const DScene& S, const Ray& ray, HitRec& h, Counters& cnt,
                                     TravStack& st) {
  Query q;
  q.hi = RT_INF;
  q.hi_order = 0xffffffffu;
  q.found = false;
  q.tau = 0.0;
  q.best_rec = -1;
  q.mesh = -1;
  q.tri = -1;
  q.u = 0.0;
  q.v = 0.0;
  int prim = -1;
  const BoxRay br = make_box_ray(ray);
  for (int p = 0; p < S.n_prims; p++) {
    if (prim_culled(S, p, ray, br, 0.0, q.hi)) continue;
    const DPrim& P = S.prims[p];
    Ray lr = ray_to_prim(S, P, ray);
    double ts[4];
    int n = prim_local_intersect(P, lr, ts);
    if (STATS) cnt.prim_tests++;
    for (int k = 0; k < n; k++) {
      double t = ts[k];
      if (t > 0.0 && (t < q.hi || (t == q.hi && P.order < q.hi_order))) {
        q.hi = t;
        q.hi_order = P.order;
        prim = p;
        q.found = true;
      }
    }
  }
  traverse_meshes<Q_CLOSEST, STATS>(S, ray, q, cnt, st);
  h.t = q.hi;
  h.order = q.hi_order;
  h.u = q.u;
  h.v = q.v;
  h.mesh = q.mesh;
  h.tri = q.tri;
  h.prim = (q.mesh >= 0) ? -1 : prim;
  return q.found;
}

// is_shadowed (world.rs:116-134): the nearest positive hit is closer than `distance`
// exactly when some intersection has 0 < t < distance.
template <bool STATS>
__device__ inline bool trace_any(const DScene& S, const Ray& ray, double distance, Counters& cnt,
                                 TravStack& st) {
  const BoxRay br = make_box_ray(ray);
  for (int p = 0; p < S.n_prims; p++) {
    if (prim_culled(S, p, ray, br, 0.0, distance)) continue;
    const DPrim& P = S.prims[p];
    Ray lr = ray_to_prim(S, P, ray);
    double ts[4];
    int n = prim_local_intersect(P, lr, ts);
    if (STATS) cnt.prim_tests++;
    for (int k = 0; k < n; k++)
      if (ts[k] > 0.0 && ts[k] < distance) return true;
  }
  Query q;
  q.hi = distance;
  q.hi_order = 0;
  q.found = false;
  q.tau = 0.0;
  q.best_rec = -1;
  q.mesh = -1;
  q.tri = -1;
  return traverse_meshes<Q_ANY, STATS>(S, ray, q, cnt, st);
}

// The key a predecessor query actually has to search below.  The hit is the smallest
// (t, leaf order) over t > 0 of ALL shapes (intersect.rs:125-139), and the trigger K is that
// hit or, for a closed-form shape, an entry of it within EPSILON below; so no triangle
// intersects the ray anywhere in 0 < t with a key below K: it would have been the hit.  With
// K.t > 0 the triangle predecessor is therefore the largest key with t <= 0 — the nearest
// intersection behind the origin, ties to the highest leaf order — and the walk can skip the
// whole stretch between the origin and the hit.
__device__ __forceinline__ Key predecessor_bound(const Key& K) {
  if (K.t > 0.0) return Key{0.0, 0xffffffffu, 0};
  return K;
}

// Largest key below K among all triangle intersections of the infinite line (t may be
// negative).  Every triangle is its own shape with at most one entry, so this is the last
// triangle still "open" when the containers walk of intersect.rs:188-222 reaches K.
template <bool STATS>
__device__ inline bool tri_predecessor(const DScene& S, const Ray& ray, const Key& K, int own_tri,
                                       Key& best, int& best_mesh, Counters& cnt, TravStack& st) {
  Query q;
  q.hi = K.t;
  q.hi_order = K.order;
  q.found = false;
  q.tau = 0.0;
  q.best_rec = -1;
  q.mesh = -1;
  q.tri = own_tri;
  traverse_meshes<Q_PRED, STATS>(S, ray, q, cnt, st);
  if (q.found) {
    best = q.best;
    best_mesh = q.mesh;
  }
  return q.found;
}

// The three queries on the cooperative walk.  Called by all 64 lanes; `enabled` = this lane
// has a query.  Closed-form shapes are tested per lane as in the per-lane versions.
template <bool STATS, bool FAST>
__device__ inline bool trace_closest_coop(const DScene& S, const Ray& ray, bool enabled, HitRec& h,
                                          Counters& cnt, TravStack& st) {
  Query q;
  q.hi = RT_INF;
  q.hi_order = 0xffffffffu;
  q.found = false;
  q.tau = 0.0;
  q.best_rec = -1;
  q.mesh = -1;
  q.tri = -1;
  q.u = 0.0;
  q.v = 0.0;
  q.best = Key{0.0, 0u, 0};
  int prim = -1;
  BoxRay br = {};
  if (enabled) {
    br = make_box_ray(ray);
    for (int p = 0; p < S.n_prims; p++) {
      if (prim_culled(S, p, ray, br, 0.0, q.hi)) continue;
      const DPrim& P = S.prims[p];
      Ray lr = ray_to_prim(S, P, ray);
      double ts[4];
      int n = prim_local_intersect(P, lr, ts);
      if (STATS) cnt.prim_tests++;
      for (int k = 0; k < n; k++) {
        double t = ts[k];
        if (t > 0.0 && (t < q.hi || (t == q.hi && P.order < q.hi_order))) {
          q.hi = t;
          q.hi_order = P.order;
          prim = p;
          q.found = true;
        }
      }
    }
  }
  traverse_meshes_coop<Q_CLOSEST, STATS, FAST>(S, ray, br, q, enabled, cnt, st);
  if (!FAST && enabled && q.mesh >= 0) {
    // u, v of the triangle that won: the same test on the same operands as the worker ran,
    // hence the same bits (and the same t)
    const Ray lr = ray_to_group(S, S.meshes[q.mesh].group, ray);
    double t;
    moller_trumbore(S.tris[q.tri], lr, t, q.u, q.v);
  }
  h.t = q.hi;
  h.order = q.hi_order;
  h.u = q.u;
  h.v = q.v;
  h.mesh = q.mesh;
  h.tri = q.tri;
  h.prim = (q.mesh >= 0) ? -1 : prim;
  return enabled && q.found;
}

template <bool STATS, bool FAST>
__device__ inline bool trace_any_coop(const DScene& S, const Ray& ray, double distance, bool enabled,
                                      Counters& cnt, TravStack& st) {
  bool found = false;
  BoxRay br = {};
  if (enabled) {
    br = make_box_ray(ray);
    for (int p = 0; p < S.n_prims && !found; p++) {
      if (prim_culled(S, p, ray, br, 0.0, distance)) continue;
      const DPrim& P = S.prims[p];
      Ray lr = ray_to_prim(S, P, ray);
      double ts[4];
      int n = prim_local_intersect(P, lr, ts);
      if (STATS) cnt.prim_tests++;
      for (int k = 0; k < n; k++)
        if (ts[k] > 0.0 && ts[k] < distance) found = true;
    }
  }
  Query q;
  q.hi = distance;
  q.hi_order = 0;
  q.found = false;
  q.tau = 0.0;
  q.best_rec = -1;
  q.mesh = -1;
  q.tri = -1;
  q.u = q.v = 0.0;
  q.best = Key{0.0, 0u, 0};
  bool mesh_hit = traverse_meshes_coop<Q_ANY, STATS, FAST>(S, ray, br, q, enabled && !found, cnt, st);
  return found || mesh_hit;
}

template <bool STATS, bool FAST>
__device__ inline bool tri_predecessor_coop(const DScene& S, const Ray& ray, const Key& K, int own_tri,
                                            bool enabled, Key& best, int& best_mesh, Counters& cnt,
                                            TravStack& st) {
  Query q;
  q.hi = K.t;
  q.hi_order = K.order;
  q.found = false;
  q.tau = 0.0;
  q.best_rec = -1;
  q.mesh = -1;
  q.tri = own_tri;
  q.u = q.v = 0.0;
  q.best = Key{0.0, 0u, 0};
  BoxRay br = {};
  if (enabled) br = make_box_ray(ray);
  traverse_meshes_coop<Q_PRED, STATS, FAST>(S, ray, br, q, enabled, cnt, st);
  if (enabled && q.found) {
    best = q.best;
    best_mesh = q.mesh;
  }
  return enabled && q.found;
}

// n1/n2 of prepare_computations (intersect.rs:188-222) without the list.
// The walk toggles a shape in `containers` at each of its entries, and reads
// containers.last() just before and just after toggling the hit's shape at the first
// entry of that shape within EPSILON of the hit's t (hits_equal, intersect.rs:164-166).
// A shape is in the list when it has an odd number of entries before that point, and the
// list is ordered by each shape's latest entry; so last() is the odd-count shape whose
// latest entry has the largest key.

// The trigger entry: the hit itself for a triangle; for a closed-form shape the first of its
// own entries within EPSILON of the hit's t.
__device__ inline Key trigger_key(const DScene& S, const Ray& ray, const HitRec& h) {
  Key K{h.t, h.order, 0};
  if (h.prim >= 0) {
    const DPrim& P = S.prims[h.prim];
    Ray lr = ray_to_prim(S, P, ray);
    double ts[4];
    int n = prim_local_intersect(P, lr, ts);
    bool have = false;
    for (int k = 0; k < n; k++) {
      if (!feq(ts[k], h.t)) continue;
      Key c{ts[k], P.order, k};
      if (!have || key_less(c, K)) {
        K = c;
        have = true;
      }
    }
  }
  return K;
}

// Closed-form shapes are enumerated exactly; the triangles' contribution is their
// predecessor below K (tri_found / tri_best / tri_mesh), found by tri_predecessor.
template <bool STATS>
__device__ inline void containers_from(const DScene& S, const Ray& ray, const HitRec& h, const Key& K,
                                       double own_ri, bool tri_found, const Key& tri_best,
                                       int tri_mesh, double& n1, double& n2, Counters& cnt) {
  bool have_all = false, have_others = false, own_odd = false;
  Key best_all{0.0, 0u, 0}, best_others{0.0, 0u, 0};
  double ri_all = 1.0, ri_others = 1.0;
  const BoxRay br = make_box_ray(ray);
  for (int p = 0; p < S.n_prims; p++) {
    // only entries below K count, and they have t <= K.t
    if (prim_culled(S, p, ray, br, -RT_INF, K.t)) continue;
    const DPrim& P = S.prims[p];
    Ray lr = ray_to_prim(S, P, ray);
    double ts[4];
    int n = prim_local_intersect(P, lr, ts);
    if (STATS) cnt.prim_tests++;
    int count = 0;
    bool have = false;
    Key last{0.0, 0u, 0};
    for (int k = 0; k < n; k++) {
      Key c{ts[k], P.order, k};
      if (!key_less(c, K)) continue;
      count++;
      if (!have || key_less(last, c)) {
        last = c;
        have = true;
      }
    }
    if ((count & 1) == 0) continue;
    double ri = S.mats[P.material].refractive_index;
    if (!have_all || key_less(best_all, last)) {
      best_all = last;
      ri_all = ri;
      have_all = true;
    }
    if (p == h.prim) {
      own_odd = true;
    } else if (!have_others || key_less(best_others, last)) {
      best_others = last;
      ri_others = ri;
      have_others = true;
    }
  }
  if (tri_found) {
    double ri = S.mats[S.meshes[tri_mesh].material].refractive_index;
    if (!have_all || key_less(best_all, tri_best)) {
      best_all = tri_best;
      ri_all = ri;
      have_all = true;
    }
    if (!have_others || key_less(best_others, tri_best)) {
      best_others = tri_best;
      ri_others = ri;
      have_others = true;
    }
  }
  n1 = have_all ? ri_all : 1.0;
  if (own_odd)
    n2 = have_others ? ri_others : 1.0;  // the hit leaves its own shape
  else
    n2 = own_ri;  // the hit enters its own shape, which becomes the last container
}

template <bool STATS>
__device__ inline void containers_n1_n2(const DScene& S, const Ray& ray, const HitRec& h,
                                        double own_ri, double& n1, double& n2, Counters& cnt,
                                        TravStack& st) {
  if (STATS) cnt.pred_queries++;
  Key K = trigger_key(S, ray, h);
  Key tb{0.0, 0u, 0};
  int tm = -1;
  bool tf = (S.n_meshes > 0) &&
            tri_predecessor<STATS>(S, ray, predecessor_bound(K), h.prim >= 0 ? -1 : h.tri, tb, tm, cnt, st);
  containers_from<STATS>(S, ray, h, K, own_ri, tf, tb, tm, n1, n2, cnt);
}

// ---------------------------------------------------------------------------------------
// Rust `x as i64`: saturating, NaN -> 0.  Only the parity is used.
__device__ __forceinline__ bool floor_is_even(double f) {
  if (f != f) return true;                       // NaN -> 0
  if (f >= 9223372036854775807.0) return false;  // i64::MAX is odd
  if (f <= -9223372036854775808.0) return true;  // i64::MIN is even
  long long i = (long long)f;
  return (i % 2) == 0;
}
__device__ inline C3 pattern_color(const DMaterial& M, V3 p) {  // patterns.rs
  C3 a = {M.a[0], M.a[1], M.a[2]};
  C3 b = {M.b[0], M.b[1], M.b[2]};
  switch (M.pattern_kind) {
    case 1:
      return floor_is_even(floor(p.x)) ? a : b;  // patterns.rs:77-83
    case 2: {                                    // patterns.rs:145-149
      double fr = p.x - floor(p.x);
      return {a.r + (b.r - a.r) * fr, a.g + (b.g - a.g) * fr, a.b + (b.b - a.b) * fr};
    }
    case 3:
      return floor_is_even(floor(sqrt(p.x * p.x + p.z * p.z))) ? a : b;  // patterns.rs:223-229
    case 4:
      return floor_is_even(floor(p.x) + floor(p.y) + floor(p.z)) ? a : b;  // patterns.rs:277-283
    default:
      return a;  // patterns.rs:30-32
  }
}

// Result of shading one hit (shade_hit, world.rs:34-65, up to the two recursive calls).
struct ShadeOut {
  C3 surface;
  double reflective, transparency, reflectance;
  bool use_schlick;  // reflective > 0 && transparency > 0
  bool do_reflect;   // !f64_eq(reflective, 0)
  bool do_refract;   // !f64_eq(transparency, 0) and no total internal reflection
  Ray reflect_ray, refract_ray;
  // for the known-answer interface
  V3 point, over_point, under_point, eyev, normalv, reflectv;
  double n1, n2;
  bool inside;
};

// The geometric half of prepare_computations (intersect.rs:168-186).
struct Prep {
  V3 normal, over, under;
  int32_t mat_index, parent_group, leaf_prim;  // leaf_prim >= 0: closed-form shape index
  bool inside;
};
__device__ inline Prep prepare_hit(const DScene& S, const Ray& ray, const HitRec& h, V3* point_out = nullptr) {
  Prep pr;
  V3 point = ray_at(ray, h.t);
  V3 normal;
  pr.leaf_prim = h.prim;
  if (h.prim >= 0) {
    const DPrim& P = S.prims[h.prim];
    pr.mat_index = P.material;
    pr.parent_group = P.parent;
    V3 lp = point;
    if (P.parent >= 0) lp = point_to_group(S, P.parent, lp);
    lp = xf_point(P.inv, lp);
    V3 ln = prim_local_normal(P, lp);
    normal = vnormalize(xf_normal(P.inv, ln));  // intersect.rs:81-83
  } else {
    const DMesh& M = S.meshes[h.mesh];
    const DTri& T = S.tris[h.tri];
    const DTriN& TN = S.trin[h.tri];
    pr.mat_index = M.material;
    pr.parent_group = M.group;
    V3 ln;
    if (T.smooth) {  // smooth_triangle.rs:114-118: (n2*u + n3*v) + n1*((1-u)-v)
      double w = 1.0 - h.u - h.v;
      ln = v3(TN.n[3] * h.u + TN.n[6] * h.v + TN.n[0] * w,
              TN.n[4] * h.u + TN.n[7] * h.v + TN.n[1] * w,
              TN.n[5] * h.u + TN.n[8] * h.v + TN.n[2] * w);
    } else {
      ln = v3(TN.n[0], TN.n[1], TN.n[2]);  // triangle.rs:95-97
    }
    normal = vnormalize(ln);  // the triangle's own (identity) level of intersect.rs:79-92
  }
  normal = normal_up_groups(S, pr.parent_group, normal);
  V3 eyev = vneg(ray.d);
  pr.inside = vdot(normal, eyev) < 0.0;
  if (pr.inside) normal = vmul(normal, -1.0);
  pr.normal = normal;
  pr.over = vadd(point, vmul(normal, EPS));
  pr.under = vsub(point, vmul(normal, EPS));
  if (point_out) *point_out = point;
  return pr;
}

// pattern colour at the hit: pattern.inv * object.world_to_object(over_point) (world.rs:50,
// light.rs:34-35)
__device__ inline C3 hit_pattern_color(const DScene& S, const DMaterial& M, const Prep& pr) {
  C3 pc = {M.a[0], M.a[1], M.a[2]};
  if (M.pattern_kind != 0) {
    V3 op = pr.over;
    if (pr.parent_group >= 0) op = point_to_group(S, pr.parent_group, op);
    if (pr.leaf_prim >= 0) op = xf_point(S.prims[pr.leaf_prim].inv, op);
    pc = pattern_color(M, xf_point(M.pinv, op));
  }
  return pc;
}

// the shadow ray of is_shadowed (world.rs:116-125)
__device__ __forceinline__ void shadow_ray(const DLight& L, V3 over, Ray& sr, double& distance) {
  V3 lpos = v3(L.pos[0], L.pos[1], L.pos[2]);
  V3 v = vsub(lpos, over);
  distance = vmag(v);
  sr = {over, v3(v.x / distance, v.y / distance, v.z / distance)};
}

// lighting() of light.rs:23-85 for one light, added to `surface` (world.rs:44)
// The two values add_light can add for one light, both evaluated before the shadow query: the
// same expressions, so whichever the query selects is the same bits (a shadowed light adds
// ambient + 0 + 0).
__device__ inline void light_terms(const DMaterial& M, const DLight& L, C3 pc, V3 lightv, V3 normal, V3 eyev,
                                   C3& lit, C3& dark) {
  C3 eff = {pc.r * L.intensity[0], pc.g * L.intensity[1], pc.b * L.intensity[2]};
  C3 ambient = {eff.r * M.ambient, eff.g * M.ambient, eff.b * M.ambient};
  double ldn = vdot(lightv, normal);
  C3 diffuse = {0.0, 0.0, 0.0}, specular = {0.0, 0.0, 0.0};
  if (!(ldn < 0.0)) {
    diffuse = {eff.r * M.diffuse * ldn, eff.g * M.diffuse * ldn, eff.b * M.diffuse * ldn};
    V3 rv = vreflect(vneg(lightv), normal);
    double rde = vdot(rv, eyev);
    if (!(rde <= 0.0)) {
      double factor = pow_specular(rde, M.shininess);
      specular = {L.intensity[0] * M.specular * factor, L.intensity[1] * M.specular * factor,
                  L.intensity[2] * M.specular * factor};
    }
  }
  lit = {ambient.r + diffuse.r + specular.r, ambient.g + diffuse.g + specular.g, ambient.b + diffuse.b + specular.b};
  dark = {ambient.r + 0.0 + 0.0, ambient.g + 0.0 + 0.0, ambient.b + 0.0 + 0.0};
}

__device__ inline void add_light(const DMaterial& M, const DLight& L, C3 pc, V3 lightv, V3 normal,
                                 V3 eyev, bool shadow, C3& surface) {
  C3 eff = {pc.r * L.intensity[0], pc.g * L.intensity[1], pc.b * L.intensity[2]};
  C3 ambient = {eff.r * M.ambient, eff.g * M.ambient, eff.b * M.ambient};
  double ldn = vdot(lightv, normal);
  C3 diffuse = {0.0, 0.0, 0.0}, specular = {0.0, 0.0, 0.0};
  if (!(shadow || ldn < 0.0)) {
    diffuse = {eff.r * M.diffuse * ldn, eff.g * M.diffuse * ldn, eff.b * M.diffuse * ldn};
    V3 rv = vreflect(vneg(lightv), normal);
    double rde = vdot(rv, eyev);
    if (!(rde <= 0.0)) {
#if RT_FAST_POW
      double factor = pow_specular(rde, M.shininess);
#else
      double factor = pow(rde, M.shininess);
#endif
      specular = {L.intensity[0] * M.specular * factor, L.intensity[1] * M.specular * factor,
                  L.intensity[2] * M.specular * factor};
    }
  }
  surface.r += ambient.r + diffuse.r + specular.r;
  surface.g += ambient.g + diffuse.g + specular.g;
  surface.b += ambient.b + diffuse.b + specular.b;
}

// material flags of shade_hit / reflected_color / refracted_color (world.rs:57-59, 79, 90)
__device__ __forceinline__ bool needs_n1_n2(const DMaterial& M) {
  return !feq(M.transparency, 0.0) || ((M.reflective > 0.) && (M.transparency > 0.));
}

// ray set-up of reflected_color / refracted_color and schlick (world.rs:78-165)
__device__ inline void finish_optics(const DMaterial& M, const Ray& ray, const Prep& pr, double n1,
                                     double n2, ShadeOut& o) {
  V3 eyev = vneg(ray.d);
  V3 normal = pr.normal;
  o.reflective = M.reflective;
  o.transparency = M.transparency;
  o.do_reflect = !feq(M.reflective, 0.0);  // world.rs:79
  o.reflect_ray = {pr.over, vreflect(ray.d, normal)};
  o.use_schlick = (M.reflective > 0.) && (M.transparency > 0.);  // world.rs:57-59
  o.do_refract = false;
  o.reflectance = 0.0;
  o.n1 = n1;
  o.n2 = n2;
  if (!feq(M.transparency, 0.0)) {  // world.rs:90-112
    double n_ratio = n1 / n2;
    double cos_i = vdot(eyev, normal);
    double sin2_t = (n_ratio * n_ratio) * (1. - cos_i * cos_i);
    if (!(sin2_t > 1.)) {
      double cos_t = sqrt(1. - sin2_t);
      V3 dir = vsub(vmul(normal, n_ratio * cos_i - cos_t), vmul(eyev, n_ratio));
      o.refract_ray = {pr.under, dir};
      o.do_refract = true;
    }
  }
  if (o.use_schlick) {  // world.rs:151-165
    double c = vdot(eyev, normal);
    bool total = false;
    if (n1 > n2) {
      double n = n1 / n2;
      double sin2_t = n * n * (1. - c * c);
      if (sin2_t > 1.)
        total = true;
      else
        c = sqrt(1. - sin2_t);
    }
    if (total) {
      o.reflectance = 1.;
    } else {
      double q = (n1 - n2) / (n1 + n2);
      double r0 = q * q;
      double x = 1. - c;
      double x2 = x * x;
      double x5 = x * (x2 * x2);  // powi(5) by repeated squaring
      o.reflectance = r0 + (1. - r0) * x5;
    }
  }
}

// prepare_computations + the light loop of shade_hit + the child-ray set-up, in one lane.
// children_traced: the call's reflected / refracted rays will be traced (remaining depth > 1).
// When they will not — color_at(ray, 0) is black whatever the ray — n1 / n2 only weigh black
// colours (black times a reflectance in [0, 1] is +0): the container query is left out, the
// call's value is the same bits.  With counters it runs all the same (total internal
// reflection decides whether a depth-0 child is counted) and stays out of the work counts.
template <bool STATS>
__device__ inline void shade_hit_local(const DScene& S, const Ray& ray, const HitRec& h,
                                       ShadeOut& o, Counters& cnt, TravStack& st, bool children_traced = true) {
  if (STATS) cnt.shade_calls++;
  V3 point;
  Prep pr = prepare_hit(S, ray, h, &point);
  const DMaterial& M = S.mats[pr.mat_index];
  V3 eyev = vneg(ray.d);
  o.point = point;
  o.over_point = pr.over;
  o.under_point = pr.under;
  o.eyev = eyev;
  o.normalv = pr.normal;
  o.reflectv = vreflect(ray.d, pr.normal);
  o.inside = pr.inside;
  C3 surface = {0.0, 0.0, 0.0};
  C3 pc = hit_pattern_color(S, M, pr);
  for (int li = 0; li < S.n_lights; li++) {
    const DLight& L = S.lights[li];
    Ray sr;
    double distance;
    shadow_ray(L, pr.over, sr, distance);
    if (STATS) cnt.rays_shadow++;
    bool shadow = trace_any<STATS>(S, sr, distance, cnt, st);
    // lightv = normalize(light.position - position) is the shadow ray's direction
    add_light(M, L, pc, sr.d, pr.normal, eyev, shadow, surface);
  }
  o.surface = surface;
  double n1 = 1.0, n2 = 1.0;
  if (needs_n1_n2(M) && (STATS || children_traced)) {
    Counters uncounted{};
    if (STATS && !children_traced) uncounted = cnt;
    containers_n1_n2<STATS>(S, ray, h, M.refractive_index, n1, n2, cnt, st);
    if (STATS && !children_traced) cnt = uncounted;
  }
  finish_optics(M, ray, pr, n1, n2, o);
}

// world.rs:57-64
__device__ __forceinline__ C3 combine(const C3& surface, const C3& reflected, const C3& refracted,
                                      bool use_schlick, double reflectance) {
  if (use_schlick) {
    double k = 1. - reflectance;
    return {surface.r + reflected.r * reflectance + refracted.r * k,
            surface.g + reflected.g * reflectance + refracted.g * k,
            surface.b + reflected.b * reflectance + refracted.b * k};
  }
  return {surface.r + reflected.r + refracted.r, surface.g + reflected.g + refracted.g,
          surface.b + reflected.b + refracted.b};
}

// One saved level of the color_at recursion (world.rs:67-114).
struct Frame {
  C3 surface, reflected;
  double reflective, transparency, reflectance;
  Ray refract_ray;
  uint8_t use_schlick, do_refract, state;  // state 1: waiting for the reflected child, 2: refracted
};

constexpr int MAX_DEPTH = 8;

// World::color_at(ray, depth) with the recursion unrolled onto an explicit stack.
template <bool STATS>
__device__ inline C3 color_at(const DScene& S, Ray ray, int depth, Counters& cnt, TravStack& st) {
  Frame frames[MAX_DEPTH];
  int level = 0;
  C3 ret = {0.0, 0.0, 0.0};
  if (depth > MAX_DEPTH) depth = MAX_DEPTH;
  int remaining = depth;  // depth argument of the current color_at call
  if (STATS) cnt.rays_primary++;
  while (true) {
    // ---- enter color_at(ray, remaining), remaining >= 1 or the top-level call
    HitRec h;
    bool found = trace_closest<STATS>(S, ray, h, cnt, st);
    bool returning = true;
    if (found && remaining > 0) {
      ShadeOut so;
      shade_hit_local<STATS>(S, ray, h, so, cnt, st, remaining > 1);
      {
        Frame& F = frames[level];
        F.surface = so.surface;
        F.reflected = {0.0, 0.0, 0.0};
        F.reflective = so.reflective;
        F.transparency = so.transparency;
        F.reflectance = so.reflectance;
        F.use_schlick = so.use_schlick;
        F.do_refract = so.do_refract;
        F.refract_ray = so.refract_ray;
        int child = remaining - 1;
        if (so.do_reflect && child > 0) {
          F.state = 1;
          ray = so.reflect_ray;
          level++;
          remaining = child;
          if (STATS) cnt.rays_reflect++;
          continue;
        }
        if (so.do_reflect && STATS) cnt.depth0_skipped++;
        if (so.do_refract && child > 0) {
          F.state = 2;
          ray = so.refract_ray;
          level++;
          remaining = child;
          if (STATS) cnt.rays_refract++;
          continue;
        }
        if (so.do_refract && STATS) cnt.depth0_skipped++;
        C3 zero = {0.0, 0.0, 0.0};
        ret = combine(F.surface, zero, zero, F.use_schlick, F.reflectance);
      }
    } else {
      ret = {0.0, 0.0, 0.0};  // world.rs:74 (no hit) or world.rs:37-39 (depth 0)
    }
    // ---- return `ret` to the callers
    while (returning) {
      if (level == 0) return ret;
      level--;
      remaining++;
      Frame& F = frames[level];
      if (F.state == 1) {
        // reflected_color: color * reflective (world.rs:85)
        F.reflected = {ret.r * F.reflective, ret.g * F.reflective, ret.b * F.reflective};
        int child = remaining - 1;
        if (F.do_refract && child > 0) {
          F.state = 2;
          ray = F.refract_ray;
          level++;
          remaining = child;
          if (STATS) cnt.rays_refract++;
          returning = false;
          break;
        }
        if (F.do_refract && STATS) cnt.depth0_skipped++;
        C3 zero = {0.0, 0.0, 0.0};
        ret = combine(F.surface, F.reflected, zero, F.use_schlick, F.reflectance);
      } else {
        // refracted_color: color * transparency (world.rs:112)
        C3 refracted = {ret.r * F.transparency, ret.g * F.transparency, ret.b * F.transparency};
        ret = combine(F.surface, F.reflected, refracted, F.use_schlick, F.reflectance);
      }
    }
  }
}

// camera.rs:64-80 with the inverse precomputed
__device__ __forceinline__ Ray ray_for_pixel(const DCamera& cam, int px, int py) {
  double x_offset = ((double)px + 0.5) * cam.pixel_size;
  double y_offset = ((double)py + 0.5) * cam.pixel_size;
  double world_x = cam.half_width - x_offset;
  double world_y = cam.half_height - y_offset;
  V3 pixel = xf_point(cam.inv, v3(world_x, world_y, -1.0));
  V3 origin = v3(cam.inv[3], cam.inv[7], cam.inv[11]);  // inv * point(0,0,0)
  V3 direction = vnormalize(vsub(pixel, origin));
  return {origin, direction};
}

// color.rs:148-155: `(val * 255.0) as u8` saturates, truncates, maps NaN to 0
__device__ __forceinline__ uint8_t quantise(double val) {
  if (val < 0.0) return 0;
  double s = val * 255.0;
  if (s != s) return 0;
  if (s >= 255.0) return 255;
  if (s <= 0.0) return 0;
  return (uint8_t)(int)s;
}

}  // namespace rtd
