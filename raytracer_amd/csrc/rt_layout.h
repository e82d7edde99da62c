// rt_layout.h — records of the committed scene as they lie in HBM (plain structs shared
// by the host-side builders and the device code; no HIP dependency).
#pragma once
#include <stdint.h>

namespace rtd {

// ---------------------------------------------------------------------------------------
// Scene records in HBM.  Per-kind tables ("SoA over shape kinds"); each record is a
// 16-byte-aligned struct so a lane fetches it with dwordx4 loads.
struct alignas(16) DGroup {
  double inv[12];  // rows 0..2 of the inverse transform (row 3 is 0,0,0,1)
  int32_t parent;
  int32_t nchain;    // number of transforms from the root down to this group
  int32_t chain[6];  // group indices, root first, this group last
};
struct alignas(16) DPrim {
  double inv[12];
  double minimum, maximum;
  int32_t kind, material, parent, closed;
  uint32_t order;  // position in the reference's depth-first leaf order
  uint32_t pad[3];
};
struct alignas(16) DMesh {
  double lo[3], hi[3];  // padded bounds of the whole mesh (object space)
  double cell[3];       // size of one step of the node boxes' 16-bit grid: plane = lo + q*cell
  double scale;         // largest |coordinate| of the mesh
  int32_t group, material;
  uint32_t order_base;  // leaf order of triangle 0
  int32_t root;         // index of the root node in the node table
  int32_t tri_base;     // first record in the triangle table
  int32_t ntris;
  int32_t n_nodes;      // nodes of this mesh; the first ones are the top of the tree, breadth first
  int32_t pad;
  // the same bounds in world space (through the mesh's groups), padded; infinite when unknown:
  // a query whose ray misses them need not be taken into the mesh's space at all
  double wlo[3], whi[3];
};
// One triangle as the intersection test reads it: p1, e1 = p2-p1, e2 = p3-p1
// (triangle.rs:49-50).  80 bytes.
struct alignas(16) DTri {
  double p1[3], e1[3], e2[3];
  uint32_t orig;    // index in face order (for the leaf order and tie-breaks)
  uint32_t smooth;  // 1: interpolate vertex normals, 0: flat normal
};
// The same triangle in f32 for the fast arithmetic mode.  48 bytes.  p1 is kept to about 2^-31
// of its magnitude: p1_lo holds the upper 16 bits of the three f32 residuals p1 - f32(p1)
// (x | y << 16, z), so that (origin - p1) can be formed without the f32 rounding of either.
struct alignas(16) DTriF {
  float p1[3], e1[3], e2[3];
  uint32_t orig, p1_lo_xy, p1_lo_z;
};
// Normals, fetched only for the final hit: n1,n2,n3 (smooth) or the flat normal in n[0..2].
struct alignas(8) DTriN {
  double n[9];
};
// Binary BVH node holding the boxes of BOTH children, so one 32-byte fetch decides both.
// Box planes are 16-bit steps of a grid laid over the mesh's padded bounds (DMesh lo/cell),
// rounded outward: q[a] = lo_q | hi_q << 16 for child 0 on axis a, q[3+a] for child 1.
// Child c >= 0: inner node index; c == -1: no child; c <= -2: leaf, code = -2 - c =
// (first << LEAF_BITS) | count with `first` relative to the mesh's tri_base.
#ifndef RT_LEAF_BITS
#define RT_LEAF_BITS 4
#endif
constexpr int LEAF_BITS = RT_LEAF_BITS;            // bits of a leaf code that hold the count
constexpr int LEAF_MAX = (1 << LEAF_BITS) - 1;     // most triangles one leaf can hold
struct alignas(32) DNode {
  uint32_t q[6];
  int32_t c0, c1;
};
// The node the kernels walk: four children (a binary node with its inner children folded
// in), so a ray takes half as many dependent steps.  q[3*k + a] = lo_q | hi_q << 16 of child
// k on axis a (same grid as DNode); c[k] as in DNode (-1: no child).  64 bytes, four
// 16-byte loads.
struct alignas(64) DNode4 {
  uint32_t q[12];
  int32_t c[4];
};
struct alignas(16) DMaterial {
  double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
  double a[3], b[3];
  double pinv[12];
  int32_t pattern_kind;
  int32_t pad[3];
};
struct alignas(16) DLight {
  double intensity[3];
  double pos[3];
};
// World-space box of one closed-form shape (padded; infinite for planes and uncapped
// cylinders/cones): rays are tested against it before the shape's own transform and solve.
struct alignas(16) DPrimBox {
  double lo[3], hi[3];
};
struct DScene {
  const DGroup* groups;
  const DPrim* prims;
  const DPrimBox* prim_boxes;
  const DMesh* meshes;
  const DTri* tris;
  const DTriF* trisf;  // same order as tris
  const DTriN* trin;
  const DNode4* nodes;
  const DMaterial* mats;
  const DLight* lights;
  int32_t n_groups, n_prims, n_meshes, n_lights;
};
struct DCamera {
  double inv[12];
  double half_width, half_height, pixel_size;
  int32_t hsize, vsize;
};

struct Counters {
  unsigned long long rays_primary, rays_shadow, rays_reflect, rays_refract, depth0_skipped,
      pred_queries, nodes_visited, tri_tests, prim_tests, inst_entries, shade_calls,
      node_wave_steps, tri_wave_steps,  // iterations of the node / triangle loop, per wave
      fast_candidates, fast_exact;      // fast mode: f32 candidates handed to their owners / f64 evaluations
  unsigned long long max_nodes;  // not a sum: the longest single traversal (nodes)
};
constexpr int N_COUNTERS = 15;  // summed counters; max_nodes is stored after them

}  // namespace rtd
