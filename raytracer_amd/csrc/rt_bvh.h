// rt_bvh.h — host-side BVH construction for one triangle mesh (object space).
// The hierarchy is a pure culling structure: it decides which triangles a ray tests,
// never what a test returns.  Boxes are padded so that rounding in the slab test can
// not drop a triangle the exhaustive loop of the reference (shapes/group.rs:64-71)
// would have hit.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "rt_layout.h"

namespace rtb {

struct BuildResult {
  std::vector<rtd::DNode> nodes;  // nodes[0] is the root; child indices are local to this vector
  std::vector<uint32_t> order;    // order[k] = face index of the k-th triangle record
  int max_depth = 0;
  double lo[3], hi[3];  // padded bounds of the whole mesh = origin and far end of the node grid
  double cell[3];       // step of the 16-bit grid the node boxes are stored on
  double scale = 0.0;   // largest |coordinate|
};

// p1,p2,p3: n x 3 doubles.  max_leaf: triangles per leaf (<= rtd::LEAF_MAX).
void build_sah(const double* p1, const double* p2, const double* p3, int64_t n, int max_leaf,
               BuildResult& out);

// A mesh's hierarchy built on the current HIP device (Morton codes, radix sort, Karras' radix
// tree, bottom-up refit, fold into four-wide nodes breadth first) and left there: the wide nodes
// with their child indices already offset by `node_base`, the triangle order, and the device
// copies of the vertices that build_tri_records_device() turns into the triangle records.  Only
// the mesh bounds and a few counts reach the host.  The caller owns the device pointers
// (free_device_build).
struct DeviceBuild {
  rtd::DNode4* d_nodes = nullptr;
  int n_nodes = 0;
  int stack_need = 0;  // most stack entries a walk of the wide tree can hold
  uint32_t* d_order = nullptr;
  double *d_p1 = nullptr, *d_p2 = nullptr, *d_p3 = nullptr;
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, cell[3] = {0, 0, 0};
  double scale = 0.0;
};
// Returns 0, or 1 with `err` set.
int build_lbvh_resident(const double* p1, const double* p2, const double* p3, int64_t n, int max_leaf,
                        int node_base, DeviceBuild& out, std::string& err);
// Writes the n triangle records of the mesh (leaf order) straight into the scene's device tables.
// n1,n2,n3,smooth: host arrays, or all null for a mesh of flat triangles.
int build_tri_records_device(const DeviceBuild& b, int64_t n, const double* n1, const double* n2,
                             const double* n3, const uint8_t* smooth, rtd::DTri* d_tris, rtd::DTriF* d_trisf,
                             rtd::DTriN* d_trin, std::string& err);
void free_device_build(DeviceBuild& b);

// Folds a binary tree (nodes[0] = root) into four-wide nodes: the inner children of a binary
// node are opened, largest surface first (cell = the mesh's grid steps), until four slots are
// taken.  Returns the largest number of stack entries a walk of the wide tree can hold.
int collapse_to_wide(const std::vector<rtd::DNode>& nodes, const double* cell, std::vector<rtd::DNode4>& out);

// Moves the top of the tree to the front of `nodes`: the first `top` entries become the
// nodes nearest the root in breadth-first order (the root stays entry 0); the others keep
// their relative order.  Kernels may stage those first entries in LDS.
void reorder_top_breadth_first(std::vector<rtd::DNode4>& nodes, int top);

// Relative padding applied to every box, times the largest |coordinate| of the mesh.
constexpr double BOX_PAD_REL = 1e-7;

}  // namespace rtb
