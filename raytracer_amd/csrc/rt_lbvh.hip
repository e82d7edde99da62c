// rt_lbvh.hip — BVH construction on the GPU for one triangle mesh (object space).
//
// Replaces the bounding-box tree of shapes/group.rs (the reference has none for its flat
// child list, group.rs:64-71) with a linear BVH:
//   triangle boxes + mesh bounds  ->  63-bit Morton code of each centroid (21 bits per axis)
//   ->  radix sort of (code, face)  ->  Karras' radix tree over the sorted codes (equal codes are
//       told apart by their position)
//   ->  the subtrees of at most 64 triangles built again by the surface-area heuristic
//   ->  bottom-up refit  ->  collapse of subtrees with <= max_leaf triangles into leaves
//   ->  binary nodes (both children's boxes on the mesh's 16-bit grid, rounded outward)
//   ->  four-wide nodes, level by level from the root (so the top of the tree comes first,
//       breadth first), with the stack a walk can need
//   ->  the triangle records (rt_layout.h) in leaf order.
// Nothing of the tree or of the triangle order returns to the host: only the mesh bounds (56
// bytes) and a few counts do.
// Like the host builder it produces a pure culling structure; which triangle a ray hits is
// decided by the exact test in rt_device.h.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <limits>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_bvh.h"

namespace rtb {
namespace {

#define LB_TRY(expr)                                                        \
  do {                                                                      \
    hipError_t e_ = (expr);                                                 \
    if (e_ != hipSuccess) {                                                 \
      err = std::string(#expr) + ": " + hipGetErrorString(e_);              \
      return 1;                                                             \
    }                                                                       \
  } while (0)

struct Box6 {
  double lo[3], hi[3];
};
// reduction cell: order-preserving integer images of doubles, so integer atomics give min/max
struct Bounds {
  unsigned long long lo[3], hi[3], scale;
};
__host__ __device__ inline unsigned long long enc(double x) {
  unsigned long long b;
  memcpy(&b, &x, 8);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double dec(unsigned long long e) {
  unsigned long long b = (e >> 63) ? (e & 0x7fffffffffffffffull) : ~e;
  double x;
  memcpy(&x, &b, 8);
  return x;
}

__device__ inline double wave_min(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

__global__ void k_tri_bounds(const double* p1, const double* p2, const double* p3, int n, Box6* boxes,
                             Bounds* bounds) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  const double inf = __builtin_huge_val();
  double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf}, sc = 0.0;
  if (t < n) {
    Box6 b;
    for (int a = 0; a < 3; a++) {
      double x = p1[3 * t + a], y = p2[3 * t + a], z = p3[3 * t + a];
      b.lo[a] = fmin(fmin(x, y), z);
      b.hi[a] = fmax(fmax(x, y), z);
      lo[a] = b.lo[a];
      hi[a] = b.hi[a];
      sc = fmax(sc, fmax(fabs(b.lo[a]), fabs(b.hi[a])));
    }
    boxes[t] = b;
  }
  for (int a = 0; a < 3; a++) {
    lo[a] = wave_min(lo[a]);
    hi[a] = wave_max(hi[a]);
  }
  sc = wave_max(sc);
  if ((threadIdx.x & 63) == 0) {
    for (int a = 0; a < 3; a++) {
      atomicMin(&bounds->lo[a], enc(lo[a]));
      atomicMax(&bounds->hi[a], enc(hi[a]));
    }
    atomicMax(&bounds->scale, enc(sc));
  }
}

__device__ inline unsigned long long expand21(unsigned long long x) {  // 21 bits -> every third bit
  x &= 0x1fffffull;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

__global__ void k_morton(const Box6* boxes, int n, const Bounds* bounds, unsigned long long* keys, uint32_t* faces) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  unsigned long long q[3];
  for (int a = 0; a < 3; a++) {
    double lo = dec(bounds->lo[a]), hi = dec(bounds->hi[a]);
    double c = 0.5 * (boxes[t].lo[a] + boxes[t].hi[a]);
    double ext = hi - lo;
    double x = ext > 0.0 ? (c - lo) / ext : 0.0;
    long long v = (long long)(x * 2097152.0);
    q[a] = (unsigned long long)min(max(v, 0ll), 2097151ll);
  }
  keys[t] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
  faces[t] = (uint32_t)t;
}

constexpr int LEAF_BIT = 0x40000000;

// Karras 2012: internal node i of the radix tree over n sorted keys; equal keys are ordered by
// their position (64 + the common prefix of the positions), which keeps the keys distinct.
__global__ void k_radix_tree(const unsigned long long* keys, int n, int* child_l, int* child_r,
                             int* range_first, int* range_last, int* parent_inner, int* parent_leaf) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  auto delta = [&](int a, int b) -> int {
    if (b < 0 || b >= n) return -1;
    const unsigned long long x = keys[a] ^ keys[b];
    return x ? __clzll((long long)x) : 64 + __clz(a ^ b);
  };
  int d = (delta(i, i + 1) - delta(i, i - 1)) >= 0 ? 1 : -1;
  int dmin = delta(i, i - d);
  int lmax = 2;
  while (delta(i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(i, i + (l + t) * d) > dmin) l += t;
  int j = i + l * d;
  int dnode = delta(i, j);
  int s = 0;
  int t = l;
  do {
    t = (t + 1) / 2;
    if (delta(i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  int gamma = i + s * d + min(d, 0);
  int first = min(i, j), last = max(i, j);
  range_first[i] = first;
  range_last[i] = last;
  if (first == gamma) {
    child_l[i] = gamma | LEAF_BIT;
    parent_leaf[gamma] = i;
  } else {
    child_l[i] = gamma;
    parent_inner[gamma] = i;
  }
  if (last == gamma + 1) {
    child_r[i] = (gamma + 1) | LEAF_BIT;
    parent_leaf[gamma + 1] = i;
  } else {
    child_r[i] = gamma + 1;
    parent_inner[gamma + 1] = i;
  }
  if (i == 0) parent_inner[0] = -1;
}

// ---- surface-area rebuild of the small subtrees ---------------------------------------------------
// The radix tree splits where the Morton codes' bits change: spatial medians on a fixed cycle of
// axes.  High up that is a fair tree; near the leaves, where a few dozen triangles are dealt out,
// it is not.  Every maximal subtree of at most TREELET triangles is therefore built again by the
// surface-area heuristic, top down, one wave per subtree with one triangle per lane: for each
// axis the lanes rank their centroids, every lane prices the split behind its own rank
// (area of the left box x its count + the same on the right), the cheapest of the three axes
// wins, and the lanes trade places so that the two sides are contiguous again.  The subtree
// keeps its root, its range of positions in the order array and the inner-node indices the
// radix tree gave it: an inner left child is numbered by the last position of its range, an
// inner right child by the first of its, which no two nodes of one subtree can share.
constexpr int TREELET = 64;

// the maximal subtrees of at least `least` and at most `most` triangles
__global__ void k_find_treelets(const int* range_first, const int* range_last, const int* parent_inner,
                                int n_inner, int least, int most, int* list, int* n_list) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_inner) return;
  const int count = range_last[i] - range_first[i] + 1;
  if (count > most || count < least) return;
  const int p = parent_inner[i];
  if (p >= 0 && range_last[p] - range_first[p] + 1 <= most) return;  // part of a larger one
  list[atomicAdd(n_list, 1)] = i;
}

__device__ inline float wave_min_f32(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ inline float lane_value(float v, int lane) {  // `lane` is the same for the whole wave
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane));
}
__device__ inline float send_to(float v, int dest_lane) {  // every lane receives what was sent to it
  return __uint_as_float((unsigned)__builtin_amdgcn_ds_permute(dest_lane * 4, (int)__float_as_uint(v)));
}

__global__ void __launch_bounds__(64)
k_rebuild_treelets(const int* list, int n_list, uint32_t* faces, const Box6* tri_boxes, int* child_l, int* child_r,
                   int* range_first, int* range_last, int* parent_inner, int* parent_leaf) {
  if ((int)blockIdx.x >= n_list) return;
  const int root = list[blockIdx.x];
  const int F = range_first[root], m = range_last[root] - F + 1;  // 3 <= m <= 64
  const int lane = threadIdx.x;
  __shared__ int todo_first[TREELET], todo_last[TREELET], todo_node[TREELET];
  const float inf = __builtin_huge_valf();
  float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
  unsigned face = 0;
  if (lane < m) {
    face = faces[F + lane];
    const Box6 b = tri_boxes[face];
    for (int a = 0; a < 3; a++) {
      lo[a] = (float)b.lo[a];
      hi[a] = (float)b.hi[a];
    }
  }
  int sp = 0;
  if (lane == 0) {
    todo_first[0] = 0;
    todo_last[0] = m - 1;
    todo_node[0] = root;
  }
  sp = 1;
  __syncthreads();
  while (sp > 0) {  // the same for every lane
    sp--;
    const int f = __builtin_amdgcn_readfirstlane(todo_first[sp]), l = __builtin_amdgcn_readfirstlane(todo_last[sp]),
              node = __builtin_amdgcn_readfirstlane(todo_node[sp]);
    __syncthreads();
    const int count = l - f + 1;
    const bool in = lane >= f && lane <= l;
    float best_cost = inf;
    int best_left = count / 2;   // triangles that go left; without a finite price: half, as they stand
    int best_rank = lane - f;
    for (int axis = 0; axis < 3; axis++) {
      float key = lo[axis] + hi[axis];
      if (!(key == key)) key = 0.0f;  // the ranks must be a permutation whatever the input
      int rank = 0;
      for (int j = f; j <= l; j++) {
        const float kj = lane_value(key, j);
        rank += (kj < key || (kj == key && j < lane)) ? 1 : 0;
      }
      // this lane prices the split that puts the ranks 0..rank on the left
      float ll[3] = {inf, inf, inf}, lh[3] = {-inf, -inf, -inf}, rl[3] = {inf, inf, inf}, rh[3] = {-inf, -inf, -inf};
      for (int j = f; j <= l; j++) {
        const bool left = __builtin_amdgcn_readlane(rank, j) <= rank;
        for (int a = 0; a < 3; a++) {
          const float bl = lane_value(lo[a], j), bh = lane_value(hi[a], j);
          ll[a] = left ? fminf(ll[a], bl) : ll[a];
          lh[a] = left ? fmaxf(lh[a], bh) : lh[a];
          rl[a] = left ? rl[a] : fminf(rl[a], bl);
          rh[a] = left ? rh[a] : fmaxf(rh[a], bh);
        }
      }
      const float lx = lh[0] - ll[0], ly = lh[1] - ll[1], lz = lh[2] - ll[2];
      const float rx = rh[0] - rl[0], ry = rh[1] - rl[1], rz = rh[2] - rl[2];
      float cost = (lx * ly + ly * lz + lz * lx) * (float)(rank + 1) +
                   (rx * ry + ry * rz + rz * rx) * (float)(count - rank - 1);
      if (!in || rank >= count - 1 || !(cost == cost)) cost = inf;
      const float cheapest = wave_min_f32(cost);
      if (cheapest < best_cost) {  // the same for every lane
        const int from = __ffsll((long long)__ballot(cost == cheapest)) - 1;
        best_cost = cheapest;
        best_left = __builtin_amdgcn_readlane(rank, from) + 1;
        best_rank = rank;
      }
    }
    // the lanes of this range trade places: rank order along the chosen axis
    const int dest = in ? f + best_rank : lane;
    for (int a = 0; a < 3; a++) {
      lo[a] = send_to(lo[a], dest);
      hi[a] = send_to(hi[a], dest);
    }
    face = (unsigned)__builtin_amdgcn_ds_permute(dest * 4, (int)face);
    const int left_last = f + best_left - 1;
    const bool left_leaf = best_left == 1, right_leaf = count - best_left == 1;
    if (lane == 0) {
      range_first[node] = F + f;
      range_last[node] = F + l;
      if (left_leaf) {
        child_l[node] = (F + f) | LEAF_BIT;
        parent_leaf[F + f] = node;
      } else {
        child_l[node] = F + left_last;
        parent_inner[F + left_last] = node;
        todo_first[sp] = f;
        todo_last[sp] = left_last;
        todo_node[sp] = F + left_last;
      }
      const int at = sp + (left_leaf ? 0 : 1);
      if (right_leaf) {
        child_r[node] = (F + l) | LEAF_BIT;
        parent_leaf[F + l] = node;
      } else {
        child_r[node] = F + left_last + 1;
        parent_inner[F + left_last + 1] = node;
        todo_first[at] = left_last + 1;
        todo_last[at] = l;
        todo_node[at] = F + left_last + 1;
      }
    }
    sp += (left_leaf ? 0 : 1) + (right_leaf ? 0 : 1);
    __syncthreads();
  }
  if (lane < m) faces[F + lane] = face;
}


// ---- surface-area rebuild of the large subtrees ---------------------------------------------------
// Above the 64-triangle subtrees the radix tree's median splits still cost a few per cent of
// frame time on meshes of a few ten thousand triangles (the pawn).  Every maximal subtree of at
// most BIG_TREELET triangles is therefore split again from its root down by the binned
// surface-area heuristic (16 bins per axis on the centroids, the host builder's rule for large
// ranges), one workgroup per subtree with the subtree's slice of the order array in LDS, until
// the ranges have at most TREELET triangles: those are left, as ranges with a root and nothing
// below it, to k_rebuild_treelets.  Node numbering as there: an inner left child takes the last
// position of its range, an inner right child the first of its.  The inner nodes the radix tree
// had inside the subtree are marked empty first (a range of one position), so that no stale
// node is ever taken for a subtree's root.
constexpr int BIG_TREELET = 8192;
constexpr int BIG_THREADS = 1024;
constexpr int BIG_BINS = 16;

__device__ inline unsigned f32_ordered(float x) {  // order-preserving image, for integer min / max
  const unsigned b = __float_as_uint(x);
  return (b >> 31) ? ~b : (b | 0x80000000u);
}
__device__ inline float f32_from_ordered(unsigned e) {
  return __uint_as_float((e >> 31) ? (e & 0x7fffffffu) : ~e);
}

__global__ void __launch_bounds__(BIG_THREADS)
k_rebuild_big_treelets(const int* list, int n_list, uint32_t* faces, const Box6* tri_boxes, int* child_l, int* child_r,
                       int* range_first, int* range_last, int* parent_inner, int* parent_leaf, int n_inner) {
  if ((int)blockIdx.x >= n_list) return;
  const int root = list[blockIdx.x];
  const int F = range_first[root], m = range_last[root] - F + 1;  // TREELET < m <= BIG_TREELET
  const int tid = threadIdx.x;
  __shared__ uint32_t s_face[BIG_TREELET], s_tmp[BIG_TREELET];
  __shared__ unsigned short s_bin[BIG_TREELET];  // the three axes' bins of a triangle, 4 bits each
  __shared__ unsigned s_cmin[3], s_cmax[3];
  __shared__ unsigned s_blo[3][BIG_BINS][3], s_bhi[3][BIG_BINS][3];
  __shared__ int s_bcount[3][BIG_BINS];
  __shared__ float s_cost[3 * BIG_BINS];
  __shared__ int s_scan[BIG_THREADS];
  __shared__ int todo_first[128], todo_last[128], todo_node[128];
  __shared__ int s_sp, s_axis, s_split;
  for (int j = tid; j < m; j += BIG_THREADS) s_face[j] = faces[F + j];
  for (int j = F + 1 + tid; j <= F + m - 2; j += BIG_THREADS)  // the radix tree's nodes inside: empty
    if (j < n_inner && j != root) {
      range_first[j] = j;
      range_last[j] = j;
    }
  if (tid == 0) {
    todo_first[0] = 0;
    todo_last[0] = m - 1;
    todo_node[0] = root;
    s_sp = 1;
  }
  __syncthreads();
  const float inf = __builtin_huge_valf();
  while (true) {
    __syncthreads();
    const int sp = s_sp;
    if (sp == 0) break;
    const int f = todo_first[sp - 1], l = todo_last[sp - 1], node = todo_node[sp - 1];
    const int count = l - f + 1;
    __syncthreads();
    // centroid bounds of the range
    if (tid < 3) {
      s_cmin[tid] = 0xffffffffu;
      s_cmax[tid] = 0u;
    }
    for (int k = tid; k < 3 * BIG_BINS * 3; k += BIG_THREADS) {
      (&s_blo[0][0][0])[k] = 0xffffffffu;
      (&s_bhi[0][0][0])[k] = 0u;
    }
    for (int k = tid; k < 3 * BIG_BINS; k += BIG_THREADS) (&s_bcount[0][0])[k] = 0;
    __syncthreads();
    for (int j = f + tid; j <= l; j += BIG_THREADS) {
      const Box6 b = tri_boxes[s_face[j]];
      for (int a = 0; a < 3; a++) {
        const unsigned c = f32_ordered((float)(b.lo[a] + b.hi[a]));
        atomicMin(&s_cmin[a], c);
        atomicMax(&s_cmax[a], c);
      }
    }
    __syncthreads();
    float cmin[3], scale[3];
    for (int a = 0; a < 3; a++) {
      cmin[a] = f32_from_ordered(s_cmin[a]);
      const float ext = f32_from_ordered(s_cmax[a]) - cmin[a];
      scale[a] = (ext > 0.0f && ext < inf) ? (float)BIG_BINS / ext : 0.0f;
    }
    // bins: box and count per (axis, bin)
    for (int j = f + tid; j <= l; j += BIG_THREADS) {
      const Box6 b = tri_boxes[s_face[j]];
      unsigned packed = 0;
      for (int a = 0; a < 3; a++) {
        const float c = (float)(b.lo[a] + b.hi[a]);
        int bin = (int)((c - cmin[a]) * scale[a]);
        bin = min(max(bin, 0), BIG_BINS - 1);
        packed |= (unsigned)bin << (4 * a);
        atomicAdd(&s_bcount[a][bin], 1);
        for (int d = 0; d < 3; d++) {
          atomicMin(&s_blo[a][bin][d], f32_ordered((float)b.lo[d]));
          atomicMax(&s_bhi[a][bin][d], f32_ordered((float)b.hi[d]));
        }
      }
      s_bin[j] = (unsigned short)packed;
    }
    __syncthreads();
    // one thread per (axis, split behind bin k): surface-area price of the two sides
    if (tid < 3 * BIG_BINS) {
      const int a = tid / BIG_BINS, k = tid % BIG_BINS;
      float cost = inf;
      if (k < BIG_BINS - 1) {
        float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
        int nl = 0, nr = 0;
        for (int q = 0; q <= k; q++)
          if (s_bcount[a][q]) {
            nl += s_bcount[a][q];
            for (int d = 0; d < 3; d++) {
              lo[d] = fminf(lo[d], f32_from_ordered(s_blo[a][q][d]));
              hi[d] = fmaxf(hi[d], f32_from_ordered(s_bhi[a][q][d]));
            }
          }
        const float lx = hi[0] - lo[0], ly = hi[1] - lo[1], lz = hi[2] - lo[2];
        for (int d = 0; d < 3; d++) {
          lo[d] = inf;
          hi[d] = -inf;
        }
        for (int q = k + 1; q < BIG_BINS; q++)
          if (s_bcount[a][q]) {
            nr += s_bcount[a][q];
            for (int d = 0; d < 3; d++) {
              lo[d] = fminf(lo[d], f32_from_ordered(s_blo[a][q][d]));
              hi[d] = fmaxf(hi[d], f32_from_ordered(s_bhi[a][q][d]));
            }
          }
        const float rx = hi[0] - lo[0], ry = hi[1] - lo[1], rz = hi[2] - lo[2];
        if (nl > 0 && nr > 0) {
          cost = (lx * ly + ly * lz + lz * lx) * (float)nl + (rx * ry + ry * rz + rz * rx) * (float)nr;
          if (!(cost == cost)) cost = inf;
        }
      }
      s_cost[tid] = cost;
    }
    __syncthreads();
    if (tid == 0) {
      float best = inf;
      int at = -1;
      for (int k = 0; k < 3 * BIG_BINS; k++)
        if (s_cost[k] < best) {
          best = s_cost[k];
          at = k;
        }
      s_axis = at < 0 ? -1 : at / BIG_BINS;
      s_split = at < 0 ? 0 : at % BIG_BINS;
    }
    __syncthreads();
    const int axis = s_axis, split = s_split;
    int n_left = count / 2;  // no split has a finite price (all centroids in one bin): halves, as they stand
    if (axis >= 0) {
      // stable partition of the range: the triangles of bins <= split first
      n_left = 0;
      for (int q = 0; q <= split; q++) n_left += s_bcount[axis][q];
      // pass 2: positions, chunk by chunk of BIG_THREADS consecutive triangles (keeps their order)
      int done_l = 0, done_r = 0;
      for (int c0 = f; c0 <= l; c0 += BIG_THREADS) {
        const int j = c0 + tid;
        const bool valid = j <= l;
        const bool left = valid && (((s_bin[j] >> (4 * axis)) & 15) <= split);
        s_scan[tid] = left ? 1 : 0;
        __syncthreads();
        for (int off = 1; off < BIG_THREADS; off <<= 1) {  // inclusive scan
          const int v = tid >= off ? s_scan[tid - off] : 0;
          __syncthreads();
          s_scan[tid] += v;
          __syncthreads();
        }
        const int incl = s_scan[tid];
        const int chunk_left = s_scan[BIG_THREADS - 1];
        const int chunk_n = min(BIG_THREADS, l - c0 + 1);
        if (valid) {
          const int pos = left ? (f + done_l + incl - 1) : (f + n_left + done_r + (tid - incl));
          s_tmp[pos] = s_face[j];
        }
        done_l += chunk_left;
        done_r += chunk_n - chunk_left;
        __syncthreads();
      }
      for (int j = f + tid; j <= l; j += BIG_THREADS) s_face[j] = s_tmp[j];
      __syncthreads();
    }
    // the node and its two children
    if (tid == 0) {
      int spn = sp - 1;
      range_first[node] = F + f;
      range_last[node] = F + l;
      const int left_last = f + n_left - 1;
      const int sides[2][2] = {{f, left_last}, {left_last + 1, l}};
      // the larger side is queued first, so the smaller is split first: the list of waiting ranges
      // never grows beyond the logarithm of the subtree's size
      const int first_side = (n_left >= count - n_left) ? 0 : 1;
      for (int turn = 0; turn < 2; turn++) {
        const int w = turn == 0 ? first_side : 1 - first_side;
        const int a = sides[w][0], b = sides[w][1], cnt = b - a + 1;
        int* slot = w == 0 ? child_l : child_r;
        if (cnt == 1) {
          slot[node] = (F + a) | LEAF_BIT;
          parent_leaf[F + a] = node;
          continue;
        }
        const int child = w == 0 ? F + b : F + a;  // last position of a left child, first of a right one
        slot[node] = child;
        parent_inner[child] = node;
        range_first[child] = F + a;
        range_last[child] = F + b;
        if (cnt == 2) {  // the only way to split two triangles
          child_l[child] = (F + a) | LEAF_BIT;
          child_r[child] = (F + b) | LEAF_BIT;
          parent_leaf[F + a] = child;
          parent_leaf[F + b] = child;
        } else if (cnt > TREELET) {
          todo_first[spn] = a;
          todo_last[spn] = b;
          todo_node[spn] = child;
          spn++;
        }  // 3 .. TREELET triangles: a root with its range, rebuilt below it by k_rebuild_treelets
      }
      s_sp = spn;
    }
  }
  __syncthreads();
  for (int j = tid; j < m; j += BIG_THREADS) faces[F + j] = s_face[j];
}

// Node boxes travel between workgroups: written and read with agent-scope atomics (a plain
// load could be served from this CU's L1, which other CUs' stores never refresh).
__device__ inline void box_store(double* dst, const double* src) {
  for (int k = 0; k < 6; k++) __hip_atomic_store(dst + k, src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void box_load(const double* src, double* dst) {
  for (int k = 0; k < 6; k++)
    dst[k] = __hip_atomic_load(const_cast<double*>(src) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One thread per leaf walks towards the root; the second thread to reach a node owns it.
__global__ void k_refit(const uint32_t* faces, const Box6* tri_boxes, int n, const int* child_l,
                        const int* child_r, const int* parent_inner, const int* parent_leaf,
                        unsigned int* arrivals, double* node_boxes, int* node_depth, const int* range_first,
                        const int* range_last, int max_leaf, float trav_cost, float* node_cost, int* leafify) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int node = parent_leaf[k];
  while (node >= 0) {
    __threadfence();
    unsigned prev = __hip_atomic_fetch_add(&arrivals[node], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == 0) return;
    __threadfence();
    double b[6] = {__builtin_huge_val(), __builtin_huge_val(), __builtin_huge_val(),
                   -__builtin_huge_val(), -__builtin_huge_val(), -__builtin_huge_val()};
    int depth = 0;
    const int cs[2] = {child_l[node], child_r[node]};
    double child_area[2];
    float child_cost[2];
    for (int w = 0; w < 2; w++) {
      double cb[6];
      int cd = 0;
      child_cost[w] = 1.0f;  // one triangle test
      if (cs[w] & LEAF_BIT) {
        const Box6& tb = tri_boxes[faces[cs[w] & ~LEAF_BIT]];
        for (int a = 0; a < 3; a++) {
          cb[a] = tb.lo[a];
          cb[3 + a] = tb.hi[a];
        }
      } else {
        box_load(node_boxes + 6 * (size_t)cs[w], cb);
        cd = __hip_atomic_load(node_depth + cs[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        child_cost[w] = __hip_atomic_load(node_cost + cs[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (int a = 0; a < 3; a++) {
        b[a] = fmin(b[a], cb[a]);
        b[3 + a] = fmax(b[3 + a], cb[3 + a]);
      }
      const double ex = cb[3] - cb[0], ey = cb[4] - cb[1], ez = cb[5] - cb[2];
      child_area[w] = ex * ey + ey * ez + ez * ex;
      depth = max(depth, cd);
    }
    // Surface-area heuristic, bottom up: this subtree as ONE leaf costs its triangle count;
    // split, it costs a node step plus each child's cost weighted by the chance that a ray
    // which meets this box meets the child's.  Small subtrees take the cheaper of the two.
    {
      const double ex = b[3] - b[0], ey = b[4] - b[1], ez = b[5] - b[2];
      const double area = ex * ey + ey * ez + ez * ex;
      const float w0 = area > 0.0 ? (float)fmin(child_area[0] / area, 1.0) : 1.0f;
      const float w1 = area > 0.0 ? (float)fmin(child_area[1] / area, 1.0) : 1.0f;
      const float split = trav_cost + w0 * child_cost[0] + w1 * child_cost[1];
      const int count = range_last[node] - range_first[node] + 1;
      const bool as_leaf = count <= max_leaf && (float)count <= split;
      leafify[node] = as_leaf ? 1 : 0;
      __hip_atomic_store(node_cost + node, as_leaf ? (float)count : split, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    box_store(node_boxes + 6 * (size_t)node, b);
    __hip_atomic_store(node_depth + node, depth + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    node = parent_inner[node];
  }
}

// An inner node stays unless it, or one of its ancestors, became a leaf.  Only subtrees of at
// most max_leaf triangles can be leaves, so the walk up ends at the first larger ancestor.  The
// root always stays.
__global__ void k_mark_live(const int* range_first, const int* range_last, const int* parent_inner,
                            const int* leafify, int n_inner, int max_leaf, int* live) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_inner) return;
  int alive = 1;
  for (int a = i; a > 0 && (range_last[a] - range_first[a] + 1) <= max_leaf; a = parent_inner[a])
    if (leafify[a]) {
      alive = 0;
      break;
    }
  live[i] = alive;
}

struct GridD {
  double base[3], cell[3], pad;
};
__device__ inline unsigned grid_down(const GridD& G, int a, double x) {
  if (!(G.cell[a] > 0.0)) return 0;
  double q = floor((x - G.base[a]) / G.cell[a]);
  if (!(q > 0.0)) return 0;
  if (q > 65535.0) q = 65535.0;
  unsigned k = (unsigned)q;
  while (k > 0 && G.base[a] + k * G.cell[a] > x) k--;
  return k;
}
__device__ inline unsigned grid_up(const GridD& G, int a, double x) {
  if (!(G.cell[a] > 0.0)) return 65535;
  double q = ceil((x - G.base[a]) / G.cell[a]);
  if (!(q > 0.0)) q = 0.0;
  if (q > 65535.0) return 65535;
  unsigned k = (unsigned)q;
  while (k < 65535 && G.base[a] + k * G.cell[a] < x) k++;
  return k;
}

__global__ void k_emit_nodes(const uint32_t* faces, const Box6* tri_boxes, const int* child_l,
                             const int* child_r, const int* range_first, const int* range_last,
                             const double* node_boxes, const int* live, const int* new_index, const int* leafify,
                             int n_inner, int max_leaf, GridD G, rtd::DNode* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_inner || !live[i]) return;
  rtd::DNode N;
  const int cs[2] = {child_l[i], child_r[i]};
  for (int w = 0; w < 2; w++) {
    double b[6];
    int code;
    if (cs[w] & LEAF_BIT) {
      int pos = cs[w] & ~LEAF_BIT;
      const Box6& tb = tri_boxes[faces[pos]];
      for (int a = 0; a < 3; a++) {
        b[a] = tb.lo[a];
        b[3 + a] = tb.hi[a];
      }
      code = -2 - ((pos << rtd::LEAF_BITS) | 1);
    } else {
      int c = cs[w];
      for (int k = 0; k < 6; k++) b[k] = node_boxes[6 * (size_t)c + k];
      int cnt = range_last[c] - range_first[c] + 1;
      code = (cnt <= max_leaf && leafify[c]) ? -2 - ((range_first[c] << rtd::LEAF_BITS) | cnt) : new_index[c];
    }
    for (int a = 0; a < 3; a++) {
      unsigned lo = grid_down(G, a, b[a] - G.pad), hi = grid_up(G, a, b[3 + a] + G.pad);
      N.q[3 * w + a] = lo | (hi << 16);
    }
    (w == 0 ? N.c0 : N.c1) = code;
  }
  out[new_index[i]] = N;
}

// ---- four-wide nodes -------------------------------------------------------------------------
// One entry of the breadth-first front: a binary node that becomes a wide node.
struct WideTask {
  int binary;  // index in the binary node array
  int wide;    // position of the wide node (local to the mesh)
  int above;   // stack entries the walk may hold when it arrives here
};
__device__ inline double slot_area(const uint32_t* q, const double* cell) {
  double e[3];
  for (int a = 0; a < 3; a++) {
    const double lo = (double)(q[a] & 0xffffu), hi = (double)(q[a] >> 16);
    e[a] = (hi > lo ? hi - lo : 0.0) * cell[a];
  }
  return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
}
// One level of the fold: every front entry opens its binary node's inner children, largest
// surface first, until four slots are taken (the rule of collapse_to_wide on the host), writes
// its wide node and puts its inner slots on the next front.
__global__ void k_wide_level(const rtd::DNode* bn, const WideTask* front, int n_front, WideTask* next,
                             int* n_next, int* n_wide, int* max_need, GridD G, int node_base, rtd::DNode4* out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_front) return;
  const WideTask w = front[t];
  uint32_t q[4][3];
  int code[4];
  int n = 0;
  auto add_children = [&](const rtd::DNode& N) {
    const int cs[2] = {N.c0, N.c1};
    for (int k = 0; k < 2; k++) {
      if (cs[k] == -1) continue;
      for (int a = 0; a < 3; a++) q[n][a] = N.q[3 * k + a];
      code[n++] = cs[k];
    }
  };
  add_children(bn[w.binary]);
  while (n < 4) {
    int best = -1;
    double best_area = -1.0;
    for (int k = 0; k < n; k++)
      if (code[k] >= 0) {
        const double area = slot_area(q[k], G.cell);
        if (area > best_area) {
          best_area = area;
          best = k;
        }
      }
    if (best < 0) break;
    const rtd::DNode C = bn[code[best]];
    const int kids = (C.c0 != -1) + (C.c1 != -1);
    if (n - 1 + kids > 4) break;
    n--;
    for (int a = 0; a < 3; a++) q[best][a] = q[n][a];
    code[best] = code[n];
    add_children(C);
  }
  const int here = w.above + (n > 0 ? n - 1 : 0);  // while one child is walked the others wait on the stack
  atomicMax(max_need, here);
  // the inner children of one node take consecutive positions: siblings share cache lines, and
  // the one that waits on the stack is fetched with the one that is walked first
  int inner = 0;
  for (int k = 0; k < n; k++) inner += code[k] >= 0;
  int child = inner ? atomicAdd(n_wide, inner) : 0;
  int slot = inner ? atomicAdd(n_next, inner) : 0;
  rtd::DNode4 N;
  for (int k = 0; k < 4; k++) {
    if (k < n) {
      for (int a = 0; a < 3; a++) N.q[3 * k + a] = q[k][a];
      if (code[k] >= 0) {
        N.c[k] = node_base + child;
        next[slot++] = WideTask{code[k], child++, here};
      } else {
        N.c[k] = code[k];
      }
    } else {
      for (int a = 0; a < 3; a++) N.q[3 * k + a] = 0xffffu;  // lo = 65535, hi = 0: never hit
      N.c[k] = -1;
    }
  }
  out[w.wide] = N;
}

// ---- triangle records --------------------------------------------------------------------------
// Record k of the mesh is face order[k]: edges (triangle.rs:49-50), the flat normal
// normalize(cross(e2, e1)) (triangle.rs:51) or the vertex normals, and the f32 copy of the fast
// mode.  The same f64 operations the host path performs (this file is built without contraction).
__global__ void k_tri_records(const double* p1, const double* p2, const double* p3, const double* n1,
                              const double* n2, const double* n3, const uint8_t* smooth, const uint32_t* order,
                              int n, rtd::DTri* tris, rtd::DTriF* trisf, rtd::DTriN* trin) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t f = order[k];
  rtd::DTri T;
  rtd::DTriN TN;
  double e1[3], e2[3];
  for (int a = 0; a < 3; a++) {
    T.p1[a] = p1[3 * (size_t)f + a];
    e1[a] = p2[3 * (size_t)f + a] - p1[3 * (size_t)f + a];
    e2[a] = p3[3 * (size_t)f + a] - p1[3 * (size_t)f + a];
    T.e1[a] = e1[a];
    T.e2[a] = e2[a];
  }
  T.orig = f;
  T.smooth = smooth ? smooth[f] : 0u;
  if (T.smooth) {
    for (int a = 0; a < 3; a++) {
      TN.n[a] = n1[3 * (size_t)f + a];
      TN.n[3 + a] = n2[3 * (size_t)f + a];
      TN.n[6 + a] = n3[3 * (size_t)f + a];
    }
  } else {
    const double cx = e2[1] * e1[2] - e2[2] * e1[1];
    const double cy = e2[2] * e1[0] - e2[0] * e1[2];
    const double cz = e2[0] * e1[1] - e2[1] * e1[0];
    const double mag = sqrt(cx * cx + cy * cy + cz * cz);
    TN.n[0] = cx / mag;
    TN.n[1] = cy / mag;
    TN.n[2] = cz / mag;
    for (int a = 3; a < 9; a++) TN.n[a] = 0.0;
  }
  rtd::DTriF TF;
  uint32_t lo16[3];
  for (int a = 0; a < 3; a++) {
    TF.p1[a] = (float)T.p1[a];
    TF.e1[a] = (float)T.e1[a];
    TF.e2[a] = (float)T.e2[a];
    lo16[a] = __float_as_uint((float)(T.p1[a] - (double)TF.p1[a])) >> 16;
  }
  TF.orig = T.orig;
  TF.p1_lo_xy = lo16[0] | (lo16[1] << 16);
  TF.p1_lo_z = lo16[2];
  tris[k] = T;
  trisf[k] = TF;
  trin[k] = TN;
}

struct Scratch {  // frees what it holds when the build ends, whichever way
  std::vector<void*> ptrs;
  ~Scratch() {
    for (void* p : ptrs)
      if (p) (void)hipFree(p);
  }
  void keep(void* p) {  // the caller takes this one over
    for (void*& q : ptrs)
      if (q == p) q = nullptr;
  }
  template <typename T>
  hipError_t get(T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipSuccess) ptrs.push_back(q);
    *p = static_cast<T*>(q);
    return e;
  }
};

}  // namespace

void free_device_build(DeviceBuild& b) {
  void* ptrs[] = {b.d_nodes, b.d_order, b.d_p1, b.d_p2, b.d_p3};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  b.d_nodes = nullptr;
  b.d_order = nullptr;
  b.d_p1 = b.d_p2 = b.d_p3 = nullptr;
}

int build_lbvh_resident(const double* p1, const double* p2, const double* p3, int64_t n64, int max_leaf,
                        int node_base, DeviceBuild& out, std::string& err) {
  out = DeviceBuild();
  max_leaf = std::max(1, std::min(max_leaf, rtd::LEAF_MAX));
  if (n64 > (1 << (30 - rtd::LEAF_BITS))) {
    err = "mesh too large for the device builder";
    return 1;
  }
  const int n = (int)n64;
  Scratch S;
  rtd::DNode4 single;  // a mesh of no triangles, or of one leaf: one node, made here
  for (int k = 0; k < 4; k++) {
    for (int a = 0; a < 3; a++) single.q[3 * k + a] = 0xffffu;
    single.c[k] = -1;
  }
  rtd::DNode4* d_wide = nullptr;
  if (n == 0) {
    LB_TRY(S.get(&d_wide, 1));
    LB_TRY(hipMemcpy(d_wide, &single, sizeof single, hipMemcpyHostToDevice));
    S.keep(d_wide);
    out.d_nodes = d_wide;
    out.n_nodes = 1;
    return 0;
  }
  double *d1, *d2, *d3;
  Box6* boxes;
  Bounds* bounds;
  unsigned long long *keys_in, *keys;
  uint32_t *faces_in, *d_order;  // d_order: face of the k-th triangle in Morton order
  LB_TRY(S.get(&faces_in, (size_t)n));
  LB_TRY(S.get(&d_order, (size_t)n));
  LB_TRY(S.get(&d1, 3 * (size_t)n));
  LB_TRY(S.get(&d2, 3 * (size_t)n));
  LB_TRY(S.get(&d3, 3 * (size_t)n));
  LB_TRY(S.get(&boxes, (size_t)n));
  LB_TRY(S.get(&bounds, 1));
  LB_TRY(S.get(&keys_in, (size_t)n));
  LB_TRY(S.get(&keys, (size_t)n));
  LB_TRY(hipMemcpy(d1, p1, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  LB_TRY(hipMemcpy(d2, p2, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  LB_TRY(hipMemcpy(d3, p3, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  Bounds init;
  const double inf = std::numeric_limits<double>::infinity();
  for (int a = 0; a < 3; a++) {
    init.lo[a] = enc(inf);
    init.hi[a] = enc(-inf);
  }
  init.scale = enc(0.0);
  LB_TRY(hipMemcpy(bounds, &init, sizeof init, hipMemcpyHostToDevice));
  const dim3 block(256), grid((unsigned)((n + 255) / 256));
  hipLaunchKernelGGL(k_tri_bounds, grid, block, 0, 0, d1, d2, d3, n, boxes, bounds);
  hipLaunchKernelGGL(k_morton, grid, block, 0, 0, boxes, n, bounds, keys_in, faces_in);
  LB_TRY(hipGetLastError());
  size_t temp_bytes = 0;
  LB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, keys_in, keys, faces_in, d_order, n, 0, 63));
  char* temp;
  LB_TRY(S.get(&temp, temp_bytes));
  LB_TRY(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys, faces_in, d_order, n, 0, 63));

  // the mesh bounds come to the host (56 bytes): they size the node grid and go into the
  // mesh's record
  Bounds hb;
  LB_TRY(hipMemcpy(&hb, bounds, sizeof hb, hipMemcpyDeviceToHost));
  out.scale = dec(hb.scale);
  const double pad = out.scale * BOX_PAD_REL + 1e-300;
  GridD G;
  G.pad = pad;
  for (int a = 0; a < 3; a++) {
    double lo = dec(hb.lo[a]) - 2.0 * pad, hi = dec(hb.hi[a]) + 2.0 * pad;
    G.base[a] = lo;
    G.cell[a] = (hi - lo) / 65535.0 * (1.0 + 1e-12);
    out.lo[a] = lo;
    out.hi[a] = std::max(hi, lo + 65535.0 * G.cell[a]);
    out.cell[a] = G.cell[a];
  }
  auto hand_over = [&]() {  // what the caller keeps: vertices and order for the record kernel
    S.keep(d1);
    S.keep(d2);
    S.keep(d3);
    S.keep(d_order);
    out.d_p1 = d1;
    out.d_p2 = d2;
    out.d_p3 = d3;
    out.d_order = d_order;
  };

  if (n <= max_leaf) {  // the whole mesh is one leaf
    for (int a = 0; a < 3; a++) single.q[a] = 0u | (65535u << 16);
    single.c[0] = -2 - ((0 << rtd::LEAF_BITS) | n);
    LB_TRY(S.get(&d_wide, 1));
    LB_TRY(hipMemcpy(d_wide, &single, sizeof single, hipMemcpyHostToDevice));
    LB_TRY(hipDeviceSynchronize());
    S.keep(d_wide);
    out.d_nodes = d_wide;
    out.n_nodes = 1;
    hand_over();
    return 0;
  }

  const int n_inner = n - 1;
  int *child_l, *child_r, *range_first, *range_last, *parent_inner, *parent_leaf, *node_depth, *live,
      *new_index;
  unsigned int* arrivals;
  double* node_boxes;
  float* node_cost;
  int* leafify;
  LB_TRY(S.get(&node_cost, (size_t)n_inner));
  LB_TRY(S.get(&leafify, (size_t)n_inner));
  LB_TRY(hipMemset(leafify, 0, (size_t)n_inner * sizeof(int)));
  // cost of a step down the binary tree, in triangle tests (measured: 2 renders the pawn and the
  // tea set fastest; the 1 M-triangle soup wants smaller leaves: 1.5 there is 3 % faster than 2)
  float trav_cost = n >= 100000 ? 1.5f : 2.0f;
  if (const char* e = getenv("RT_LBVH_TRAV")) trav_cost = (float)atof(e);
  LB_TRY(S.get(&child_l, (size_t)n_inner));
  LB_TRY(S.get(&child_r, (size_t)n_inner));
  LB_TRY(S.get(&range_first, (size_t)n_inner));
  LB_TRY(S.get(&range_last, (size_t)n_inner));
  LB_TRY(S.get(&parent_inner, (size_t)n_inner));
  LB_TRY(S.get(&parent_leaf, (size_t)n));
  LB_TRY(S.get(&node_depth, (size_t)n_inner));
  LB_TRY(S.get(&live, (size_t)n_inner));
  LB_TRY(S.get(&new_index, (size_t)n_inner));
  LB_TRY(S.get(&arrivals, (size_t)n_inner));
  LB_TRY(S.get(&node_boxes, 6 * (size_t)n_inner));
  LB_TRY(hipMemset(arrivals, 0, (size_t)n_inner * sizeof(unsigned int)));
  LB_TRY(hipMemset(node_depth, 0, (size_t)n_inner * sizeof(int)));
  const dim3 igrid((unsigned)((n_inner + 255) / 256));
  hipLaunchKernelGGL(k_radix_tree, igrid, block, 0, 0, keys, n, child_l, child_r, range_first, range_last,
                     parent_inner, parent_leaf);
  bool treelets = true;
  if (const char* e = getenv("RT_LBVH_TREELETS")) treelets = atoi(e) != 0;
  if (treelets) {
    int *list, *n_list;
    LB_TRY(S.get(&list, (size_t)n_inner));
    LB_TRY(S.get(&n_list, 1));
    int count = 0;
    bool big = true;  // RT_LBVH_BIG_TREELETS=0: the radix tree's splits stay above the 64-triangle subtrees
    if (const char* e = getenv("RT_LBVH_BIG_TREELETS")) big = atoi(e) != 0;
    if (big) {
      LB_TRY(hipMemset(n_list, 0, sizeof(int)));
      hipLaunchKernelGGL(k_find_treelets, igrid, block, 0, 0, range_first, range_last, parent_inner, n_inner, TREELET + 1,
                         BIG_TREELET, list, n_list);
      LB_TRY(hipGetLastError());
      LB_TRY(hipMemcpy(&count, n_list, sizeof(int), hipMemcpyDeviceToHost));
      if (count > 0) {
        hipLaunchKernelGGL(k_rebuild_big_treelets, dim3((unsigned)count), dim3(BIG_THREADS), 0, 0, list, count, d_order,
                           boxes, child_l, child_r, range_first, range_last, parent_inner, parent_leaf, n_inner);
        LB_TRY(hipGetLastError());
      }
    }
    LB_TRY(hipMemset(n_list, 0, sizeof(int)));
    // (two triangles can be split one way only: from three on)
    hipLaunchKernelGGL(k_find_treelets, igrid, block, 0, 0, range_first, range_last, parent_inner, n_inner, 3, TREELET, list,
                       n_list);
    LB_TRY(hipGetLastError());
    count = 0;
    LB_TRY(hipMemcpy(&count, n_list, sizeof(int), hipMemcpyDeviceToHost));
    if (count > 0) {
      hipLaunchKernelGGL(k_rebuild_treelets, dim3((unsigned)count), dim3(64), 0, 0, list, count, d_order, boxes, child_l,
                         child_r, range_first, range_last, parent_inner, parent_leaf);
      LB_TRY(hipGetLastError());
    }
  }
  hipLaunchKernelGGL(k_refit, grid, block, 0, 0, d_order, boxes, n, child_l, child_r, parent_inner, parent_leaf,
                     arrivals, node_boxes, node_depth, range_first, range_last, max_leaf, trav_cost, node_cost, leafify);
  hipLaunchKernelGGL(k_mark_live, igrid, block, 0, 0, range_first, range_last, parent_inner, leafify, n_inner, max_leaf,
                     live);
  LB_TRY(hipGetLastError());
  size_t scan_bytes = 0;
  LB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, live, new_index, n_inner));
  char* scan_temp;
  LB_TRY(S.get(&scan_temp, scan_bytes));
  LB_TRY(hipcub::DeviceScan::ExclusiveSum(scan_temp, scan_bytes, live, new_index, n_inner));
  int last_live = 0, last_index = 0;
  LB_TRY(hipMemcpy(&last_live, live + n_inner - 1, sizeof(int), hipMemcpyDeviceToHost));
  LB_TRY(hipMemcpy(&last_index, new_index + n_inner - 1, sizeof(int), hipMemcpyDeviceToHost));
  const int n_live = last_index + last_live;
  rtd::DNode* d_nodes;
  LB_TRY(S.get(&d_nodes, (size_t)n_live));
  hipLaunchKernelGGL(k_emit_nodes, igrid, block, 0, 0, d_order, boxes, child_l, child_r, range_first, range_last,
                     node_boxes, live, new_index, leafify, n_inner, max_leaf, G, d_nodes);
  LB_TRY(hipGetLastError());

  // fold into four-wide nodes, one level of the wide tree per launch, from the root (binary node
  // 0 -> wide node 0); each level's length is the only thing the host reads
  WideTask *front, *next;
  int* counters;  // [0] entries on the next front, [1] wide nodes so far, [2] stack need
  LB_TRY(S.get(&front, (size_t)n_live));
  LB_TRY(S.get(&next, (size_t)n_live));
  LB_TRY(S.get(&counters, 3));
  LB_TRY(S.get(&d_wide, (size_t)n_live));
  const WideTask root{0, 0, 0};
  const int init3[3] = {0, 1, 0};
  LB_TRY(hipMemcpy(front, &root, sizeof root, hipMemcpyHostToDevice));
  LB_TRY(hipMemcpy(counters, init3, sizeof init3, hipMemcpyHostToDevice));
  int n_front = 1, levels = 0;
  while (n_front > 0) {
    if (++levels > 4096) {
      err = "wide collapse did not terminate";
      return 1;
    }
    hipLaunchKernelGGL(k_wide_level, dim3((unsigned)((n_front + 127) / 128)), dim3(128), 0, 0, d_nodes, front, n_front,
                       next, counters, counters + 1, counters + 2, G, node_base, d_wide);
    LB_TRY(hipGetLastError());
    LB_TRY(hipMemcpy(&n_front, counters, sizeof(int), hipMemcpyDeviceToHost));
    LB_TRY(hipMemset(counters, 0, sizeof(int)));
    std::swap(front, next);
  }
  int tail[2];
  LB_TRY(hipMemcpy(tail, counters + 1, sizeof tail, hipMemcpyDeviceToHost));
  S.keep(d_wide);
  out.d_nodes = d_wide;
  out.n_nodes = tail[0];
  out.stack_need = tail[1];
  hand_over();
  return 0;
}

int build_tri_records_device(const DeviceBuild& b, int64_t n64, const double* n1, const double* n2,
                             const double* n3, const uint8_t* smooth, rtd::DTri* d_tris, rtd::DTriF* d_trisf,
                             rtd::DTriN* d_trin, std::string& err) {
  const int n = (int)n64;
  if (n == 0) return 0;
  Scratch S;
  double *dn1 = nullptr, *dn2 = nullptr, *dn3 = nullptr;
  uint8_t* dsm = nullptr;
  if (n1 && n2 && n3 && smooth) {
    LB_TRY(S.get(&dn1, 3 * (size_t)n));
    LB_TRY(S.get(&dn2, 3 * (size_t)n));
    LB_TRY(S.get(&dn3, 3 * (size_t)n));
    LB_TRY(S.get(&dsm, (size_t)n));
    LB_TRY(hipMemcpy(dn1, n1, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    LB_TRY(hipMemcpy(dn2, n2, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    LB_TRY(hipMemcpy(dn3, n3, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    LB_TRY(hipMemcpy(dsm, smooth, (size_t)n, hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(k_tri_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, b.d_p1, b.d_p2, b.d_p3, dn1, dn2,
                     dn3, dsm, b.d_order, n, d_tris, d_trisf, d_trin);
  LB_TRY(hipGetLastError());
  LB_TRY(hipDeviceSynchronize());
  return 0;
}

}  // namespace rtb
