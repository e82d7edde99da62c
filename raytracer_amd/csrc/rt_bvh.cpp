// rt_bvh.cpp — binned-SAH BVH2 over one mesh, emitted in the "children's boxes in the
// parent" node layout of rt_layout.h.
#include "rt_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>

namespace rtb {
namespace {

struct Box {
  double lo[3], hi[3];
  void reset() {
    for (int a = 0; a < 3; a++) {
      lo[a] = std::numeric_limits<double>::infinity();
      hi[a] = -std::numeric_limits<double>::infinity();
    }
  }
  void grow(const double* p) {
    for (int a = 0; a < 3; a++) {
      lo[a] = std::min(lo[a], p[a]);
      hi[a] = std::max(hi[a], p[a]);
    }
  }
  void grow(const Box& b) {
    for (int a = 0; a < 3; a++) {
      lo[a] = std::min(lo[a], b.lo[a]);
      hi[a] = std::max(hi[a], b.hi[a]);
    }
  }
  double half_area() const {
    double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (dx < 0 || dy < 0 || dz < 0) return 0.0;
    return dx * dy + dy * dz + dz * dx;
  }
};

struct TmpNode {
  Box box;
  int left = -1, right = -1;  // inner: indices into tmp
  int first = 0, count = 0;   // leaf range
};

struct Builder {
  const double *p1, *p2, *p3;
  int64_t n;
  int max_leaf;
  // cost of one node step in units of one triangle test: triangle tests are shared by the
  // lanes of a wave, node steps are not (measured best of 0.5, 1, 2, 3, 5 on the pawn with the
  // greedy stop; the bottom-up decision takes 2, see build_sah)
  double trav_cost = 3.0;
  std::vector<Box> tri_box;
  std::vector<double> cen;  // 3 per triangle
  std::vector<uint32_t> idx;
  std::vector<TmpNode> tmp;
  int max_depth = 0;

  static constexpr int BINS = 16;
  static constexpr int DEPTH_LIMIT = 40;  // beyond this, split by median so depth stays bounded
  // bottom_up: split down to single triangles and decide afterwards, from the leaves up, which
  // subtrees are cheaper as one leaf (settle()); otherwise a node stops when its own split does
  // not pay.  sweep_below: ranges of at most this many triangles try every split position of
  // the sorted centroids instead of the bin borders.
  // (measured on MI355X: bottom-up + sweeps below 64 renders the tea set 3 % faster than the
  // greedy stop and the pawn as fast; sweeping every range is slower on the pawn)
  bool bottom_up = true;
  int sweep_below = 64;
  std::vector<double> suffix;

  // exact search over the sorted order of one axis; leaves idx[first..first+count) sorted by it
  void sweep_axis(int first, int count, int a, double& best_cost, int& best_axis, int& best_left) {
    std::sort(idx.begin() + first, idx.begin() + first + count, [&](uint32_t x, uint32_t y) {
      const double cx = cen[3 * x + a], cy = cen[3 * y + a];
      return cx < cy || (cx == cy && x < y);
    });
    suffix.resize((size_t)count + 1);
    Box acc;
    acc.reset();
    for (int i = count - 1; i > 0; i--) {
      acc.grow(tri_box[idx[first + i]]);
      suffix[i] = acc.half_area();
    }
    acc.reset();
    for (int i = 0; i < count - 1; i++) {
      acc.grow(tri_box[idx[first + i]]);
      const double cost = acc.half_area() * (i + 1) + suffix[i + 1] * (count - i - 1);
      if (cost < best_cost) {
        best_cost = cost;
        best_axis = a;
        best_left = i + 1;
      }
    }
  }

  // Cost of the subtree in triangle tests per ray that meets its box, deciding on the way up
  // whether it stays split or becomes one leaf.
  double settle(int node) {
    TmpNode& N = tmp[node];
    if (N.left < 0) return (double)N.count;
    const double cl = settle(N.left), cr = settle(N.right);
    const double area = N.box.half_area();
    const double wl = area > 0.0 ? std::min(tmp[N.left].box.half_area() / area, 1.0) : 1.0;
    const double wr = area > 0.0 ? std::min(tmp[N.right].box.half_area() / area, 1.0) : 1.0;
    const double split = trav_cost + wl * cl + wr * cr;
    if (N.count <= max_leaf && (double)N.count <= split) {
      N.left = N.right = -1;
      return (double)N.count;
    }
    return split;
  }

  int build(int first, int count, int depth) {
    max_depth = std::max(max_depth, depth);
    TmpNode node;
    node.box.reset();
    Box cb;
    cb.reset();
    for (int i = 0; i < count; i++) {
      uint32_t t = idx[first + i];
      node.box.grow(tri_box[t]);
      cb.grow(&cen[3 * t]);
    }
    node.first = first;
    node.count = count;
    int self = (int)tmp.size();
    tmp.push_back(node);
    if (count <= 1) return self;

    int best_axis = -1, best_split = -1;
    double best_cost = std::numeric_limits<double>::infinity();
    if (depth < DEPTH_LIMIT && count <= sweep_below) {
      int left = -1;
      for (int a = 0; a < 3; a++) sweep_axis(first, count, a, best_cost, best_axis, left);
      if (best_axis >= 0) {
        if (best_axis != 2) {  // the range is sorted by the last axis tried: sort by the winner again
          double c = std::numeric_limits<double>::infinity();
          int ax = -1, lf = -1;
          sweep_axis(first, count, best_axis, c, ax, lf);
        }
        if (!bottom_up && count <= max_leaf &&
            best_cost + node.box.half_area() * trav_cost >= node.box.half_area() * count)
          return self;
        int l = build(first, left, depth + 1);
        int r = build(first + left, count - left, depth + 1);
        tmp[self].left = l;
        tmp[self].right = r;
        return self;
      }
    }
    if (depth < DEPTH_LIMIT) {
      for (int a = 0; a < 3; a++) {
        double ext = cb.hi[a] - cb.lo[a];
        if (!(ext > 0.0)) continue;
        Box bins[BINS];
        int cnt[BINS];
        for (int b = 0; b < BINS; b++) {
          bins[b].reset();
          cnt[b] = 0;
        }
        double k = BINS / ext;
        for (int i = 0; i < count; i++) {
          uint32_t t = idx[first + i];
          int b = (int)((cen[3 * t + a] - cb.lo[a]) * k);
          b = std::min(std::max(b, 0), BINS - 1);
          bins[b].grow(tri_box[t]);
          cnt[b]++;
        }
        double right_area[BINS];
        int right_cnt[BINS];
        Box acc;
        acc.reset();
        int c = 0;
        for (int b = BINS - 1; b > 0; b--) {
          acc.grow(bins[b]);
          c += cnt[b];
          right_area[b] = acc.half_area();
          right_cnt[b] = c;
        }
        acc.reset();
        c = 0;
        for (int b = 0; b < BINS - 1; b++) {
          acc.grow(bins[b]);
          c += cnt[b];
          if (c == 0 || right_cnt[b + 1] == 0) continue;
          double cost = acc.half_area() * c + right_area[b + 1] * right_cnt[b + 1];
          if (cost < best_cost) {
            best_cost = cost;
            best_axis = a;
            best_split = b;
          }
        }
      }
    }
    // leaf if small enough and splitting does not pay
    if (count <= max_leaf && !bottom_up) {
      double leaf_cost = node.box.half_area() * count;
      if (best_axis < 0 || best_cost + node.box.half_area() * trav_cost >= leaf_cost) return self;
    }
    int mid;
    if (best_axis >= 0) {
      double ext = cb.hi[best_axis] - cb.lo[best_axis];
      double k = BINS / ext;
      double lo = cb.lo[best_axis];
      int axis = best_axis, split = best_split;
      auto it = std::partition(idx.begin() + first, idx.begin() + first + count, [&](uint32_t t) {
        int b = (int)((cen[3 * t + axis] - lo) * k);
        b = std::min(std::max(b, 0), BINS - 1);
        return b <= split;
      });
      mid = (int)(it - (idx.begin() + first));
    } else {
      mid = 0;
    }
    if (mid <= 0 || mid >= count) {
      // degenerate (coincident centroids) or depth limit: median split on the widest axis
      int axis = 0;
      double ext = -1.0;
      for (int a = 0; a < 3; a++)
        if (node.box.hi[a] - node.box.lo[a] > ext) {
          ext = node.box.hi[a] - node.box.lo[a];
          axis = a;
        }
      mid = count / 2;
      std::nth_element(idx.begin() + first, idx.begin() + first + mid, idx.begin() + first + count,
                       [&](uint32_t x, uint32_t y) { return cen[3 * x + axis] < cen[3 * y + axis]; });
    }
    int l = build(first, mid, depth + 1);
    int r = build(first + mid, count - mid, depth + 1);
    tmp[self].left = l;
    tmp[self].right = r;
    return self;
  }
};

// The 16-bit grid of one mesh: plane(q) = base + q*cell, q in [0, 65535].
struct Grid {
  double base[3], cell[3];
  // grid step at or below x / at or above x, verified against the f64 plane it stands for
  uint32_t down(int a, double x) const {
    if (!(cell[a] > 0.0)) return 0;
    double q = std::floor((x - base[a]) / cell[a]);
    if (!(q > 0.0)) return 0;
    if (q > 65535.0) q = 65535.0;
    uint32_t k = (uint32_t)q;
    while (k > 0 && base[a] + k * cell[a] > x) k--;
    return k;
  }
  uint32_t up(int a, double x) const {
    if (!(cell[a] > 0.0)) return 65535;
    double q = std::ceil((x - base[a]) / cell[a]);
    if (!(q > 0.0)) q = 0.0;
    if (q > 65535.0) return 65535;
    uint32_t k = (uint32_t)q;
    while (k < 65535 && base[a] + k * cell[a] < x) k++;
    return k;
  }
};

void set_child(rtd::DNode& N, int which, const Box& b, double pad, const Grid& G, int child) {
  for (int a = 0; a < 3; a++) {
    uint32_t lo = G.down(a, b.lo[a] - pad), hi = G.up(a, b.hi[a] + pad);
    N.q[3 * which + a] = lo | (hi << 16);
  }
  (which == 0 ? N.c0 : N.c1) = child;
}

void set_empty(rtd::DNode& N, int which) {
  for (int a = 0; a < 3; a++) N.q[3 * which + a] = 0xffffu;  // lo = 65535, hi = 0
  (which == 0 ? N.c0 : N.c1) = -1;
}

}  // namespace

namespace {
struct WideSlot {
  uint32_t q[3];
  int code;
};
double slot_area(const WideSlot& s, const double* cell) {
  double e[3];
  for (int a = 0; a < 3; a++) {
    const double lo = (double)(s.q[a] & 0xffffu), hi = (double)(s.q[a] >> 16);
    e[a] = (hi > lo ? hi - lo : 0.0) * cell[a];
  }
  return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
}
int collapse_rec(const std::vector<rtd::DNode>& in, int i, std::vector<rtd::DNode4>& out, int self,
                 const double* cell) {
  // Start from the two children of the binary node and, while a slot is free, open the inner
  // child with the largest surface (the one a ray is most likely to enter anyway).
  WideSlot slot[4];
  int n = 0;
  auto add_children = [&](const rtd::DNode& N) {
    const int cs[2] = {N.c0, N.c1};
    for (int w = 0; w < 2; w++) {
      if (cs[w] == -1) continue;
      for (int a = 0; a < 3; a++) slot[n].q[a] = N.q[3 * w + a];
      slot[n++].code = cs[w];
    }
  };
  add_children(in[i]);
  while (n < 4) {
    int best = -1;
    double best_area = -1.0;
    for (int k = 0; k < n; k++)
      if (slot[k].code >= 0) {
        const double area = slot_area(slot[k], cell);
        if (area > best_area) {
          best_area = area;
          best = k;
        }
      }
    if (best < 0) break;
    const rtd::DNode& C = in[slot[best].code];
    const int kids = (C.c0 != -1) + (C.c1 != -1);
    if (n - 1 + kids > 4) break;
    slot[best] = slot[--n];
    add_children(C);
  }
  // stack entries a walk of this subtree can hold at once: while one child is walked, the
  // other n - 1 may all be waiting on the stack
  int below = 0;
  for (int k = 0; k < 4; k++) {
    if (k < n) {
      for (int a = 0; a < 3; a++) out[self].q[3 * k + a] = slot[k].q[a];
      if (slot[k].code >= 0) {
        const int child = (int)out.size();
        out.emplace_back();
        out[self].c[k] = child;
        below = std::max(below, collapse_rec(in, slot[k].code, out, child, cell));
      } else {
        out[self].c[k] = slot[k].code;
      }
    } else {
      for (int a = 0; a < 3; a++) out[self].q[3 * k + a] = 0xffffu;  // lo = 65535, hi = 0
      out[self].c[k] = -1;
    }
  }
  return std::max(n - 1, 0) + below;
}
}  // namespace

int collapse_to_wide(const std::vector<rtd::DNode>& nodes, const double* cell, std::vector<rtd::DNode4>& out) {
  out.clear();
  out.reserve(nodes.size() / 2 + 2);
  out.emplace_back();
  if (nodes.empty()) {
    for (int k = 0; k < 4; k++) {
      for (int a = 0; a < 3; a++) out[0].q[3 * k + a] = 0xffffu;
      out[0].c[k] = -1;
    }
    return 0;
  }
  return collapse_rec(nodes, 0, out, 0, cell);
}

void reorder_top_breadth_first(std::vector<rtd::DNode4>& nodes, int top) {
  const int n = (int)nodes.size();
  if (n <= 2 || top <= 1) return;
  std::vector<int> order;  // new position -> old index
  order.reserve(n);
  std::vector<char> taken(n, 0);
  order.push_back(0);
  taken[0] = 1;
  for (size_t head = 0; head < order.size() && (int)order.size() < top; head++) {
    const rtd::DNode4& N = nodes[order[head]];
    for (int c : N.c)
      if (c >= 0 && c < n && !taken[c] && (int)order.size() < top) {
        order.push_back(c);
        taken[c] = 1;
      }
  }
  for (int i = 0; i < n; i++)
    if (!taken[i]) order.push_back(i);
  std::vector<int> where(n);
  for (int k = 0; k < n; k++) where[order[k]] = k;
  std::vector<rtd::DNode4> out(n);
  for (int k = 0; k < n; k++) {
    rtd::DNode4 N = nodes[order[k]];
    for (int j = 0; j < 4; j++)
      if (N.c[j] >= 0) N.c[j] = where[N.c[j]];
    out[k] = N;
  }
  nodes.swap(out);
}

void build_sah(const double* p1, const double* p2, const double* p3, int64_t n, int max_leaf,
               BuildResult& out) {
  out.nodes.clear();
  out.order.clear();
  out.max_depth = 0;
  Builder B;
  B.p1 = p1;
  B.p2 = p2;
  B.p3 = p3;
  B.n = n;
  B.max_leaf = std::max(1, std::min(max_leaf, rtd::LEAF_MAX));
  if (const char* e = getenv("RT_SAH_BOTTOMUP")) B.bottom_up = atoi(e) != 0;
  if (const char* e = getenv("RT_SAH_SWEEP")) B.sweep_below = atoi(e);
  if (B.bottom_up) B.trav_cost = 2.0;
  if (const char* e = getenv("RT_SAH_TRAV")) B.trav_cost = atof(e);
  B.tri_box.resize(n);
  B.cen.resize(3 * n);
  B.idx.resize(n);
  double scale = 0.0;
  for (int64_t t = 0; t < n; t++) {
    Box& b = B.tri_box[t];
    b.reset();
    b.grow(p1 + 3 * t);
    b.grow(p2 + 3 * t);
    b.grow(p3 + 3 * t);
    for (int a = 0; a < 3; a++) {
      B.cen[3 * t + a] = 0.5 * (b.lo[a] + b.hi[a]);
      scale = std::max(scale, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
    }
    B.idx[t] = (uint32_t)t;
  }
  double pad = scale * BOX_PAD_REL + 1e-300;
  out.scale = scale;
  for (int a = 0; a < 3; a++) {
    out.lo[a] = 0.0;
    out.hi[a] = 0.0;
    out.cell[a] = 0.0;
  }
  rtd::DNode zero{};
  if (n == 0) {
    rtd::DNode N = zero;
    set_empty(N, 0);
    set_empty(N, 1);
    out.nodes.push_back(N);
    return;
  }
  B.tmp.reserve(2 * n);
  int root = B.build(0, (int)n, 0);
  if (B.bottom_up) B.settle(root);
  out.max_depth = B.max_depth;
  out.order = B.idx;
  Grid G;
  for (int a = 0; a < 3; a++) {
    // the grid spans the root box padded twice: every child box (padded once) lies strictly
    // inside it, so clamping to [0, 65535] never cuts a box short
    double lo = B.tmp[root].box.lo[a] - 2.0 * pad, hi = B.tmp[root].box.hi[a] + 2.0 * pad;
    G.base[a] = lo;
    G.cell[a] = (hi - lo) / 65535.0 * (1.0 + 1e-12);
    out.lo[a] = lo;
    out.hi[a] = std::max(hi, lo + 65535.0 * G.cell[a]);
    out.cell[a] = G.cell[a];
  }

  // Convert: every inner tmp node becomes one DNode carrying its two children's boxes.
  const TmpNode& R = B.tmp[root];
  if (R.left < 0) {  // the whole mesh is one leaf
    rtd::DNode N = zero;
    set_child(N, 0, R.box, pad, G, -2 - ((R.first << rtd::LEAF_BITS) | R.count));
    set_empty(N, 1);
    out.nodes.push_back(N);
    return;
  }
  // iterative conversion, DFS order (left subtree right after its parent)
  std::vector<int> map(B.tmp.size(), -1);
  std::vector<int> stack{root};
  std::vector<int> order_inner;
  while (!stack.empty()) {
    int t = stack.back();
    stack.pop_back();
    map[t] = (int)order_inner.size();
    order_inner.push_back(t);
    const TmpNode& T = B.tmp[t];
    if (B.tmp[T.right].left >= 0) stack.push_back(T.right);
    if (B.tmp[T.left].left >= 0) stack.push_back(T.left);
  }
  out.nodes.resize(order_inner.size(), zero);
  for (size_t i = 0; i < order_inner.size(); i++) {
    const TmpNode& T = B.tmp[order_inner[i]];
    rtd::DNode& N = out.nodes[i];
    const TmpNode& L = B.tmp[T.left];
    const TmpNode& Rr = B.tmp[T.right];
    set_child(N, 0, L.box, pad, G, L.left >= 0 ? map[T.left] : -2 - ((L.first << rtd::LEAF_BITS) | L.count));
    set_child(N, 1, Rr.box, pad, G, Rr.left >= 0 ? map[T.right] : -2 - ((Rr.first << rtd::LEAF_BITS) | Rr.count));
  }
}

}  // namespace rtb
