// rt_core.hip — the C ABI of include/rt_core.h: scene flattening, commit (BVH build +
// upload), the render kernels and their launchers.  gfx950 only; there is no CPU path —
// every entry point that needs the GPU fails with RT_ERR_NO_DEVICE / RT_ERR_HIP without one.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_core.h"
#include "rt_bvh.h"
#include "rt_device.h"
#include "rt_internal.h"

using namespace rtd;

// ---------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
  } while (0)

struct MeshH {
  int32_t group, material;
  int64_t ntris;
  uint32_t order_base;
  std::vector<double> p1, p2, p3, n1, n2, n3;
  std::vector<uint8_t> smooth;
  bool has_normals;
};

// Everything that fixes a frame's ray-queue lengths.  Plans and pipeline choices are found by a
// hash of these fields and then compared field by field, so two frames that happen to share a
// hash never share a plan.
struct FrameId {
  DCamera dc;
  int32_t tile_first, tile_step, depth, fast, compact;
};
static FrameId frame_id(const DCamera& dc, int tile_first, int tile_step, int depth, bool fast, bool compact) {
  FrameId f;
  std::memset(&f, 0, sizeof f);  // padding too: the identities are compared as bytes
  f.dc = dc;
  f.tile_first = tile_first;
  f.tile_step = tile_step;
  f.depth = depth;
  f.fast = fast ? 1 : 0;
  f.compact = compact ? 1 : 0;
  return f;
}
static bool same_frame(const FrameId& a, const FrameId& b) { return std::memcmp(&a, &b, sizeof a) == 0; }

// What the host builder made of one mesh: the four-wide nodes (child indices relative to the
// mesh's first node) and the triangle records in leaf order.  rt_scene_commit_multi keeps these
// while it commits the replicas, so the tree is built once, not once per GPU.
struct HostBuilt {
  std::vector<DNode4> nodes;
  std::vector<DTri> tris;
  std::vector<DTriF> trisf;
  std::vector<DTriN> trin;
  double lo[3], hi[3], cell[3], scale;
  int stack_need;
};

struct Multi;
struct rt_scene {
  std::vector<rt_material> mats;
  std::vector<rt_group> groups;
  std::vector<rt_primitive> prims;
  std::vector<uint32_t> prim_order;
  std::vector<MeshH> meshes;
  std::vector<DLight> lights;
  uint32_t next_order = 0;
  bool committed = false;
  int device = -1;
  // device side
  std::vector<void*> allocs;
  DScene d{};
  rt_scene_info info{};
  unsigned long long* d_counters = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timing_pending = false;
  uint64_t launches = 0;
  // scratch for the host-buffer entry point (grow only)
  double* d_out_f64 = nullptr;
  uint8_t* d_out_u8 = nullptr;
  size_t out_cap_px = 0;
  // wavefront pipeline: per-level ray queues and result nodes (grow only)
  // One set per stream that renders with the wavefront pipeline, so that frames issued on
  // different streams may overlap (the tail of one frame's level under the next frame).
  struct WfState {
    hipStream_t stream = nullptr;
    void* wf_tasks[MAX_DEPTH + 1] = {};
    void* wf_nodes[MAX_DEPTH + 1] = {};
    void* wf_live[MAX_DEPTH + 1] = {};  // one byte per node: 1 = children will add to it
    void* wf_shadow[MAX_DEPTH + 1] = {};  // the level's queued light loops (WfShadow), one per hit
    size_t wf_shadow_cap[MAX_DEPTH + 1] = {};
    void* wf_keys[MAX_DEPTH + 1] = {};  // coherence sort: key per task slot / task slot per sorted position
    void* wf_perm[MAX_DEPTH + 1] = {};
    size_t wf_keys_cap[MAX_DEPTH + 1] = {};
    size_t wf_perm_cap[MAX_DEPTH + 1] = {};
    unsigned int* wf_hist = nullptr;  // SORT_BINS counters (zero between uses) + SORT_BINS cursors
    size_t wf_task_cap[MAX_DEPTH + 1] = {};
    size_t wf_node_cap[MAX_DEPTH + 1] = {};
    size_t wf_live_cap[MAX_DEPTH + 1] = {};
    // device: two counters per level, the overflow flag, one shadow-queue length per level; then, for
    // placed emission, SORT_BINS tickets for each of the levels 1..MAX_DEPTH (one allocation, so that
    // one fill at the start of a frame zeroes what the frame needs)
    unsigned int* wf_counts = nullptr;
    unsigned int* wf_counts_host = nullptr;  // pinned
    // a frame issued on an estimate whose counters are on their way to wf_counts_host
    bool estimate_pending = false;
    uint64_t estimate_key = 0;
    FrameId estimate_id{};
    int estimate_depth = 0;
    long long estimate_n0 = 0;
    bool estimate_fast = false;
    hipEvent_t done = nullptr;  // recorded behind the last frame issued with this set
    uint64_t last_use = 0;      // tick of that frame: the set used longest ago is the one handed on
  };
  uint64_t wf_tick = 0;
  // Results of estimated frames whose buffer set was handed to another stream before the caller
  // asked (estimated_frame_result): stream -> complete
  std::vector<std::pair<hipStream_t, int>> orphan_results;
  // set around a call by rt_render_device_estimated: unseen frames may be issued on an estimate
  bool allow_estimates = false;
  // camera sweeps (rti::sweep_slot): streams and image buffers of the scene's own, grow only
  struct SweepSlot {
    hipStream_t stream = nullptr;
    uint8_t* device = nullptr;
    uint8_t* pinned = nullptr;
    size_t bytes = 0;
  };
  std::vector<SweepSlot> sweep_slots;
  long long n_estimated = 0, n_estimate_missed = 0;  // frames issued on an estimate / of those, rendered again
  std::vector<WfState*> wf_states;  // at most WF_MAX_STREAMS; beyond that streams share the last one
  // queue lengths of the last frame rendered with host read-backs: the same frame (camera,
  // shard, depth, mode) is then issued again without any synchronisation
  // Wavefront kernels share the triangle tests of a wave's leaves between all its lanes
  // (traverse_meshes_coop): pawn 1080p 7.0 -> 5.5 ms, tea set 1.08 -> 1.01 ms.
  // RT_SHARE_LEAVES=0 selects the per-lane leaf tests instead.
  struct AutoChoice {
    uint64_t key;
    FrameId id;
    int pipeline;
  };
  std::vector<AutoChoice> auto_choice;  // frame -> pipeline picked by RT_PIPELINE_AUTO
  bool share_leaves = true;
  // RT_SHADOW_PASS=1: the light loops of a level's hits run as a pass of their own over a
  // compacted queue (full waves, nothing parked in scratch while a shadow walk runs; the pass
  // rides behind the next level's waves in that level's launch) instead of inside the level
  // kernel.  Same frames.  Off by default: measured on MI355X the queue's traffic and the second
  // set-up of every hit cost more than the idle lanes it removes (pawn 1080p 2.54 vs 2.42 ms,
  // pawn 8K 37.3 vs 35.0 ms, tea set 0.67 vs 0.57 ms, 1M triangles 9.64 vs 9.08 ms with four
  // frames in flight) — waves whose calls all miss skip the fused shadow walk anyway.
  bool shadow_pass = false;
  // Queue lengths of frames rendered with host read-backs: the same frame (camera, shard,
  // depth, mode) is then issued again without any synchronisation.
  struct WfPlan {
    uint64_t key = 0;
    FrameId id{};
    int levels = 0;
    int depth = 0;
    bool fast = false;
    long long n[MAX_DEPTH + 1] = {}, front[MAX_DEPTH + 1] = {};
    long long shadow[MAX_DEPTH + 1] = {};  // hits per level = entries of the level's shadow queue (-1: not known)
    // device, per sorted level: start of every key's run in the level's queue (SORT_BINS entries),
    // left behind by the frame that was rendered with sort passes; owned by the plan list
    unsigned int* place[MAX_DEPTH + 1] = {};
  };
  std::vector<WfPlan> wf_plans;  // most recent last, at most WF_MAX_PLANS
  // A wavefront frame is rendered as frame_split interleaved tile shards on streams of the
  // scene's own, forked from and joined to the caller's stream, so that the thin tail of one
  // shard's level runs under another shard's work (RT_FRAME_SPLIT, 1 = off).
  int frame_split = 0;  // 0: decided per frame (see render_device_fixed)
  // coherence sort of the ray queues (see SortGrid): grid over the world-space bounds of the
  // geometry; unless RT_SORT_RAYS=0 the queues of level >= sort_min_level (RT_SORT_MIN_LEVEL) are kept in key order
  double sort_lo[3] = {0, 0, 0}, sort_inv_cell[3] = {0, 0, 0};
  bool sort_enabled = false;
  int sort_min_level = 1;
  std::vector<hipStream_t> split_streams;
  std::vector<hipStream_t> recent_streams;  // caller streams of the last four wavefront frames
  struct Multi* multi = nullptr;  // rt_scene_commit_multi: replicas on other GPUs, RCCL, buffers
  // per mesh, during rt_scene_commit_multi only: the host builder's result, shared with the replicas
  std::vector<std::shared_ptr<HostBuilt>> host_built;
  bool keep_host_built = false;
  std::vector<hipEvent_t> split_joined;
  hipEvent_t split_fork = nullptr;
};
static constexpr size_t WF_MAX_PLANS = 64;
// counters of one wavefront frame on the device: [2l], [2l+1] reflected / refracted rays queued for
// level l; [WF_OVERFLOW_SLOT] a queue outgrew its estimate; [WF_SHADOW_SLOT0 + l] light loops
// queued by level l
constexpr int WF_OVERFLOW_SLOT = 2 * (MAX_DEPTH + 1);
constexpr int WF_SHADOW_SLOT0 = 2 * (MAX_DEPTH + 2);
constexpr int WF_N_COUNTS = WF_SHADOW_SLOT0 + MAX_DEPTH + 1;
constexpr size_t WF_COUNTS_PAD = 64;  // u32 slots the counters take in their allocation (the tickets follow, 256-byte aligned)
static_assert(WF_N_COUNTS <= (int)WF_COUNTS_PAD, "the counters outgrew their slots");
// a plan that leaves the list takes its device tables with it (the scene's device must be current)
static void drop_plan(rt_scene::WfPlan& p) {
  for (int l = 0; l <= MAX_DEPTH; l++) {
    if (p.place[l]) (void)hipFree(p.place[l]);
    p.place[l] = nullptr;
  }
}

static bool affine_rows(const double* m16, double* out12, std::string& why) {
  if (!(m16[12] == 0.0 && m16[13] == 0.0 && m16[14] == 0.0 && m16[15] == 1.0)) {
    why = "only affine transforms are supported (last row of the inverse must be 0,0,0,1)";
    return false;
  }
  for (int i = 0; i < 12; i++) {
    if (!std::isfinite(m16[i])) {
      why = "non-finite matrix entry";
      return false;
    }
    out12[i] = m16[i];
  }
  return true;
}

// World-space bounds of a closed-form shape: the corners of its object-space bounds
// (sphere.rs:57-76 and cube.rs:48-82 live in [-1,1]^3; cylinder.rs:53-117 has radius 1
// between minimum and maximum) pushed through the inverse of (shape inverse x parent
// inverses), then padded.  Planes, uncapped cylinders and anything numerically doubtful get
// an infinite box, i.e. are never culled.  So do cones: the single-root branch of
// cone.rs:93-95 (ray parallel to one half) reports its hit without the minimum/maximum
// test, anywhere on the infinite double cone.
static void mat4_mul(const double* a, const double* b, double* o) {
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) {
      double v = 0.0;
      for (int k = 0; k < 4; k++) v += a[4 * r + k] * b[4 * k + c];
      o[4 * r + c] = v;
    }
}
static bool affine_inverse(const double* m, double* o) {
  // rows 0..2 hold [A | b]; inverse is [A^-1 | -A^-1 b]
  double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
  double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
  if (!(std::fabs(det) > 1e-300) || !std::isfinite(det)) return false;
  double inv[9] = {(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det,
                   (f * g - d * i) / det, (a * i - c * g) / det, (c * d - a * f) / det,
                   (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
  for (int r = 0; r < 3; r++) {
    for (int k = 0; k < 3; k++) o[4 * r + k] = inv[3 * r + k];
    o[4 * r + 3] = -(inv[3 * r] * m[3] + inv[3 * r + 1] * m[7] + inv[3 * r + 2] * m[11]);
  }
  o[12] = o[13] = o[14] = 0.0;
  o[15] = 1.0;
  return true;
}
static DPrimBox world_box_of(const rt_scene* s, const rt_primitive& P) {
  const double inf = std::numeric_limits<double>::infinity();
  DPrimBox B;
  for (int a = 0; a < 3; a++) {
    B.lo[a] = -inf;
    B.hi[a] = inf;
  }
  double lo[3] = {-1.0, -1.0, -1.0}, hi[3] = {1.0, 1.0, 1.0};
  if (P.kind == RT_PLANE || P.kind == RT_CONE) return B;
  if (P.kind == RT_CYLINDER) {
    if (!std::isfinite(P.minimum) || !std::isfinite(P.maximum)) return B;
    lo[1] = std::fmin(P.minimum, P.maximum);
    hi[1] = std::fmax(P.minimum, P.maximum);
  }
  // world -> object: shape inverse applied after the parents' inverses, root first
  double total[16];
  std::memcpy(total, P.inv, sizeof total);
  for (int g = P.parent; g >= 0; g = s->groups[g].parent) {
    double t[16];
    mat4_mul(total, s->groups[g].inv, t);
    std::memcpy(total, t, sizeof total);
  }
  double fwd[16], chk[16];
  if (!affine_inverse(total, fwd)) return B;
  mat4_mul(total, fwd, chk);
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++)
      if (!(std::fabs(chk[4 * r + c] - (r == c ? 1.0 : 0.0)) < 1e-9)) return B;
  double wlo[3] = {inf, inf, inf}, whi[3] = {-inf, -inf, -inf};
  for (int k = 0; k < 8; k++) {
    double p[3] = {(k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 4) ? hi[2] : lo[2]};
    for (int a = 0; a < 3; a++) {
      double v = fwd[4 * a] * p[0] + fwd[4 * a + 1] * p[1] + fwd[4 * a + 2] * p[2] + fwd[4 * a + 3];
      wlo[a] = std::fmin(wlo[a], v);
      whi[a] = std::fmax(whi[a], v);
    }
  }
  for (int a = 0; a < 3; a++) {
    if (!std::isfinite(wlo[a]) || !std::isfinite(whi[a])) return B;
  }
  for (int a = 0; a < 3; a++) {
    double pad = 1e-6 * ((whi[a] - wlo[a]) + std::fmax(std::fabs(wlo[a]), std::fabs(whi[a]))) + 1e-9;
    B.lo[a] = wlo[a] - pad;
    B.hi[a] = whi[a] + pad;
  }
  return B;
}

// World-space bounds of an object-space box under group g (corners through the inverse of the
// chain of inverse transforms); false when the chain cannot be inverted.
static bool world_bounds_of_box(const rt_scene* s, int g, const double* lo, const double* hi, double* wlo,
                                double* whi) {
  double total[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (; g >= 0; g = s->groups[g].parent) {
    double t[16];
    mat4_mul(total, s->groups[g].inv, t);
    std::memcpy(total, t, sizeof total);
  }
  double fwd[16];
  if (!affine_inverse(total, fwd)) return false;
  for (int a = 0; a < 3; a++) {
    wlo[a] = std::numeric_limits<double>::infinity();
    whi[a] = -std::numeric_limits<double>::infinity();
  }
  for (int k = 0; k < 8; k++) {
    double p[3] = {(k & 1) ? hi[0] : lo[0], (k & 2) ? hi[1] : lo[1], (k & 4) ? hi[2] : lo[2]};
    for (int a = 0; a < 3; a++) {
      double v = fwd[4 * a] * p[0] + fwd[4 * a + 1] * p[1] + fwd[4 * a + 2] * p[2] + fwd[4 * a + 3];
      wlo[a] = std::fmin(wlo[a], v);
      whi[a] = std::fmax(whi[a], v);
    }
  }
  for (int a = 0; a < 3; a++)
    if (!std::isfinite(wlo[a]) || !std::isfinite(whi[a])) return false;
  return true;
}

// ---------------------------------------------------------------------------------------
// kernels

// Lane -> pixel: the shard's k-th tile is global tile tile_first + k*tile_step; a
// 64-thread block (one wave) shades one 8x8 pixel patch of it.
struct PixelMap {
  int hsize, vsize, tiles_x;
  int tile_first, tile_step;  // this launch renders tiles tile_first, tile_first + tile_step, ...
  // Where local tile k of this launch goes: into the row-major image, or to slot
  // k * out_k_mul + out_k_add of a compact buffer of whole tiles (a shard's output).
  int out_row_major, out_k_mul, out_k_add;
};

__device__ __forceinline__ bool lane_pixel(const PixelMap& pm, int& px, int& py, long long& out_index,
                                           long long b = blockIdx.x) {
  constexpr int PATCHES = (RT_TILE / 8) * (RT_TILE / 8);
  long long k = b / PATCHES;  // local tile
  int patch = (int)(b % PATCHES);
  long long tile = pm.tile_first + k * pm.tile_step;
  int ty = (int)(tile / pm.tiles_x), tx = (int)(tile % pm.tiles_x);
  int lane = threadIdx.x;
  int ix = (patch % (RT_TILE / 8)) * 8 + (lane & 7);
  int iy = (patch / (RT_TILE / 8)) * 8 + (lane >> 3);
  px = tx * RT_TILE + ix;
  py = ty * RT_TILE + iy;
  if (pm.out_row_major)
    out_index = (long long)py * pm.hsize + px;
  else
    out_index = (k * pm.out_k_mul + pm.out_k_add) * (RT_TILE * RT_TILE) + iy * RT_TILE + ix;
  return px < pm.hsize && py < pm.vsize;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <bool STATS>
__device__ __forceinline__ void flush_counters(const Counters& c, unsigned long long* g) {
  if (!STATS) return;
  const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&c);
  for (int i = 0; i < N_COUNTERS; i++) {
    unsigned long long s = wave_sum(src[i]);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(&g[i], s);
  }
  if (c.max_nodes) atomicMax(&g[N_COUNTERS], c.max_nodes);
}

// render_thread (camera.rs:165-184): ray_for_pixel + color_at(ray, depth) + write_pixel,
// one lane per pixel, recursion on a per-lane stack.
template <bool STATS>
__global__ void __launch_bounds__(64)
k_render_mega(DScene S, DCamera cam, PixelMap pm, int depth, double* out_f64, uint8_t* out_u8,
              unsigned long long* counters) {
  RT_TRAV_STACK(st, S);
  Counters cnt{};
  int px, py;
  long long oi;
  bool active = lane_pixel(pm, px, py, oi);
  if (active) {
    Ray r = ray_for_pixel(cam, px, py);
    C3 c = color_at<STATS>(S, r, depth, cnt, st);
    if (out_f64) {
      out_f64[oi * 3 + 0] = c.r;
      out_f64[oi * 3 + 1] = c.g;
      out_f64[oi * 3 + 2] = c.b;
    }
    if (out_u8) {
      out_u8[oi * 3 + 0] = quantise(c.r);
      out_u8[oi * 3 + 1] = quantise(c.g);
      out_u8[oi * 3 + 2] = quantise(c.b);
    }
  }
  flush_counters<STATS>(cnt, counters);
}

__global__ void __launch_bounds__(64) k_trace_rays(DScene S, long long n, const double* origins, const double* dirs,
                             int depth, double* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  RT_TRAV_STACK(st, S);
  if (i >= n) return;
  Counters cnt{};
  Ray r = {v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]),
           v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2])};
  double* o = out + 32 * i;
  for (int k = 0; k < 32; k++) o[k] = 0.0;
  HitRec h;
  if (!trace_closest<false>(S, r, h, cnt, st)) return;
  ShadeOut so;
  shade_hit_local<false>(S, r, h, so, cnt, st);
  C3 c = color_at<false>(S, r, depth, cnt, st);
  o[0] = 1.0;
  o[1] = h.t;
  o[2] = (double)h.order;
  o[3] = h.u;
  o[4] = h.v;
  o[5] = so.inside ? 1.0 : 0.0;
  o[6] = so.n1;
  o[7] = so.n2;
  const V3* vs[6] = {&so.point, &so.over_point, &so.under_point, &so.eyev, &so.normalv, &so.reflectv};
  for (int k = 0; k < 6; k++) {
    o[8 + 3 * k] = vs[k]->x;
    o[9 + 3 * k] = vs[k]->y;
    o[10 + 3 * k] = vs[k]->z;
  }
  o[26] = so.reflectance;
  o[27] = c.r;
  o[28] = c.g;
  o[29] = c.b;
  if (S.n_lights > 0) {
    V3 lpos = v3(S.lights[0].pos[0], S.lights[0].pos[1], S.lights[0].pos[2]);
    V3 v = vsub(lpos, so.over_point);
    double distance = vmag(v);
    Ray sr = {so.over_point, v3(v.x / distance, v.y / distance, v.z / distance)};
    o[30] = trace_any<false>(S, sr, distance, cnt, st) ? 1.0 : 0.0;
  }
}

// stitch of tile shards into the row-major image (canvas.rs:47-64 for tiles)
template <typename T>
__global__ void k_detile(const T* tiles, int nshards, long long tiles_per_shard, int hsize,
                         int vsize, int tiles_x, T* image) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // pixel index
  long long npx = (long long)hsize * vsize;
  if (i >= npx) return;
  int px = (int)(i % hsize), py = (int)(i / hsize);
  long long tile = (long long)(py / RT_TILE) * tiles_x + (px / RT_TILE);
  int shard = (int)(tile % nshards);
  long long k = tile / nshards;
  long long src = ((long long)shard * tiles_per_shard + k) * (RT_TILE * RT_TILE) +
                  (py % RT_TILE) * RT_TILE + (px % RT_TILE);
  image[3 * i + 0] = tiles[3 * src + 0];
  image[3 * i + 1] = tiles[3 * src + 1];
  image[3 * i + 2] = tiles[3 * src + 2];
}


// =======================================================================================
// Wavefront pipeline.  The recursion tree of color_at (world.rs:67-114) is evaluated level
// by level: level L holds every color_at call whose remaining depth is depth-L.  One lane
// per call: trace, shade, append the reflected / refracted child rays to the next level's
// queue.  When the deepest level is done the results flow back up, one combine pass per
// level, each node evaluating exactly the expression of world.rs:57-64 on its children's
// results — so the colours are the same bits as the recursive evaluation, while expensive
// pixels (glass: ~18 rays) are spread over many lanes instead of serialising one.
struct alignas(16) WfTask {
  double o[3], d[3];
  double coef;     // what the parent multiplies this call's colour by (reflective / transparency)
  int32_t parent;  // node index in the previous level
  uint32_t slot;   // 0: the parent's reflected colour, 1: its refracted colour
};
struct alignas(16) WfNode {
  double surface[3], reflected[3], refracted[3];
  double reflectance, coef;
  int32_t parent;  // previous-level node, or the output pixel index at level 0 (-1: no pixel)
  uint32_t flags;  // bit0: hit (otherwise the call returned black), bit1: Schlick mix, bit2: slot
};
// One shade_hit call's light loop (world.rs:43-52), queued by the level kernel for the shadow
// pass: where the shadow rays start, what lighting() needs (light.rs:23-85), and where the
// summed surface colour goes — into the call's result node when children will add to it, or,
// with the rest of world.rs:57-64 evaluated on it, straight to the pixel / the parent's slot.
struct alignas(16) WfShadow {
  double over[3], normal[3], eyev[3], pc[3];
  double coef, reflectance;
  int32_t mat;
  uint32_t node;   // index of the call in its level
  int32_t parent;  // as WfNode::parent
  uint32_t flags;  // WfNode::flags, bit3: the call has children (its node is stored)
};
static_assert(sizeof(WfShadow) == 128, "two cache lines per queued light loop");
constexpr int WF_BLOCK = 64;

#ifndef WF_WAVES
#define WF_WAVES 4
#endif
// One node on the way back up: ret = shade_hit's final expression (world.rs:57-64); level 0
// writes the pixel (canvas.rs:21 + color.rs:148-155), other levels hand `ret * coef`
// (world.rs:85 / 112) to the parent's slot.
template <bool LEVEL0>
__device__ __forceinline__ void combine_node(const WfNode& N, WfNode* parents, double* out_f64,
                                             uint8_t* out_u8) {
  if (N.parent < 0) return;
  C3 ret = {0.0, 0.0, 0.0};
  bool hit = N.flags & 1u;
  if (hit) {
    C3 s = {N.surface[0], N.surface[1], N.surface[2]};
    C3 rl = {N.reflected[0], N.reflected[1], N.reflected[2]};
    C3 rr = {N.refracted[0], N.refracted[1], N.refracted[2]};
    ret = combine(s, rl, rr, (N.flags & 2u) != 0, N.reflectance);
  }
  if (LEVEL0) {
    long long oi = N.parent;
    if (out_f64) {
      out_f64[oi * 3 + 0] = ret.r;
      out_f64[oi * 3 + 1] = ret.g;
      out_f64[oi * 3 + 2] = ret.b;
    }
    if (out_u8) {
      out_u8[oi * 3 + 0] = quantise(ret.r);
      out_u8[oi * 3 + 1] = quantise(ret.g);
      out_u8[oi * 3 + 2] = quantise(ret.b);
    }
  } else if (hit) {
    WfNode& P = parents[N.parent];
    double* dst = (N.flags & 4u) ? P.refracted : P.reflected;
    dst[0] = ret.r * N.coef;
    dst[1] = ret.g * N.coef;
    dst[2] = ret.b * N.coef;
  }
}

// Coherence sort of a level's ray queue.  The calls of level >= 1 arrive in the order the
// parents' atomics happened to run; a wave of 64 consecutive tasks then holds rays that start
// all over the scene and walk unrelated subtrees (the fan-out of world.rs:78-114).  Every task
// gets a key when it is appended — ray kind, direction octant, Morton code of the origin's cell
// in a grid over the world-space bounds of the geometry — and a counting sort over the keys
// (histogram in the emitting kernel, one scan block, one scatter pass) yields the order the
// next level's kernel takes its tasks in.  The order changes which lanes share a wave, never a
// result: every call writes into its own node and its parent's slot.
#ifndef RT_SORT_CELL_BITS
#define RT_SORT_CELL_BITS 4  // per axis
#endif
#ifndef RT_SORT_DIR_BITS
#define RT_SORT_DIR_BITS 3  // 3: direction octant; 7: cube-map face and a 4 x 4 grid on it
#endif
constexpr int SORT_CELL_BITS = RT_SORT_CELL_BITS;
constexpr int SORT_DIR_BITS = RT_SORT_DIR_BITS;
constexpr int SORT_KEY_BITS = 3 * SORT_CELL_BITS + SORT_DIR_BITS + 1;  // kind, direction, cell
constexpr unsigned int SORT_BINS = 1u << SORT_KEY_BITS;
static_assert(SORT_BINS >= 1024 && SORT_KEY_BITS <= 20, "the scan block handles 2^10 .. 2^20 bins");
struct SortGrid {
  double lo[3], inv_cell[3];
  int enabled;
};
__device__ __forceinline__ unsigned int spread3(unsigned int v) {  // bits of v to every third position
  unsigned int r = 0;
#pragma unroll
  for (int b = 0; b < SORT_CELL_BITS; b++) r |= ((v >> b) & 1u) << (3 * b);
  return r;
}
// refracted rays (inside glass: the longest walks) get the low keys and are dispatched first
__device__ __forceinline__ unsigned int sort_key(const SortGrid& g, const Ray& r, unsigned int reflected) {
  constexpr double top = (double)((1 << SORT_CELL_BITS) - 1);
  const double cx = fmin(fmax((r.o.x - g.lo[0]) * g.inv_cell[0], 0.0), top);
  const double cy = fmin(fmax((r.o.y - g.lo[1]) * g.inv_cell[1], 0.0), top);
  const double cz = fmin(fmax((r.o.z - g.lo[2]) * g.inv_cell[2], 0.0), top);
  const unsigned int cell = spread3((unsigned int)(int)cx) | (spread3((unsigned int)(int)cy) << 1) |
                            (spread3((unsigned int)(int)cz) << 2);
  unsigned int dir;
  if (SORT_DIR_BITS == 3) {
    dir = (r.d.x < 0.0 ? 1u : 0u) | (r.d.y < 0.0 ? 2u : 0u) | (r.d.z < 0.0 ? 4u : 0u);
  } else {
    const float dx = (float)r.d.x, dy = (float)r.d.y, dz = (float)r.d.z;
    const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
    float m, u, v;
    unsigned int face;
    if (ax >= ay && ax >= az) {
      m = ax; u = dy; v = dz; face = dx < 0.f ? 1u : 0u;
    } else if (ay >= az) {
      m = ay; u = dx; v = dz; face = dy < 0.f ? 3u : 2u;
    } else {
      m = az; u = dx; v = dy; face = dz < 0.f ? 5u : 4u;
    }
    const float s = 2.0f / fmaxf(m, 1e-30f);  // u/m in [-1, 1] -> [0, 4)
    const int qu = min(max((int)(u * s + 2.0f), 0), 3), qv = min(max((int)(v * s + 2.0f), 0), 3);
    dir = (face << 4) | ((unsigned int)qu << 2) | (unsigned int)qv;
  }
  return (reflected << (SORT_KEY_BITS - 1)) | (dir << (3 * SORT_CELL_BITS)) | cell;
}
// One histogram update per distinct key among the wave's active lanes instead of one per lane:
// rays of one wave mostly share a key, and atomics of many lanes on one address serialise.
// (The loop runs over a mask of lanes still to serve, which every iteration shrinks; a loop
// whose only exit is `key == readfirstlane(key)` has no per-lane progress and is compiled away.)
__device__ __forceinline__ void hist_add(unsigned int* hist, unsigned int key) {
  unsigned long long todo = __ballot(1);
  const int lane = (int)__lane_id();
  while (todo) {  // uniform over the lanes that came in
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned int k = (unsigned int)__builtin_amdgcn_readlane((int)key, leader);
    const unsigned long long same = __ballot(key == k);
    if (lane == leader) atomicAdd(&hist[k], (unsigned int)__popcll(same));
    todo &= ~same;
  }
}
// A frame rendered before with its queues sorted has left, per level, where every key's run of
// tasks starts (the counting sort's exclusive prefix).  The same frame again needs no sort
// passes: the emitting kernel writes each task straight to its place in key order — start of the
// key's run plus a ticket from the key's cursor — and the next level reads its queue front to back.
struct Placement {
  const unsigned int* off;  // next level: start of every key's run (null: tasks are appended as they come)
  unsigned int* cur;        // next level: tickets per key, zero when the frame starts
  int ordered_in;           // this level's queue is already in key order: task i is call i
};
// The lanes among `want` that share this lane's key (0 for a lane outside `want`).  Called by all
// the lanes of a converged region.
__device__ __forceinline__ unsigned long long lanes_with_my_key(bool want, unsigned int key) {
  unsigned long long todo = __ballot(want);
  unsigned long long mine = 0ull;
  while (todo) {  // uniform
    const int first = __ffsll((long long)todo) - 1;
    const unsigned int k = (unsigned int)__builtin_amdgcn_readlane((int)key, first);
    const unsigned long long same = __ballot(want && key == k);
    if (want && key == k) mine = same;
    todo &= ~same;
  }
  return mine;
}
// The places of a call's two child rays.  The lanes that share a key take their tickets with one
// atomic of the group's first lane; the groups are found with ballots and lane reads alone, so
// all of them — of the reflected and of the refracted rays — issue their atomics before any is
// waited for: one memory round trip per call on the wave's dependent chain (a round trip per
// group, as first written, cost the pawn frame 2 %).
__device__ __forceinline__ void place_two(const Placement& P, bool want_a, unsigned int key_a, bool want_b,
                                          unsigned int key_b, unsigned int& pos_a, unsigned int& pos_b) {
  const int lane = (int)__lane_id();
  const unsigned long long mine_a = lanes_with_my_key(want_a, key_a), mine_b = lanes_with_my_key(want_b, key_b);
  const int lead_a = want_a ? __ffsll((long long)mine_a) - 1 : lane, lead_b = want_b ? __ffsll((long long)mine_b) - 1 : lane;
  unsigned int start_a = 0, start_b = 0, base_a = 0, base_b = 0;
  if (want_a) start_a = P.off[key_a];
  if (want_b) start_b = P.off[key_b];
  if (want_a && lane == lead_a) base_a = atomicAdd(&P.cur[key_a], (unsigned int)__popcll(mine_a));
  if (want_b && lane == lead_b) base_b = atomicAdd(&P.cur[key_b], (unsigned int)__popcll(mine_b));
  base_a = (unsigned int)__shfl((int)base_a, lead_a, 64);
  base_b = (unsigned int)__shfl((int)base_b, lead_b, 64);
  const unsigned long long below = (1ull << lane) - 1ull;
  pos_a = start_a + base_a + (unsigned int)__popcll(mine_a & below);
  pos_b = start_b + base_b + (unsigned int)__popcll(mine_b & below);
}
// exclusive prefix sum of the histogram into `cursor` (one block); leaves the histogram zeroed
__global__ void __launch_bounds__(1024) k_wf_sort_scan(unsigned int* hist, unsigned int* cursor) {
  constexpr int PER = SORT_BINS / 1024;
  __shared__ unsigned int part[1024];
  const int t = threadIdx.x;
  unsigned int local[PER];
  unsigned int sum = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    local[k] = hist[t * PER + k];
    hist[t * PER + k] = 0u;
    sum += local[k];
  }
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    unsigned int v = (t >= off) ? part[t - off] : 0u;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  unsigned int run = part[t] - sum;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    cursor[t * PER + k] = run;
    run += local[k];
  }
}
// position in key order -> slot of the task in the queue (reflected rays fill it from the front,
// refracted rays from the back); the order inside one bin is whatever the atomics give
__global__ void k_wf_sort_scatter(const unsigned int* keys, long long n, long long n_front, unsigned int cap,
                                  unsigned int* cursor, unsigned int* perm, const unsigned int* counts_in) {
  if (counts_in) {  // a frame issued on an estimate: the true lengths are on the device
    const long long have_front = counts_in[0], have = have_front + counts_in[1];
    if (have > n) return;  // the level kernel raises the overflow flag
    n = have;
    n_front = have_front;
  }
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned int slot = (i < n_front) ? (unsigned int)i : cap - 1u - (unsigned int)(i - n_front);
  const unsigned int key = keys[slot];
  // one atomic per distinct key among the wave's lanes; each lane takes its rank inside the group
  unsigned long long todo = __ballot(1);
  const int lane = (int)__lane_id();
  unsigned int pos = 0;
  while (todo) {  // uniform over the lanes that came in
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned int k = (unsigned int)__builtin_amdgcn_readlane((int)key, leader);
    const unsigned long long same = __ballot(key == k);
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(&cursor[k], (unsigned int)__popcll(same));
    base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
    if (key == k) pos = base + (unsigned int)__popcll(same & ((1ull << lane) - 1ull));
    todo &= ~same;
  }
  perm[pos] = slot;
}

// Fills the level's result node and appends the child rays to the next level's queue:
// reflected rays from the front, refracted rays from the back (world.rs:78-114).
template <bool STATS>
__device__ __forceinline__ bool emit_result(const ShadeOut& so, const C3& surface, WfNode& nd, long long i,
                                            WfTask* next, unsigned int* next_count, unsigned int next_cap,
                                            int spawn, Counters& cnt, const SortGrid& grid,
                                            unsigned int* next_keys, unsigned int* hist, const Placement& place) {
  nd.surface[0] = surface.r;
  nd.surface[1] = surface.g;
  nd.surface[2] = surface.b;
  nd.reflectance = so.reflectance;
  nd.flags |= 1u | (so.use_schlick ? 2u : 0u);
  for (int k = 0; k < 3; k++) {
    nd.reflected[k] = 0.0;
    nd.refracted[k] = 0.0;
  }
  // (placed: the queue holds exactly the tasks of the frame that left the plan behind)
  unsigned int placed_reflect = 0, placed_refract = 0;
  if (place.off && spawn)
    place_two(place, so.do_reflect, so.do_reflect ? sort_key(grid, so.reflect_ray, 1u) : 0u, so.do_refract,
              so.do_refract ? sort_key(grid, so.refract_ray, 0u) : 0u, placed_reflect, placed_refract);
  if (so.do_reflect) {
    if (spawn) {
      const unsigned int idx = place.off ? placed_reflect : atomicAdd(next_count, 1u);
      if (idx < next_cap) {
        WfTask& T = next[idx];
        T.o[0] = so.reflect_ray.o.x;
        T.o[1] = so.reflect_ray.o.y;
        T.o[2] = so.reflect_ray.o.z;
        T.d[0] = so.reflect_ray.d.x;
        T.d[1] = so.reflect_ray.d.y;
        T.d[2] = so.reflect_ray.d.z;
        T.coef = so.reflective;
        T.parent = (int32_t)i;
        T.slot = 0;
        if (next_keys) {
          const unsigned int key = sort_key(grid, so.reflect_ray, 1u);
          next_keys[idx] = key;
          hist_add(hist, key);
        }
      }
    } else if (STATS) {
      cnt.depth0_skipped++;
    }
  }
  if (so.do_refract) {
    if (spawn) {
      const unsigned int idx = place.off ? placed_refract : atomicAdd(next_count + 1, 1u);
      if (idx < next_cap) {
        WfTask& T = next[place.off ? idx : next_cap - 1 - idx];
        T.o[0] = so.refract_ray.o.x;
        T.o[1] = so.refract_ray.o.y;
        T.o[2] = so.refract_ray.o.z;
        T.d[0] = so.refract_ray.d.x;
        T.d[1] = so.refract_ray.d.y;
        T.d[2] = so.refract_ray.d.z;
        T.coef = so.transparency;
        T.parent = (int32_t)i;
        T.slot = 1;
        if (next_keys) {
          const unsigned int key = sort_key(grid, so.refract_ray, 0u);
          next_keys[next_cap - 1 - idx] = key;
          hist_add(hist, key);
        }
      }
    } else if (STATS) {
      cnt.depth0_skipped++;
    }
  }
  return spawn && (so.do_reflect || so.do_refract);  // children will add to this node later
}

#ifdef RT_SECTION_TIMES
// Diagnostic build: wave cycles per section of k_wf_level, summed over all waves.
__device__ unsigned long long g_section_cycles[8];
// (summed per wave in scalar registers and added to the totals once, when the wave ends: an
// atomic at every mark would make the next section wait for it)
#define RT_SECTION(k)                                                              \
  do {                                                                             \
    const unsigned long long now_ = __builtin_readcyclecounter();                  \
    section_sum_[k] += now_ - section_t_;                                          \
    section_t_ = now_;                                                             \
  } while (0)
#define RT_SECTION_FLUSH()                                                         \
  do {                                                                             \
    if (threadIdx.x == 0)                                                          \
      for (int k_ = 0; k_ < 8; k_++)                                               \
        if (section_sum_[k_]) atomicAdd(&g_section_cycles[k_], section_sum_[k_]);  \
  } while (0)
#else
#define RT_SECTION_FLUSH() do { } while (0)
#define RT_SECTION(k) do { } while (0)
#endif

// The hardware deals consecutive workgroups of a launch round-robin to the 8 XCDs, each
// with an L2 of its own.  Taken as they come, the 16 waves of one 32x32 tile (and the runs of
// neighbouring rays of a deeper level) would be spread over all eight L2s, and each would
// fetch the same nodes and triangles.  This permutation of the block index gives every XCD
// runs of RT_XCD_CHUNK consecutive waves instead: within each group of 8 * CHUNK blocks, the
// blocks that land on XCD x (index % 8 == x) take the x-th run.  The incomplete group at the
// end of a launch keeps its order.  0 switches the permutation off.
#ifndef RT_XCD_CHUNK
#define RT_XCD_CHUNK 16
#endif
__device__ __forceinline__ long long wf_block_index() {
  const long long b = blockIdx.x;
  if (RT_XCD_CHUNK == 0) return b;
  if (RT_XCD_CHUNK < 0) {  // one run per XCD: an eighth of the launch each
    const long long run = gridDim.x / 8;
    return b < run * 8 ? (b % 8) * run + b / 8 : b;
  }
  constexpr long long G = 8ll * (RT_XCD_CHUNK > 0 ? RT_XCD_CHUNK : 1);
  const long long g = b / G;
  if ((g + 1) * G > (long long)gridDim.x) return b;
  const long long r = b - g * G;
  return g * G + (r % 8) * (G / 8) + r / 8;
}

// SHQ: the light loop of every hit is queued for k_wf_shadow instead of being run here.
template <bool STATS, bool PRIMARY, bool FAST, bool COOP, bool SHQ>
__device__ __forceinline__ void
wf_level_body(const DScene& S, const DCamera& cam, const PixelMap& pm, const WfTask* tasks, long long n, long long n_front,
              unsigned int cap, WfNode* nodes, uint8_t* live, WfNode* parents, double* out_f64, uint8_t* out_u8,
              WfTask* next,
              unsigned int* next_count, unsigned int next_cap, int spawn, unsigned long long* counters,
              const unsigned int* counts_in, unsigned int* overflow, const SortGrid& grid, const unsigned int* perm,
              unsigned int* next_keys, unsigned int* hist, const Placement& place, WfShadow* shadow_q,
              unsigned int* shadow_count, TravStack& st, const long long block) {
  static_assert(WF_BLOCK == 64, "the traversal stack is laid out for 64-lane blocks");
  static_assert(COOP || !FAST, "the f32 search exists on the shared leaf tests only");
  static_assert(COOP || !SHQ, "the shadow pass exists on the cooperative kernels only");
#ifdef RT_SECTION_TIMES
  unsigned long long section_sum_[8] = {};
  unsigned long long section_t_ = __builtin_readcyclecounter();
#endif
  Counters cnt{};
  long long i = block * WF_BLOCK + threadIdx.x;
  bool active;
  Ray ray = {v3(0.0, 0.0, 0.0), v3(0.0, 0.0, 1.0)};
  WfNode nd;
  nd.flags = 0;
  nd.coef = 0.0;
  nd.parent = -1;
  bool has_children = false;
  if (PRIMARY) {
    int px, py;
    long long oi;
    active = lane_pixel(pm, px, py, oi, block);
    if (active) {
      ray = ray_for_pixel(cam, px, py);
      nd.parent = (int32_t)oi;
      if (STATS) cnt.rays_primary++;
    }
  } else {
    // A frame issued on an estimate (see render_wavefront) learns the queue's length here, from
    // the counters the previous level left on the device; `n` is then what the grid and the
    // buffers were sized for.  A queue longer than that cannot be rendered by this launch: the
    // level is skipped and the frame marked as incomplete for the host to render again.
    if (counts_in) {
      const long long have_front = counts_in[0], have = have_front + counts_in[1];
      if (have > n) {
        if (i == 0) *overflow = 1u;
        n = 0;
      } else {
        n = have;
        n_front = have_front;
      }
      if (block * WF_BLOCK >= n) return;  // wave-uniform
    }
    active = i < n;
    if (active) {
      // In key order when the queue has been sorted.  Otherwise: reflected rays fill the queue
      // from the front, refracted rays from the back, and the refracted ones (inside glass: the
      // longest walks) take the lowest indices, so they are dispatched first and the level's
      // tail is made of cheap rays.
      const long long n_back = n - n_front;
      const WfTask& T = tasks[place.ordered_in ? i : perm ? (long long)perm[i] : (i < n_back ? (long long)cap - 1 - i : i - n_back)];
      ray = {v3(T.o[0], T.o[1], T.o[2]), v3(T.d[0], T.d[1], T.d[2])};
      nd.parent = T.parent;
      nd.coef = T.coef;
      nd.flags = T.slot << 2;
      if (STATS) {
        if (T.slot)
          cnt.rays_refract++;
        else
          cnt.rays_reflect++;
      }
    }
  }
  if (COOP) {
    // Every lane of the wave goes through the three queries together (the lanes share the
    // triangle tests of each other's leaves); `active`, `hit` and `need` say who has one.
    HitRec h;
    RT_SECTION(0);
    const bool hit = trace_closest_coop<STATS, FAST>(S, ray, active, h, cnt, st);
    RT_SECTION(1);
    // The container query comes first: it needs only the ray and the hit, so nothing of the
    // shading state is alive (and parked in scratch) while it walks.
    int mat_index = 0;
    if (hit) mat_index = h.prim >= 0 ? S.prims[h.prim].material : S.meshes[h.mesh].material;
    // n1 / n2 decide the refracted ray (world.rs:90-112) and weigh the two child colours
    // (schlick, world.rs:151-165).  At the deepest level the children are color_at(ray, 0) —
    // black whatever their rays — and black times any reflectance in [0, 1] is +0: the call's
    // value is the same bits without them, so the walk that finds them is left out there.  (The
    // kernels with counters keep it: total internal reflection decides whether a depth-0 child
    // is counted.)
    const bool need = hit && needs_n1_n2(S.mats[mat_index]) && (STATS || spawn != 0);
    double n1 = 1.0, n2 = 1.0;
    // (... and leave its work out of their counts, which describe what the other kernels do)
    Counters uncounted{};
    if (STATS && spawn == 0) uncounted = cnt;
    {
      Key K{0.0, 0u, 0}, tb{0.0, 0u, 0};
      int tm = -1;
      if (need) {
        if (STATS) cnt.pred_queries++;
        K = trigger_key(S, ray, h);
      }
      bool tf = false;
      RT_SECTION(2);
      if (S.n_meshes > 0)
        tf = tri_predecessor_coop<STATS, FAST>(S, ray, predecessor_bound(K), h.prim >= 0 ? -1 : h.tri, need, tb,
                                               tm, cnt, st);
      RT_SECTION(4);
      if (need) containers_from<STATS>(S, ray, h, K, S.mats[mat_index].refractive_index, tf, tb, tm, n1, n2, cnt);
    }
    if (STATS && spawn == 0) cnt = uncounted;
    Prep pr;
    pr.mat_index = 0;
    C3 surface = {0.0, 0.0, 0.0};
    C3 pc = {0.0, 0.0, 0.0};
    // Everything of a call that does not hang on the shadow queries is done before them: the
    // child rays (they depend on the optics alone, world.rs:78-114) are queued, and each light's
    // contribution is evaluated both ways (lit and shadowed), so that a query leaves one
    // select and three additions behind and nothing of the hit has to live through its walk.
    V3 eyev = {0.0, 0.0, 0.0}, normal = {0.0, 0.0, 0.0}, over = {0.0, 0.0, 0.0};
    if (hit) {
      if (STATS) cnt.shade_calls++;
      pr = prepare_hit(S, ray, h);
      pc = hit_pattern_color(S, S.mats[pr.mat_index], pr);
      eyev = vneg(ray.d);
      normal = pr.normal;
      over = pr.over;
      ShadeOut so;
      finish_optics(S.mats[pr.mat_index], ray, pr, n1, n2, so);
      has_children = emit_result<STATS>(so, surface, nd, i, next, next_count, next_cap, spawn, cnt, grid, next_keys, hist, place);
    }
    if (SHQ) {
      // The light loop goes to the shadow pass: the hits of the wave take consecutive slots of
      // the level's shadow queue (one atomic per wave), so that pass walks with full waves and
      // nothing of this call waits, parked, for a shadow walk.
      const unsigned long long hm = __ballot(hit);
      if (hm != 0ull) {  // wave-uniform
        const int lane = (int)__lane_id();
        const int leader = __ffsll((long long)hm) - 1;
        unsigned int base = 0;
        if (lane == leader) base = atomicAdd(shadow_count, (unsigned int)__popcll(hm));
        base = (unsigned int)__shfl((int)base, leader, 64);
        if (hit) {
          WfShadow& Q = shadow_q[base + (unsigned int)__popcll(hm & ((1ull << lane) - 1ull))];
          Q.over[0] = over.x; Q.over[1] = over.y; Q.over[2] = over.z;
          Q.normal[0] = normal.x; Q.normal[1] = normal.y; Q.normal[2] = normal.z;
          Q.eyev[0] = eyev.x; Q.eyev[1] = eyev.y; Q.eyev[2] = eyev.z;
          Q.pc[0] = pc.r; Q.pc[1] = pc.g; Q.pc[2] = pc.b;
          Q.coef = nd.coef;
          Q.reflectance = nd.reflectance;
          Q.mat = pr.mat_index;
          Q.node = (uint32_t)i;
          Q.parent = nd.parent;
          Q.flags = nd.flags | (has_children ? 8u : 0u);
        }
      }
    } else {
      for (int li = 0; li < S.n_lights; li++) {
        const DLight& L = S.lights[li];
        Ray sr = {over, v3(0.0, 0.0, 1.0)};
        double distance = 0.0;
        C3 lit = {0.0, 0.0, 0.0}, dark = {0.0, 0.0, 0.0};
        if (hit) {
          shadow_ray(L, over, sr, distance);
          if (STATS) cnt.rays_shadow++;
          light_terms(S.mats[pr.mat_index], L, pc, sr.d, normal, eyev, lit, dark);
        }
        RT_SECTION(2);
        const bool shadow = trace_any_coop<STATS, FAST>(S, sr, distance, hit, cnt, st);
        RT_SECTION(3);
        if (hit) {
          surface.r += shadow ? dark.r : lit.r;
          surface.g += shadow ? dark.g : lit.g;
          surface.b += shadow ? dark.b : lit.b;
        }
      }
#ifdef RT_SECTION_TIMES
      {  // how much of the shadow walks runs in waves with few hits (slots 6, 7: under 16, under 32 of 64)
        const int nh = __popcll(__ballot(hit));
        if (nh < 16) section_sum_[6] += section_sum_[3];
        if (nh < 32) section_sum_[7] += section_sum_[3];
      }
#endif
      if (hit) {
        nd.surface[0] = surface.r;
        nd.surface[1] = surface.g;
        nd.surface[2] = surface.b;
      }
    }
  } else {
    if (active) {
      HitRec h;
      if (trace_closest<STATS>(S, ray, h, cnt, st)) {
        ShadeOut so;
        shade_hit_local<STATS>(S, ray, h, so, cnt, st, spawn != 0);
        has_children = emit_result<STATS>(so, so.surface, nd, i, next, next_count, next_cap, spawn, cnt, grid, next_keys, hist, place);
      }
    }
  }
  // A call without children (a miss, a surface that spawns nothing, the deepest level) has its
  // final value now: it goes straight to the pixel or into the parent's slot — the expression
  // k_wf_combine would evaluate on the stored node — and the way back up skips the node.
  if (has_children) {
    nodes[i] = nd;
  } else if ((PRIMARY || active) && !(SHQ && (nd.flags & 1u))) {  // (a queued hit is finished by k_wf_shadow)
    combine_node<PRIMARY>(nd, parents, out_f64, out_u8);
  }
  // one dense byte per call tells the level's combine pass which nodes to touch at all
  if (spawn && (PRIMARY || active)) live[i] = has_children ? 1 : 0;
  RT_SECTION(5);
  RT_SECTION_FLUSH();
  flush_counters<STATS>(cnt, counters);
}

// One level of the way back up (see k_wf_combine below).
template <bool LEVEL0>
__global__ void k_wf_combine(const WfNode* nodes, const uint8_t* live, long long n, WfNode* parents,
                             double* out_f64, uint8_t* out_u8, const unsigned int* counts_in,
                             const unsigned int* overflow) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // A frame issued on an estimate: the level's true length is on the device; after a level
  // has outgrown its estimate the nodes of the levels behind it are not this frame's.
  if (overflow && *overflow) return;
  if (counts_in) n = min(n, (long long)counts_in[0] + (long long)counts_in[1]);
  if (i >= n) return;
  if (!live[i]) return;  // no children: the level kernel has written this call's value already
  combine_node<LEVEL0>(nodes[i], parents, out_f64, out_u8);
}


// The shadow pass of one level: one lane per queued light loop, full waves.  Per light, in the
// scene's order (world.rs:43-52): the shadow ray of is_shadowed (world.rs:116-134), the any-hit
// query, lighting() with its answer — the expressions and their order are those of the fused
// kernel, so the surface colour is the same bits.  It then goes where the fused kernel would
// have put it.  `block`: this wave's index within the pass.
struct ShadowPass {
  const WfShadow* q;
  const unsigned int* count;  // left by the level kernel that queued the loops
  WfNode* nodes;              // the level's result nodes
  WfNode* parents;            // the nodes of the level above (null at level 0)
  int level0;                 // the level above is the image
};
template <bool STATS, bool FAST>
__device__ __forceinline__ void wf_shadow_body(const DScene& S, const ShadowPass& P, double* out_f64, uint8_t* out_u8,
                                               unsigned long long* counters, TravStack& st, const long long block) {
  Counters cnt{};
  const WfShadow* q = P.q;
  const long long n = *P.count;
  if (block * WF_BLOCK >= n) return;  // wave-uniform
  const long long i = block * WF_BLOCK + threadIdx.x;
  const bool active = i < n;
  // Only the shadow ray and the running sum live through a walk; what lighting() needs is read
  // from the queue again when the walk is over (add_light with the query's answer evaluates the
  // expressions light_terms evaluates both ways: the same bits).
  C3 surface = {0.0, 0.0, 0.0};
  for (int li = 0; li < S.n_lights; li++) {
    const DLight& L = S.lights[li];
    Ray sr = {v3(0.0, 0.0, 0.0), v3(0.0, 0.0, 1.0)};
    double distance = 0.0;
    if (active) {
      const WfShadow& Q = q[i];
      shadow_ray(L, v3(Q.over[0], Q.over[1], Q.over[2]), sr, distance);
      if (STATS) cnt.rays_shadow++;
    }
    const bool shadow = trace_any_coop<STATS, FAST>(S, sr, distance, active, cnt, st);
    if (active) {
      const WfShadow& Q = q[i];
      const C3 pc = {Q.pc[0], Q.pc[1], Q.pc[2]};
      add_light(S.mats[Q.mat], L, pc, sr.d, v3(Q.normal[0], Q.normal[1], Q.normal[2]),
                v3(Q.eyev[0], Q.eyev[1], Q.eyev[2]), shadow, surface);
    }
  }
  if (active) {
    const WfShadow& Q = q[i];
    if (Q.flags & 8u) {  // children will add to the node: only the surface colour is missing
      WfNode& N = P.nodes[Q.node];
      N.surface[0] = surface.r;
      N.surface[1] = surface.g;
      N.surface[2] = surface.b;
    } else {  // the call's final value: shade_hit's last expression on a node without children
      WfNode nd;
      nd.surface[0] = surface.r;
      nd.surface[1] = surface.g;
      nd.surface[2] = surface.b;
      for (int k = 0; k < 3; k++) {
        nd.reflected[k] = 0.0;
        nd.refracted[k] = 0.0;
      }
      nd.reflectance = Q.reflectance;
      nd.coef = Q.coef;
      nd.parent = Q.parent;
      nd.flags = Q.flags;
      if (P.level0)
        combine_node<true>(nd, P.parents, out_f64, out_u8);
      else
        combine_node<false>(nd, P.parents, out_f64, out_u8);
    }
  }
  flush_counters<STATS>(cnt, counters);
}

// One recursion level.  The launch's first `level_blocks` waves run the level's calls; with SHQ
// the waves behind them run the shadow pass of the level BEFORE (its light loops were queued by
// that level's launch), so the pass costs no launch of its own and its short, even walks fill
// the end of this launch, where the level's longest walks would otherwise run alone.
template <bool STATS, bool PRIMARY, bool FAST, bool COOP, bool SHQ>
__global__ void __launch_bounds__(WF_BLOCK, WF_WAVES)
k_wf_level(DScene S, DCamera cam, PixelMap pm, const WfTask* tasks, long long n, long long n_front,
           unsigned int cap, WfNode* nodes, uint8_t* live, WfNode* parents, double* out_f64, uint8_t* out_u8,
           WfTask* next,
           unsigned int* next_count, unsigned int next_cap, int spawn, unsigned long long* counters,
           const unsigned int* counts_in, unsigned int* overflow, SortGrid grid, const unsigned int* perm,
           unsigned int* next_keys, unsigned int* hist, Placement place, WfShadow* shadow_q,
           unsigned int* shadow_count, ShadowPass before, long long level_blocks) {
  RT_TRAV_STACK_COOP(st, S);
  const long long block = wf_block_index();
  if (!SHQ || PRIMARY || block < level_blocks)
    wf_level_body<STATS, PRIMARY, FAST, COOP, SHQ>(S, cam, pm, tasks, n, n_front, cap, nodes, live, parents, out_f64,
                                                   out_u8, next, next_count, next_cap, spawn, counters, counts_in,
                                                   overflow, grid, perm, next_keys, hist, place, shadow_q, shadow_count,
                                                   st, block);
  else
    wf_shadow_body<STATS, FAST>(S, before, out_f64, out_u8, counters, st, block - level_blocks);
}

// The shadow pass of the last level of a frame (nothing follows that could carry it).
template <bool STATS, bool FAST>
__global__ void __launch_bounds__(WF_BLOCK, WF_WAVES)
k_wf_shadow(DScene S, ShadowPass P, double* out_f64, uint8_t* out_u8, unsigned long long* counters) {
  RT_TRAV_STACK_COOP(st, S);
  wf_shadow_body<STATS, FAST>(S, P, out_f64, out_u8, counters, st, wf_block_index());
}

template <typename T>
static int upload(rt_scene* s, const std::vector<T>& v, const T** out) {
  *out = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, bytes));
  s->allocs.push_back(p);
  s->info.device_bytes += (int64_t)bytes;
  if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = static_cast<const T*>(p);
  return RT_OK;
}

static int scratch_for_device(int device, unsigned char** out);

// ---------------------------------------------------------------------------------------
extern "C" {
#ifdef RT_TRI_PLANE_STATS
extern "C" int rt_debug_tri_plane(unsigned long long* out8, int reset) {
  if (out8) HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(rtd::g_tri_plane), 8 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[8] = {};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(rtd::g_tri_plane), z, sizeof z));
  }
  return RT_OK;
}
#endif
#ifdef RT_SECTION_TIMES
int rt_debug_section_cycles(unsigned long long* out8, int reset) {
  if (out8) HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_section_cycles), 8 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[8] = {};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_section_cycles), z, sizeof z));
  }
  return RT_OK;
}
#endif

const char* rt_last_error(void) { return g_err.c_str(); }

int rt_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

rt_scene* rt_scene_create(void) { return new rt_scene(); }

static void multi_destroy(struct Multi* m);
void rt_scene_destroy(rt_scene* s) {
  if (!s) return;
  if (s->multi) {
    Multi* m = s->multi;
    s->multi = nullptr;
    multi_destroy(m);  // the replicas on the other GPUs, the communicators, the exchange buffers
  }
  if (s->device >= 0) {  // also after a commit that failed half way: what it uploaded is freed
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(s->device);
    for (void* p : s->allocs) (void)hipFree(p);
    if (s->d_out_f64) (void)hipFree(s->d_out_f64);
    if (s->d_out_u8) (void)hipFree(s->d_out_u8);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->split_fork) (void)hipEventDestroy(s->split_fork);
    for (hipEvent_t e : s->split_joined) (void)hipEventDestroy(e);
    for (hipStream_t q : s->split_streams) (void)hipStreamDestroy(q);
    for (rt_scene::SweepSlot& q : s->sweep_slots) {
      if (q.stream) (void)hipStreamDestroy(q.stream);
      if (q.device) (void)hipFree(q.device);
      if (q.pinned) (void)hipHostFree(q.pinned);
    }
    for (rt_scene::WfPlan& p : s->wf_plans) drop_plan(p);
    for (rt_scene::WfState* w : s->wf_states) {
      for (int l = 0; l <= MAX_DEPTH; l++) {
        if (w->wf_tasks[l]) (void)hipFree(w->wf_tasks[l]);
        if (w->wf_nodes[l]) (void)hipFree(w->wf_nodes[l]);
        if (w->wf_live[l]) (void)hipFree(w->wf_live[l]);
        if (w->wf_shadow[l]) (void)hipFree(w->wf_shadow[l]);
        if (w->wf_keys[l]) (void)hipFree(w->wf_keys[l]);
        if (w->wf_perm[l]) (void)hipFree(w->wf_perm[l]);
      }
      if (w->wf_counts) (void)hipFree(w->wf_counts);
      if (w->wf_hist) (void)hipFree(w->wf_hist);
      if (w->wf_counts_host) (void)hipHostFree(w->wf_counts_host);
      if (w->done) (void)hipEventDestroy(w->done);
      delete w;
    }
    (void)hipSetDevice(cur);
  }
  delete s;
}

int rt_scene_add_material(rt_scene* s, const rt_material* m) {
  if (!s || !m) return -fail(RT_ERR_ARG, "null argument");
  if (s->committed) return -fail(RT_ERR_STATE, "scene already committed");
  if (m->pattern_kind < RT_SOLID || m->pattern_kind > RT_CHECKERED)
    return -fail(RT_ERR_ARG, "unknown pattern kind");
  double tmp[12];
  std::string why;
  if (!affine_rows(m->pattern_inv, tmp, why)) return -fail(RT_ERR_UNSUPPORTED, "pattern_inv: " + why);
  s->mats.push_back(*m);
  return (int)s->mats.size() - 1;
}

int rt_scene_add_group(rt_scene* s, const rt_group* g) {
  if (!s || !g) return -fail(RT_ERR_ARG, "null argument");
  if (s->committed) return -fail(RT_ERR_STATE, "scene already committed");
  if (g->parent < -1 || g->parent >= (int)s->groups.size())
    return -fail(RT_ERR_ARG, "group parent out of range");
  double tmp[12];
  std::string why;
  if (!affine_rows(g->inv, tmp, why)) return -fail(RT_ERR_UNSUPPORTED, "group inv: " + why);
  int depth = 1;
  for (int p = g->parent; p >= 0; p = s->groups[p].parent) depth++;
  if (depth > 6) return -fail(RT_ERR_UNSUPPORTED, "groups nested deeper than 6 levels");
  s->groups.push_back(*g);
  return (int)s->groups.size() - 1;
}

int rt_scene_add_primitive(rt_scene* s, const rt_primitive* p) {
  if (!s || !p) return -fail(RT_ERR_ARG, "null argument");
  if (s->committed) return -fail(RT_ERR_STATE, "scene already committed");
  if (p->kind < RT_SPHERE || p->kind > RT_CONE) return -fail(RT_ERR_ARG, "unknown primitive kind");
  if (p->material < 0 || p->material >= (int)s->mats.size())
    return -fail(RT_ERR_ARG, "primitive material out of range");
  if (p->parent < -1 || p->parent >= (int)s->groups.size())
    return -fail(RT_ERR_ARG, "primitive parent out of range");
  double tmp[12];
  std::string why;
  if (!affine_rows(p->inv, tmp, why)) return -fail(RT_ERR_UNSUPPORTED, "primitive inv: " + why);
  s->prims.push_back(*p);
  s->prim_order.push_back(s->next_order++);
  return (int)s->prims.size() - 1;
}

int rt_scene_add_mesh(rt_scene* s, const rt_mesh* m) {
  if (!s || !m) return -fail(RT_ERR_ARG, "null argument");
  if (s->committed) return -fail(RT_ERR_STATE, "scene already committed");
  if (m->group < 0 || m->group >= (int)s->groups.size())
    return -fail(RT_ERR_ARG, "mesh group out of range");
  if (m->material < 0 || m->material >= (int)s->mats.size())
    return -fail(RT_ERR_ARG, "mesh material out of range");
  if (m->ntris < 0 || m->ntris > 0x0fffff00LL) return -fail(RT_ERR_ARG, "bad triangle count (max 2^28 per mesh)");
  if (m->ntris > 0 && (!m->p1 || !m->p2 || !m->p3)) return -fail(RT_ERR_ARG, "null vertex array");
  bool has_n = m->n1 || m->n2 || m->n3;
  if (has_n && !(m->n1 && m->n2 && m->n3))
    return -fail(RT_ERR_ARG, "n1,n2,n3 must be all given or all NULL");
  if (m->smooth && !has_n) return -fail(RT_ERR_ARG, "smooth flags without normals");
  MeshH h;
  h.group = m->group;
  h.material = m->material;
  h.ntris = m->ntris;
  h.order_base = s->next_order;
  h.has_normals = has_n;
  size_t n3 = (size_t)m->ntris * 3;
  h.p1.assign(m->p1, m->p1 + n3);
  h.p2.assign(m->p2, m->p2 + n3);
  h.p3.assign(m->p3, m->p3 + n3);
  if (has_n) {
    h.n1.assign(m->n1, m->n1 + n3);
    h.n2.assign(m->n2, m->n2 + n3);
    h.n3.assign(m->n3, m->n3 + n3);
    if (m->smooth)
      h.smooth.assign(m->smooth, m->smooth + m->ntris);
    else
      h.smooth.assign((size_t)m->ntris, 1);
  } else {
    h.smooth.assign((size_t)m->ntris, 0);
  }
  for (size_t i = 0; i < n3; i++)
    if (!std::isfinite(h.p1[i]) || !std::isfinite(h.p2[i]) || !std::isfinite(h.p3[i]))
      return -fail(RT_ERR_ARG, "non-finite vertex");
  s->next_order += (uint32_t)m->ntris;
  s->meshes.push_back(std::move(h));
  return (int)s->meshes.size() - 1;
}

int rt_scene_add_light(rt_scene* s, const double intensity[3], const double position[3]) {
  if (!s || !intensity || !position) return -fail(RT_ERR_ARG, "null argument");
  if (s->committed) return -fail(RT_ERR_STATE, "scene already committed");
  DLight L;
  for (int i = 0; i < 3; i++) {
    L.intensity[i] = intensity[i];
    L.pos[i] = position[i];
  }
  s->lights.push_back(L);
  return (int)s->lights.size() - 1;
}

int rt_scene_commit(rt_scene* s, int device, int bvh_builder) {
  if (!s) return fail(RT_ERR_ARG, "null scene");
  if (s->committed) return fail(RT_ERR_STATE, "scene already committed");
  if (bvh_builder < 0 || bvh_builder > 2)
    return fail(RT_ERR_ARG, "bvh_builder must be 0 (host SAH), 1 (device LBVH) or 2 (by mesh size)");
  int ndev = rt_device_count();
  if (ndev <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device visible; the render core has no CPU path");
  if (device < 0) HIP_TRY(hipGetDevice(&device));
  if (device >= ndev) return fail(RT_ERR_ARG, "device index out of range");
  auto t0 = std::chrono::steady_clock::now();
  HIP_TRY(hipSetDevice(device));
  s->device = device;
  {
    // the address of this device's copy of the fixed array rt_render_device_noalloc works in:
    // looked up here, where allocating is allowed, so that the entry point itself never has to
    unsigned char* unused = nullptr;
    int rc_scratch = scratch_for_device(device, &unused);
    if (rc_scratch) return rc_scratch;
  }

  std::string why;
  std::vector<DGroup> groups(s->groups.size());
  for (size_t g = 0; g < s->groups.size(); g++) {
    DGroup& G = groups[g];
    affine_rows(s->groups[g].inv, G.inv, why);
    G.parent = s->groups[g].parent;
    int chain[8], n = 0;
    for (int p = (int)g; p >= 0; p = s->groups[p].parent) chain[n++] = p;
    G.nchain = n;
    for (int k = 0; k < 6; k++) G.chain[k] = 0;
    for (int k = 0; k < n; k++) G.chain[k] = chain[n - 1 - k];
  }
  std::vector<DPrim> prims(s->prims.size());
  for (size_t p = 0; p < s->prims.size(); p++) {
    DPrim& P = prims[p];
    const rt_primitive& src = s->prims[p];
    affine_rows(src.inv, P.inv, why);
    P.minimum = src.minimum;
    P.maximum = src.maximum;
    P.kind = src.kind;
    P.material = src.material;
    P.parent = src.parent;
    P.closed = src.closed;
    P.order = s->prim_order[p];
    P.pad[0] = P.pad[1] = P.pad[2] = 0;
  }
  std::vector<DPrimBox> prim_boxes(s->prims.size());
  for (size_t p = 0; p < s->prims.size(); p++) prim_boxes[p] = world_box_of(s, s->prims[p]);
  std::vector<DMaterial> mats(s->mats.size());
  for (size_t m = 0; m < s->mats.size(); m++) {
    DMaterial& M = mats[m];
    const rt_material& src = s->mats[m];
    M.ambient = src.ambient;
    M.diffuse = src.diffuse;
    M.specular = src.specular;
    M.shininess = src.shininess;
    M.reflective = src.reflective;
    M.transparency = src.transparency;
    M.refractive_index = src.refractive_index;
    for (int i = 0; i < 3; i++) {
      M.a[i] = src.color_a[i];
      M.b[i] = src.color_b[i];
    }
    affine_rows(src.pattern_inv, M.pinv, why);
    M.pattern_kind = src.pattern_kind;
    M.pad[0] = M.pad[1] = M.pad[2] = 0;
  }
  // meshes: edges (triangle.rs:49-51), BVH, records in leaf order.  The triangle tables are
  // device arrays from the start: the host builder copies each mesh's records into its slice,
  // the device builder writes them there itself (nothing of its tree or triangle order comes
  // back to the host).
  std::vector<DMesh> meshes(s->meshes.size());
  int64_t total_tris = 0;
  for (const MeshH& h : s->meshes) total_tris += h.ntris;
  if (total_tris > 0x7fffff00LL) return fail(RT_ERR_UNSUPPORTED, "too many triangles");
  for (const MeshH& h : s->meshes)
    if (h.ntris >= (1LL << (30 - LEAF_BITS)))
      return fail(RT_ERR_UNSUPPORTED, "too many triangles in one mesh for the leaf code");
  s->info.device_bytes = 0;
  auto device_table = [&](size_t bytes, void** out) -> int {
    *out = nullptr;
    HIP_TRY(hipMalloc(out, std::max<size_t>(bytes, 16)));
    s->allocs.push_back(*out);
    s->info.device_bytes += (int64_t)bytes;
    return RT_OK;
  };
  DTri* d_tris = nullptr;
  DTriF* d_trisf = nullptr;
  DTriN* d_trin = nullptr;
  int rc;
  if ((rc = device_table((size_t)total_tris * sizeof(DTri), (void**)&d_tris))) return rc;
  if ((rc = device_table((size_t)total_tris * sizeof(DTriF), (void**)&d_trisf))) return rc;
  if ((rc = device_table((size_t)total_tris * sizeof(DTriN), (void**)&d_trin))) return rc;
  struct MeshNodes {
    std::vector<DNode4> host;  // host builder
    rtb::DeviceBuild dev;      // device builder
    int count = 0;
  };
  std::vector<MeshNodes> mesh_nodes(s->meshes.size());
  struct FreeBuilds {  // whatever a device build still holds is released when commit ends
    std::vector<MeshNodes>& v;
    ~FreeBuilds() {
      for (MeshNodes& m : v) rtb::free_device_build(m.dev);
    }
  } free_builds{mesh_nodes};
  int64_t node_total = 0, tri_total = 0;
  for (size_t mi = 0; mi < s->meshes.size(); mi++) {
    const MeshH& h = s->meshes[mi];
    int max_leaf = 12;  // measured best with the shared leaf tests (pawn 1080p; 4..15 tried)
    if (const char* e = getenv("RT_MAX_LEAF")) max_leaf = atoi(e);
    DMesh& M = meshes[mi];
    M.group = h.group;
    M.material = h.material;
    M.order_base = h.order_base;
    M.root = (int32_t)node_total;
    M.tri_base = (int32_t)tri_total;
    M.ntris = (int32_t)h.ntris;
    M.pad = 0;
    MeshNodes& mn = mesh_nodes[mi];
    // 2: the host's binned SAH gives small meshes slightly faster frames (pawn: 8 %) for a few
    // milliseconds of build; from 100 000 triangles on the device build wins on every count
    // (1M triangles: 21 ms instead of 0.9 s, frames within 2 %)
    const bool on_device = bvh_builder == 1 || (bvh_builder == 2 && h.ntris >= 100000);
    if (on_device) {
      std::string berr;
      if (rtb::build_lbvh_resident(h.p1.data(), h.p2.data(), h.p3.data(), h.ntris, max_leaf, (int)node_total, mn.dev,
                                   berr))
        return fail(RT_ERR_HIP, "device BVH build: " + berr);
      if (mn.dev.stack_need > STACK_MAX) return fail(RT_ERR_UNSUPPORTED, "BVH deeper than the traversal stack");
      if (rtb::build_tri_records_device(mn.dev, h.ntris, h.has_normals ? h.n1.data() : nullptr,
                                        h.has_normals ? h.n2.data() : nullptr, h.has_normals ? h.n3.data() : nullptr,
                                        h.has_normals ? h.smooth.data() : nullptr, d_tris + tri_total,
                                        d_trisf + tri_total, d_trin + tri_total, berr))
        return fail(RT_ERR_HIP, "device triangle records: " + berr);
      mn.count = mn.dev.n_nodes;
      for (int a = 0; a < 3; a++) {
        M.lo[a] = mn.dev.lo[a];
        M.hi[a] = mn.dev.hi[a];
        M.cell[a] = mn.dev.cell[a];
      }
      M.scale = mn.dev.scale;
    } else {
      std::shared_ptr<HostBuilt> hb = mi < s->host_built.size() ? s->host_built[mi] : nullptr;
      if (!hb) {
        hb = std::make_shared<HostBuilt>();
        rtb::BuildResult br;
        rtb::build_sah(h.p1.data(), h.p2.data(), h.p3.data(), h.ntris, max_leaf, br);
        // the kernels walk four-wide nodes; a step leaves up to three children on the stack
        hb->stack_need = rtb::collapse_to_wide(br.nodes, br.cell, hb->nodes);
        rtb::reorder_top_breadth_first(hb->nodes, TOP_NODES);
        for (int a = 0; a < 3; a++) {
          hb->lo[a] = br.lo[a];
          hb->hi[a] = br.hi[a];
          hb->cell[a] = br.cell[a];
        }
        hb->scale = br.scale;
        hb->tris.resize((size_t)h.ntris);
        hb->trisf.resize((size_t)h.ntris);
        hb->trin.resize((size_t)h.ntris);
        for (int64_t k = 0; k < h.ntris; k++) {
          uint32_t f = br.order[k];
          DTri T;
          DTriN TN;
          double e1[3], e2[3];
          for (int a = 0; a < 3; a++) {
            T.p1[a] = h.p1[3 * f + a];
            e1[a] = h.p2[3 * f + a] - h.p1[3 * f + a];
            e2[a] = h.p3[3 * f + a] - h.p1[3 * f + a];
            T.e1[a] = e1[a];
            T.e2[a] = e2[a];
          }
          T.orig = f;
          T.smooth = h.smooth[f];
          if (T.smooth) {
            for (int a = 0; a < 3; a++) {
              TN.n[a] = h.n1[3 * f + a];
              TN.n[3 + a] = h.n2[3 * f + a];
              TN.n[6 + a] = h.n3[3 * f + a];
            }
          } else {
            // triangle.rs:51: normalize(cross(e2, e1))
            double cx = e2[1] * e1[2] - e2[2] * e1[1];
            double cy = e2[2] * e1[0] - e2[0] * e1[2];
            double cz = e2[0] * e1[1] - e2[1] * e1[0];
            double mag = std::sqrt(cx * cx + cy * cy + cz * cz);
            TN.n[0] = cx / mag;
            TN.n[1] = cy / mag;
            TN.n[2] = cz / mag;
            for (int a = 3; a < 9; a++) TN.n[a] = 0.0;
          }
          DTriF TF;
          for (int a = 0; a < 3; a++) {
            TF.p1[a] = (float)T.p1[a];
            TF.e1[a] = (float)T.e1[a];
            TF.e2[a] = (float)T.e2[a];
          }
          TF.orig = T.orig;
          uint32_t lo16[3];
          for (int a = 0; a < 3; a++) {
            const float r = (float)(T.p1[a] - (double)TF.p1[a]);  // what f32 lost
            uint32_t bits;
            std::memcpy(&bits, &r, sizeof bits);
            lo16[a] = bits >> 16;  // truncated: at most 2^-7 of the residual is dropped
          }
          TF.p1_lo_xy = lo16[0] | (lo16[1] << 16);
          TF.p1_lo_z = lo16[2];
          hb->tris[(size_t)k] = T;
          hb->trisf[(size_t)k] = TF;
          hb->trin[(size_t)k] = TN;
        }
        if (s->keep_host_built) {
          if (s->host_built.size() < s->meshes.size()) s->host_built.resize(s->meshes.size());
          s->host_built[mi] = hb;
        }
      }
      if (hb->stack_need > STACK_MAX) return fail(RT_ERR_UNSUPPORTED, "BVH deeper than the traversal stack");
      mn.host = hb->nodes;
      for (DNode4& N : mn.host)
        for (int k = 0; k < 4; k++)
          if (N.c[k] >= 0) N.c[k] += (int32_t)node_total;
      mn.count = (int)mn.host.size();
      for (int a = 0; a < 3; a++) {
        M.lo[a] = hb->lo[a];
        M.hi[a] = hb->hi[a];
        M.cell[a] = hb->cell[a];
      }
      M.scale = hb->scale;
      if (h.ntris > 0) {
        HIP_TRY(hipMemcpy(d_tris + tri_total, hb->tris.data(), hb->tris.size() * sizeof(DTri), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_trisf + tri_total, hb->trisf.data(), hb->trisf.size() * sizeof(DTriF), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_trin + tri_total, hb->trin.data(), hb->trin.size() * sizeof(DTriN), hipMemcpyHostToDevice));
      }
    }
    M.n_nodes = mn.count;
    node_total += mn.count;
    tri_total += h.ntris;
    {
      const double inf = std::numeric_limits<double>::infinity();
      double wlo[3], whi[3];
      const bool known = world_bounds_of_box(s, M.group, M.lo, M.hi, wlo, whi);
      for (int a = 0; a < 3; a++) {
        // padded like the boxes of the closed-form shapes: 1e-6 of the size, far above the
        // rounding of the f64 slab test and of the corner transforms
        const double pad = known ? 1e-6 * (whi[a] - wlo[a]) + 1e-9 * std::fmax(std::fabs(wlo[a]), std::fabs(whi[a])) + 1e-300 : 0.0;
        M.wlo[a] = known ? wlo[a] - pad : -inf;
        M.whi[a] = known ? whi[a] + pad : inf;
      }
    }
  }
  // the node table: every mesh's nodes back to back (indices were offset when they were made)
  DNode4* d_nodes = nullptr;
  if ((rc = device_table((size_t)node_total * sizeof(DNode4), (void**)&d_nodes))) return rc;
  {
    int64_t at = 0;
    for (MeshNodes& mn : mesh_nodes) {
      if (mn.count > 0) {
        if (mn.dev.d_nodes)
          HIP_TRY(hipMemcpy(d_nodes + at, mn.dev.d_nodes, (size_t)mn.count * sizeof(DNode4), hipMemcpyDeviceToDevice));
        else
          HIP_TRY(hipMemcpy(d_nodes + at, mn.host.data(), (size_t)mn.count * sizeof(DNode4), hipMemcpyHostToDevice));
      }
      at += mn.count;
      rtb::free_device_build(mn.dev);
    }
  }
  s->d.tris = d_tris;
  s->d.trisf = d_trisf;
  s->d.trin = d_trin;
  s->d.nodes = d_nodes;

  {
    // grid of the coherence sort: over the meshes' world-space bounds, or, in a scene without
    // meshes, over the finite boxes of the closed-form shapes
    const double inf = std::numeric_limits<double>::infinity();
    double glo[3] = {inf, inf, inf}, ghi[3] = {-inf, -inf, -inf};
    bool any = false;
    for (size_t mi = 0; mi < meshes.size(); mi++) {
      if (meshes[mi].ntris == 0) continue;
      double wlo[3], whi[3];
      if (!world_bounds_of_box(s, meshes[mi].group, meshes[mi].lo, meshes[mi].hi, wlo, whi)) continue;
      for (int a = 0; a < 3; a++) {
        glo[a] = std::fmin(glo[a], wlo[a]);
        ghi[a] = std::fmax(ghi[a], whi[a]);
      }
      any = true;
    }
    if (!any)
      for (const DPrimBox& B : prim_boxes) {
        bool finite = true;
        for (int a = 0; a < 3; a++) finite = finite && std::isfinite(B.lo[a]) && std::isfinite(B.hi[a]);
        if (!finite) continue;
        for (int a = 0; a < 3; a++) {
          glo[a] = std::fmin(glo[a], B.lo[a]);
          ghi[a] = std::fmax(ghi[a], B.hi[a]);
        }
        any = true;
      }
    s->sort_enabled = any;
    for (int a = 0; a < 3 && any; a++) {
      const double ext = ghi[a] - glo[a];
      s->sort_lo[a] = glo[a];
      s->sort_inv_cell[a] = (ext > 0.0 && std::isfinite(ext)) ? (double)(1 << SORT_CELL_BITS) / ext : 0.0;
    }
    // On unless RT_SORT_RAYS=0.  Measured on MI355X the order raises the node loop's lane occupancy
    // at the deep levels of the pawn frame from 24-32 to 32-37 of 64 and shortens those levels by
    // 9-11 %.  The sort's own passes (30 + 45-80 us per level) give that back, but only the first
    // exact render of a frame runs them: it leaves the start of every key's run with the frame's
    // plan, and the same frame afterwards places its tasks in key order as it emits them
    // (pawn 1080p 2.26 instead of 2.40 ms, pawn 8K 31.2 instead of 34.9, tea set 0.555 / 0.567).
    const char* want_sort = getenv("RT_SORT_RAYS");
    s->sort_enabled = s->sort_enabled && (!want_sort || atoi(want_sort) != 0);
    if (const char* e = getenv("RT_SORT_MIN_LEVEL")) s->sort_min_level = std::max(1, atoi(e));
  }
  if ((rc = upload(s, groups, &s->d.groups))) return rc;
  if ((rc = upload(s, prims, &s->d.prims))) return rc;
  if ((rc = upload(s, prim_boxes, &s->d.prim_boxes))) return rc;
  if ((rc = upload(s, meshes, &s->d.meshes))) return rc;
  if ((rc = upload(s, mats, &s->d.mats))) return rc;
  if ((rc = upload(s, s->lights, &s->d.lights))) return rc;
  s->share_leaves = true;
  if (const char* e = getenv("RT_SHARE_LEAVES")) s->share_leaves = atoi(e) != 0;
  if (const char* e = getenv("RT_SHADOW_PASS")) s->shadow_pass = atoi(e) != 0;
  if (const char* e = getenv("RT_FRAME_SPLIT")) s->frame_split = std::min(std::max(atoi(e), 0), 8);
  s->d.n_groups = (int32_t)groups.size();
  s->d.n_prims = (int32_t)prims.size();
  s->d.n_meshes = (int32_t)meshes.size();
  s->d.n_lights = (int32_t)s->lights.size();
  void* pc = nullptr;
  HIP_TRY(hipMalloc(&pc, (N_COUNTERS + 1) * sizeof(unsigned long long)));
  s->allocs.push_back(pc);
  s->d_counters = static_cast<unsigned long long*>(pc);
  HIP_TRY(hipMemset(pc, 0, (N_COUNTERS + 1) * sizeof(unsigned long long)));
  HIP_TRY(hipEventCreate(&s->ev0));
  HIP_TRY(hipEventCreate(&s->ev1));

  s->info.n_groups = (int64_t)groups.size();
  s->info.n_primitives = (int64_t)prims.size();
  s->info.n_meshes = (int64_t)meshes.size();
  s->info.n_triangles = total_tris;
  s->info.n_materials = (int64_t)mats.size();
  s->info.n_lights = (int64_t)s->lights.size();
  s->info.n_bvh_nodes = node_total;
  s->info.node_bytes = (int32_t)sizeof(DNode4);
  s->info.tri_bytes = (int32_t)sizeof(DTri);
  s->info.device = device;
  s->info.bvh_builder = bvh_builder;
  s->info.commit_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  s->committed = true;
  return RT_OK;
}

int rt_scene_get_info(const rt_scene* s, rt_scene_info* info) {
  if (!s || !info) return fail(RT_ERR_ARG, "null argument");
  if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
  *info = s->info;
  return RT_OK;
}

int64_t rt_shard_tiles(int hsize, int vsize, int tile_first, int tile_step) {
  if (hsize <= 0 || vsize <= 0 || tile_step <= 0 || tile_first < 0 || tile_first >= tile_step) return -1;
  int64_t tx = (hsize + RT_TILE - 1) / RT_TILE, ty = (vsize + RT_TILE - 1) / RT_TILE;
  int64_t total = tx * ty;
  if (tile_first >= total) return 0;
  return (total - tile_first + tile_step - 1) / tile_step;
}

static int camera_to_device(const rt_camera* c, DCamera& d) {
  if (c->hsize <= 0 || c->vsize <= 0) return fail(RT_ERR_ARG, "camera size must be positive");
  std::string why;
  if (!affine_rows(c->inv, d.inv, why)) return fail(RT_ERR_UNSUPPORTED, "camera inv: " + why);
  d.half_width = c->half_width;
  d.half_height = c->half_height;
  d.pixel_size = c->pixel_size;
  d.hsize = c->hsize;
  d.vsize = c->vsize;
  return RT_OK;
}

// A whole call's pixel map: all the image row-major, or a shard into its compact tile buffer.
static PixelMap pixel_map(const rt_camera* cam, int tile_first, int tile_step, bool compact_always = false) {
  PixelMap pm;
  pm.hsize = cam->hsize;
  pm.vsize = cam->vsize;
  pm.tiles_x = (cam->hsize + RT_TILE - 1) / RT_TILE;
  pm.tile_first = tile_first;
  pm.tile_step = tile_step;
  pm.out_row_major = (tile_step == 1 && !compact_always) ? 1 : 0;
  pm.out_k_mul = 1;
  pm.out_k_add = 0;
  return pm;
}

static int wf_ensure(void** p, size_t* cap, size_t need, size_t elem) {
  if (need <= *cap) return RT_OK;
  if (*p) HIP_TRY(hipFree(*p));
  *p = nullptr;
  *cap = 0;
  size_t want = need + need / 8 + 1024;
  HIP_TRY(hipMalloc(p, want * elem));
  *cap = want;
  return RT_OK;
}

static uint64_t frame_key(const DCamera& dc, const PixelMap& pm, int depth, bool fast) {
  uint64_t h = 1469598103934665603ull;  // FNV-1a over everything that fixes the queue lengths
  auto mix = [&h](const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
  };
  mix(dc.inv, sizeof dc.inv);
  mix(&dc.half_width, sizeof(double));
  mix(&dc.half_height, sizeof(double));
  mix(&dc.pixel_size, sizeof(double));
  mix(&dc.hsize, sizeof(int32_t));
  mix(&dc.vsize, sizeof(int32_t));
  mix(&pm.tile_first, sizeof(int));
  mix(&pm.tile_step, sizeof(int));
  mix(&depth, sizeof(int));
  int f = fast ? 1 : 0;
  mix(&f, sizeof(int));
  return h ? h : 1;
}

// Verdict on the frame last issued with set `w` on an estimate, once its counters have reached
// the host (the caller has synchronised): *complete = 0 if a ray queue outgrew its estimate.
// A frame that is good leaves its true queue lengths behind as a plan.
static void resolve_estimate(rt_scene* s, rt_scene::WfState* w, int* complete) {
  *complete = 1;
  if (!w->estimate_pending) return;
  w->estimate_pending = false;
  const unsigned int* h = w->wf_counts_host;
  if (h[2 * (MAX_DEPTH + 1)] != 0u) {
    *complete = 0;
    s->n_estimate_missed++;
    return;
  }
  rt_scene::WfPlan plan;
  plan.key = w->estimate_key;
  plan.id = w->estimate_id;
  plan.depth = w->estimate_depth;
  plan.fast = w->estimate_fast;
  plan.n[0] = plan.front[0] = w->estimate_n0;
  plan.levels = 1;
  for (int l = 0; l <= MAX_DEPTH; l++) plan.shadow[l] = -1;
  for (int l = 1; l < w->estimate_depth; l++) {
    plan.front[l] = h[2 * l];
    plan.n[l] = (long long)h[2 * l] + (long long)h[2 * l + 1];
    if (plan.n[l] == 0) break;
    plan.levels = l + 1;
  }
  for (int l = 0; l < plan.levels; l++) plan.shadow[l] = h[WF_SHADOW_SLOT0 + l];
  for (size_t i = 0; i < s->wf_plans.size(); i++)
    if (s->wf_plans[i].key == plan.key && same_frame(s->wf_plans[i].id, plan.id)) {
      drop_plan(s->wf_plans[i]);
      s->wf_plans.erase(s->wf_plans.begin() + (long)i);
      break;
    }
  if (s->wf_plans.size() >= WF_MAX_PLANS) {
    drop_plan(s->wf_plans.front());
    s->wf_plans.erase(s->wf_plans.begin());
  }
  s->wf_plans.push_back(plan);
}

static int render_wavefront(rt_scene* s, const DCamera& dc, const PixelMap& pm, int64_t ntiles,
                            int depth, double* d_f64, uint8_t* d_u8, hipStream_t st, bool stats,
                            bool fast) {
  const uint64_t key = frame_key(dc, pm, depth, fast);
  const FrameId id = frame_id(dc, pm.tile_first, pm.tile_step, depth, fast, false);
  rt_scene::WfPlan plan;  // a copy: the list may be reordered below
  bool planned = false;
  for (size_t i = 0; i < s->wf_plans.size(); i++)
    if (s->wf_plans[i].key == key && same_frame(s->wf_plans[i].id, id)) {
      plan = s->wf_plans[i];
      planned = true;
      // most recently rendered last: the oldest is the one forgotten, the newest the one
      // estimates are taken from
      s->wf_plans.erase(s->wf_plans.begin() + (long)i);
      s->wf_plans.push_back(plan);
      break;
    }
  if (!planned) {
    plan.key = key;
    plan.id = id;
    plan.depth = depth;
    plan.fast = fast;
    plan.n[0] = plan.front[0] = ntiles * (long long)(RT_TILE * RT_TILE);
  }
  // a plan under construction owns the tables it allocates until it joins the list
  struct PlanGuard {
    rt_scene::WfPlan* p;
    bool armed;
    ~PlanGuard() {
      if (armed) drop_plan(*p);
    }
  } plan_guard{&plan, !planned};
  // A frame never seen before normally costs a host read-back per level (below).  When the
  // caller has asked for it (rt_render_device_estimated) and the scene has rendered a frame of
  // the same size, depth and mode, the new frame is issued without any read-back instead: grids
  // and buffers are sized from that frame's queue lengths plus a margin, the kernels take the
  // true lengths from the device, and a level that outgrows its estimate marks the frame as
  // incomplete (rt_estimated_frame_result tells; the caller renders it again the exact way).
  constexpr int OVERFLOW_SLOT = WF_OVERFLOW_SLOT;
  const rt_scene::WfPlan* estimate = nullptr;
  if (!planned && s->allow_estimates && !stats)
    for (const rt_scene::WfPlan& c : s->wf_plans)
      if (c.levels > 0 && c.n[0] == ntiles * (long long)(RT_TILE * RT_TILE) && c.depth == depth && c.fast == fast)
        estimate = &c;  // the most recent one wins
  rt_scene::WfPlan estimate_copy;
  if (estimate) {
    estimate_copy = *estimate;
    estimate = &estimate_copy;
  }
  // a new frame on this stream: whatever verdict an earlier one left unasked is void now
  for (size_t i = s->orphan_results.size(); i-- > 0;)
    if (s->orphan_results[i].first == st) s->orphan_results.erase(s->orphan_results.begin() + (long)i);
  rt_scene::WfState* w = nullptr;
  for (rt_scene::WfState* c : s->wf_states)
    if (c->stream == st) w = c;
  if (!w) {
    // 8 sweep slots of the scene's own plus as many caller streams before sets are shared
    constexpr size_t WF_MAX_STREAMS = 16;
    if (s->wf_states.size() < WF_MAX_STREAMS) {
      w = new rt_scene::WfState;
      w->stream = st;
      s->wf_states.push_back(w);
    } else {
      // More streams than buffer sets: this one takes over the set used longest ago, ordered
      // behind that set's last frame through the event recorded after it (the stream it ran on
      // may be gone by now; its event is not).  A frame issued on an estimate that nobody has
      // asked about yet is resolved first, so its verdict is not lost with the set.
      for (rt_scene::WfState* c : s->wf_states)
        if (!w || (c->estimate_pending ? 1 : 0) < (w->estimate_pending ? 1 : 0) ||
            ((c->estimate_pending == w->estimate_pending) && c->last_use < w->last_use))
          w = c;
      if (w->done) HIP_TRY(hipEventSynchronize(w->done));
      if (w->estimate_pending) {
        int complete = 1;
        resolve_estimate(s, w, &complete);
        s->orphan_results.push_back({w->stream, complete});
      }
      w->stream = st;
    }
  }
  if (!w->done) HIP_TRY(hipEventCreateWithFlags(&w->done, hipEventDisableTiming));
  w->last_use = ++s->wf_tick;
  w->estimate_pending = false;  // a new frame on this stream: an unasked verdict of an earlier one is void
  if (!w->wf_counts) {
    HIP_TRY(hipMalloc(&w->wf_counts, (WF_COUNTS_PAD + (size_t)MAX_DEPTH * SORT_BINS) * sizeof(unsigned int)));
    HIP_TRY(hipHostMalloc(&w->wf_counts_host, WF_N_COUNTS * sizeof(unsigned int)));
  }
  {
    int placed_levels = 0;  // the tickets of levels 1..placed_levels are cleared with the counters
    if (planned && s->sort_enabled)
      for (int l = 1; l < depth; l++)
        if (plan.place[l]) placed_levels = l;
    HIP_TRY(hipMemsetAsync(w->wf_counts, 0, (WF_COUNTS_PAD + (size_t)placed_levels * SORT_BINS) * sizeof(unsigned int), st));
  }
  long long n = ntiles * (RT_TILE * RT_TILE);
  long long n_front = n;     // level 0 has no queue
  unsigned int cap_in = 0;
  long long level_n[MAX_DEPTH + 1];
  int levels = 0;
  bool spawned[MAX_DEPTH + 1] = {};
  bool sorted_in = false;   // the queue the current level reads has been sorted (read through a permutation)
  bool ordered_in = false;  // ... or was filled in key order (read front to back)
  bool captured = false;    // this frame has added tables to its plan
  const long long level_n0 = n;  // calls of level 0 = pixels of the shard's tiles
  int rc;
  for (int l = 0; l < depth; l++) {
    bool spawn = (l + 1 < depth);
    if ((rc = wf_ensure(&w->wf_nodes[l], &w->wf_node_cap[l], (size_t)n, sizeof(WfNode)))) return rc;
    if ((rc = wf_ensure(&w->wf_live[l], &w->wf_live_cap[l], (size_t)n, 1))) return rc;
    const bool shq = s->shadow_pass && (fast || s->share_leaves);
    if (shq && (rc = wf_ensure(&w->wf_shadow[l], &w->wf_shadow_cap[l], (size_t)n, sizeof(WfShadow)))) return rc;
    if (spawn) {
      if (2 * n > 0xfffffff0LL) return fail(RT_ERR_UNSUPPORTED, "ray queue exceeds 2^32 entries");
      if ((rc = wf_ensure(&w->wf_tasks[l + 1], &w->wf_task_cap[l + 1], (size_t)(2 * n), sizeof(WfTask))))
        return rc;
    }
    // the queue this level fills is sorted before the next level reads it
    // (a frame issued on an estimate runs unsorted: its one render would pay the passes for nothing)
    // ... nor is a short queue kept in order (under a sixteenth of the shard's pixels: the order of
    // emission, pixel patch by pixel patch, is as good for the few rays of a scene with little
    // recursion — the 64 mirror spheres of the 1 M-triangle frame: 1 % slower sorted); a planned
    // frame knows the length beforehand, a new one learns it after the level
    const long long sort_least = level_n0 / 16;
    const bool sort_next = spawn && s->sort_enabled && (l + 1) >= s->sort_min_level && !estimate &&
                           !(planned && (l + 1 >= plan.levels || plan.n[l + 1] < sort_least));
    if (sort_next) {
      if ((rc = wf_ensure(&w->wf_keys[l + 1], &w->wf_keys_cap[l + 1], w->wf_task_cap[l + 1], sizeof(unsigned int))))
        return rc;
      if (!w->wf_hist) {
        HIP_TRY(hipMalloc(&w->wf_hist, 2 * (size_t)SORT_BINS * sizeof(unsigned int)));
        HIP_TRY(hipMemsetAsync(w->wf_hist, 0, 2 * (size_t)SORT_BINS * sizeof(unsigned int), st));
      }
    }
    SortGrid sgrid;
    for (int a = 0; a < 3; a++) {
      sgrid.lo[a] = s->sort_lo[a];
      sgrid.inv_cell[a] = s->sort_inv_cell[a];
    }
    sgrid.enabled = sort_next ? 1 : 0;
    // A planned frame whose plan knows where every key's run starts places its tasks in key
    // order as it emits them: no keys, no histogram, no sort passes.
    Placement place{nullptr, nullptr, ordered_in ? 1 : 0};
    const bool place_next = sort_next && planned && plan.place[l + 1] != nullptr;
    if (place_next) {
      place.off = plan.place[l + 1];
      place.cur = w->wf_counts + WF_COUNTS_PAD + (size_t)l * SORT_BINS;  // cleared with the counters above
    }
    const unsigned int* perm_in = sorted_in ? static_cast<const unsigned int*>(w->wf_perm[l]) : nullptr;
    unsigned int* keys_next = (sort_next && !place_next) ? static_cast<unsigned int*>(w->wf_keys[l + 1]) : nullptr;
    // With the shadow pass, this launch also carries the light loops the level before has queued
    // (behind the level's own waves): as many waves as that level had hits when the count is
    // known (a planned frame), as many as it had calls otherwise; the kernel reads the true count.
    const long long level_blocks = (n + WF_BLOCK - 1) / WF_BLOCK;
    ShadowPass before{};
    long long before_blocks = 0;
    if (shq && l > 0) {
      const long long ns = (planned && plan.shadow[l - 1] >= 0) ? plan.shadow[l - 1] : level_n[l - 1];
      before_blocks = (ns + WF_BLOCK - 1) / WF_BLOCK;
      before.q = static_cast<const WfShadow*>(w->wf_shadow[l - 1]);
      before.count = w->wf_counts + WF_SHADOW_SLOT0 + (l - 1);
      before.nodes = static_cast<WfNode*>(w->wf_nodes[l - 1]);
      before.parents = (l > 1) ? static_cast<WfNode*>(w->wf_nodes[l - 2]) : nullptr;
      before.level0 = (l == 1) ? 1 : 0;
    }
    if (level_blocks + before_blocks > 0x7fffffffLL) return fail(RT_ERR_UNSUPPORTED, "too many waves in one launch");
    dim3 grid((unsigned)(level_blocks + before_blocks)), block(WF_BLOCK);
    const WfTask* tin = static_cast<const WfTask*>(w->wf_tasks[l]);
    WfNode* nodes = static_cast<WfNode*>(w->wf_nodes[l]);
    WfTask* tnext = spawn ? static_cast<WfTask*>(w->wf_tasks[l + 1]) : nullptr;
    unsigned int cap = spawn ? (unsigned int)std::min<size_t>(w->wf_task_cap[l + 1], 0xffffffffu) : 0u;
    unsigned int* cnt_next = w->wf_counts + 2 * (l + 1);
    // calls without children write into their parents (or pixels) directly; a level that
    // spawns nothing needs no pass of its own on the way up
    WfNode* parents = (l > 0) ? static_cast<WfNode*>(w->wf_nodes[l - 1]) : nullptr;
    spawned[l] = spawn;
#define WF_LAUNCH(ST, PR, FA, CO, SQ)                                                                    \
  hipLaunchKernelGGL((k_wf_level<ST, PR, FA, CO, SQ>), grid, block, 0, st, s->d, dc, pm, tin, n, n_front, cap_in, \
                     nodes, static_cast<uint8_t*>(w->wf_live[l]), parents, d_f64, d_u8, tnext, cnt_next, cap,   \
                     spawn ? 1 : 0, s->d_counters,                                                              \
                     (estimate && l > 0) ? w->wf_counts + 2 * l : (const unsigned int*)nullptr,                \
                     w->wf_counts + OVERFLOW_SLOT, sgrid, perm_in, keys_next, w->wf_hist, place,                \
                     static_cast<WfShadow*>(w->wf_shadow[l]), w->wf_counts + WF_SHADOW_SLOT0 + l, before,      \
                     level_blocks)
#define WF_LAUNCH2(ST, PR)                    \
  do {                                        \
    if (fast && shq)                          \
      WF_LAUNCH(ST, PR, true, true, true);    \
    else if (fast)                            \
      WF_LAUNCH(ST, PR, true, true, false);   \
    else if (s->share_leaves && shq)          \
      WF_LAUNCH(ST, PR, false, true, true);   \
    else if (s->share_leaves)                 \
      WF_LAUNCH(ST, PR, false, true, false);  \
    else                                      \
      WF_LAUNCH(ST, PR, false, false, false); \
  } while (0)
    if (l == 0) {
      if (stats) WF_LAUNCH2(true, true); else WF_LAUNCH2(false, true);
    } else {
      if (stats) WF_LAUNCH2(true, false); else WF_LAUNCH2(false, false);
    }
#undef WF_LAUNCH2
#undef WF_LAUNCH
    HIP_TRY(hipGetLastError());
    s->launches++;
    level_n[l] = n;
    levels = l + 1;
    if (!planned && !estimate) {
      // the level's hit count, for the plan (it travels with the queue lengths read below, or,
      // for the last level, is waited for after the loop)
      HIP_TRY(hipMemcpyAsync(w->wf_counts_host + WF_SHADOW_SLOT0 + l, w->wf_counts + WF_SHADOW_SLOT0 + l,
                             sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    }
    if (!spawn) break;
    if (planned) {
      if (l + 1 >= plan.levels) break;
      n = plan.n[l + 1];
      n_front = plan.front[l + 1];
    } else if (estimate) {
      // what the next launch and its buffers are sized for: a quarter more than the frame the
      // estimate comes from, never more than two children per call of this level
      const long long guess = (l + 1 < estimate->levels) ? estimate->n[l + 1] : 0;
      n = std::min(2 * n, guess + guess / 4 + 16384);
      n_front = n;  // the kernel reads both from the device
    } else {
      HIP_TRY(hipMemcpyAsync(w->wf_counts_host + 2 * (l + 1), cnt_next, 2 * sizeof(unsigned int),
                             hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      n_front = w->wf_counts_host[2 * (l + 1)];
      n = n_front + w->wf_counts_host[2 * (l + 1) + 1];
      plan.n[l + 1] = n;
      plan.front[l + 1] = n_front;
    }
    cap_in = cap;
    sorted_in = false;
    ordered_in = place_next;
    if (sort_next && !place_next && n < sort_least) {
      // a new frame's queue turned out short: it stays as it was filled; the histogram is cleared
      // for its next use
      HIP_TRY(hipMemsetAsync(w->wf_hist, 0, (size_t)SORT_BINS * sizeof(unsigned int), st));
    } else if (sort_next && !place_next) {
      // counting sort of the queue just filled: the histogram is complete, one block turns it
      // into start positions (and zeroes it for its next use), one pass deals out the positions
      hipLaunchKernelGGL(k_wf_sort_scan, dim3(1), dim3(1024), 0, st, w->wf_hist, w->wf_hist + SORT_BINS);
      HIP_TRY(hipGetLastError());
      s->launches++;
      if (n > 0 && !plan.place[l + 1]) {
        // the start positions stay with the frame's plan: the same frame again places its tasks
        // there as it emits them (a plan that came from an estimated frame gets them on the
        // frame's first exact render)
        HIP_TRY(hipMalloc(&plan.place[l + 1], (size_t)SORT_BINS * sizeof(unsigned int)));
        captured = true;
        HIP_TRY(hipMemcpyAsync(plan.place[l + 1], w->wf_hist + SORT_BINS, (size_t)SORT_BINS * sizeof(unsigned int),
                               hipMemcpyDeviceToDevice, st));
      }
      if (n > 0) {
        if ((rc = wf_ensure(&w->wf_perm[l + 1], &w->wf_perm_cap[l + 1], (size_t)n, sizeof(unsigned int)))) return rc;
        hipLaunchKernelGGL(k_wf_sort_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, keys_next, n, n_front,
                           cap, w->wf_hist + SORT_BINS, static_cast<unsigned int*>(w->wf_perm[l + 1]),
                           estimate ? w->wf_counts + 2 * (l + 1) : (const unsigned int*)nullptr);
        HIP_TRY(hipGetLastError());
        s->launches++;
        sorted_in = true;
      }
    }
    if (n == 0 && !estimate) break;
  }
  if (planned && captured) {  // the list's entry of this plan (moved to the back above) takes the new tables
    for (int l = 0; l <= MAX_DEPTH; l++) s->wf_plans.back().place[l] = plan.place[l];
  }
  if (levels > 0 && s->shadow_pass && (fast || s->share_leaves)) {
    // the light loops of the last level: nothing follows that could carry them
    const int l = levels - 1;
    const long long ns = (planned && plan.shadow[l] >= 0) ? plan.shadow[l] : level_n[l];
    if (ns > 0) {
      ShadowPass last{};
      last.q = static_cast<const WfShadow*>(w->wf_shadow[l]);
      last.count = w->wf_counts + WF_SHADOW_SLOT0 + l;
      last.nodes = static_cast<WfNode*>(w->wf_nodes[l]);
      last.parents = (l > 0) ? static_cast<WfNode*>(w->wf_nodes[l - 1]) : nullptr;
      last.level0 = (l == 0) ? 1 : 0;
      dim3 sgrid_dim((unsigned)((ns + WF_BLOCK - 1) / WF_BLOCK)), block(WF_BLOCK);
#define WF_SHADOW(ST, FA) \
  hipLaunchKernelGGL((k_wf_shadow<ST, FA>), sgrid_dim, block, 0, st, s->d, last, d_f64, d_u8, s->d_counters)
      if (stats) {
        if (fast) WF_SHADOW(true, true); else WF_SHADOW(true, false);
      } else {
        if (fast) WF_SHADOW(false, true); else WF_SHADOW(false, false);
      }
#undef WF_SHADOW
      HIP_TRY(hipGetLastError());
      s->launches++;
    }
  }
  if (estimate) {
    // the true queue lengths follow the frame to the host; rt_estimated_frame_result reads them
    HIP_TRY(hipMemcpyAsync(w->wf_counts_host, w->wf_counts, WF_N_COUNTS * sizeof(unsigned int),
                           hipMemcpyDeviceToHost, st));
    w->estimate_pending = true;
    s->n_estimated++;
    w->estimate_key = key;
    w->estimate_id = id;
    w->estimate_depth = depth;
    w->estimate_n0 = level_n[0];
    w->estimate_fast = fast;
  } else if (!planned) {
    plan.levels = levels;
    HIP_TRY(hipStreamSynchronize(st));  // the last level's hit count
    for (int l = 0; l <= MAX_DEPTH; l++)
      plan.shadow[l] = (l < levels && s->shadow_pass) ? (long long)w->wf_counts_host[WF_SHADOW_SLOT0 + l] : -1;
    if (s->wf_plans.size() >= WF_MAX_PLANS) {
      drop_plan(s->wf_plans.front());
      s->wf_plans.erase(s->wf_plans.begin());
    }
    s->wf_plans.push_back(plan);
    plan_guard.armed = false;
    if (getenv("RT_TRACE_LEVELS"))  // queue length of every level of a newly planned frame
      for (int l = 0; l < levels; l++) fprintf(stderr, "rt: level %d: %lld tasks\n", l, level_n[l]);
  }
  for (int l = levels - 1; l >= 0; l--) {
    if (!spawned[l]) continue;
    long long m = level_n[l];
    dim3 block(256), grid((unsigned)((m + 255) / 256));
    const WfNode* nodes = static_cast<const WfNode*>(w->wf_nodes[l]);
    const uint8_t* live = static_cast<const uint8_t*>(w->wf_live[l]);
    if (l == 0)
      hipLaunchKernelGGL(k_wf_combine<true>, grid, block, 0, st, nodes, live, m, (WfNode*)nullptr, d_f64, d_u8,
                         (const unsigned int*)nullptr,
                         estimate ? w->wf_counts + OVERFLOW_SLOT : (const unsigned int*)nullptr);
    else
      hipLaunchKernelGGL(k_wf_combine<false>, grid, block, 0, st, nodes, live, m,
                         static_cast<WfNode*>(w->wf_nodes[l - 1]), d_f64, d_u8,
                         estimate ? w->wf_counts + 2 * l : (const unsigned int*)nullptr,
                         estimate ? w->wf_counts + OVERFLOW_SLOT : (const unsigned int*)nullptr);
    HIP_TRY(hipGetLastError());
    s->launches++;
  }
  HIP_TRY(hipEventRecord(w->done, st));
  return RT_OK;
}

// `compact`: a shard of ONE (tile_step 1) still goes to a compact tile buffer instead of the
// row-major image — what rt_render_multi asks for, whose one shard of one is stitched like any
// other.  The flag belongs to the call, not to the scene: the single-GPU entry points render a
// scene committed with rt_scene_commit_multi row-major like any other scene.
static int render_device_fixed(rt_scene* s, const rt_camera* cam, int depth, int mode, int pipeline,
                               int tile_first, int tile_step, double* d_rgb_f64, uint8_t* d_rgb_u8,
                               void* hip_stream, int want_stats, bool compact) {
  if (!s || !cam) return fail(RT_ERR_ARG, "null argument");
  if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
  if (mode != RT_MODE_F64_PARITY && mode != RT_MODE_F32_FAST) return fail(RT_ERR_ARG, "unknown mode");
  if (mode == RT_MODE_F32_FAST && pipeline != RT_PIPELINE_WAVEFRONT)
    return fail(RT_ERR_UNSUPPORTED, "RT_MODE_F32_FAST runs on the wavefront pipeline only");
  if (mode == RT_MODE_F32_FAST && depth == 0)
    return fail(RT_ERR_UNSUPPORTED, "RT_MODE_F32_FAST needs depth >= 1");
  if (pipeline != RT_PIPELINE_MEGAKERNEL && pipeline != RT_PIPELINE_WAVEFRONT)
    return fail(RT_ERR_ARG, "unknown pipeline");
  if (depth < 0 || depth > MAX_DEPTH) return fail(RT_ERR_ARG, "depth out of range (0..8)");
  if (!d_rgb_f64 && !d_rgb_u8) return fail(RT_ERR_ARG, "no output buffer");
  DCamera dc;
  int rc = camera_to_device(cam, dc);
  if (rc) return rc;
  int64_t ntiles = rt_shard_tiles(cam->hsize, cam->vsize, tile_first, tile_step);
  if (ntiles < 0) return fail(RT_ERR_ARG, "bad tile shard");
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  HIP_TRY(hipSetDevice(s->device));
  const PixelMap pm = pixel_map(cam, tile_first, tile_step, compact);
  if (want_stats) HIP_TRY(hipMemsetAsync(s->d_counters, 0, (N_COUNTERS + 1) * sizeof(unsigned long long), st));
  HIP_TRY(hipEventRecord(s->ev0, st));
  s->launches = 0;
  if (ntiles > 0 && pipeline == RT_PIPELINE_WAVEFRONT && depth > 0) {
    // interleaved sub-shards of this call's tiles, each a wavefront frame of its own
    int split = s->frame_split;
    if (split == 0) {
      // Not set by the host.  A split frame's parts overlap: the thin end of one part's recursion
      // level runs under another part's work.  Sweeps that issue frames on estimates keep whole
      // frames.
      // A host that issues its frames on ONE stream renders one image at a time: each frame is
      // split in three.  A host that spreads frames over several streams keeps frames in flight,
      // which fills those gaps already; splitting would only add launches.  Decided from the
      // streams of the last few frames, so the choice does not depend on timing.
      bool one_stream = !s->allow_estimates && s->recent_streams.size() >= 4;
      for (hipStream_t q : s->recent_streams) one_stream = one_stream && q == st;
      // ... and only a frame with much recursion has levels whose ends are worth hiding (the
      // pawn: 3.28 -> 2.85 ms; the tea set, whose deeper levels are short, loses 4 %): the
      // queue lengths the whole frame left behind say which kind this is
      bool deep = false;
      if (one_stream) {
        const uint64_t whole = frame_key(dc, pm, depth, mode == RT_MODE_F32_FAST);
        const FrameId whole_id = frame_id(dc, pm.tile_first, pm.tile_step, depth, mode == RT_MODE_F32_FAST, false);
        for (const rt_scene::WfPlan& c : s->wf_plans)
          if (c.key == whole && same_frame(c.id, whole_id)) {
            long long below = 0;
            for (int l = 1; l < c.levels; l++) below += c.n[l];
            deep = below >= c.n[0];
          }
      }
      split = (one_stream && deep) ? 3 : 1;
      if (getenv("RT_TRACE_SPLIT")) fprintf(stderr, "rt: split %d\n", split);
    }
    s->recent_streams.push_back(st);
    if (s->recent_streams.size() > 4) s->recent_streams.erase(s->recent_streams.begin());
    if (ntiles < 64 * (int64_t)split) split = 1;  // too small to be worth the extra launches
    if (split > 1) s->allow_estimates = false;     // estimates are kept per caller stream
    if (split > 1) {
      // every part runs on a stream of the scene's own (so the caller stream's ray queues stay
      // sized for whole frames), forked from and joined to the caller's stream
      while ((int)s->split_streams.size() < split) {
        hipStream_t q = nullptr;
        hipEvent_t e = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
        s->split_streams.push_back(q);
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        s->split_joined.push_back(e);
      }
      if (!s->split_fork) HIP_TRY(hipEventCreateWithFlags(&s->split_fork, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(s->split_fork, st));
    }
    for (int i = 0; i < split; i++) {
      PixelMap sub = pm;
      sub.tile_first = tile_first + i * tile_step;
      sub.tile_step = tile_step * split;
      sub.out_k_mul = split;
      sub.out_k_add = i;
      const int64_t sub_tiles = rt_shard_tiles(cam->hsize, cam->vsize, sub.tile_first, sub.tile_step);
      if (sub_tiles <= 0) continue;
      hipStream_t q = (split == 1) ? st : s->split_streams[i];
      if (split > 1) HIP_TRY(hipStreamWaitEvent(q, s->split_fork, 0));
      rc = render_wavefront(s, dc, sub, sub_tiles, depth, d_rgb_f64, d_rgb_u8, q, want_stats != 0,
                            mode == RT_MODE_F32_FAST);
      if (rc) return rc;
      if (split > 1) {
        HIP_TRY(hipEventRecord(s->split_joined[i], q));
        HIP_TRY(hipStreamWaitEvent(st, s->split_joined[i], 0));
      }
    }
  } else if (ntiles > 0) {
    constexpr int PATCHES = (RT_TILE / 8) * (RT_TILE / 8);
    dim3 grid((unsigned)(ntiles * PATCHES)), block(64);
    if (want_stats)
      hipLaunchKernelGGL(k_render_mega<true>, grid, block, 0, st, s->d, dc, pm, depth, d_rgb_f64,
                         d_rgb_u8, s->d_counters);
    else
      hipLaunchKernelGGL(k_render_mega<false>, grid, block, 0, st, s->d, dc, pm, depth, d_rgb_f64,
                         d_rgb_u8, s->d_counters);
    HIP_TRY(hipGetLastError());
    s->launches++;
  }
  HIP_TRY(hipEventRecord(s->ev1, st));
  s->timing_pending = true;
  return RT_OK;
}

// ---------------------------------------------------------------------------------------
// rt_render_device_noalloc: the wavefront frame with its ray queues, result nodes and counters
// in one fixed device array instead of buffers from hipMalloc.  Every GPU the library runs on
// holds its own copy of the array at its own address (found once per device with
// hipGetSymbolAddress, with that device current; in host code the array's NAME is not a device
// address).  Calls on one GPU share the array: a call clears the counters it needs at its
// start, and calls that overlap on one GPU (different streams) race.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];
constexpr int NOALLOC_MAX_DEVICES = 64;
static unsigned char* g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};

// The device's copy of `scratch`; the device must be current.
static int scratch_for_device(int device, unsigned char** out) {
  if (device < 0 || device >= NOALLOC_MAX_DEVICES) return fail(RT_ERR_ARG, "device index out of range");
  if (!g_scratch_ptr[device]) {
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(RT_ERR_HIP, "hipGetSymbolAddress returned no address for the scratch array");
    g_scratch_ptr[device] = static_cast<unsigned char*>(p);
  }
  *out = g_scratch_ptr[device];
  return RT_OK;
}

// Layout of one batch of `px` primary rays at `depth` in the scratch array, sized for the worst
// case (every call spawns a reflected and a refracted ray: level l holds at most px << l calls),
// so that no level can outgrow its queue.  Returns the bytes needed.
struct NoallocLayout {
  size_t counts, tasks[MAX_DEPTH + 1], nodes[MAX_DEPTH + 1], live[MAX_DEPTH + 1];
  long long cap[MAX_DEPTH + 1];
};
static size_t noalloc_layout(long long px, int depth, NoallocLayout& L) {
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  size_t off = 0;
  L.counts = off;
  off = up(off + 2 * (MAX_DEPTH + 2) * sizeof(unsigned int));
  for (int l = 0; l < depth; l++) {
    const bool spawn = l + 1 < depth;
    L.cap[l] = px << l;
    L.live[l] = off;
    off = up(off + (size_t)L.cap[l]);
    L.nodes[l] = off;
    if (spawn) off = up(off + (size_t)L.cap[l] * sizeof(WfNode));
    L.tasks[l + 1] = off;
    if (spawn) off = up(off + 2 * (size_t)L.cap[l] * sizeof(WfTask));
  }
  L.tasks[0] = 0;
  return off;
}

int64_t rt_noalloc_max_tiles(int depth) {
  if (depth < 0 || depth > MAX_DEPTH) return -1;
  NoallocLayout L;
  int64_t fits = 0, over = RT_NOALLOC_MAX_TILES + 1;  // the layout grows with the tile count: bisect
  while (over - fits > 1) {
    const int64_t t = fits + (over - fits) / 2;
    if (noalloc_layout(t * (RT_TILE * RT_TILE), depth, L) > (size_t)RT_NOALLOC_SCRATCH_BYTES)
      over = t;
    else
      fits = t;
  }
  return fits;
}

int rt_render_device_noalloc(rt_scene* s, const rt_camera* cam, int depth, int mode, int tile_first,
                             int tile_step, double* d_rgb_f64, uint8_t* d_rgb_u8, void* hip_stream) {
  if (!s || !cam) return fail(RT_ERR_ARG, "null argument");
  if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
  if (mode != RT_MODE_F64_PARITY && mode != RT_MODE_F32_FAST) return fail(RT_ERR_ARG, "unknown mode");
  if (depth < 1 || depth > MAX_DEPTH) return fail(RT_ERR_ARG, "depth out of range (1..8)");
  if (!d_rgb_f64 && !d_rgb_u8) return fail(RT_ERR_ARG, "no output buffer");
  DCamera dc;
  int rc = camera_to_device(cam, dc);
  if (rc) return rc;
  const int64_t ntiles = rt_shard_tiles(cam->hsize, cam->vsize, tile_first, tile_step);
  if (ntiles < 0) return fail(RT_ERR_ARG, "bad tile shard");
  if (ntiles > RT_NOALLOC_MAX_TILES)
    return fail(RT_ERR_UNSUPPORTED,
                "rt_render_device_noalloc: this shard has " + std::to_string(ntiles) + " tiles, the maximum per call is " +
                    std::to_string(RT_NOALLOC_MAX_TILES) + " (render fewer tiles per call: a larger tile_step)");
  const int64_t batch_tiles = rt_noalloc_max_tiles(depth);
  if (batch_tiles < 1) return fail(RT_ERR_UNSUPPORTED, "rt_render_device_noalloc: depth too large for the scratch array");
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  HIP_TRY(hipSetDevice(s->device));
  unsigned char* base = nullptr;
  if ((rc = scratch_for_device(s->device, &base))) return rc;
  const bool fast = mode == RT_MODE_F32_FAST;
  constexpr int OVERFLOW_SLOT = 2 * (MAX_DEPTH + 1);
  for (int64_t k0 = 0; k0 < ntiles; k0 += batch_tiles) {
    const int64_t bt = std::min<int64_t>(batch_tiles, ntiles - k0);
    NoallocLayout L;
    noalloc_layout(bt * (RT_TILE * RT_TILE), depth, L);
    PixelMap pm = pixel_map(cam, tile_first, tile_step);
    pm.tile_first = (int)(tile_first + k0 * tile_step);  // local tile k of this batch = shard tile k0 + k
    pm.out_k_mul = 1;
    pm.out_k_add = (int)k0;
    unsigned int* counts = reinterpret_cast<unsigned int*>(base + L.counts);
    HIP_TRY(hipMemsetAsync(counts, 0, 2 * (MAX_DEPTH + 2) * sizeof(unsigned int), st));
    for (int l = 0; l < depth; l++) {
      const bool spawn = l + 1 < depth;
      const long long n = L.cap[l];
      dim3 grid((unsigned)((n + WF_BLOCK - 1) / WF_BLOCK)), block(WF_BLOCK);
      const WfTask* tin = reinterpret_cast<const WfTask*>(base + L.tasks[l]);
      WfNode* nodes = reinterpret_cast<WfNode*>(base + L.nodes[l]);
      uint8_t* live = base + L.live[l];
      WfNode* parents = l > 0 ? reinterpret_cast<WfNode*>(base + L.nodes[l - 1]) : nullptr;
      WfTask* tnext = spawn ? reinterpret_cast<WfTask*>(base + L.tasks[l + 1]) : nullptr;
      const unsigned int cap_in = l > 0 ? (unsigned int)(2 * L.cap[l - 1]) : 0u;
      const unsigned int cap = spawn ? (unsigned int)(2 * n) : 0u;
      const unsigned int* counts_in = l > 0 ? counts + 2 * l : nullptr;
#define NA_LAUNCH(PR, FA)                                                                                   \
  hipLaunchKernelGGL((k_wf_level<false, PR, FA, true, false>), grid, block, 0, st, s->d, dc, pm, tin, n, n, cap_in, \
                     nodes, live, parents, d_rgb_f64, d_rgb_u8, tnext, counts + 2 * (l + 1), cap,            \
                     spawn ? 1 : 0, s->d_counters, counts_in, counts + OVERFLOW_SLOT, SortGrid{}, (const unsigned int*)nullptr, \
                     (unsigned int*)nullptr, (unsigned int*)nullptr, Placement{nullptr, nullptr, 0}, (WfShadow*)nullptr, \
                     (unsigned int*)nullptr, ShadowPass{}, (long long)grid.x)
      if (l == 0) {
        if (fast) NA_LAUNCH(true, true); else NA_LAUNCH(true, false);
      } else {
        if (fast) NA_LAUNCH(false, true); else NA_LAUNCH(false, false);
      }
#undef NA_LAUNCH
      HIP_TRY(hipGetLastError());
    }
    for (int l = depth - 2; l >= 0; l--) {  // the levels that spawned: back up the recursion tree
      const long long m = L.cap[l];
      dim3 block(256), grid((unsigned)((m + 255) / 256));
      const WfNode* nodes = reinterpret_cast<const WfNode*>(base + L.nodes[l]);
      const uint8_t* live = base + L.live[l];
      if (l == 0)
        hipLaunchKernelGGL(k_wf_combine<true>, grid, block, 0, st, nodes, live, m, (WfNode*)nullptr, d_rgb_f64,
                           d_rgb_u8, (const unsigned int*)nullptr, (const unsigned int*)nullptr);
      else
        hipLaunchKernelGGL(k_wf_combine<false>, grid, block, 0, st, nodes, live, m,
                           reinterpret_cast<WfNode*>(base + L.nodes[l - 1]), d_rgb_f64, d_rgb_u8,
                           (const unsigned int*)(counts + 2 * l), (const unsigned int*)nullptr);
      HIP_TRY(hipGetLastError());
    }
  }
  return RT_OK;
}

}  // extern "C"

// Internal to the library (csrc/rt_internal.h; used by the host mirror's camera sweeps).
namespace rti {
int render_device_estimated(rt_scene* s, const rt_camera* cam, int depth, int mode, int pipeline,
                            uint8_t* d_rgb_u8, void* hip_stream) {
  if (!s) return fail(RT_ERR_ARG, "null argument");
  s->allow_estimates = (pipeline == RT_PIPELINE_WAVEFRONT);
  const int rc = rt_render_device(s, cam, depth, mode, pipeline, 0, 1, nullptr, d_rgb_u8, hip_stream, 0);
  s->allow_estimates = false;
  return rc;
}

int estimated_frame_result(rt_scene* s, void* hip_stream, int* complete) {
  if (!s || !complete) return fail(RT_ERR_ARG, "null argument");
  *complete = 1;
  const hipStream_t st = static_cast<hipStream_t>(hip_stream);
  // the frame's buffer set may have moved on to another stream (or frame) meanwhile: its verdict
  // was taken then
  for (size_t i = 0; i < s->orphan_results.size(); i++)
    if (s->orphan_results[i].first == st) {
      *complete = s->orphan_results[i].second;
      s->orphan_results.erase(s->orphan_results.begin() + (long)i);
      return RT_OK;
    }
  rt_scene::WfState* w = nullptr;
  for (rt_scene::WfState* c : s->wf_states)
    if (c->stream == st) w = c;
  if (!w || !w->estimate_pending) return RT_OK;  // the last frame on this stream was issued the exact way
  resolve_estimate(s, w, complete);
  return RT_OK;
}
int sweep_slot(rt_scene* s, int i, size_t bytes, void** stream, uint8_t** device, uint8_t** pinned) {
  if (!s || !stream || !device || !pinned || i < 0 || i >= 8) return fail(RT_ERR_ARG, "bad sweep slot");
  if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
  HIP_TRY(hipSetDevice(s->device));
  if ((int)s->sweep_slots.size() <= i) s->sweep_slots.resize((size_t)i + 1);
  rt_scene::SweepSlot& q = s->sweep_slots[i];
  if (!q.stream) HIP_TRY(hipStreamCreateWithFlags(&q.stream, hipStreamNonBlocking));
  if (q.bytes < bytes) {
    HIP_TRY(hipStreamSynchronize(q.stream));
    if (q.device) (void)hipFree(q.device);
    if (q.pinned) (void)hipHostFree(q.pinned);
    q.device = q.pinned = nullptr;
    q.bytes = 0;
    HIP_TRY(hipMalloc((void**)&q.device, bytes));
    HIP_TRY(hipHostMalloc((void**)&q.pinned, bytes, hipHostMallocDefault));
    q.bytes = bytes;
  }
  *stream = q.stream;
  *device = q.device;
  *pinned = q.pinned;
  return RT_OK;
}

void estimate_counters(const rt_scene* s, long long* issued, long long* missed) {
  if (issued) *issued = s ? s->n_estimated : 0;
  if (missed) *missed = s ? s->n_estimate_missed : 0;
}
}  // namespace rti

extern "C" {
// RT_PIPELINE_AUTO: the first time a frame (camera, shard, depth, mode) is seen it is rendered
// once with each pipeline and timed with HIP events (both write the same bits); the faster
// one is remembered for that frame.
static int render_device_any(rt_scene* s, const rt_camera* cam, int depth, int mode, int pipeline,
                             int tile_first, int tile_step, double* d_rgb_f64, uint8_t* d_rgb_u8,
                             void* hip_stream, int want_stats, bool compact) {
  if (pipeline != RT_PIPELINE_AUTO)
    return render_device_fixed(s, cam, depth, mode, pipeline, tile_first, tile_step, d_rgb_f64, d_rgb_u8,
                               hip_stream, want_stats, compact);
  if (!s || !cam) return fail(RT_ERR_ARG, "null argument");
  if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
  if (mode == RT_MODE_F32_FAST || depth == 0)  // one pipeline serves these
    return render_device_fixed(s, cam, depth, mode, depth == 0 ? RT_PIPELINE_MEGAKERNEL : RT_PIPELINE_WAVEFRONT,
                               tile_first, tile_step, d_rgb_f64, d_rgb_u8, hip_stream, want_stats, compact);
  DCamera dc;
  int rc = camera_to_device(cam, dc);
  if (rc) return rc;
  const PixelMap pm = pixel_map(cam, tile_first, tile_step, compact);
  const uint64_t key = frame_key(dc, pm, depth, false);
  const FrameId id = frame_id(dc, tile_first, tile_step, depth, false, compact);
  for (const auto& kv : s->auto_choice)
    if (kv.key == key && same_frame(kv.id, id))
      return render_device_fixed(s, cam, depth, mode, kv.pipeline, tile_first, tile_step, d_rgb_f64, d_rgb_u8,
                                 hip_stream, want_stats, compact);
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  float best_ms = 0.f;
  int best = RT_PIPELINE_WAVEFRONT;
  const int candidates[2] = {RT_PIPELINE_WAVEFRONT, RT_PIPELINE_MEGAKERNEL};
  for (int c = 0; c < 2; c++) {
    float ms = 0.f;
    for (int rep = 0; rep < 2; rep++) {  // the first run of a pipeline sizes its buffers
      rc = render_device_fixed(s, cam, depth, mode, candidates[c], tile_first, tile_step, d_rgb_f64, d_rgb_u8,
                               hip_stream, 0, compact);
      if (rc) return rc;
      HIP_TRY(hipStreamSynchronize(st));
      HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    }
    if (c == 0 || ms < best_ms) {
      best_ms = ms;
      best = candidates[c];
    }
  }
  s->auto_choice.push_back({key, id, best});
  return render_device_fixed(s, cam, depth, mode, best, tile_first, tile_step, d_rgb_f64, d_rgb_u8, hip_stream,
                             want_stats, compact);
}

int rt_render_device(rt_scene* s, const rt_camera* cam, int depth, int mode, int pipeline,
                     int tile_first, int tile_step, double* d_rgb_f64, uint8_t* d_rgb_u8,
                     void* hip_stream, int want_stats) {
  return render_device_any(s, cam, depth, mode, pipeline, tile_first, tile_step, d_rgb_f64, d_rgb_u8, hip_stream,
                           want_stats, false);
}

int rt_render_stats(rt_scene* s, rt_stats* st) {
  if (!s || !st) return fail(RT_ERR_ARG, "null argument");
  if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
  HIP_TRY(hipSetDevice(s->device));
  memset(st, 0, sizeof(*st));
  if (s->timing_pending) {
    HIP_TRY(hipEventSynchronize(s->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    st->kernel_ms = ms;
  }
  unsigned long long c[N_COUNTERS + 1];
  HIP_TRY(hipMemcpy(c, s->d_counters, sizeof(c), hipMemcpyDeviceToHost));
  st->max_nodes_per_query = c[N_COUNTERS];
  st->rays_primary = c[0];
  st->rays_shadow = c[1];
  st->rays_reflect = c[2];
  st->rays_refract = c[3];
  st->depth0_skipped = c[4];
  st->pred_queries = c[5];
  st->nodes_visited = c[6];
  st->tri_tests = c[7];
  st->prim_tests = c[8];
  st->inst_entries = c[9];
  st->shade_calls = c[10];
  st->node_wave_steps = c[11];
  st->tri_wave_steps = c[12];
  st->fast_candidates = c[13];
  st->fast_exact = c[14];
  st->launches = s->launches;
  return RT_OK;
}

int rt_render(rt_scene* s, const rt_camera* cam, int depth, int mode, int pipeline,
              double* out_rgb_f64, uint8_t* out_rgb_u8, rt_stats* stats) {
  if (!s || !cam) return fail(RT_ERR_ARG, "null argument");
  if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
  if (cam->hsize <= 0 || cam->vsize <= 0) return fail(RT_ERR_ARG, "camera size must be positive");
  auto t0 = std::chrono::steady_clock::now();
  HIP_TRY(hipSetDevice(s->device));
  size_t npx = (size_t)cam->hsize * cam->vsize;
  if (npx > s->out_cap_px) {
    if (s->d_out_f64) (void)hipFree(s->d_out_f64);
    if (s->d_out_u8) (void)hipFree(s->d_out_u8);
    s->d_out_f64 = nullptr;
    s->d_out_u8 = nullptr;
    s->out_cap_px = 0;
    HIP_TRY(hipMalloc(&s->d_out_f64, npx * 3 * sizeof(double)));
    HIP_TRY(hipMalloc(&s->d_out_u8, npx * 3));
    s->out_cap_px = npx;
  }
  int rc = rt_render_device(s, cam, depth, mode, pipeline, 0, 1, out_rgb_f64 ? s->d_out_f64 : nullptr,
                            (out_rgb_u8 || !out_rgb_f64) ? s->d_out_u8 : nullptr, nullptr,
                            stats ? 1 : 0);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  if (out_rgb_f64)
    HIP_TRY(hipMemcpy(out_rgb_f64, s->d_out_f64, npx * 3 * sizeof(double), hipMemcpyDeviceToHost));
  if (out_rgb_u8) HIP_TRY(hipMemcpy(out_rgb_u8, s->d_out_u8, npx * 3, hipMemcpyDeviceToHost));
  if (stats) {
    rc = rt_render_stats(s, stats);
    if (rc) return rc;
    stats->total_ms =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  return RT_OK;
}

}  // extern "C"
template <typename T>
static int detile_any(const T* d_tiles, int nshards, int64_t tiles_per_shard, int hsize, int vsize, T* d_image,
                      void* hip_stream) {
  if (!d_tiles || !d_image || nshards <= 0 || hsize <= 0 || vsize <= 0 || tiles_per_shard < 0)
    return fail(RT_ERR_ARG, "bad argument");
  int tiles_x = (hsize + RT_TILE - 1) / RT_TILE;
  int64_t total = (int64_t)tiles_x * ((vsize + RT_TILE - 1) / RT_TILE);
  if ((total + nshards - 1) / nshards > tiles_per_shard)
    return fail(RT_ERR_ARG, "tiles_per_shard too small for this image");
  long long npx = (long long)hsize * vsize;
  dim3 block(256), grid((unsigned)((npx + 255) / 256));
  hipLaunchKernelGGL(k_detile<T>, grid, block, 0, static_cast<hipStream_t>(hip_stream), d_tiles,
                     nshards, (long long)tiles_per_shard, hsize, vsize, tiles_x, d_image);
  HIP_TRY(hipGetLastError());
  return RT_OK;
}
extern "C" {
int rt_detile_u8(const uint8_t* d_tiles, int nshards, int64_t tiles_per_shard, int hsize, int vsize,
                 uint8_t* d_image, void* hip_stream) {
  return detile_any(d_tiles, nshards, tiles_per_shard, hsize, vsize, d_image, hip_stream);
}
int rt_detile_f64(const double* d_tiles, int nshards, int64_t tiles_per_shard, int hsize, int vsize,
                  double* d_image, void* hip_stream) {
  return detile_any(d_tiles, nshards, tiles_per_shard, hsize, vsize, d_image, hip_stream);
}

int rt_trace_rays(rt_scene* s, int64_t n, const double* origins, const double* directions,
                  int depth, double* out) {
  if (!s || !origins || !directions || !out || n < 0) return fail(RT_ERR_ARG, "bad argument");
  if (!s->committed) return fail(RT_ERR_STATE, "scene not committed");
  if (depth < 0 || depth > MAX_DEPTH) return fail(RT_ERR_ARG, "depth out of range (0..8)");
  if (n == 0) return RT_OK;
  HIP_TRY(hipSetDevice(s->device));
  double *d_o = nullptr, *d_d = nullptr, *d_out = nullptr;
  HIP_TRY(hipMalloc(&d_o, n * 3 * sizeof(double)));
  HIP_TRY(hipMalloc(&d_d, n * 3 * sizeof(double)));
  HIP_TRY(hipMalloc(&d_out, n * 32 * sizeof(double)));
  HIP_TRY(hipMemcpy(d_o, origins, n * 3 * sizeof(double), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_d, directions, n * 3 * sizeof(double), hipMemcpyHostToDevice));
  dim3 block(64), grid((unsigned)((n + 63) / 64));
  hipLaunchKernelGGL(k_trace_rays, grid, block, 0, nullptr, s->d, (long long)n, d_o, d_d, depth, d_out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(out, d_out, n * 32 * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(d_o);
  (void)hipFree(d_d);
  (void)hipFree(d_out);
  if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("rt_trace_rays: ") + hipGetErrorString(e));
  return RT_OK;
}

// =======================================================================================
// Several GPUs of one node, one host thread.  The reference shares World and Camera between
// threads (Arc, camera.rs:95-100), gives each a band of rows (camera.rs:114-143), joins them and
// stitches the bands (camera.rs:147-162, canvas.rs:47-64).  Here the scene is replicated on
// every GPU, the frame's 32x32-pixel tiles are dealt round-robin to the GPUs (bands would be
// badly balanced: glass pixels cost several times what floor pixels do), and ONE exchange ends
// the frame: every GPU sends its compact tile buffer to the first one (grouped ncclSend/ncclRecv:
// each peer has its own xGMI link into it), which stitches them (k_detile).
// RCCL is loaded when the first multi-GPU scene is committed (dlopen), so a process that
// renders on one GPU through the other entry points never touches it.
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static Rccl g_rccl;
static int rccl_load() {
  if (g_rccl.lib) return RT_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) return fail(RT_ERR_UNSUPPORTED, std::string("RCCL not found (librccl.so.1): ") + dlerror());
  Rccl r;
  r.lib = h;
#define RT_SYM(field, name)                                                                \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                            \
  if (!r.field) return fail(RT_ERR_UNSUPPORTED, std::string("RCCL lacks ") + name)
  RT_SYM(CommInitAll, "ncclCommInitAll");
  RT_SYM(CommDestroy, "ncclCommDestroy");
  RT_SYM(GroupStart, "ncclGroupStart");
  RT_SYM(GroupEnd, "ncclGroupEnd");
  RT_SYM(Send, "ncclSend");
  RT_SYM(Recv, "ncclRecv");
  RT_SYM(GetErrorString, "ncclGetErrorString");
#undef RT_SYM
  g_rccl = r;
  return RT_OK;
}
#define NCCL_TRY(expr)                                                                              \
  do {                                                                                              \
    ncclResult_t r_ = (expr);                                                                       \
    if (r_ != ncclSuccess) return fail(RT_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
  } while (0)

struct Multi {
  std::vector<int> devs;
  std::vector<rt_scene*> scenes;  // scenes[0] is the scene the caller holds; the others are its replicas
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> streams;
  std::vector<uint8_t*> d_tiles;  // per GPU: its shard's compact tile buffer
  size_t tiles_cap = 0;           // bytes of each
  uint8_t* d_all = nullptr;       // first GPU: every shard's buffer back to back
  size_t all_cap = 0;
  uint8_t* d_image = nullptr;     // first GPU: the stitched image, when the caller gave no device buffer
  size_t image_cap = 0;
  // the same three for the f64 colours (allocated when a frame first asks for them)
  std::vector<double*> d_tiles_f64;
  size_t tiles_f64_cap = 0;
  double* d_all_f64 = nullptr;
  size_t all_f64_cap = 0;
  double* d_image_f64 = nullptr;
  size_t image_f64_cap = 0;
  // Frames this scene has rendered before: their queue lengths (or pipeline choice) are known on
  // every GPU, so the shards are issued one after the other without any wait.  A frame not in
  // this list is planned with read-backs per recursion level; its shards are issued by one host
  // thread per GPU, so the GPUs plan side by side instead of one after the other.
  struct Seen {
    FrameId id;
    int pipeline;
  };
  std::vector<Seen> seen;
  bool ready = false;             // rt_scene_commit_multi ran to its end
};

// Puts the caller's current device back when an entry point that visits several GPUs returns,
// whichever way it returns.
struct CurrentDeviceKeeper {
  int device = -1;
  CurrentDeviceKeeper() {
    if (hipGetDevice(&device) != hipSuccess) device = -1;
  }
  ~CurrentDeviceKeeper() {
    if (device >= 0) (void)hipSetDevice(device);
  }
};

static void multi_destroy(Multi* m) {
  if (!m) return;
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (size_t d = 0; d < m->devs.size(); d++) {
    (void)hipSetDevice(m->devs[d]);
    if (d < m->comms.size() && m->comms[d] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->comms[d]);
    if (d < m->streams.size() && m->streams[d]) (void)hipStreamDestroy(m->streams[d]);
    if (d < m->d_tiles.size() && m->d_tiles[d]) (void)hipFree(m->d_tiles[d]);
    if (d < m->d_tiles_f64.size() && m->d_tiles_f64[d]) (void)hipFree(m->d_tiles_f64[d]);
    if (d == 0) {
      if (m->d_all) (void)hipFree(m->d_all);
      if (m->d_image) (void)hipFree(m->d_image);
      if (m->d_all_f64) (void)hipFree(m->d_all_f64);
      if (m->d_image_f64) (void)hipFree(m->d_image_f64);
    }
    if (d > 0 && d < m->scenes.size() && m->scenes[d]) rt_scene_destroy(m->scenes[d]);
  }
  (void)hipSetDevice(cur);
  delete m;
}

}  // extern "C"
// Runs job(d) for d = first .. ndev-1, each on a host thread of its own with devs[d] current,
// and returns the first failure (its message becomes the caller's rt_last_error()).
template <typename Job>
static int per_device_threads(const std::vector<int>& devs, int first, Job job) {
  const int ndev = (int)devs.size();
  std::vector<int> rcs((size_t)ndev, RT_OK);
  std::vector<std::string> errs((size_t)ndev);
  std::vector<std::thread> threads;
  for (int d = first; d < ndev; d++)
    threads.emplace_back([&, d]() {
      if (hipSetDevice(devs[(size_t)d]) != hipSuccess) {
        rcs[(size_t)d] = RT_ERR_HIP;
        errs[(size_t)d] = "hipSetDevice failed";
        return;
      }
      rcs[(size_t)d] = job(d);
      if (rcs[(size_t)d]) errs[(size_t)d] = g_err;  // the error text is thread-local
    });
  for (std::thread& t : threads) t.join();
  for (int d = first; d < ndev; d++)
    if (rcs[(size_t)d]) return fail(rcs[(size_t)d], "GPU " + std::to_string(devs[(size_t)d]) + ": " + errs[(size_t)d]);
  return RT_OK;
}
// grow-only device buffer on the current device; *cap is 0 while the pointer is not valid
template <typename T>
static int multi_ensure(T** p, size_t* cap, size_t need, hipStream_t wait_for) {
  if (need <= *cap && *p) return RT_OK;
  HIP_TRY(hipStreamSynchronize(wait_for));
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  HIP_TRY(hipMalloc((void**)p, need * sizeof(T)));
  *cap = need;
  return RT_OK;
}
extern "C" {

int rt_scene_commit_multi(rt_scene* s, const int* device_ids, int ndev, int bvh_builder) {
  if (!s || !device_ids) return fail(RT_ERR_ARG, "null argument");
  if (s->committed) return fail(RT_ERR_STATE, "scene already committed");
  const int visible = rt_device_count();
  if (visible <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device visible; the render core has no CPU path");
  if (ndev < 1 || ndev > 64) return fail(RT_ERR_ARG, "ndev must be 1..64");
  for (int a = 0; a < ndev; a++) {
    if (device_ids[a] < 0 || device_ids[a] >= visible) return fail(RT_ERR_ARG, "device index out of range");
    for (int b = 0; b < a; b++)
      if (device_ids[a] == device_ids[b]) return fail(RT_ERR_ARG, "device listed twice");
  }
  int rc = rccl_load();
  if (rc) return rc;
  CurrentDeviceKeeper keep;
  Multi* m = new Multi;
  m->devs.assign(device_ids, device_ids + ndev);
  m->scenes.assign((size_t)ndev, nullptr);
  m->scenes[0] = s;
  s->multi = m;  // from here on rt_scene_destroy(s) releases everything below
  // The first GPU's commit builds the host-built trees and keeps them; the replicas upload those
  // (one host thread per GPU, side by side) instead of building them again.  Meshes that go to the
  // device builder are built on every GPU, also side by side.
  s->keep_host_built = true;
  rc = rt_scene_commit(s, m->devs[0], bvh_builder);
  s->keep_host_built = false;
  if (rc) {
    s->host_built.clear();
    return rc;
  }
  for (int d = 1; d < ndev; d++) {  // replicas: the same description, committed to the other GPUs
    rt_scene* r = new rt_scene();
    r->mats = s->mats;
    r->groups = s->groups;
    r->prims = s->prims;
    r->prim_order = s->prim_order;
    r->meshes = s->meshes;
    r->lights = s->lights;
    r->next_order = s->next_order;
    r->host_built = s->host_built;
    m->scenes[(size_t)d] = r;
  }
  rc = per_device_threads(m->devs, 1, [&](int d) { return rt_scene_commit(m->scenes[(size_t)d], m->devs[(size_t)d], bvh_builder); });
  s->host_built.clear();
  for (int d = 1; d < ndev; d++) {
    rt_scene* r = m->scenes[(size_t)d];
    r->host_built.clear();
    std::vector<MeshH>().swap(r->meshes);  // a replica's host copy of the geometry has served its purpose
  }
  if (rc) return rc;
  m->comms.assign((size_t)ndev, nullptr);
  NCCL_TRY(g_rccl.CommInitAll(m->comms.data(), ndev, m->devs.data()));
  m->streams.assign((size_t)ndev, nullptr);
  m->d_tiles.assign((size_t)ndev, nullptr);
  m->d_tiles_f64.assign((size_t)ndev, nullptr);
  for (int d = 0; d < ndev; d++) {
    HIP_TRY(hipSetDevice(m->devs[(size_t)d]));
    HIP_TRY(hipStreamCreateWithFlags(&m->streams[(size_t)d], hipStreamNonBlocking));
  }
  m->ready = true;
  return RT_OK;
}

int rt_scene_device_count(const rt_scene* s) {
  if (!s || !s->committed) return 0;
  if (s->multi && !s->multi->ready) return 0;  // a multi-GPU commit that failed half way
  return s->multi ? (int)s->multi->devs.size() : 1;
}

int rt_render_multi_f64(rt_scene* s, const rt_camera* cam, int depth, int mode, int pipeline, double* out_rgb_f64,
                        double* d_rgb_f64, uint8_t* out_rgb_u8, uint8_t* d_rgb_u8, rt_stats* stats) {
  if (!s || !cam) return fail(RT_ERR_ARG, "null argument");
  if (!s->committed || !s->multi || !s->multi->ready)
    return fail(RT_ERR_STATE, "scene not committed with rt_scene_commit_multi (or that commit failed)");
  if (cam->hsize <= 0 || cam->vsize <= 0) return fail(RT_ERR_ARG, "camera size must be positive");
  const bool want_u8 = out_rgb_u8 || d_rgb_u8, want_f64 = out_rgb_f64 || d_rgb_f64;
  if (!want_u8 && !want_f64) return fail(RT_ERR_ARG, "no output buffer");
  DCamera dc;
  int rc = camera_to_device(cam, dc);
  if (rc) return rc;
  Multi* m = s->multi;
  const int ndev = (int)m->devs.size();
  auto t0 = std::chrono::steady_clock::now();
  CurrentDeviceKeeper keep;
  int64_t per = 0;
  for (int d = 0; d < ndev; d++) per = std::max(per, rt_shard_tiles(cam->hsize, cam->vsize, d, ndev));
  const size_t elems_per = (size_t)per * RT_TILE * RT_TILE * 3;  // colour components per shard buffer
  const size_t npx = (size_t)cam->hsize * cam->vsize;
  // Each capacity is zeroed before its buffers are touched: an allocation that fails half way
  // leaves nothing behind that a later call would trust.
  if (want_u8 && elems_per > m->tiles_cap) {
    m->tiles_cap = 0;
    for (int d = 0; d < ndev; d++) {
      HIP_TRY(hipSetDevice(m->devs[(size_t)d]));
      size_t cap = 0;
      if ((rc = multi_ensure(&m->d_tiles[(size_t)d], &cap, elems_per, m->streams[(size_t)d]))) return rc;
    }
    m->tiles_cap = elems_per;
  }
  if (want_f64 && elems_per > m->tiles_f64_cap) {
    m->tiles_f64_cap = 0;
    for (int d = 0; d < ndev; d++) {
      HIP_TRY(hipSetDevice(m->devs[(size_t)d]));
      size_t cap = 0;
      if ((rc = multi_ensure(&m->d_tiles_f64[(size_t)d], &cap, elems_per, m->streams[(size_t)d]))) return rc;
    }
    m->tiles_f64_cap = elems_per;
  }
  HIP_TRY(hipSetDevice(m->devs[0]));
  uint8_t* image = d_rgb_u8;
  double* image_f64 = d_rgb_f64;
  if (want_u8) {
    if ((rc = multi_ensure(&m->d_all, &m->all_cap, elems_per * (size_t)ndev, m->streams[0]))) return rc;
    if (!image) {
      if ((rc = multi_ensure(&m->d_image, &m->image_cap, npx * 3, m->streams[0]))) return rc;
      image = m->d_image;
    }
  }
  if (want_f64) {
    if ((rc = multi_ensure(&m->d_all_f64, &m->all_f64_cap, elems_per * (size_t)ndev, m->streams[0]))) return rc;
    if (!image_f64) {
      if ((rc = multi_ensure(&m->d_image_f64, &m->image_f64_cap, npx * 3, m->streams[0]))) return rc;
      image_f64 = m->d_image_f64;
    }
  }
  // every GPU renders its tiles (scene replicated, no communication) ...
  const FrameId id = frame_id(dc, 0, ndev, depth, mode == RT_MODE_F32_FAST, true);
  bool seen = false;
  for (const Multi::Seen& q : m->seen) seen = seen || (q.pipeline == pipeline && same_frame(q.id, id));
  auto issue = [&](int d) {
    return render_device_any(m->scenes[(size_t)d], cam, depth, mode, pipeline, d, ndev,
                             want_f64 ? m->d_tiles_f64[(size_t)d] : nullptr, want_u8 ? m->d_tiles[(size_t)d] : nullptr,
                             m->streams[(size_t)d], stats ? 1 : 0, true);
  };
  if (seen || ndev == 1) {
    for (int d = 0; d < ndev; d++) {
      HIP_TRY(hipSetDevice(m->devs[(size_t)d]));
      if ((rc = issue(d))) return rc;
    }
  } else {
    // a frame not rendered before costs a host read-back per recursion level (or the timing runs
    // of RT_PIPELINE_AUTO) on every GPU: one host thread per GPU, so they plan side by side
    if ((rc = per_device_threads(m->devs, 0, issue))) return rc;
  }
  if (!seen) {
    if (m->seen.size() >= 64) m->seen.erase(m->seen.begin());
    m->seen.push_back({id, pipeline});
  }
  // ... and ONE exchange: all shards to the first GPU.  The first GPU's own shard travels the
  // same way (a send to itself), so one GPU alone runs the code path of eight.
  NCCL_TRY(g_rccl.GroupStart());
  for (int d = 0; d < ndev; d++) {
    ncclResult_t r1 = ncclSuccess, r2 = ncclSuccess, r3 = ncclSuccess, r4 = ncclSuccess;
    if (want_u8) {
      r1 = g_rccl.Send(m->d_tiles[(size_t)d], elems_per, ncclUint8, 0, m->comms[(size_t)d], m->streams[(size_t)d]);
      r2 = g_rccl.Recv(m->d_all + (size_t)d * elems_per, elems_per, ncclUint8, d, m->comms[0], m->streams[0]);
    }
    if (want_f64) {
      r3 = g_rccl.Send(m->d_tiles_f64[(size_t)d], elems_per, ncclDouble, 0, m->comms[(size_t)d], m->streams[(size_t)d]);
      r4 = g_rccl.Recv(m->d_all_f64 + (size_t)d * elems_per, elems_per, ncclDouble, d, m->comms[0], m->streams[0]);
    }
    for (ncclResult_t r : {r1, r2, r3, r4})
      if (r != ncclSuccess) {
        (void)g_rccl.GroupEnd();
        return fail(RT_ERR_HIP, std::string("ncclSend/ncclRecv: ") + g_rccl.GetErrorString(r));
      }
  }
  NCCL_TRY(g_rccl.GroupEnd());
  HIP_TRY(hipSetDevice(m->devs[0]));
  if (want_u8) {
    if ((rc = rt_detile_u8(m->d_all, ndev, per, cam->hsize, cam->vsize, image, m->streams[0]))) return rc;
    if (out_rgb_u8) HIP_TRY(hipMemcpyAsync(out_rgb_u8, image, npx * 3, hipMemcpyDeviceToHost, m->streams[0]));
  }
  if (want_f64) {
    if ((rc = rt_detile_f64(m->d_all_f64, ndev, per, cam->hsize, cam->vsize, image_f64, m->streams[0]))) return rc;
    if (out_rgb_f64)
      HIP_TRY(hipMemcpyAsync(out_rgb_f64, image_f64, npx * 3 * sizeof(double), hipMemcpyDeviceToHost, m->streams[0]));
  }
  for (int d = ndev - 1; d >= 0; d--) {
    HIP_TRY(hipSetDevice(m->devs[(size_t)d]));
    HIP_TRY(hipStreamSynchronize(m->streams[(size_t)d]));
  }
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    for (int d = 0; d < ndev; d++) {
      rt_stats one;
      if ((rc = rt_render_stats(m->scenes[(size_t)d], &one))) return rc;
      stats->rays_primary += one.rays_primary;
      stats->rays_shadow += one.rays_shadow;
      stats->rays_reflect += one.rays_reflect;
      stats->rays_refract += one.rays_refract;
      stats->depth0_skipped += one.depth0_skipped;
      stats->pred_queries += one.pred_queries;
      stats->nodes_visited += one.nodes_visited;
      stats->tri_tests += one.tri_tests;
      stats->prim_tests += one.prim_tests;
      stats->inst_entries += one.inst_entries;
      stats->shade_calls += one.shade_calls;
      stats->launches += one.launches;
      stats->node_wave_steps += one.node_wave_steps;
      stats->tri_wave_steps += one.tri_wave_steps;
      stats->fast_candidates += one.fast_candidates;
      stats->fast_exact += one.fast_exact;
      stats->kernel_ms = std::max(stats->kernel_ms, one.kernel_ms);  // the GPUs render side by side
      stats->max_nodes_per_query = std::max(stats->max_nodes_per_query, one.max_nodes_per_query);
    }
    stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  return RT_OK;
}

int rt_render_multi(rt_scene* s, const rt_camera* cam, int depth, int mode, int pipeline, uint8_t* out_rgb_u8,
                    uint8_t* d_rgb_u8, rt_stats* stats) {
  if (!out_rgb_u8 && !d_rgb_u8) return fail(RT_ERR_ARG, "no output buffer");
  return rt_render_multi_f64(s, cam, depth, mode, pipeline, nullptr, nullptr, out_rgb_u8, d_rgb_u8, stats);
}

}  // extern "C"
