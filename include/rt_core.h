/* rt_core.h — C ABI of the MI355X render core.
 *
 * The core replaces ONE path of the reference ray tracer: everything that runs
 * below `scene::camera::render` (src/scene/camera.rs:83), i.e. the per-pixel
 * loop `render_thread` (camera.rs:165-184) with `ray_for_pixel`
 * (camera.rs:64-80), `World::color_at` and everything it calls
 * (src/scene/world.rs:22-165, src/shapes/intersect.rs:57-237,
 * src/scene/light.rs:23-85, every shape's local_intersect/local_normal_at).
 *
 * The reference has no FFI of its own; the entry points below are what a
 * binding for that seam would call: the host flattens `World` (world.rs:9-12)
 * and `Camera` (camera.rs:11-19) into the plain structs declared here, commits
 * them to a GPU once, and asks for frames.  Plain C structs, no callbacks, no
 * exceptions across the boundary; every call returns a status (or a
 * non-negative index), and rt_last_error() gives the thread's last message.
 *
 * Ordering contract.  The reference resolves equal-t intersections by list
 * position (stable sorts in group.rs:69 and world.rs:29; `hit`,
 * intersect.rs:125-139, keeps the first of equal minima), and list position is
 * the depth-first order of the leaf shapes in `World.objects`.  Leaves
 * (primitives and meshes) must therefore be added in that depth-first order;
 * a mesh occupies `ntris` consecutive positions in face order.
 *
 * Transforms.  Every `inv` is the INVERSE transform as the reference stores it
 * (`inverse_transform`, e.g. sphere.rs:13, computed by Matrix::inverse,
 * math/matrix.rs:154-168), row-major.  The transposed copy
 * (`inverse_transform_transpose`) is derived from it.  Only affine matrices
 * (last row exactly 0,0,0,1) are accepted.
 */
#ifndef RT_CORE_H
#define RT_CORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_scene rt_scene;

/* status codes (returned negated by the functions that return an index) */
enum {
  RT_OK = 0,
  RT_ERR_ARG = 1,         /* NULL / out-of-range argument */
  RT_ERR_STATE = 2,       /* call order violated (e.g. add after commit) */
  RT_ERR_HIP = 3,         /* a HIP runtime call failed; see rt_last_error() */
  RT_ERR_UNSUPPORTED = 4, /* e.g. projective transform, mode not built */
  RT_ERR_NO_DEVICE = 5    /* no GPU visible: the core has no CPU fallback */
};

/* shapes/{sphere,plane,cube,cylinder,cone}.rs */
enum { RT_SPHERE = 0, RT_PLANE = 1, RT_CUBE = 2, RT_CYLINDER = 3, RT_CONE = 4 };
/* draw/patterns.rs:29-306 */
enum { RT_SOLID = 0, RT_STRIPE = 1, RT_GRADIENT = 2, RT_RINGS = 3, RT_CHECKERED = 4 };
/* Arithmetic mode of the kernels.  RT_MODE_F32_SEARCH (historic name RT_MODE_F32_FAST): the
 * triangle search runs in f32 with error bounds and everything it cannot decide, and all
 * shading, in f64; it renders the parity mode's frames bit for bit and is NOT faster on CDNA4
 * (non-packed f32 and f64 vector instructions issue at the same rate there, and the bounds double
 * the test's instructions): about 20 % slower than RT_MODE_F64_PARITY.  Kept for its counters. */
enum { RT_MODE_F64_PARITY = 0, RT_MODE_F32_FAST = 1, RT_MODE_F32_SEARCH = 1 };
/* kernel organisation: one lane per pixel with an explicit recursion stack, or per-depth
 * ray queues with one fused trace+shade kernel per depth ("wavefront") */
enum { RT_PIPELINE_MEGAKERNEL = 0, RT_PIPELINE_WAVEFRONT = 1,
       /* whichever of the two is faster for this frame: measured the first time a (camera,
        * shard, depth) is rendered, remembered afterwards; the frames are the same bits */
       RT_PIPELINE_AUTO = 2 };

/* draw/material.rs:6-15 + the pattern it boxes (patterns.rs).  pattern_inv is the
 * pattern's inverse transform (identity for Solid, patterns.rs:42-45). */
typedef struct rt_material {
  double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
  int32_t pattern_kind;
  int32_t reserved;
  double color_a[3];
  double color_b[3];
  double pattern_inv[16];
} rt_material;

/* shapes/group.rs:10-18: a transform node.  parent = index of the enclosing group or -1. */
typedef struct rt_group {
  double inv[16];
  int32_t parent;
  int32_t reserved;
} rt_group;

/* One closed-form shape.  minimum/maximum/closed are read for cylinders and cones
 * only (cylinder.rs:17-19, cone.rs:17-19).  parent = enclosing group or -1. */
typedef struct rt_primitive {
  int32_t kind;
  int32_t material;
  int32_t parent;
  int32_t closed;
  double minimum, maximum;
  double inv[16];
} rt_primitive;

/* The triangle children of one group (triangle.rs:13-26, smooth_triangle.rs:14-30),
 * each with an identity transform of its own, all sharing one material
 * (group.rs:48-53 copies the group's material into every child).
 * p1,p2,p3: ntris x 3 doubles each (xyz per triangle).  n1,n2,n3: per-vertex normals,
 * or all three NULL for flat triangles.  `smooth` may be NULL (all smooth when normals
 * are given) or ntris bytes selecting per triangle (obj_parser.rs:80-96 decides per fan
 * triangle).  The library copies the arrays during the call. */
typedef struct rt_mesh {
  int32_t group;    /* index from rt_scene_add_group; required (>= 0) */
  int32_t material; /* index from rt_scene_add_material */
  int64_t ntris;
  const double* p1;
  const double* p2;
  const double* p3;
  const double* n1;
  const double* n2;
  const double* n3;
  const uint8_t* smooth;
} rt_mesh;

/* scene/camera.rs:11-19 after Camera::new (33-59); inv = transform.inverse() which
 * ray_for_pixel recomputes per pixel (camera.rs:73). */
typedef struct rt_camera {
  int32_t hsize, vsize;
  double half_width, half_height, pixel_size;
  double inv[16];
} rt_camera;

/* Counters of one render call.  A "ray" is one scene query of type primary, shadow,
 * reflection or refraction.  Reflection/refraction rays whose remaining depth is 0
 * (the reference traces them and discards the result, world.rs:37-39) are not traced
 * and are counted in depth0_skipped only. */
typedef struct rt_stats {
  uint64_t rays_primary, rays_shadow, rays_reflect, rays_refract;
  uint64_t depth0_skipped;
  uint64_t pred_queries;  /* refraction-container predecessor queries (intersect.rs:188-222); the wavefront
                           * pipeline leaves them out at the deepest level, where n1 / n2 weigh black children */
  uint64_t nodes_visited; /* BVH nodes fetched */
  uint64_t tri_tests;     /* Möller–Trumbore evaluations (ttriangle.rs:9-34) */
  uint64_t prim_tests;    /* closed-form shape evaluations */
  uint64_t inst_entries;  /* ray-into-group-space transforms */
  uint64_t shade_calls;   /* shade_hit evaluations (world.rs:34-65) */
  uint64_t launches;      /* kernel launches issued */
  double kernel_ms;       /* HIP-event time of the render kernels on their stream */
  double total_ms;        /* host wall time of the call */
  uint64_t max_nodes_per_query; /* largest BVH node count one traversal needed */
  /* iterations of the node loop and of the triangle loop counted once per 64-lane wave:
   * nodes_visited / (64 * node_wave_steps) is the fraction of lanes doing useful work there */
  uint64_t node_wave_steps, tri_wave_steps;
  /* RT_MODE_F32_FAST only: candidates the f32 triangle search handed on, and how many triangle
   * evaluations in f64 it took to settle what the f32 values could not */
  uint64_t fast_candidates, fast_exact;
} rt_stats;

/* scene info after commit */
typedef struct rt_scene_info {
  int64_t n_groups, n_primitives, n_meshes, n_triangles, n_materials, n_lights;
  int64_t n_bvh_nodes;
  int64_t device_bytes;
  int32_t node_bytes; /* size of one BVH node record */
  int32_t tri_bytes;  /* size of one triangle record read by the intersection test */
  int32_t device;
  int32_t bvh_builder; /* as given to rt_scene_commit */
  double commit_ms;
} rt_scene_info;

const char* rt_last_error(void);
int rt_device_count(void);

/* ---- scene construction (host side; no GPU needed) --------------------------------- */
rt_scene* rt_scene_create(void);
void rt_scene_destroy(rt_scene*);
/* each returns the new index (>= 0) or -RT_ERR_* */
int rt_scene_add_material(rt_scene*, const rt_material*);
int rt_scene_add_group(rt_scene*, const rt_group*);
int rt_scene_add_primitive(rt_scene*, const rt_primitive*);
int rt_scene_add_mesh(rt_scene*, const rt_mesh*);
/* scene/light.rs:6-9 */
int rt_scene_add_light(rt_scene*, const double intensity[3], const double position[3]);

/* Builds the per-mesh BVHs and uploads everything to `device` (-1: current device).
 * bvh_builder: 0 = host binned-SAH build, 1 = device LBVH build (Morton codes, radix tree, fold
 * into four-wide nodes and triangle records all on the GPU), 2 = by mesh size (device build from
 * 100 000 triangles on).  Frames are the same bits with any of them.  After commit the scene is
 * immutable. */
int rt_scene_commit(rt_scene*, int device, int bvh_builder);
int rt_scene_get_info(const rt_scene*, rt_scene_info*);

/* ---- rendering ------------------------------------------------------------------------ */
/* Replaces scene::camera::render (camera.rs:83-163) + Color::clamp (draw/color.rs:148-155).
 * depth is the literal 5 of camera.rs:178.  Outputs are HOST buffers, row-major,
 * hsize*vsize*3 elements; either may be NULL.  stats may be NULL; when given, the
 * kernels run with counters enabled. */
int rt_render(rt_scene*, const rt_camera*, int depth, int mode, int pipeline,
              double* out_rgb_f64, uint8_t* out_rgb_u8, rt_stats* stats);

/* Same, DEVICE buffers, asynchronous on `hip_stream` (a hipStream_t, NULL = default
 * stream).  The image is cut into RT_TILE x RT_TILE pixel tiles numbered row-major;
 * this call renders the tiles t with t % tile_step == tile_first.  With tile_step == 1
 * the outputs are the row-major image.  With tile_step > 1 they are the compact tile
 * buffer of this shard: local tile k (global tile tile_first + k*tile_step) occupies
 * elements [k*RT_TILE*RT_TILE*3, (k+1)*RT_TILE*RT_TILE*3), pixels row-major inside the
 * tile; rt_shard_tiles() gives the tile count.  want_stats != 0 enables counters
 * (read them with rt_render_stats after synchronising the stream).
 * The wavefront pipeline remembers every frame it has rendered (camera, shard, depth, mode; a
 * bounded list per scene): the first render of a frame reads its queue lengths back level by
 * level (the call waits for the GPU then) and sorts its ray queues; the same frame afterwards is
 * issued without a wait and writes its rays in sorted order as it goes.  Same bits either way. */
#define RT_TILE 32
int rt_render_device(rt_scene*, const rt_camera*, int depth, int mode, int pipeline,
                     int tile_first, int tile_step, double* d_rgb_f64, uint8_t* d_rgb_u8,
                     void* hip_stream, int want_stats);
int rt_render_stats(rt_scene*, rt_stats* stats);
int64_t rt_shard_tiles(int hsize, int vsize, int tile_first, int tile_step);

/* Counterpart of draw::canvas::stitch_canvases (canvas.rs:47-64) for tile shards:
 * d_tiles holds `nshards` compact u8 tile buffers back to back, each
 * tiles_per_shard*RT_TILE*RT_TILE*3 bytes (shard s = tile_first s of tile_step nshards);
 * writes the row-major hsize*vsize*3 image to d_image.  Asynchronous on hip_stream. */
int rt_detile_u8(const uint8_t* d_tiles, int nshards, int64_t tiles_per_shard, int hsize,
                 int vsize, uint8_t* d_image, void* hip_stream);
/* The same for shards of f64 colours (the Canvas of draw/canvas.rs:6-27 holds f64 colours):
 * buffers of tiles_per_shard*RT_TILE*RT_TILE*3 doubles. */
int rt_detile_f64(const double* d_tiles, int nshards, int64_t tiles_per_shard, int hsize,
                  int vsize, double* d_image, void* hip_stream);

/* ---- several GPUs of one node ------------------------------------------------------------ */
/* Replaces what scene::camera::render does with threads: the shared World/Camera
 * (Arc, camera.rs:95-100), the bands of rows (camera.rs:114-143), join and stitch_canvases
 * (camera.rs:147-162, canvas.rs:47-64).  rt_scene_commit_multi commits the scene to every
 * listed GPU (device_ids distinct; the first one assembles the frames) and sets up RCCL between
 * them (librccl is loaded on this call, not before).  rt_render_multi deals the frame's
 * RT_TILE x RT_TILE tiles round-robin to the GPUs, every GPU renders its tiles, ONE exchange
 * (grouped ncclSend/ncclRecv of the compact u8 tile buffers to the first GPU) and a de-tile
 * kernel there produce the row-major image: out_rgb_u8 (host) and/or d_rgb_u8 (device memory on
 * device_ids[0]); at least one must be given.  The image is the one rt_render produces, byte for
 * byte, for every ndev; ndev = 1 runs the same steps (its shard goes through RCCL to itself).
 * stats (may be NULL): counters summed over the GPUs, kernel_ms the slowest GPU's.
 * One host thread per scene calls in; the call returns when the image is complete.  (Inside,
 * the replicas are committed, and a frame not rendered before is planned, by one short-lived
 * host thread per GPU, so that the GPUs work side by side; trees the host builds are built once
 * and uploaded to every GPU.)  After rt_scene_commit_multi the single-GPU entry points
 * (rt_render, rt_render_device, rt_render_device_noalloc) keep working on the scene: they use
 * device_ids[0] alone.  A scene whose rt_scene_commit_multi failed can only be destroyed
 * (rt_render_multi refuses it).  All calls leave the caller's current HIP device as they found it.
 * Verified on one GPU only (ndev = 1, where the shard is sent to itself); ndev > 1 has not run
 * on hardware. */
int rt_scene_commit_multi(rt_scene*, const int* device_ids, int ndev, int bvh_builder);
int rt_render_multi(rt_scene*, const rt_camera*, int depth, int mode, int pipeline,
                    uint8_t* out_rgb_u8, uint8_t* d_rgb_u8, rt_stats* stats);
/* The same frame with the f64 colours the reference's Canvas holds (draw/canvas.rs:6-27) as
 * well as, or instead of, the quantised ones: out_rgb_f64 (host) and/or d_rgb_f64 (device memory
 * on device_ids[0]), hsize*vsize*3 doubles, row-major; any of the four outputs may be NULL, not
 * all.  The f64 shards travel in the same grouped exchange (ncclDouble). */
int rt_render_multi_f64(rt_scene*, const rt_camera*, int depth, int mode, int pipeline,
                        double* out_rgb_f64, double* d_rgb_f64, uint8_t* out_rgb_u8,
                        uint8_t* d_rgb_u8, rt_stats* stats);
/* GPUs a committed scene lives on (1 after rt_scene_commit), 0 if not committed */
int rt_scene_device_count(const rt_scene*);

/* ---- a render entry point that allocates nothing --------------------------------------- */
/* No counterpart in the reference.  For callers that cannot allocate device memory while they
 * render: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * Same frame and same output layout as rt_render_device on the wavefront pipeline (the same
 * kernels), but the ray queues, result nodes and counters live in ONE fixed device array of
 * RT_NOALLOC_SCRATCH_BYTES that each GPU the library runs on holds until the process ends,
 * and the call issues no hipMalloc/hipFree, no read-back and no synchronisation: it only
 * enqueues work on hip_stream.  The shard is rendered in batches of rt_noalloc_max_tiles(depth)
 * tiles, each sized for the worst case (every call spawning two rays), so no queue can overflow.
 * Limits: 1 <= depth <= 8; at most RT_NOALLOC_MAX_TILES tiles per call (RT_ERR_UNSUPPORTED with
 * a message otherwise: pass a larger tile_step and call once per shard).  All calls on one GPU
 * share the array: calls that overlap on one GPU (different streams, different scenes) race;
 * calls on one stream, or ordered by events, do not.  It always runs the wave-shared kernels
 * (the RT_SHARE_LEAVES and RT_SHADOW_PASS knobs do not apply to it; the frames are the same bits
 * either way), and serves a scene committed with rt_scene_commit_multi like any other (on its
 * first GPU, row-major).  The per-GPU address of the array is looked up when a scene is
 * committed: two host threads committing different scenes at the same moment would both store
 * the same value. */
#define RT_NOALLOC_SCRATCH_BYTES (64u << 20)
#define RT_NOALLOC_MAX_TILES 8192
int rt_render_device_noalloc(rt_scene*, const rt_camera*, int depth, int mode, int tile_first,
                             int tile_step, double* d_rgb_f64, uint8_t* d_rgb_u8, void* hip_stream);
/* tiles per internal batch at this depth (19 at depth 5), or -1 for a depth out of range */
int64_t rt_noalloc_max_tiles(int depth);

/* ---- single-ray queries (known-answer tests of the device functions) ------------------ */
/* Traces n rays (origins/directions: n x 3 doubles, host) through the committed scene.
 * out: n x 32 doubles, per ray:
 * [0] hit flag, [1] t, [2] leaf order of the hit shape, [3] u, [4] v, [5] inside,
 * [6] n1, [7] n2, [8..10] point, [11..13] over_point, [14..16] under_point,
 * [17..19] eyev, [20..22] normalv, [23..25] reflectv, [26] schlick reflectance,
 * [27..29] colour of color_at(ray, depth), [30] shadowed flag of over_point for light 0,
 * [31] reserved. */
int rt_trace_rays(rt_scene*, int64_t n, const double* origins, const double* directions,
                  int depth, double* out);

#ifdef __cplusplus
}
#endif
#endif /* RT_CORE_H */
