/* rt_host_c.h — C view of the host mirror (raytracer_amd/host/rt_host.h) for callers
 * without a C++ compiler: Python's ctypes here, a Rust `extern "C"` block for the
 * reference's own host.  Each function names the reference item it stands for
 * (file:line under the reference's src/).
 *
 * A rth_world bundles a `World` (scene/world.rs:9-12), a `Camera`
 * (scene/camera.rs:11-19) and, after rth_commit, the committed rt_scene.
 */
#ifndef RT_HOST_C_H
#define RT_HOST_C_H

#include <stdint.h>

#include "rt_core.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_world rth_world;

/* draw/material.rs:6-15 with the pattern's FORWARD transform (patterns.rs set_transform) */
typedef struct rth_material {
  double ambient, diffuse, specular, shininess, reflective, transparency, refractive_index;
  int32_t pattern_kind;          /* RT_SOLID .. RT_CHECKERED */
  int32_t has_pattern_transform; /* 0: identity */
  double color_a[3], color_b[3];
  double pattern_transform[16];
} rth_material;

/* shape kinds accepted by rth_add_shape: RT_SPHERE..RT_CONE, or RTH_GROUP */
#define RTH_GROUP 7

const char* rth_last_error(void);

rth_world* rth_world_new(void);                 /* World::new, world.rs:15-20 */
void rth_world_free(rth_world*);
/* examples.rs: "cover" (134-268), "pawn" (92-132), "tea set" (19-90), "test_scene"
 * (270-404), or "synthetic" (the 1M-triangle stress scene).  hsize/vsize 0 keep the
 * example's own size.  Replaces the world's content and camera. */
int rth_load_example(rth_world*, const char* name, const char* obj_dir, int hsize, int vsize,
                     int synthetic_tris);
/* PointLight::new, light.rs:12-18 */
int rth_add_light(rth_world*, const double intensity[3], const double position[3]);
/* Sphere/Plane/Cube/Cylinder/Cone::new(transform) or Group::new(transform, material).
 * parent: handle of a group (Group::add_object, group.rs:48-53: the child takes the
 * group's material) or -1 for World.objects.  transform16 NULL = None.  material NULL =
 * Material::default_material().  Returns a shape handle >= 0, or -1 on error. */
int rth_add_shape(rth_world*, int parent, int kind, const double* transform16,
                  const rth_material* material, double minimum, double maximum, int closed);
/* Triangle::new / SmoothTriangle::new (triangle.rs:29, smooth_triangle.rs:33);
 * p: p1,p2,p3 (9 doubles); n: n1,n2,n3 or NULL.  parent must be a group. */
int rth_add_triangle(rth_world*, int parent, const double* p, const double* n);
/* obj_parser::parse_obj_file(text, transform, material), obj_parser.rs:9 */
int rth_add_obj(rth_world*, int parent, const char* obj_text, const double* transform16,
                const rth_material* material);
int rth_group_child_count(rth_world*, int handle);
/* Camera::new_with_transform, camera.rs:22-31 */
int rth_set_camera(rth_world*, int hsize, int vsize, double field_of_view, const double* transform16);
int rth_get_camera(rth_world*, rt_camera* out);

/* flatten + rt_scene_commit; the world must not change afterwards */
int rth_commit(rth_world*, int device, int bvh_builder);
/* the same on several GPUs (rt_scene_commit_multi); rth_render then uses all of them (rt_render_multi_f64) */
int rth_commit_multi(rth_world*, const int* device_ids, int ndev, int bvh_builder);
rt_scene* rth_scene(rth_world*);
/* scene::camera::render, camera.rs:83 (host buffers; see rt_render) */
int rth_render(rth_world*, int depth, int mode, int pipeline, double* out_rgb_f64,
               uint8_t* out_rgb_u8, rt_stats* stats);
/* A camera sweep over the committed scene (no counterpart in the reference, which renders one
 * frame per process): frame k for cameras[k], up to `inflight` (1..7) frames at a time on
 * streams of their own.  out_rgb_u8 receives the n images back to back (frame k:
 * hsize*vsize*3 bytes of cameras[k]); they equal what rth_render gives for each camera. */
int rth_render_sequence(rth_world*, const rt_camera* cameras, int n, int depth, int mode,
                        int pipeline, int inflight, uint8_t* out_rgb_u8);
/* A sweep issues a camera the scene has not seen without waiting for its ray-queue lengths,
 * sizing launches from the latest frame of the same size instead, and renders the frame again
 * if a queue outgrew that estimate.  Totals since commit: frames issued that way, and how many
 * of them were rendered again. */
int rth_sweep_counters(rth_world*, int64_t* issued_on_estimate, int64_t* rendered_again);
/* Canvas::write_to_ppm, canvas.rs:29-44 */
int rth_write_ppm(const char* path, int width, int height, const uint8_t* rgb);
/* The same image as an 8-bit RGB PNG, the format of the reference's renders/ directory
 * (the reference itself only writes PPM, canvas.rs:29-44). */
int rth_write_png(const char* path, int width, int height, const uint8_t* rgb);

/* math pre-computation (known-answer tests): Matrix::inverse (matrix.rs:154-168),
 * determinant (117-128), mat x mat (33-53), constructors (170-240; kind 0 translation,
 * 1 scaling, 2 rotation_x, 3 rotation_y, 4 rotation_z, 5 shear), view_transform
 * (camera.rs:189-209) */
int rth_mat_inverse(const double* m, int size, double* out);
double rth_mat_determinant(const double* m, int size);
void rth_mat_mul(const double* a16, const double* b16, double* out16);
void rth_mat_make(int kind, const double* params, double* out16);
void rth_view_transform(const double* from3, const double* to3, const double* up3, double* out16);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_C_H */
