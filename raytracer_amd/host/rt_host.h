// rt_host.h — host-side mirror of the reference's scene API (C++17).
//
// The reference's host is Rust (src/main.rs, src/examples.rs, src/obj_parser.rs and the
// World/Camera/Shape types they build).  This header restates that API surface with the
// same names and argument meaning so a scene is written the same way, but it holds DATA
// only: there is no intersection or shading code on the host.  `render()` flattens the
// World into the C ABI of include/rt_core.h and the GPU does the rest.
//
// Citations are file:line under the reference's src/.
#pragma once
#include <cstdint>
#include <memory>
#include <optional>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rt_core.h"

namespace rt {

// math/tuples.rs:6-11
struct Tuple {
  double x = 0, y = 0, z = 0, w = 0;
  static Tuple point(double x, double y, double z) { return {x, y, z, 1.0}; }
  static Tuple vector(double x, double y, double z) { return {x, y, z, 0.0}; }
};
Tuple operator+(const Tuple& a, const Tuple& b);
Tuple operator-(const Tuple& a, const Tuple& b);
Tuple operator*(const Tuple& a, double s);
double dot(const Tuple& a, const Tuple& b);
Tuple cross(const Tuple& a, const Tuple& b);
Tuple normalize(const Tuple& a);

// math/matrix.rs:12-241 (square, size 2..4, row-major)
struct Matrix {
  int size = 4;
  double m[4][4] = {{0}};
  static Matrix identity(int size = 4);
  static Matrix translation(double x, double y, double z);
  static Matrix scaling(double x, double y, double z);
  static Matrix rotation_x(double radians);
  static Matrix rotation_y(double radians);
  static Matrix rotation_z(double radians);
  static Matrix shear(double xy, double xz, double yx, double yz, double zx, double zy);
  static Matrix from_rows(const double* v, int size);
  double determinant() const;
  double cofactor(int i, int j) const;
  Matrix sub_matrix(int row, int col) const;
  Matrix inverse() const;     // matrix.rs:154-168: cofactor(n,m) / det
  Matrix transposed() const;  // matrix.rs:107-115
  bool is_identity() const;
  void to_rows(double* out16) const;
};
Matrix operator*(const Matrix& a, const Matrix& b);  // matrix.rs:33-53
Tuple operator*(const Matrix& a, const Tuple& t);    // matrix.rs:55-81

// draw/color.rs:6-10
struct Color {
  double r = 0, g = 0, b = 0;
  static Color white() { return {1.0, 1.0, 1.0}; }
  static Color black() { return {0.0, 0.0, 0.0}; }
};

// draw/patterns.rs
struct Pattern {
  int kind = RT_SOLID;
  Color a = Color::white(), b = Color::black();
  Matrix inverse_transform = Matrix::identity();
  static Pattern Solid(Color c);
  static Pattern Stripe(Color a, Color b);
  static Pattern Gradient(Color a, Color b);
  static Pattern Rings(Color a, Color b);
  static Pattern Checkered(Color a, Color b);
  void set_transform(const Matrix& t);  // ignored for Solid (patterns.rs:38-40)
};

// draw/material.rs:6-43
struct Material {
  Pattern pattern;
  double ambient = 0.1, diffuse = 0.9, specular = 0.9, shininess = 200.0;
  double reflective = 0.0, transparency = 0.0, refractive_index = 1.0;
  static Material default_material() { return Material(); }
  bool same_values(const Material& o) const;
};

enum class ShapeKind { Sphere, Plane, Cube, Cylinder, Cone, Triangle, SmoothTriangle, Group };

struct Group;

// shapes/intersect.rs:45-56: what every shape carries
struct Shape {
  ShapeKind kind;
  Material material;
  Group* parent = nullptr;
  explicit Shape(ShapeKind k, const std::optional<Matrix>& transform);
  virtual ~Shape() = default;
  Shape(Shape&&) = default;
  Shape& operator=(Shape&&) = default;
  bool has_transform() const { return xf != nullptr; }
  Matrix get_transform() const;
  Matrix get_inverse_transform() const;  // Matrix::inverse of the transform, identity when none

 private:
  std::unique_ptr<Matrix[]> xf;  // [0] transform, [1] inverse; null = identity
};

struct Sphere : Shape {  // shapes/sphere.rs:19-54
  explicit Sphere(const std::optional<Matrix>& t = std::nullopt) : Shape(ShapeKind::Sphere, t) {}
  static Sphere new_glass_sphere(const std::optional<Matrix>& t = std::nullopt);
};
struct Plane : Shape {  // shapes/plane.rs:19-46
  explicit Plane(const std::optional<Matrix>& t = std::nullopt) : Shape(ShapeKind::Plane, t) {}
};
struct Cube : Shape {  // shapes/cube.rs:19-46
  explicit Cube(const std::optional<Matrix>& t = std::nullopt) : Shape(ShapeKind::Cube, t) {}
};
struct Cylinder : Shape {  // shapes/cylinder.rs:10-51
  double minimum, maximum;
  bool closed = false;
  explicit Cylinder(const std::optional<Matrix>& t = std::nullopt);
};
struct Cone : Shape {  // shapes/cone.rs:10-51
  double minimum, maximum;
  bool closed = false;
  explicit Cone(const std::optional<Matrix>& t = std::nullopt);
};
struct Triangle : Shape {  // shapes/triangle.rs:13-70
  Tuple p1, p2, p3;
  Triangle(Tuple p1, Tuple p2, Tuple p3, const std::optional<Matrix>& t = std::nullopt);

 protected:
  Triangle(ShapeKind k, Tuple p1, Tuple p2, Tuple p3, const std::optional<Matrix>& t);
};
struct SmoothTriangle : Triangle {  // shapes/smooth_triangle.rs:14-89
  Tuple n1, n2, n3;
  SmoothTriangle(Tuple p1, Tuple p2, Tuple p3, Tuple n1, Tuple n2, Tuple n3,
                 const std::optional<Matrix>& t = std::nullopt);
};
struct Group : Shape {  // shapes/group.rs:10-61
  std::vector<std::unique_ptr<Shape>> objects;
  explicit Group(const std::optional<Matrix>& t = std::nullopt,
                 const std::optional<Material>& m = std::nullopt);
  // group.rs:48-53: the child takes this group as parent and a copy of its material
  Shape* add_object(std::unique_ptr<Shape> shape);
};

// scene/light.rs:6-18
struct PointLight {
  Color intensity;
  Tuple position;
};

// scene/world.rs:9-20
struct World {
  std::vector<std::unique_ptr<Shape>> objects;
  std::vector<PointLight> light_sources;
};

// scene/camera.rs:11-59
struct Camera {
  int hsize, vsize;
  double field_of_view;
  Matrix transform = Matrix::identity();
  double pixel_size, half_width, half_height;
  Camera(int hsize, int vsize, double field_of_view);
  static Camera new_with_transform(int hsize, int vsize, double fov, const Matrix& transform);
  rt_camera to_abi() const;
};
Matrix view_transform(const Tuple& from, const Tuple& to, const Tuple& up);  // camera.rs:189-209

// draw/canvas.rs:6-45: the finished image, already quantised by Color::clamp
// (draw/color.rs:148-155) on the device; `rgb` keeps the f64 colours when asked for.
struct Canvas {
  int width = 0, height = 0;
  std::vector<uint8_t> rgb8;
  std::vector<double> rgb;
  bool write_to_ppm(const std::string& file_name) const;  // P3 text, canvas.rs:29-44
  // 8-bit RGB PNG (the format of the reference's renders/): the quantised image, unfiltered
  // rows in stored deflate blocks, so writing costs one pass over the bytes.
  bool write_to_png(const std::string& file_name) const;
};

// obj_parser.rs:9-68
std::unique_ptr<Group> parse_obj_file(const std::string& s,
                                      const std::optional<Matrix>& transform = std::nullopt,
                                      const std::optional<Material>& material = std::nullopt);

// A World committed to one GPU; reusable across frames and cameras.
class DeviceScene {
 public:
  DeviceScene() = default;
  ~DeviceScene();
  DeviceScene(const DeviceScene&) = delete;
  DeviceScene& operator=(const DeviceScene&) = delete;
  // Flattens `world` depth-first (the order the reference's sorts see) and commits it.
  // Returns false and sets error() on failure.  bvh_builder (rt_scene_commit): 1 = the trees are
  // built on the GPU (the default: as fast to walk as the host's on every scene measured, and
  // built in a fraction of the time), 0 = on the host, 2 = by mesh size.
  bool commit(const World& world, int device = -1, int bvh_builder = 1);
  // The same scene on several GPUs of the node (the first one assembles the frames): what the
  // reference does with `thread_count` threads (camera.rs:83-163).  render() then deals the
  // frame's tiles to all of them.
  bool commit(const World& world, const std::vector<int>& devices, int bvh_builder = 1);
  bool multi() const { return multi_; }
  rt_scene* handle() const { return scene_; }
  const std::string& error() const { return error_; }

 private:
  bool flatten(const World& world);
  rt_scene* scene_ = nullptr;
  bool multi_ = false;
  std::string error_;
};

struct RenderOptions {
  int device = -1;  // which GPU renders ...
  int gpus = 0;     // ... or, when > 0, how many (devices 0 .. gpus-1, through rt_scene_commit_multi /
                    // rt_render_multi even for one): takes the place of `thread_count` (camera.rs:83)
  int depth = 5;    // camera.rs:178
  int mode = RT_MODE_F64_PARITY;
  int pipeline = RT_PIPELINE_WAVEFRONT;
  bool keep_f64 = false;
  rt_stats* stats = nullptr;
};
// scene::camera::render (camera.rs:83-163).  Throws std::runtime_error on failure (the
// reference panics).
Canvas render(const Camera& camera, const World& world, const RenderOptions& opt = RenderOptions());
Canvas render(const Camera& camera, DeviceScene& scene, const RenderOptions& opt = RenderOptions());

// A camera sweep over one committed scene (not in the reference, whose command line renders
// one frame per process): frame k is rendered for cameras[k], up to `inflight` (1..7) frames at a
// time on streams of their own, each with its own device and pinned host buffer, so the thin
// tail of one frame's recursion level runs under the next frame and the copy back to the
// host under both.  The frames are the ones render() returns for the same cameras.  Start
// the process with GPU_MAX_HW_QUEUES=8 to give four frames in flight a hardware queue each.
// A camera the scene has not rendered before normally costs one host read-back per recursion
// level (the ray-queue lengths size the next launch).  A sweep issues it without them instead,
// sized from the latest finished frame of the same size plus a quarter, and renders the frame
// again the exact way if a queue outgrew that (csrc/rt_internal.h), so new cameras overlap too.
// RT_TRACE_SWEEP=1 prints where the host's time went (waiting, copying, issuing).
std::vector<Canvas> render_sequence(const std::vector<Camera>& cameras, DeviceScene& scene,
                                    const RenderOptions& opt = RenderOptions(), int inflight = 4);
// The same over plain camera structs, into caller-provided images (out[k]: hsize*vsize*3
// bytes of frame k).  Returns RT_OK or an RT_ERR_* code (rt_last_error() has the text).
int render_sequence_raw(rt_scene* scene, const rt_camera* cameras, int n, int depth, int mode,
                        int pipeline, int inflight, uint8_t* const* out);

// examples.rs:19-404.  obj_dir holds pawn-chess.obj, teapot.obj, teacup.obj (the reference
// reads ./obj/ relative to the working directory).  hsize/vsize 0 keep the example's size.
namespace examples {
std::pair<Camera, World> tea_set(const std::string& obj_dir, int hsize = 0, int vsize = 0);
std::pair<Camera, World> pawn_chess(const std::string& obj_dir, int hsize = 0, int vsize = 0);
std::pair<Camera, World> book_cover(int hsize = 0, int vsize = 0);
std::pair<Camera, World> test_scene(int hsize = 0, int vsize = 0);
// Not in the reference: the 1M-triangle stress scene (see rt_host.cpp for the definition).
std::pair<Camera, World> synthetic(int ntris = 1000000, int hsize = 0, int vsize = 0);
// "cover" | "pawn" | "tea set" | "test_scene" | "synthetic"
std::pair<Camera, World> by_name(const std::string& name, const std::string& obj_dir, int hsize,
                                 int vsize, int synthetic_tris = 0);
}  // namespace examples

}  // namespace rt
