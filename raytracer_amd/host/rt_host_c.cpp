// rt_host_c.cpp — C wrappers of the host mirror (include/rt_host_c.h).
#include "../../include/rt_host_c.h"

#include <limits>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "rt_host.h"
#include "../csrc/rt_internal.h"

using namespace rt;

static thread_local std::string g_herr;

struct rth_world {
  World world;
  Camera camera{1, 1, 1.0};
  bool has_camera = false;
  bool committed = false;
  DeviceScene device;
  std::vector<Shape*> handles;
};

static Material to_material(const rth_material* m) {
  Material out;
  if (!m) return out;
  out.ambient = m->ambient;
  out.diffuse = m->diffuse;
  out.specular = m->specular;
  out.shininess = m->shininess;
  out.reflective = m->reflective;
  out.transparency = m->transparency;
  out.refractive_index = m->refractive_index;
  out.pattern.kind = m->pattern_kind;
  out.pattern.a = {m->color_a[0], m->color_a[1], m->color_a[2]};
  out.pattern.b = {m->color_b[0], m->color_b[1], m->color_b[2]};
  if (m->has_pattern_transform) out.pattern.set_transform(Matrix::from_rows(m->pattern_transform, 4));
  return out;
}
static std::optional<Matrix> to_matrix(const double* t16) {
  if (!t16) return std::nullopt;
  return Matrix::from_rows(t16, 4);
}

template <typename F>
static int guarded(F&& f) {
  try {
    return f();
  } catch (const std::exception& e) {
    g_herr = e.what();
    return -1;
  }
}

static int attach(rth_world* w, int parent, std::unique_ptr<Shape> s, bool keep_material) {
  Shape* raw;
  if (parent >= 0) {
    if (parent >= (int)w->handles.size() || w->handles[parent]->kind != ShapeKind::Group) {
      g_herr = "parent is not a group handle";
      return -1;
    }
    (void)keep_material;
    raw = static_cast<Group*>(w->handles[parent])->add_object(std::move(s));
  } else {
    w->world.objects.push_back(std::move(s));
    raw = w->world.objects.back().get();
  }
  w->handles.push_back(raw);
  return (int)w->handles.size() - 1;
}

extern "C" {

const char* rth_last_error(void) { return g_herr.c_str(); }

rth_world* rth_world_new(void) { return new rth_world(); }
void rth_world_free(rth_world* w) { delete w; }

int rth_load_example(rth_world* w, const char* name, const char* obj_dir, int hsize, int vsize,
                     int synthetic_tris) {
  if (!w || !name) return -1;
  if (w->committed) {
    g_herr = "world already committed";
    return -1;
  }
  return guarded([&] {
    auto cw = examples::by_name(name, obj_dir ? obj_dir : "./obj", hsize, vsize, synthetic_tris);
    w->camera = cw.first;
    w->world = std::move(cw.second);
    w->has_camera = true;
    w->handles.clear();
    return 0;
  });
}

int rth_add_light(rth_world* w, const double intensity[3], const double position[3]) {
  if (!w || !intensity || !position) return -1;
  if (w->committed) {
    g_herr = "world already committed";
    return -1;
  }
  w->world.light_sources.push_back(
      {{intensity[0], intensity[1], intensity[2]}, Tuple::point(position[0], position[1], position[2])});
  return (int)w->world.light_sources.size() - 1;
}

int rth_add_shape(rth_world* w, int parent, int kind, const double* transform16,
                  const rth_material* material, double minimum, double maximum, int closed) {
  if (!w) return -1;
  if (w->committed) {
    g_herr = "world already committed";
    return -1;
  }
  return guarded([&] {
    std::optional<Matrix> t = to_matrix(transform16);
    std::unique_ptr<Shape> s;
    switch (kind) {
      case RT_SPHERE: s = std::make_unique<Sphere>(t); break;
      case RT_PLANE: s = std::make_unique<Plane>(t); break;
      case RT_CUBE: s = std::make_unique<Cube>(t); break;
      case RT_CYLINDER: {
        auto c = std::make_unique<Cylinder>(t);
        c->minimum = minimum;
        c->maximum = maximum;
        c->closed = closed != 0;
        s = std::move(c);
        break;
      }
      case RT_CONE: {
        auto c = std::make_unique<Cone>(t);
        c->minimum = minimum;
        c->maximum = maximum;
        c->closed = closed != 0;
        s = std::move(c);
        break;
      }
      case RTH_GROUP: s = std::make_unique<Group>(t, std::nullopt); break;
      default: g_herr = "unknown shape kind"; return -1;
    }
    s->material = to_material(material);
    return attach(w, parent, std::move(s), false);
  });
}

int rth_add_triangle(rth_world* w, int parent, const double* p, const double* n) {
  if (!w || !p) return -1;
  if (w->committed) {
    g_herr = "world already committed";
    return -1;
  }
  return guarded([&] {
    Tuple p1 = Tuple::point(p[0], p[1], p[2]), p2 = Tuple::point(p[3], p[4], p[5]),
          p3 = Tuple::point(p[6], p[7], p[8]);
    std::unique_ptr<Shape> s;
    if (n)
      s = std::make_unique<SmoothTriangle>(p1, p2, p3, Tuple::vector(n[0], n[1], n[2]),
                                           Tuple::vector(n[3], n[4], n[5]), Tuple::vector(n[6], n[7], n[8]));
    else
      s = std::make_unique<Triangle>(p1, p2, p3);
    return attach(w, parent, std::move(s), false);
  });
}

int rth_add_obj(rth_world* w, int parent, const char* obj_text, const double* transform16,
                const rth_material* material) {
  if (!w || !obj_text) return -1;
  if (w->committed) {
    g_herr = "world already committed";
    return -1;
  }
  return guarded([&] {
    std::optional<Material> m;
    if (material) m = to_material(material);
    return attach(w, parent, parse_obj_file(obj_text, to_matrix(transform16), m), false);
  });
}

int rth_group_child_count(rth_world* w, int handle) {
  if (!w || handle < 0 || handle >= (int)w->handles.size() || w->handles[handle]->kind != ShapeKind::Group)
    return -1;
  return (int)static_cast<Group*>(w->handles[handle])->objects.size();
}

int rth_set_camera(rth_world* w, int hsize, int vsize, double fov, const double* transform16) {
  if (!w || hsize <= 0 || vsize <= 0) return -1;
  w->camera = Camera::new_with_transform(hsize, vsize, fov,
                                         transform16 ? Matrix::from_rows(transform16, 4) : Matrix::identity());
  w->has_camera = true;
  return 0;
}

int rth_get_camera(rth_world* w, rt_camera* out) {
  if (!w || !out || !w->has_camera) return -1;
  return guarded([&] {
    *out = w->camera.to_abi();
    return 0;
  });
}

int rth_commit(rth_world* w, int device, int bvh_builder) {
  if (!w) return -1;
  if (w->committed) {
    g_herr = "world already committed";
    return -1;
  }
  return guarded([&] {
    if (!w->device.commit(w->world, device, bvh_builder)) {
      g_herr = w->device.error();
      return -1;
    }
    w->committed = true;
    return 0;
  });
}

int rth_commit_multi(rth_world* w, const int* device_ids, int ndev, int bvh_builder) {
  if (!w || !device_ids || ndev < 1) return -1;
  if (w->committed) {
    g_herr = "world already committed";
    return -1;
  }
  return guarded([&] {
    if (!w->device.commit(w->world, std::vector<int>(device_ids, device_ids + ndev), bvh_builder)) {
      g_herr = w->device.error();
      return -1;
    }
    w->committed = true;
    return 0;
  });
}

rt_scene* rth_scene(rth_world* w) { return (w && w->committed) ? w->device.handle() : nullptr; }

int rth_render(rth_world* w, int depth, int mode, int pipeline, double* out_rgb_f64,
               uint8_t* out_rgb_u8, rt_stats* stats) {
  if (!w || !w->committed || !w->has_camera) {
    g_herr = "world not committed or no camera";
    return -1;
  }
  return guarded([&] {
    rt_camera cam = w->camera.to_abi();
    int rc;
    if (w->device.multi()) {
      rc = rt_render_multi_f64(w->device.handle(), &cam, depth, mode, pipeline, out_rgb_f64, nullptr, out_rgb_u8,
                               nullptr, stats);
    } else {
      rc = rt_render(w->device.handle(), &cam, depth, mode, pipeline, out_rgb_f64, out_rgb_u8, stats);
    }
    if (rc != RT_OK) {
      g_herr = rt_last_error();
      return -rc;
    }
    return 0;
  });
}

int rth_render_sequence(rth_world* w, const rt_camera* cameras, int n, int depth, int mode, int pipeline,
                        int inflight, uint8_t* out_rgb_u8) {
  if (!w || !w->committed) {
    g_herr = "world not committed";
    return -1;
  }
  if (n < 0 || (n > 0 && (!cameras || !out_rgb_u8))) {
    g_herr = "bad argument";
    return -1;
  }
  return guarded([&] {
    std::vector<uint8_t*> out((size_t)n);
    size_t off = 0;
    for (int k = 0; k < n; k++) {
      out[k] = out_rgb_u8 + off;
      if (cameras[k].hsize > 0 && cameras[k].vsize > 0) off += (size_t)cameras[k].hsize * cameras[k].vsize * 3;
    }
    int rc = render_sequence_raw(w->device.handle(), cameras, n, depth, mode, pipeline, inflight, out.data());
    if (rc != RT_OK) {
      g_herr = rc == RT_ERR_HIP ? "HIP runtime call failed" : rt_last_error();
      return -rc;
    }
    return 0;
  });
}

int rth_sweep_counters(rth_world* w, int64_t* issued_on_estimate, int64_t* rendered_again) {
  if (!w || !w->committed) {
    g_herr = "world not committed";
    return -1;
  }
  long long a = 0, b = 0;
  rti::estimate_counters(w->device.handle(), &a, &b);
  if (issued_on_estimate) *issued_on_estimate = a;
  if (rendered_again) *rendered_again = b;
  return 0;
}

int rth_write_ppm(const char* path, int width, int height, const uint8_t* rgb) {
  if (!path || !rgb || width <= 0 || height <= 0) return -1;
  Canvas c;
  c.width = width;
  c.height = height;
  c.rgb8.assign(rgb, rgb + (size_t)width * height * 3);
  return c.write_to_ppm(path) ? 0 : -1;
}

int rth_write_png(const char* path, int width, int height, const uint8_t* rgb) {
  if (!path || !rgb || width <= 0 || height <= 0) return -1;
  Canvas c;
  c.width = width;
  c.height = height;
  c.rgb8.assign(rgb, rgb + (size_t)width * height * 3);
  return c.write_to_png(path) ? 0 : -1;
}

int rth_mat_inverse(const double* m, int size, double* out) {
  if (!m || !out || size < 2 || size > 4) {
    g_herr = "matrix size must be 2, 3 or 4";
    return -1;
  }
  return guarded([&] {
    Matrix::from_rows(m, size).inverse().to_rows(out);
    return 0;
  });
}
double rth_mat_determinant(const double* m, int size) {
  if (!m || size < 2 || size > 4) {
    g_herr = "matrix size must be 2, 3 or 4";
    return std::numeric_limits<double>::quiet_NaN();
  }
  return Matrix::from_rows(m, size).determinant();
}
void rth_mat_mul(const double* a16, const double* b16, double* out16) {
  (Matrix::from_rows(a16, 4) * Matrix::from_rows(b16, 4)).to_rows(out16);
}
void rth_mat_make(int kind, const double* p, double* out16) {
  Matrix m = Matrix::identity();
  switch (kind) {
    case 0: m = Matrix::translation(p[0], p[1], p[2]); break;
    case 1: m = Matrix::scaling(p[0], p[1], p[2]); break;
    case 2: m = Matrix::rotation_x(p[0]); break;
    case 3: m = Matrix::rotation_y(p[0]); break;
    case 4: m = Matrix::rotation_z(p[0]); break;
    case 5: m = Matrix::shear(p[0], p[1], p[2], p[3], p[4], p[5]); break;
  }
  m.to_rows(out16);
}
void rth_view_transform(const double* f, const double* t, const double* u, double* out16) {
  view_transform(Tuple::point(f[0], f[1], f[2]), Tuple::point(t[0], t[1], t[2]),
                 Tuple::vector(u[0], u[1], u[2]))
      .to_rows(out16);
}

}  // extern "C"
