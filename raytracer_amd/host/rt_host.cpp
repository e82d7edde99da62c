// rt_host.cpp — host mirror of the reference's scene API: data, matrix pre-computation,
// OBJ loading, the example scenes, flattening into the C ABI, PPM output.
#include "rt_host.h"

#include <hip/hip_runtime.h>

#include "../csrc/rt_internal.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <limits>
#include <sstream>
#include <stdexcept>

namespace rt {

// ---- tuples (only what scene construction needs; math/tuples.rs) ----------------------
Tuple operator+(const Tuple& a, const Tuple& b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
Tuple operator-(const Tuple& a, const Tuple& b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
Tuple operator*(const Tuple& a, double s) { return Tuple::vector(a.x * s, a.y * s, a.z * s); }
double dot(const Tuple& a, const Tuple& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
Tuple cross(const Tuple& a, const Tuple& b) {
  return Tuple::vector(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
Tuple normalize(const Tuple& a) {
  double mag = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  return Tuple::vector(a.x / mag, a.y / mag, a.z / mag);
}

// ---- matrices --------------------------------------------------------------------------
Matrix Matrix::identity(int size) {
  Matrix r;
  r.size = size;
  for (int i = 0; i < size; i++) r.m[i][i] = 1.0;
  return r;
}
Matrix Matrix::from_rows(const double* v, int size) {
  if (size < 1 || size > 4) throw std::invalid_argument("Matrix::from_rows: size must be 1..4");
  Matrix r;
  r.size = size;
  for (int i = 0; i < size; i++)
    for (int j = 0; j < size; j++) r.m[i][j] = v[i * size + j];
  return r;
}
void Matrix::to_rows(double* out16) const {
  for (int i = 0; i < size; i++)
    for (int j = 0; j < size; j++) out16[i * size + j] = m[i][j];
}
Matrix Matrix::translation(double x, double y, double z) {
  Matrix r = identity();
  r.m[0][3] = x;
  r.m[1][3] = y;
  r.m[2][3] = z;
  return r;
}
Matrix Matrix::scaling(double x, double y, double z) {
  Matrix r;
  r.m[0][0] = x;
  r.m[1][1] = y;
  r.m[2][2] = z;
  r.m[3][3] = 1.0;
  return r;
}
Matrix Matrix::rotation_x(double a) {
  Matrix r = identity();
  double c = std::cos(a), s = std::sin(a);
  r.m[1][1] = c;
  r.m[2][2] = c;
  r.m[1][2] = -s;
  r.m[2][1] = s;
  return r;
}
Matrix Matrix::rotation_y(double a) {
  Matrix r = identity();
  double c = std::cos(a), s = std::sin(a);
  r.m[0][0] = c;
  r.m[2][2] = c;
  r.m[0][2] = s;
  r.m[2][0] = -s;
  return r;
}
Matrix Matrix::rotation_z(double a) {
  Matrix r = identity();
  double c = std::cos(a), s = std::sin(a);
  r.m[0][0] = c;
  r.m[1][1] = c;
  r.m[0][1] = -s;
  r.m[1][0] = s;
  return r;
}
Matrix Matrix::shear(double xy, double xz, double yx, double yz, double zx, double zy) {
  Matrix r = identity();
  r.m[0][1] = xy;
  r.m[0][2] = xz;
  r.m[1][0] = yx;
  r.m[1][2] = yz;
  r.m[2][0] = zx;
  r.m[2][1] = zy;
  return r;
}
Matrix operator*(const Matrix& a, const Matrix& b) {
  Matrix r;
  r.size = a.size;
  for (int row = 0; row < a.size; row++)
    for (int col = 0; col < a.size; col++) {
      double acc = 0.0;
      for (int i = 0; i < a.size; i++) acc += a.m[row][i] * b.m[i][col];
      r.m[row][col] = acc;
    }
  return r;
}
Tuple operator*(const Matrix& a, const Tuple& t) {
  double v[4] = {t.x, t.y, t.z, t.w}, o[4];
  for (int r = 0; r < 4; r++) o[r] = a.m[r][0] * v[0] + a.m[r][1] * v[1] + a.m[r][2] * v[2] + a.m[r][3] * v[3];
  return {o[0], o[1], o[2], o[3]};
}
Matrix Matrix::sub_matrix(int row, int col) const {
  Matrix r;
  r.size = size - 1;
  for (int i = 0, ri = 0; i < size; i++) {
    if (i == row) continue;
    for (int j = 0, ci = 0; j < size; j++) {
      if (j == col) continue;
      r.m[ri][ci++] = m[i][j];
    }
    ri++;
  }
  return r;
}
double Matrix::cofactor(int i, int j) const {  // matrix.rs:140-152
  double minor = sub_matrix(i, j).determinant();
  return ((i + j) & 1) ? -minor : minor;
}
double Matrix::determinant() const {  // matrix.rs:117-128
  if (size == 2) return m[0][0] * m[1][1] - m[0][1] * m[1][0];
  double det = 0.0;
  for (int col = 0; col < size; col++) det += m[0][col] * cofactor(0, col);
  return det;
}
Matrix Matrix::inverse() const {
  double det = determinant();
  if (det == 0.0) throw std::runtime_error("Matrix::inverse: matrix is not invertible");  // matrix.rs:158
  Matrix inv = *this;
  for (int n = 0; n < size; n++)
    for (int k = 0; k < size; k++) inv.m[k][n] = cofactor(n, k) / det;
  return inv;
}
Matrix Matrix::transposed() const {
  Matrix r = *this;
  for (int i = 0; i < size; i++)
    for (int j = 0; j < size; j++) r.m[i][j] = m[j][i];
  return r;
}
bool Matrix::is_identity() const {
  for (int i = 0; i < size; i++)
    for (int j = 0; j < size; j++)
      if (m[i][j] != (i == j ? 1.0 : 0.0)) return false;
  return true;
}

// ---- patterns / materials ------------------------------------------------------------
static Pattern make_pattern(int kind, Color a, Color b) {
  Pattern p;
  p.kind = kind;
  p.a = a;
  p.b = b;
  return p;
}
Pattern Pattern::Solid(Color c) { return make_pattern(RT_SOLID, c, Color::black()); }
Pattern Pattern::Stripe(Color a, Color b) { return make_pattern(RT_STRIPE, a, b); }
Pattern Pattern::Gradient(Color a, Color b) { return make_pattern(RT_GRADIENT, a, b); }
Pattern Pattern::Rings(Color a, Color b) { return make_pattern(RT_RINGS, a, b); }
Pattern Pattern::Checkered(Color a, Color b) { return make_pattern(RT_CHECKERED, a, b); }
void Pattern::set_transform(const Matrix& t) {
  if (kind == RT_SOLID) return;
  inverse_transform = t.inverse();
}
bool Material::same_values(const Material& o) const {
  return std::memcmp(&pattern.a, &o.pattern.a, sizeof(Color)) == 0 &&
         std::memcmp(&pattern.b, &o.pattern.b, sizeof(Color)) == 0 && pattern.kind == o.pattern.kind &&
         std::memcmp(pattern.inverse_transform.m, o.pattern.inverse_transform.m, sizeof(double) * 16) == 0 &&
         ambient == o.ambient && diffuse == o.diffuse && specular == o.specular &&
         shininess == o.shininess && reflective == o.reflective && transparency == o.transparency &&
         refractive_index == o.refractive_index;
}

// ---- shapes ----------------------------------------------------------------------------
Shape::Shape(ShapeKind k, const std::optional<Matrix>& transform) : kind(k) {
  if (transform) {
    if (transform->size != 4) throw std::runtime_error("shape transform must be 4x4");
    xf.reset(new Matrix[2]);
    xf[0] = *transform;
    xf[1] = transform->inverse();
  }
}
Matrix Shape::get_transform() const { return xf ? xf[0] : Matrix::identity(); }
Matrix Shape::get_inverse_transform() const { return xf ? xf[1] : Matrix::identity(); }

Sphere Sphere::new_glass_sphere(const std::optional<Matrix>& t) {
  Sphere s(t);
  s.material.transparency = 1.0;
  s.material.refractive_index = 1.5;
  s.material.pattern = Pattern::Solid(Color::black());
  return s;
}
Cylinder::Cylinder(const std::optional<Matrix>& t)
    : Shape(ShapeKind::Cylinder, t),
      minimum(-std::numeric_limits<double>::infinity()),
      maximum(std::numeric_limits<double>::infinity()) {}
Cone::Cone(const std::optional<Matrix>& t)
    : Shape(ShapeKind::Cone, t),
      minimum(-std::numeric_limits<double>::infinity()),
      maximum(std::numeric_limits<double>::infinity()) {}
Triangle::Triangle(Tuple a, Tuple b, Tuple c, const std::optional<Matrix>& t)
    : Shape(ShapeKind::Triangle, t), p1(a), p2(b), p3(c) {}
Triangle::Triangle(ShapeKind k, Tuple a, Tuple b, Tuple c, const std::optional<Matrix>& t)
    : Shape(k, t), p1(a), p2(b), p3(c) {}
SmoothTriangle::SmoothTriangle(Tuple a, Tuple b, Tuple c, Tuple na, Tuple nb, Tuple nc,
                               const std::optional<Matrix>& t)
    : Triangle(ShapeKind::SmoothTriangle, a, b, c, t), n1(na), n2(nb), n3(nc) {}
Group::Group(const std::optional<Matrix>& t, const std::optional<Material>& m)
    : Shape(ShapeKind::Group, t) {
  if (m) material = *m;
}
Shape* Group::add_object(std::unique_ptr<Shape> shape) {
  shape->parent = this;
  shape->material = material;
  objects.push_back(std::move(shape));
  return objects.back().get();
}

// ---- camera ------------------------------------------------------------------------------
Camera::Camera(int h, int v, double fov) : hsize(h), vsize(v), field_of_view(fov) {
  double half_view = std::tan(fov / 2.0);
  double aspect = (double)h / (double)v;
  if (aspect >= 1.0) {
    half_width = half_view;
    half_height = half_view / aspect;
  } else {
    half_height = half_view;
    half_width = half_view * aspect;
  }
  pixel_size = (half_width * 2.0) / (double)h;
}
Camera Camera::new_with_transform(int h, int v, double fov, const Matrix& t) {
  Camera c(h, v, fov);
  c.transform = t;
  return c;
}
rt_camera Camera::to_abi() const {
  rt_camera c;
  c.hsize = hsize;
  c.vsize = vsize;
  c.half_width = half_width;
  c.half_height = half_height;
  c.pixel_size = pixel_size;
  transform.inverse().to_rows(c.inv);  // camera.rs:73
  return c;
}
Matrix view_transform(const Tuple& from, const Tuple& to, const Tuple& up) {
  Tuple forward = normalize(to - from);
  Tuple left = cross(forward, normalize(up));
  Tuple true_up = cross(left, forward);
  Matrix o = Matrix::identity();
  const Tuple rows[3] = {left, true_up, Tuple::vector(-forward.x, -forward.y, -forward.z)};
  for (int r = 0; r < 3; r++) {
    o.m[r][0] = rows[r].x;
    o.m[r][1] = rows[r].y;
    o.m[r][2] = rows[r].z;
  }
  return o * Matrix::translation(-from.x, -from.y, -from.z);
}

// ---- canvas ------------------------------------------------------------------------------
bool Canvas::write_to_ppm(const std::string& file_name) const {
  FILE* f = std::fopen(file_name.c_str(), "w");
  if (!f) return false;
  std::fprintf(f, "P3\n%d %d\n255\n", width, height);
  std::string line;
  char buf[16];
  for (int y = 0; y < height; y++) {
    line.clear();
    const uint8_t* row = rgb8.data() + (size_t)y * width * 3;
    for (int x = 0; x < width; x++) {
      int n = std::snprintf(buf, sizeof buf, "%d %d %d ", row[3 * x], row[3 * x + 1], row[3 * x + 2]);
      line.append(buf, n);
    }
    line.push_back('\n');
    std::fwrite(line.data(), 1, line.size(), f);
  }
  return std::fclose(f) == 0;
}

namespace {
struct Crc32 {  // CRC-32 of the PNG chunk layer (polynomial 0xedb88320, reflected)
  uint32_t table[256];
  Crc32() {
    for (uint32_t n = 0; n < 256; n++) {
      uint32_t c = n;
      for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
      table[n] = c;
    }
  }
  uint32_t update(uint32_t crc, const uint8_t* p, size_t n) const {
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return crc;
  }
};
void put_be32(std::vector<uint8_t>& v, uint32_t x) {
  v.push_back((uint8_t)(x >> 24));
  v.push_back((uint8_t)(x >> 16));
  v.push_back((uint8_t)(x >> 8));
  v.push_back((uint8_t)x);
}
bool write_chunk(FILE* f, const Crc32& crc, const char type[4], const std::vector<uint8_t>& data) {
  std::vector<uint8_t> head;
  put_be32(head, (uint32_t)data.size());
  head.insert(head.end(), type, type + 4);
  uint32_t c = crc.update(0xffffffffu, head.data() + 4, 4);
  c = crc.update(c, data.data(), data.size()) ^ 0xffffffffu;
  std::vector<uint8_t> tail;
  put_be32(tail, c);
  return std::fwrite(head.data(), 1, head.size(), f) == head.size() &&
         (data.empty() || std::fwrite(data.data(), 1, data.size(), f) == data.size()) &&  // IEND has no data
         std::fwrite(tail.data(), 1, tail.size(), f) == tail.size();
}
}  // namespace

bool Canvas::write_to_png(const std::string& file_name) const {
  if (width <= 0 || height <= 0 || rgb8.size() != (size_t)width * height * 3) return false;
  // one IDAT chunk holds the whole zlib stream: its length field is 32 bits
  const size_t row = (size_t)width * 3 + 1, raw = row * (size_t)height;
  const size_t blocks = (raw + 65534) / 65535;
  if (raw + 5 * blocks + 6 > 0x7fffffffu) return false;
  FILE* f = std::fopen(file_name.c_str(), "wb");
  if (!f) return false;
  const Crc32 crc;
  static const uint8_t signature[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
  bool ok = std::fwrite(signature, 1, 8, f) == 8;
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, (uint32_t)width);
  put_be32(ihdr, (uint32_t)height);
  const uint8_t kind[5] = {8, 2, 0, 0, 0};  // 8 bits per sample, truecolour, deflate, filter set 0, no interlace
  ihdr.insert(ihdr.end(), kind, kind + 5);
  ok = ok && write_chunk(f, crc, "IHDR", ihdr);
  // scanlines with filter type 0, then a zlib stream of stored blocks around them
  std::vector<uint8_t> lines(raw);
  uint32_t a = 1, b = 0;  // Adler-32 of the scanlines
  for (int y = 0; y < height; y++) {
    uint8_t* dst = lines.data() + (size_t)y * row;
    dst[0] = 0;
    std::memcpy(dst + 1, rgb8.data() + (size_t)y * width * 3, (size_t)width * 3);
  }
  for (size_t i = 0; i < raw;) {
    const size_t n = std::min<size_t>(raw - i, 5552);  // largest run before the sums must be reduced
    for (size_t k = 0; k < n; k++) {
      a += lines[i + k];
      b += a;
    }
    a %= 65521u;
    b %= 65521u;
    i += n;
  }
  std::vector<uint8_t> z;
  z.reserve(raw + 5 * blocks + 6);
  z.push_back(0x78);
  z.push_back(0x01);
  for (size_t i = 0; i < raw; i += 65535) {
    const size_t n = std::min<size_t>(raw - i, 65535);
    z.push_back(i + n == raw ? 1 : 0);  // BFINAL, BTYPE = 00 (stored)
    z.push_back((uint8_t)(n & 0xff));
    z.push_back((uint8_t)(n >> 8));
    z.push_back((uint8_t)(~n & 0xff));
    z.push_back((uint8_t)((~n >> 8) & 0xff));
    z.insert(z.end(), lines.begin() + (long)i, lines.begin() + (long)(i + n));
  }
  put_be32(z, (b << 16) | a);
  ok = ok && write_chunk(f, crc, "IDAT", z);
  ok = ok && write_chunk(f, crc, "IEND", {});
  return (std::fclose(f) == 0) && ok;
}

// ---- OBJ ---------------------------------------------------------------------------------
namespace {
// `str::lines()`: split at '\n' and drop one trailing '\r'
void for_each_line(const std::string& s, const std::function<void(const char*, size_t)>& fn) {
  size_t pos = 0, n = s.size();
  while (pos < n) {
    size_t nl = s.find('\n', pos);
    size_t end = nl == std::string::npos ? n : nl;
    size_t len = end - pos;
    if (len > 0 && s[pos + len - 1] == '\r') len--;
    fn(s.data() + pos, len);
    if (nl == std::string::npos) break;
    pos = nl + 1;
  }
}
bool is_ws(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }
// obj_parser.rs:17-20: split on single spaces; empty tokens and tokens that contain any
// other whitespace are dropped
void tokenize(const char* p, size_t len, std::vector<std::string>& out) {
  out.clear();
  size_t i = 0;
  while (i <= len) {
    size_t j = i;
    while (j < len && p[j] != ' ') j++;
    if (j > i) {
      bool bad = false;
      for (size_t k = i; k < j; k++)
        if (is_ws((unsigned char)p[k])) bad = true;
      if (!bad) out.emplace_back(p + i, j - i);
    }
    i = j + 1;
  }
}
bool parse_index(const std::string& s, size_t& v) {
  if (s.empty()) return false;
  char* end = nullptr;
  unsigned long long x = std::strtoull(s.c_str(), &end, 10);
  if (!end || *end != 0 || s[0] == '-') return false;
  v = (size_t)x;
  return true;
}
}  // namespace

std::unique_ptr<Group> parse_obj_file(const std::string& s, const std::optional<Matrix>& transform,
                                      const std::optional<Material>& material) {
  auto group = std::make_unique<Group>(transform, material);
  std::vector<Tuple> vertices{Tuple::vector(0, 0, 0)}, normals{Tuple::vector(0, 0, 0)};  // 1-indexed
  std::vector<std::string> sym;
  std::vector<size_t> vi;
  std::vector<long> ni;
  for_each_line(s, [&](const char* p, size_t len) {
    tokenize(p, len, sym);
    if (sym.empty()) return;
    if (sym[0] == "v" || sym[0] == "vn") {
      if (sym.size() < 4) throw std::runtime_error("obj: short v/vn record");  // index panic in the reference
      double c[3];
      for (int k = 0; k < 3; k++) c[k] = std::strtod(sym[1 + k].c_str(), nullptr);
      if (sym[0] == "v")
        vertices.push_back(Tuple::point(c[0], c[1], c[2]));
      else
        normals.push_back(Tuple::vector(c[0], c[1], c[2]));
    } else if (sym[0] == "f") {
      vi.clear();
      ni.clear();
      for (size_t k = 1; k < sym.size(); k++) {
        // obj_parser.rs:41-50: "a/b/c" -> vertex a, normal c; "a" -> no normal
        std::vector<std::string> parts;
        size_t b = 0;
        const std::string& tk = sym[k];
        for (;;) {
          size_t sl = tk.find('/', b);
          parts.push_back(tk.substr(b, sl == std::string::npos ? std::string::npos : sl - b));
          if (sl == std::string::npos) break;
          b = sl + 1;
        }
        size_t v = 0;
        if (!parse_index(parts[0], v)) throw std::runtime_error("obj: bad vertex index");
        vi.push_back(v);
        long nidx = -1;
        if (parts.size() >= 2) {
          if (parts.size() < 3) throw std::runtime_error("obj: a/b face token");  // reference panics
          size_t nv;
          if (parse_index(parts[2], nv)) nidx = (long)nv;
        }
        ni.push_back(nidx);
      }
      // fan triangulation, obj_parser.rs:71-100
      for (size_t i = 1; i + 1 < vi.size(); i++) {
        if (vi[0] >= vertices.size() || vi[i] >= vertices.size() || vi[i + 1] >= vertices.size())
          throw std::runtime_error("obj: vertex index out of range");
        if (ni[i] >= 0) {
          if (ni[0] < 0 || ni[i + 1] < 0) throw std::runtime_error("obj: mixed normal indices");
          if ((size_t)ni[0] >= normals.size() || (size_t)ni[i] >= normals.size() ||
              (size_t)ni[i + 1] >= normals.size())
            throw std::runtime_error("obj: normal index out of range");
          group->add_object(std::make_unique<SmoothTriangle>(
              vertices[vi[0]], vertices[vi[i]], vertices[vi[i + 1]], normals[ni[0]], normals[ni[i]],
              normals[ni[i + 1]]));
        } else {
          group->add_object(
              std::make_unique<Triangle>(vertices[vi[0]], vertices[vi[i]], vertices[vi[i + 1]]));
        }
      }
    }
  });
  return group;
}

// ---- flattening ----------------------------------------------------------------------------
namespace {
struct Flattener {
  rt_scene* sc;
  std::string err;

  bool check(int rc, const char* what) {
    if (rc >= 0) return true;
    err = std::string(what) + ": " + rt_last_error();
    return false;
  }
  int add_material(const Material& m) {
    rt_material rm;
    std::memset(&rm, 0, sizeof rm);
    rm.ambient = m.ambient;
    rm.diffuse = m.diffuse;
    rm.specular = m.specular;
    rm.shininess = m.shininess;
    rm.reflective = m.reflective;
    rm.transparency = m.transparency;
    rm.refractive_index = m.refractive_index;
    rm.pattern_kind = m.pattern.kind;
    rm.color_a[0] = m.pattern.a.r;
    rm.color_a[1] = m.pattern.a.g;
    rm.color_a[2] = m.pattern.a.b;
    rm.color_b[0] = m.pattern.b.r;
    rm.color_b[1] = m.pattern.b.g;
    rm.color_b[2] = m.pattern.b.b;
    m.pattern.inverse_transform.to_rows(rm.pattern_inv);
    return rt_scene_add_material(sc, &rm);
  }
  bool flush_run(const Group& g, int gid, size_t a, size_t b) {
    if (a >= b) return true;
    size_t n = b - a;
    std::vector<double> p1(3 * n), p2(3 * n), p3(3 * n), n1, n2, n3;
    std::vector<uint8_t> smooth(n, 0);
    bool any_smooth = false;
    for (size_t i = a; i < b; i++)
      if (g.objects[i]->kind == ShapeKind::SmoothTriangle) any_smooth = true;
    if (any_smooth) {
      n1.assign(3 * n, 0.0);
      n2.assign(3 * n, 0.0);
      n3.assign(3 * n, 0.0);
    }
    for (size_t i = a; i < b; i++) {
      const Triangle& t = static_cast<const Triangle&>(*g.objects[i]);
      size_t k = i - a;
      const Tuple* ps[3] = {&t.p1, &t.p2, &t.p3};
      double* dst[3] = {p1.data(), p2.data(), p3.data()};
      for (int v = 0; v < 3; v++) {
        dst[v][3 * k] = ps[v]->x;
        dst[v][3 * k + 1] = ps[v]->y;
        dst[v][3 * k + 2] = ps[v]->z;
      }
      if (t.kind == ShapeKind::SmoothTriangle) {
        const SmoothTriangle& st = static_cast<const SmoothTriangle&>(t);
        const Tuple* ns[3] = {&st.n1, &st.n2, &st.n3};
        double* nd[3] = {n1.data(), n2.data(), n3.data()};
        for (int v = 0; v < 3; v++) {
          nd[v][3 * k] = ns[v]->x;
          nd[v][3 * k + 1] = ns[v]->y;
          nd[v][3 * k + 2] = ns[v]->z;
        }
        smooth[k] = 1;
      }
    }
    int mat = add_material(g.objects[a]->material);
    if (!check(mat, "add_material")) return false;
    rt_mesh m;
    std::memset(&m, 0, sizeof m);
    m.group = gid;
    m.material = mat;
    m.ntris = (int64_t)n;
    m.p1 = p1.data();
    m.p2 = p2.data();
    m.p3 = p3.data();
    if (any_smooth) {
      m.n1 = n1.data();
      m.n2 = n2.data();
      m.n3 = n3.data();
      m.smooth = smooth.data();
    }
    return check(rt_scene_add_mesh(sc, &m), "add_mesh");
  }
  bool shape(const Shape& s, int parent) {
    switch (s.kind) {
      case ShapeKind::Group: {
        const Group& g = static_cast<const Group&>(s);
        rt_group rg;
        std::memset(&rg, 0, sizeof rg);
        g.get_inverse_transform().to_rows(rg.inv);
        rg.parent = parent;
        int gid = rt_scene_add_group(sc, &rg);
        if (!check(gid, "add_group")) return false;
        // consecutive identity-transform triangles with one material form a mesh
        size_t run = 0, i = 0, n = g.objects.size();
        for (; i < n; i++) {
          const Shape& c = *g.objects[i];
          bool tri = (c.kind == ShapeKind::Triangle || c.kind == ShapeKind::SmoothTriangle);
          if (tri && c.has_transform() && !c.get_transform().is_identity()) {
            err = "triangles with a transform of their own are not supported";
            return false;
          }
          if (tri && (i == run || c.material.same_values(g.objects[run]->material))) continue;
          if (!flush_run(g, gid, run, i)) return false;
          if (tri) {
            run = i;
          } else {
            if (!shape(c, gid)) return false;
            run = i + 1;
          }
        }
        return flush_run(g, gid, run, n);
      }
      case ShapeKind::Triangle:
      case ShapeKind::SmoothTriangle: {
        // a lone triangle outside any group: wrap it in an identity group so it can be a mesh
        err = "a triangle must be the child of a group";
        return false;
      }
      default: {
        int mat = add_material(s.material);
        if (!check(mat, "add_material")) return false;
        rt_primitive p;
        std::memset(&p, 0, sizeof p);
        p.material = mat;
        p.parent = parent;
        s.get_inverse_transform().to_rows(p.inv);
        p.minimum = -std::numeric_limits<double>::infinity();
        p.maximum = std::numeric_limits<double>::infinity();
        switch (s.kind) {
          case ShapeKind::Sphere: p.kind = RT_SPHERE; break;
          case ShapeKind::Plane: p.kind = RT_PLANE; break;
          case ShapeKind::Cube: p.kind = RT_CUBE; break;
          case ShapeKind::Cylinder: {
            const Cylinder& c = static_cast<const Cylinder&>(s);
            p.kind = RT_CYLINDER;
            p.minimum = c.minimum;
            p.maximum = c.maximum;
            p.closed = c.closed;
            break;
          }
          default: {
            const Cone& c = static_cast<const Cone&>(s);
            p.kind = RT_CONE;
            p.minimum = c.minimum;
            p.maximum = c.maximum;
            p.closed = c.closed;
            break;
          }
        }
        return check(rt_scene_add_primitive(sc, &p), "add_primitive");
      }
    }
  }
};
}  // namespace

DeviceScene::~DeviceScene() {
  if (scene_) rt_scene_destroy(scene_);
}
bool DeviceScene::flatten(const World& world) {
  if (scene_) {
    rt_scene_destroy(scene_);
    scene_ = nullptr;
  }
  multi_ = false;
  scene_ = rt_scene_create();
  Flattener f{scene_, {}};
  for (const auto& o : world.objects)
    if (!f.shape(*o, -1)) {
      error_ = f.err;
      return false;
    }
  for (const PointLight& L : world.light_sources) {
    double in[3] = {L.intensity.r, L.intensity.g, L.intensity.b};
    double pos[3] = {L.position.x, L.position.y, L.position.z};
    if (rt_scene_add_light(scene_, in, pos) < 0) {
      error_ = rt_last_error();
      return false;
    }
  }
  return true;
}
bool DeviceScene::commit(const World& world, int device, int bvh_builder) {
  if (!flatten(world)) return false;
  if (rt_scene_commit(scene_, device, bvh_builder) != RT_OK) {
    error_ = rt_last_error();
    return false;
  }
  return true;
}
bool DeviceScene::commit(const World& world, const std::vector<int>& devices, int bvh_builder) {
  if (!flatten(world)) return false;
  if (rt_scene_commit_multi(scene_, devices.data(), (int)devices.size(), bvh_builder) != RT_OK) {
    error_ = rt_last_error();
    return false;
  }
  multi_ = true;
  return true;
}

Canvas render(const Camera& camera, DeviceScene& scene, const RenderOptions& opt) {
  Canvas c;
  c.width = camera.hsize;
  c.height = camera.vsize;
  size_t n = (size_t)c.width * c.height * 3;
  c.rgb8.resize(n);
  if (opt.keep_f64) c.rgb.resize(n);
  rt_camera cam = camera.to_abi();
  int rc;
  if (scene.multi()) {
    rc = rt_render_multi_f64(scene.handle(), &cam, opt.depth, opt.mode, opt.pipeline,
                             opt.keep_f64 ? c.rgb.data() : nullptr, nullptr, c.rgb8.data(), nullptr, opt.stats);
  } else {
    rc = rt_render(scene.handle(), &cam, opt.depth, opt.mode, opt.pipeline,
                   opt.keep_f64 ? c.rgb.data() : nullptr, c.rgb8.data(), opt.stats);
  }
  if (rc != RT_OK) throw std::runtime_error(std::string("render: ") + rt_last_error());
  return c;
}
Canvas render(const Camera& camera, const World& world, const RenderOptions& opt) {
  DeviceScene ds;
  bool ok;
  if (opt.gpus > 0) {
    std::vector<int> devices;
    for (int d = 0; d < opt.gpus; d++) devices.push_back(d);
    ok = ds.commit(world, devices);
  } else {
    ok = ds.commit(world, opt.device);
  }
  if (!ok) throw std::runtime_error("render: " + ds.error());
  return render(camera, ds, opt);
}

int render_sequence_raw(rt_scene* scene, const rt_camera* cameras, int n, int depth, int mode,
                        int pipeline, int inflight, uint8_t* const* out) {
  if (!scene || n < 0 || (n > 0 && (!cameras || !out))) return RT_ERR_ARG;
  rt_scene_info info;
  int rc = rt_scene_get_info(scene, &info);
  if (rc != RT_OK) return rc;
  if (n == 0) return RT_OK;
  // F frames on the GPU at a time need F + 1 slots: the host copies the image of the frame that
  // has just finished out of its slot while the frame that took its place runs in another.
  const int F = std::max(1, std::min(std::min(inflight, 7), n));
  const int S = F + 1;
  size_t largest = 0;
  for (int k = 0; k < n; k++) {
    if (cameras[k].hsize <= 0 || cameras[k].vsize <= 0 || !out[k]) return RT_ERR_ARG;
    largest = std::max(largest, (size_t)cameras[k].hsize * cameras[k].vsize * 3);
  }
  int previous = 0;
  if (hipGetDevice(&previous) != hipSuccess || hipSetDevice(info.device) != hipSuccess) return RT_ERR_HIP;
  struct Slot {
    hipStream_t stream = nullptr;
    uint8_t* device = nullptr;
    uint8_t* pinned = nullptr;
    int frame = -1;  // the frame whose image is in, or on its way into, `pinned`
  };
  std::vector<Slot> slots(S);
  bool hip_ok = true;
  for (int i = 0; i < S; i++) {  // streams and buffers belong to the scene
    Slot& s = slots[i];
    void* q = nullptr;
    hip_ok = hip_ok && rti::sweep_slot(scene, i, largest, &q, &s.device, &s.pinned) == RT_OK;
    s.stream = static_cast<hipStream_t>(q);
  }
  double t_wait = 0, t_copy = 0, t_issue = 0;  // host milliseconds, printed with RT_TRACE_SWEEP=1
  auto now = [] {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  };
  auto bytes_of = [&](int frame) { return (size_t)cameras[frame].hsize * cameras[frame].vsize * 3; };
  // waits until the slot's frame is in its pinned buffer
  auto wait = [&](Slot& s) {
    if (s.frame < 0) return true;
    const double t0 = now();
    bool ok = hipStreamSynchronize(s.stream) == hipSuccess;
    // a frame issued on an estimate of its ray-queue lengths that did not hold is rendered again
    int complete = 1;
    ok = ok && rti::estimated_frame_result(scene, s.stream, &complete) == RT_OK;
    if (ok && !complete)
      ok = rt_render_device(scene, &cameras[s.frame], depth, mode, pipeline, 0, 1, nullptr, s.device, s.stream, 0) ==
               RT_OK &&
           hipMemcpyAsync(s.pinned, s.device, bytes_of(s.frame), hipMemcpyDeviceToHost, s.stream) == hipSuccess &&
           hipStreamSynchronize(s.stream) == hipSuccess;
    t_wait += now() - t0;
    return ok;
  };
  // hands the slot's image to the caller
  auto deliver = [&](Slot& s) {
    if (s.frame < 0) return;
    const double t0 = now();
    std::memcpy(out[s.frame], s.pinned, bytes_of(s.frame));
    t_copy += now() - t0;
    s.frame = -1;
  };
  rc = hip_ok ? RT_OK : RT_ERR_HIP;
  for (int k = 0; k < n && rc == RT_OK; k++) {
    Slot& fresh = slots[k % S];         // delivered one iteration ago (or never used)
    Slot& oldest = slots[(k + 1) % S];  // holds frame k - F
    if (!wait(oldest)) {
      rc = RT_ERR_HIP;
      break;
    }
    const double t0 = now();
    rc = rti::render_device_estimated(scene, &cameras[k], depth, mode, pipeline, fresh.device, fresh.stream);
    if (rc == RT_OK && hipMemcpyAsync(fresh.pinned, fresh.device, bytes_of(k), hipMemcpyDeviceToHost,
                                      fresh.stream) != hipSuccess)
      rc = RT_ERR_HIP;
    t_issue += now() - t0;
    if (rc != RT_OK) break;
    fresh.frame = k;
    deliver(oldest);  // under frame k and the F - 1 frames before it
  }
  // drain in frame order; on an error the streams are still waited for, so that nothing is in
  // flight into the caller's or the scene's buffers when the call returns
  for (int j = 0; j < S; j++) {
    Slot& s = slots[(n + j) % S];
    if (rc == RT_OK) {
      if (wait(s))
        deliver(s);
      else
        rc = RT_ERR_HIP;
    } else if (s.stream) {
      (void)hipStreamSynchronize(s.stream);
    }
  }
  if (getenv("RT_TRACE_SWEEP"))
    fprintf(stderr, "sweep of %d frames, %d in flight: host waited %.2f ms, copied %.2f ms, issued %.2f ms\n", n, F,
            t_wait, t_copy, t_issue);
  (void)hipSetDevice(previous);
  return rc;
}

std::vector<Canvas> render_sequence(const std::vector<Camera>& cameras, DeviceScene& scene,
                                    const RenderOptions& opt, int inflight) {
  std::vector<Canvas> frames(cameras.size());
  std::vector<rt_camera> abi(cameras.size());
  std::vector<uint8_t*> out(cameras.size());
  for (size_t k = 0; k < cameras.size(); k++) {
    abi[k] = cameras[k].to_abi();
    frames[k].width = cameras[k].hsize;
    frames[k].height = cameras[k].vsize;
    frames[k].rgb8.resize((size_t)cameras[k].hsize * cameras[k].vsize * 3);
    out[k] = frames[k].rgb8.data();
  }
  int rc = render_sequence_raw(scene.handle(), abi.data(), (int)abi.size(), opt.depth, opt.mode, opt.pipeline,
                               inflight, out.data());
  if (rc != RT_OK)
    throw std::runtime_error(std::string("render_sequence: ") + (rc == RT_ERR_HIP ? "HIP runtime call failed; " : "") +
                             rt_last_error());
  return frames;
}

// ---- examples (examples.rs) -----------------------------------------------------------------
namespace examples {
namespace {
const double PI = 3.14159265358979323846264338327950288;
std::string slurp(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("Something went wrong reading the file: " + path);  // examples.rs:23
  std::stringstream ss;
  ss << f.rdbuf();
  return ss.str();
}
template <typename T>
T* push(World& w, T&& shape) {
  w.objects.push_back(std::make_unique<T>(std::move(shape)));
  return static_cast<T*>(w.objects.back().get());
}
}  // namespace

std::pair<Camera, World> tea_set(const std::string& dir, int hs, int vs) {  // examples.rs:19-90
  World world;
  Material tea_mat;
  tea_mat.pattern = Pattern::Solid({0.68359375, 0.55859375, 0.91015625});
  tea_mat.ambient = 0.35;
  std::string pot = slurp(dir + "/teapot.obj");
  auto tea_pot = parse_obj_file(pot, Matrix::rotation_x(PI / -2.0) * Matrix::rotation_z(PI / 6.0), tea_mat);
  std::string cup = slurp(dir + "/teacup.obj");
  auto cup_xf = [&](double tx, double tz) {
    return Matrix::scaling(0.08, 0.08, 0.08) *
           (Matrix::translation(tx, 4.0, tz) * (Matrix::rotation_y(PI / 4.0) * Matrix::rotation_x(PI / -2.0)));
  };
  auto tea_cup_right = parse_obj_file(cup, cup_xf(150.0, -150.0), tea_mat);
  auto tea_cup_left = parse_obj_file(cup, cup_xf(-175.0, -125.0), tea_mat);
  Plane plane(Matrix::translation(0.0, 1.0, 0.0));
  plane.material.pattern = Pattern::Solid(Color::black());
  plane.material.reflective = 0.6;
  plane.material.specular = 0.0;
  world.objects.push_back(std::move(tea_cup_left));
  world.objects.push_back(std::move(tea_pot));
  world.objects.push_back(std::move(tea_cup_right));
  push(world, std::move(plane));
  Camera camera = Camera::new_with_transform(
      hs > 0 ? hs : 1750, vs > 0 ? vs : 1000, PI / 3.0,
      view_transform(Tuple::point(0.0, 20.0, -40.0), Tuple::point(0.0, 2.0, 0.0), Tuple::vector(0.0, 1.0, 0.0)));
  world.light_sources.push_back({{1.0, 1.0, 1.0}, Tuple::point(-7.0, 11.0, -10.)});
  return {camera, std::move(world)};
}

std::pair<Camera, World> pawn_chess(const std::string& dir, int hs, int vs) {  // examples.rs:92-132
  World world;
  Material pawn_mat;
  pawn_mat.specular = 1.;
  pawn_mat.transparency = 1.0;
  pawn_mat.reflective = 0.9;
  pawn_mat.shininess = 300.;
  pawn_mat.ambient = 0.1;
  pawn_mat.diffuse = 0.1;
  pawn_mat.refractive_index = 1.52;
  world.objects.push_back(parse_obj_file(slurp(dir + "/pawn-chess.obj"), std::nullopt, pawn_mat));
  Plane plane(Matrix::scaling(2.0, 2.0, 2.0));
  plane.material.pattern = Pattern::Checkered(Color::black(), Color::white());
  plane.material.reflective = 0.3;
  push(world, std::move(plane));
  world.light_sources.push_back({{1.0, 1.0, 1.0}, Tuple::point(-10.0, 13.0, -10.)});
  Camera camera = Camera::new_with_transform(
      hs > 0 ? hs : 1000, vs > 0 ? vs : 1000, PI / 3.0,
      view_transform(Tuple::point(0.0, 4.0, -5.0), Tuple::point(0.0, 2.0, 0.0), Tuple::vector(0.0, 1.0, 0.0)));
  return {camera, std::move(world)};
}

std::pair<Camera, World> book_cover(int hs, int vs) {  // examples.rs:134-268
  World world;
  world.light_sources.push_back({{1.0, 1.0, 1.0}, Tuple::point(50.0, 100.0, -50.0)});
  world.light_sources.push_back({{0.2, 0.2, 0.2}, Tuple::point(-400.0, 50.0, -10.0)});
  Material white;
  white.pattern = Pattern::Solid(Color::white());
  white.diffuse = 0.7;
  white.ambient = 0.1;
  white.specular = 0.0;
  white.reflective = 0.1;
  Material blue = white, red = white, purple = white;
  blue.pattern = Pattern::Solid({0.537, 0.831, 0.914});
  red.pattern = Pattern::Solid({0.941, 0.322, 0.388});
  purple.pattern = Pattern::Solid({0.373, 0.404, 0.550});
  Matrix standard = Matrix::scaling(0.5, 0.5, 0.5) * Matrix::translation(1.0, -1.0, 1.0);
  Matrix large = Matrix::scaling(3.5, 3.5, 3.5) * standard;
  Matrix medium = Matrix::scaling(3.0, 3.0, 3.0) * standard;
  Matrix small = Matrix::scaling(2.0, 2.0, 2.0) * standard;

  Plane plane(Matrix::translation(0.0, 0.0, 500.0) * Matrix::rotation_x(PI / 2.0));
  plane.material.pattern = Pattern::Solid(Color::white());
  plane.material.ambient = 1.0;
  plane.material.diffuse = 0.0;
  plane.material.specular = 0.0;
  push(world, std::move(plane));

  Sphere sphere(Matrix::identity(4) * large);
  sphere.material.pattern = Pattern::Solid({0.373, 0.404, 0.550});
  sphere.material.diffuse = 0.2;
  sphere.material.ambient = 0.0;
  sphere.material.specular = 1.0;
  sphere.material.shininess = 200.0;
  sphere.material.reflective = 0.7;
  sphere.material.transparency = 0.7;
  sphere.material.refractive_index = 1.5;
  push(world, std::move(sphere));

  auto cube = [&](double x, double y, double z, const Matrix& base, const Material& m) {
    Cube c(Matrix::translation(x, y, z) * base);
    c.material = m;
    push(world, std::move(c));
  };
  cube(4.0, 0.0, 0.0, medium, white);
  cube(8.5, 1.5, -0.5, large, blue);
  cube(0.0, 0.0, 4.0, large, red);
  cube(4.0, 0.0, 4.0, small, white);
  cube(7.5, 0.5, 4.0, medium, purple);
  cube(-0.25, 0.25, 8.0, medium, white);
  cube(4.0, 1.0, 7.5, large, blue);
  cube(10.0, 2.0, 7.5, medium, red);
  cube(8.0, 2.0, 12.0, small, white);
  cube(20.0, 1.0, 9.0, small, white);
  cube(-0.5, -5.0, 0.25, large, blue);
  cube(4.0, -4.0, 0.0, large, red);
  cube(8.5, -4.0, 0.0, large, white);
  cube(0.0, -4.0, 4.0, large, white);
  cube(-0.5, -4.5, 8.0, large, purple);
  cube(0.0, -8.0, 4.0, large, purple);
  cube(-0.5, -8.5, 8.0, large, white);

  Camera camera = Camera::new_with_transform(
      hs > 0 ? hs : 2000, vs > 0 ? vs : 2000, 0.785,
      view_transform(Tuple::point(-6.0, 6.0, -10.0), Tuple::point(6.0, 0.0, 6.0), Tuple::vector(-0.45, 1.0, 0.0)));
  return {camera, std::move(world)};
}

std::pair<Camera, World> test_scene(int hs, int vs) {  // examples.rs:270-404
  World world;
  Sphere middle(Matrix::translation(-0.5, 1.0, 0.5));
  middle.material.pattern = Pattern::Solid(Color::black());
  middle.material.specular = 1.;
  middle.material.transparency = 1.0;
  middle.material.reflective = 0.9;
  middle.material.shininess = 300.;
  middle.material.ambient = 0.1;
  middle.material.diffuse = 0.1;
  middle.material.refractive_index = 1.52;

  Cube middle_behind(Matrix::translation(0.5, 1.0, 4.) * Matrix::rotation_y(PI / 3.));
  middle_behind.material.pattern = Pattern::Solid({1.0, 0.0, 0.0});
  middle_behind.material.diffuse = 0.7;
  middle_behind.material.specular = 0.3;
  middle_behind.material.shininess = 100.;
  middle_behind.material.reflective = 0.1;

  Sphere right(Matrix::translation(1.5, 0.5, -0.5) * (Matrix::scaling(0.5, 0.5, 0.5) * Matrix::rotation_z(-PI / 3.0)));
  right.material.pattern = Pattern::Checkered({0.461, 0.586, 0.336}, {0.93, 0.93, 0.82});
  right.material.pattern.set_transform(Matrix::scaling(0.5, 0.5, 0.5));
  right.material.diffuse = 0.7;
  right.material.specular = 0.3;
  right.material.reflective = 0.1;

  Sphere left(Matrix::translation(-1.5, 0.33, -0.75) * Matrix::scaling(0.33, 0.33, 0.33));
  left.material.pattern = Pattern::Rings({1.0, 0.8, 0.1}, Color::black());
  left.material.pattern.set_transform(Matrix::rotation_z(-PI / 3.0) * Matrix::scaling(0.33, 0.33, 0.33));
  left.material.diffuse = 0.7;
  left.material.specular = 0.3;
  left.material.reflective = 0.1;

  Plane floor;
  floor.material.pattern = Pattern::Checkered(Color::black(), Color::white());
  floor.material.reflective = 0.1;

  Plane ceil(Matrix::translation(0., 100., 0.));
  ceil.material.pattern = Pattern::Solid({0., 0.707, 0.882});
  ceil.material.specular = 1.;
  ceil.material.diffuse = 1.;
  ceil.material.ambient = 0.8;
  ceil.material.reflective = 0.3;

  Cylinder cylinder_outer(Matrix::translation(-2.5, 0.0, 4.0));
  cylinder_outer.minimum = 0.0;
  cylinder_outer.maximum = 1.0;
  cylinder_outer.closed = true;
  cylinder_outer.material.pattern = Pattern::Solid({1.0, 0.3, 1.0});
  cylinder_outer.material.specular = 1.;
  cylinder_outer.material.shininess = 20.;
  cylinder_outer.material.ambient = 0.5;
  cylinder_outer.material.diffuse = 0.1;
  cylinder_outer.material.reflective = 0.2;

  Cylinder cylinder_middle((Matrix::rotation_x(PI / -2.) * Matrix::scaling(0.66, 1.0, 0.66)) *
                           Matrix::translation(-4.0, -5.0, 2.5));
  cylinder_middle.minimum = 1.0;
  cylinder_middle.maximum = 1.5;
  cylinder_middle.closed = true;
  cylinder_middle.material.pattern = Pattern::Solid({0.0, 1.0, 0.0});
  cylinder_middle.material.refractive_index = 1.52;
  cylinder_middle.material.transparency = 0.7;
  cylinder_middle.material.specular = 1.;
  cylinder_middle.material.reflective = 0.9;
  cylinder_middle.material.shininess = 150.;
  cylinder_middle.material.ambient = 0.1;
  cylinder_middle.material.diffuse = 0.1;

  Cone cone((((Matrix::rotation_x(PI / 2.) * Matrix::rotation_z(PI / -3.)) * Matrix::rotation_x(PI / -7.4)) *
             Matrix::scaling(1.0, 2.0, 1.0)) *
            Matrix::translation(-1.0, 1.0, 1.0));
  cone.minimum = 0.0;
  cone.maximum = 1.0;
  cone.closed = true;
  cone.material.pattern = Pattern::Solid({1.0, 1.0, 0.0});
  cone.material.refractive_index = 1.52;
  cone.material.transparency = 0.7;
  cone.material.specular = 1.;
  cone.material.reflective = 0.9;
  cone.material.shininess = 150.;
  cone.material.ambient = 0.5;
  cone.material.diffuse = 0.1;
  cone.material.ambient = 0.2;

  push(world, std::move(left));
  push(world, std::move(middle));
  push(world, std::move(right));
  push(world, std::move(floor));
  push(world, std::move(middle_behind));
  push(world, std::move(cylinder_outer));
  push(world, std::move(cylinder_middle));
  push(world, std::move(cone));
  push(world, std::move(ceil));
  world.light_sources.push_back({{1.0, 1.0, 1.0}, Tuple::point(-10.0, 13.0, -10.)});
  Camera camera = Camera::new_with_transform(
      hs > 0 ? hs : 1920, vs > 0 ? vs : 1080, PI / 3.0,
      view_transform(Tuple::point(0.0, 3.0, -5.0), Tuple::point(0.0, 1.0, 0.0), Tuple::vector(0.0, 1.0, 0.0)));
  return {camera, std::move(world)};
}

// Stress scene for BVH traversal from HBM (no counterpart in the reference).
// SplitMix64 with seed 0x5EED5EED; unit doubles are (next() >> 11) * 2^-53.  One group
// (identity transform, default material) of `ntris` flat triangles: p1 uniform in
// [-20,20] x [0,20] x [-20,20], e1 and e2 uniform in [-0.15,0.15]^3, nine draws per
// triangle in the order p1.xyz, e1.xyz, e2.xyz; p2 = p1 + e1, p3 = p1 + e2.  Then 64 unit
// spheres at (-17.5+5i, 1, -17.5+5j), i outer, j inner, default material with reflective
// 0.8 and diffuse 0.3; a default floor plane; one white light at (-30,50,-30); camera
// 3840x2160, fov pi/3, from (0,25,-55) to (0,6,0), up (0,1,0).
std::pair<Camera, World> synthetic(int ntris, int hs, int vs) {
  World world;
  uint64_t state = 0x5EED5EEDULL;
  auto unit = [&]() {
    uint64_t z = (state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
  };
  auto g = std::make_unique<Group>();
  g->objects.reserve(ntris);
  for (int i = 0; i < ntris; i++) {
    double v[9];
    for (double& x : v) x = unit();
    Tuple p1 = Tuple::point(-20.0 + 40.0 * v[0], 20.0 * v[1], -20.0 + 40.0 * v[2]);
    Tuple e1 = Tuple::vector(-0.15 + 0.3 * v[3], -0.15 + 0.3 * v[4], -0.15 + 0.3 * v[5]);
    Tuple e2 = Tuple::vector(-0.15 + 0.3 * v[6], -0.15 + 0.3 * v[7], -0.15 + 0.3 * v[8]);
    g->add_object(std::make_unique<Triangle>(p1, p1 + e1, p1 + e2));
  }
  world.objects.push_back(std::move(g));
  for (int i = 0; i < 8; i++)
    for (int j = 0; j < 8; j++) {
      Sphere s(Matrix::translation(-17.5 + 5.0 * i, 1.0, -17.5 + 5.0 * j));
      s.material.reflective = 0.8;
      s.material.diffuse = 0.3;
      push(world, std::move(s));
    }
  push(world, Plane());
  world.light_sources.push_back({{1.0, 1.0, 1.0}, Tuple::point(-30.0, 50.0, -30.0)});
  Camera camera = Camera::new_with_transform(
      hs > 0 ? hs : 3840, vs > 0 ? vs : 2160, PI / 3.0,
      view_transform(Tuple::point(0.0, 25.0, -55.0), Tuple::point(0.0, 6.0, 0.0), Tuple::vector(0.0, 1.0, 0.0)));
  return {camera, std::move(world)};
}

std::pair<Camera, World> by_name(const std::string& name, const std::string& dir, int hs, int vs, int ntris) {
  if (name == "cover") return book_cover(hs, vs);
  if (name == "pawn") return pawn_chess(dir, hs, vs);
  if (name == "tea set" || name == "teaset") return tea_set(dir, hs, vs);
  if (name == "test_scene") return test_scene(hs, vs);
  if (name == "synthetic") return synthetic(ntris > 0 ? ntris : 1000000, hs, vs);
  throw std::runtime_error("Unrecognized scene");  // main.rs:50
}
}  // namespace examples

}  // namespace rt
