// asan_main.cpp — drives the CPU side of the product under AddressSanitizer +
// UndefinedBehaviorSanitizer (make -C raytracer_amd asan-host-run): the host mirror (matrix
// algebra, OBJ loader, example scenes, PPM and PNG writers) and the host BVH builder.  No GPU
// code runs: nothing is committed or rendered.
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../csrc/rt_bvh.h"
#include "rt_host.h"

static void collect(const rt::Shape& s, std::vector<double>& p1, std::vector<double>& p2, std::vector<double>& p3) {
  if (s.kind == rt::ShapeKind::Group) {
    for (const auto& c : static_cast<const rt::Group&>(s).objects) collect(*c, p1, p2, p3);
  } else if (s.kind == rt::ShapeKind::Triangle || s.kind == rt::ShapeKind::SmoothTriangle) {
    const rt::Triangle& t = static_cast<const rt::Triangle&>(s);
    const rt::Tuple* ps[3] = {&t.p1, &t.p2, &t.p3};
    std::vector<double>* out[3] = {&p1, &p2, &p3};
    for (int k = 0; k < 3; k++) {
      out[k]->push_back(ps[k]->x);
      out[k]->push_back(ps[k]->y);
      out[k]->push_back(ps[k]->z);
    }
  }
}

int main(int argc, char** argv) {
  const std::string obj_dir = argc > 1 ? argv[1] : "../assets/obj";
  const std::string out = argc > 2 ? argv[2] : "/tmp/host_asan";
  try {
    for (const char* name : {"cover", "pawn", "tea set", "test_scene", "synthetic"}) {
      auto scene = rt::examples::by_name(name, obj_dir, 64, 36, std::string(name) == "synthetic" ? 5000 : 0);
      (void)scene.first.to_abi();
      std::vector<double> p1, p2, p3;
      for (const auto& o : scene.second.objects) collect(*o, p1, p2, p3);
      const int64_t n = (int64_t)p1.size() / 3;
      rtb::BuildResult br;
      rtb::build_sah(p1.data(), p2.data(), p3.data(), n, 12, br);
      std::vector<rtd::DNode4> wide;
      const int need = rtb::collapse_to_wide(br.nodes, br.cell, wide);
      rtb::reorder_top_breadth_first(wide, 128);
      std::printf("%s: %zu objects, %lld triangles, %zu binary / %zu wide nodes, stack %d\n", name,
                  scene.second.objects.size(), (long long)n, br.nodes.size(), wide.size(), need);
    }
    rt::Matrix m = rt::Matrix::scaling(2, 3, 4) * (rt::Matrix::translation(1, 2, 3) * rt::Matrix::rotation_y(0.3));
    (void)m.inverse().determinant();
    (void)rt::view_transform(rt::Tuple::point(0, 1, -5), rt::Tuple::point(0, 0, 0), rt::Tuple::vector(0, 1, 0));
    rt::Canvas c;
    c.width = 33;
    c.height = 17;
    c.rgb8.resize((size_t)c.width * c.height * 3);
    for (size_t i = 0; i < c.rgb8.size(); i++) c.rgb8[i] = (uint8_t)(i * 7);
    if (!c.write_to_ppm(out + ".ppm") || !c.write_to_png(out + ".png")) throw std::runtime_error("cannot write " + out);
    try {
      (void)rt::Matrix::from_rows(nullptr, 5);
      throw std::runtime_error("Matrix::from_rows accepted size 5");
    } catch (const std::invalid_argument&) {
    }
    try {
      (void)rt::parse_obj_file("v 0 0 0\nv 1 0 0\nf 1 2 9\n");
      std::printf("out-of-range face index accepted\n");
    } catch (const std::exception& e) {
      std::printf("bad OBJ refused: %s\n", e.what());
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  std::printf("sanitizer run complete\n");
  return 0;
}
