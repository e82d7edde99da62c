// asan_main.cpp — drives the CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer
// (make -C oracle asan-run).  Test infrastructure, CPU only: loads every example scene, renders
// small frames in both search modes on several threads, traces single rays, writes a PPM.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

struct orc_scene;
extern "C" {
orc_scene* orc_scene_new();
void orc_scene_free(orc_scene*);
int orc_scene_load_example(orc_scene*, const char* name, const char* obj_dir, int hsize, int vsize, int synthetic_tris);
void orc_set_use_bvh(orc_scene*, int on);
int orc_render_rows(orc_scene*, int y0, int y1, int threads, int schedule, int depth, double* out_f64, uint8_t* out_u8,
                    uint64_t* counters8);
int orc_render_pixels(orc_scene*, int64_t n, const int32_t* xs, const int32_t* ys, int threads, int depth,
                      double* out_f64, uint8_t* out_u8, uint64_t* counters8);
void orc_trace_info(orc_scene*, const double* o3, const double* d3, double* out);
int orc_write_ppm(const char* path, int width, int height, const uint8_t* rgb);
void orc_mat_inverse(const double* m, int size, double* out);
}

int main(int argc, char** argv) {
  const std::string obj_dir = argc > 1 ? argv[1] : "../assets/obj";
  const std::string out = argc > 2 ? argv[2] : "/tmp/oracle_asan.ppm";
  struct Case {
    const char* name;
    int hs, vs, tris, bvh;
  };
  const Case cases[] = {{"cover", 48, 48, 0, 0},      {"test_scene", 48, 27, 0, 0}, {"pawn", 12, 9, 0, 0},
                        {"pawn", 64, 36, 0, 1},       {"tea set", 64, 36, 0, 1},    {"synthetic", 48, 27, 20000, 1}};
  for (const Case& c : cases) {
    orc_scene* s = orc_scene_new();
    if (orc_scene_load_example(s, c.name, obj_dir.c_str(), c.hs, c.vs, c.tris) != 0) {
      std::fprintf(stderr, "cannot load %s\n", c.name);
      return 1;
    }
    orc_set_use_bvh(s, c.bvh);
    std::vector<double> f64((size_t)c.hs * c.vs * 3);
    std::vector<uint8_t> u8((size_t)c.hs * c.vs * 3);
    uint64_t cnt[8];
    if (orc_render_rows(s, 0, c.vs, 3, 0, 5, f64.data(), u8.data(), cnt) != 0) return 1;
    std::vector<int32_t> xs, ys;
    for (int k = 0; k < 40; k++) {
      xs.push_back((k * 7) % c.hs);
      ys.push_back((k * 5) % c.vs);
    }
    if (orc_render_pixels(s, (int64_t)xs.size(), xs.data(), ys.data(), 2, 5, f64.data(), u8.data(), cnt) != 0) return 1;
    const double o[3] = {0.0, 1.5, -6.0}, d[3] = {0.0, -0.1, 1.0};
    double info[64];
    orc_trace_info(s, o, d, info);
    if (orc_write_ppm(out.c_str(), c.hs, c.vs, u8.data()) != 0) return 1;
    std::printf("%s %dx%d %s: %llu rays\n", c.name, c.hs, c.vs, c.bvh ? "culling" : "brute force",
                (unsigned long long)(cnt[0] + cnt[1] + cnt[2] + cnt[3]));
    orc_scene_free(s);
  }
  const double m[16] = {2, 0, 0, 1, 0, 3, 0, 2, 0, 0, 4, 3, 0, 0, 0, 1};
  double inv[16];
  orc_mat_inverse(m, 4, inv);
  std::printf("sanitizer run complete\n");
  return 0;
}
