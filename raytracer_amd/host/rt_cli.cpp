// rt_cli.cpp — command line of the reference (src/main.rs:13-55) on top of the GPU core.
//   -t/--threads N   accepted for compatibility (main.rs:19-26); the GPU needs no thread count
//   -e/--example X   "cover" (default) | "pawn" | "tea set"   (main.rs:27-35, 46-51)
// Additions: --width/--height (override the example's size), --gpu N, --obj-dir DIR,
// --out FILE (default canvas.ppm, main.rs:54; a name ending in .png is written as PNG), --stats.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "rt_host.h"

int main(int argc, char** argv) {
  std::string example = "cover", out = "canvas.ppm", obj_dir = "./obj";
  int width = 0, height = 0, gpu = -1;
  bool stats = false;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "-t" || a == "--threads") {
      char* end = nullptr;
      const char* v = next();
      long t = std::strtol(v, &end, 10);
      if (!*v || *end || t < 0) {
        std::printf("Invalid number of threads\n");  // main.rs:41
        return 0;
      }
    } else if (a == "-e" || a == "--example") {
      example = next();
    } else if (a == "--width") {
      width = std::atoi(next());
    } else if (a == "--height") {
      height = std::atoi(next());
    } else if (a == "--gpu") {
      gpu = std::atoi(next());
    } else if (a == "--obj-dir") {
      obj_dir = next();
    } else if (a == "--out") {
      out = next();
    } else if (a == "--stats") {
      stats = true;
    } else {
      std::fprintf(stderr, "unknown argument %s\n", a.c_str());
      return 2;
    }
  }
  if (example != "cover" && example != "pawn" && example != "tea set") {
    std::fprintf(stderr, "error: '%s' isn't a valid value for '--example'\n", example.c_str());
    return 2;
  }
  try {
    auto scene = rt::examples::by_name(example, obj_dir, width, height);
    rt::RenderOptions opt;
    opt.device = gpu;
    rt_stats st;
    if (stats) opt.stats = &st;
    std::printf("Rendering image on GPU %d\n", gpu < 0 ? 0 : gpu);
    rt::Canvas image = rt::render(scene.first, scene.second, opt);
    const bool png = out.size() >= 4 && out.compare(out.size() - 4, 4, ".png") == 0;
    if (!(png ? image.write_to_png(out) : image.write_to_ppm(out))) {
      std::fprintf(stderr, "could not create file %s\n", out.c_str());
      return 1;
    }
    if (stats) {
      unsigned long long rays = st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract;
      std::printf("rays %llu  kernel %.3f ms  %.1f Mrays/s\n", rays, st.kernel_ms,
                  rays / (st.kernel_ms * 1e3));
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
