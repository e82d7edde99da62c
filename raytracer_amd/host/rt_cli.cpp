// rt_cli.cpp — command line of the reference (src/main.rs:13-55) on top of the GPU core.
//   -t/--threads N   accepted for compatibility (main.rs:19-26); the GPU needs no thread count
//   -e/--example X   "cover" (default) | "pawn" | "tea set"   (main.rs:27-35, 46-51)
// Additions: --width/--height (override the example's size), --gpu N (which GPU), --gpus N (how many:
// devices 0..N-1 share the frame's tiles, one RCCL gather), --mode f64|f32, --obj-dir DIR,
// --out FILE (default canvas.ppm, main.rs:54; a name ending in .png is written as PNG), --stats,
// --sweep N (a turntable of N frames over one committed scene: FILE_000, FILE_001, ...).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_host.h"

int main(int argc, char** argv) {
  std::string example = "cover", out = "canvas.ppm", obj_dir = "./obj";
  int width = 0, height = 0, gpu = -1, gpus = 0, sweep = 0, mode = RT_MODE_F64_PARITY;
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
    } else if (a == "--gpus") {
      gpus = std::atoi(next());
      if (gpus < 1) {
        std::fprintf(stderr, "--gpus needs a positive number\n");
        return 2;
      }
    } else if (a == "--mode") {
      const std::string m = next();
      if (m == "f64") {
        mode = RT_MODE_F64_PARITY;
      } else if (m == "f32") {
        mode = RT_MODE_F32_FAST;
      } else {
        std::fprintf(stderr, "error: '%s' isn't a valid value for '--mode' (f64, f32)\n", m.c_str());
        return 2;
      }
    } else if (a == "--obj-dir") {
      obj_dir = next();
    } else if (a == "--out") {
      out = next();
    } else if (a == "--sweep") {
      sweep = std::atoi(next());
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
    opt.gpus = gpus;
    opt.mode = mode;
    rt_stats st;
    if (stats) opt.stats = &st;
    const bool png = out.size() >= 4 && out.compare(out.size() - 4, 4, ".png") == 0;
    if (sweep > 0) {
      // a turntable: frame k sees the world turned by k * 2 pi / N about its y axis; the scene is
      // committed once and the frames are rendered with four of them in flight
      rt::DeviceScene ds;
      if (!ds.commit(scene.second, gpu)) throw std::runtime_error("commit: " + ds.error());
      std::vector<rt::Camera> cameras;
      for (int k = 0; k < sweep; k++)
        cameras.push_back(rt::Camera::new_with_transform(
            scene.first.hsize, scene.first.vsize, scene.first.field_of_view,
            scene.first.transform * rt::Matrix::rotation_y(k * 6.283185307179586 / sweep)));
      std::printf("Rendering %d frames on GPU %d\n", sweep, gpu < 0 ? 0 : gpu);
      std::vector<rt::Canvas> frames = rt::render_sequence(cameras, ds, opt, 4);
      const size_t dot = out.rfind('.');
      const std::string stem = dot == std::string::npos ? out : out.substr(0, dot);
      const std::string ext = dot == std::string::npos ? "" : out.substr(dot);
      for (int k = 0; k < sweep; k++) {
        char num[16];
        std::snprintf(num, sizeof num, "_%03d", k);
        const std::string name = stem + num + ext;
        if (!(png ? frames[k].write_to_png(name) : frames[k].write_to_ppm(name))) {
          std::fprintf(stderr, "could not create file %s\n", name.c_str());
          return 1;
        }
      }
      return 0;
    }
    if (gpus > 0)
      std::printf("Rendering image on %d GPU(s)\n", gpus);  // camera.rs:111
    else
      std::printf("Rendering image on GPU %d\n", gpu < 0 ? 0 : gpu);
    rt::Canvas image = rt::render(scene.first, scene.second, opt);
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
