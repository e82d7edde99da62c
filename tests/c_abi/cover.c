/* A plain C11 host of include/rt_core.h (compiled with gcc, not hipcc): builds the reference's
 * default test world (scene/world.rs:180-204: a light at (-10,10,-10) and two spheres with colour
 * (0.8,1.0,0.6), diffuse 0.7, specular 0.2, the second one scaled by 0.5) through the
 * rt_scene_add_* calls, renders it with rt_render and prints the CRC-32 of the quantised image.
 * Usage: cover [out.rgb]   (out.rgb: the raw hsize*vsize*3 bytes) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_core.h"

static uint32_t crc32_of(const uint8_t* p, size_t n) {
  uint32_t c = 0xffffffffu;
  for (size_t i = 0; i < n; i++) {
    c ^= p[i];
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
  }
  return ~c;
}

static void identity(double m[16]) {
  memset(m, 0, 16 * sizeof(double));
  m[0] = m[5] = m[10] = m[15] = 1.0;
}

static int check(int rc, const char* what) {
  if (rc < 0 || (rc > 0 && strncmp(what, "add", 3) != 0)) {
    fprintf(stderr, "%s: %s\n", what, rt_last_error());
    exit(1);
  }
  return rc;
}

int main(int argc, char** argv) {
  enum { HS = 96, VS = 64 };
  rt_scene* s = rt_scene_create();
  rt_material m1;
  memset(&m1, 0, sizeof m1);
  m1.ambient = 0.1;
  m1.diffuse = 0.7;
  m1.specular = 0.2;
  m1.shininess = 200.0;
  m1.refractive_index = 1.0;
  m1.pattern_kind = RT_SOLID;
  m1.color_a[0] = 0.8;
  m1.color_a[1] = 1.0;
  m1.color_a[2] = 0.6;
  identity(m1.pattern_inv);
  const int i1 = check(rt_scene_add_material(s, &m1), "add_material");

  rt_primitive p;
  memset(&p, 0, sizeof p);
  p.kind = RT_SPHERE;
  p.parent = -1;
  p.minimum = -INFINITY;
  p.maximum = INFINITY;
  p.material = i1;
  identity(p.inv);
  check(rt_scene_add_primitive(s, &p), "add_primitive");
  p.inv[0] = p.inv[5] = p.inv[10] = 2.0; /* the inverse of scaling(0.5, 0.5, 0.5) */
  check(rt_scene_add_primitive(s, &p), "add_primitive");
  const double white[3] = {1.0, 1.0, 1.0}, lpos[3] = {-10.0, 10.0, -10.0};
  check(rt_scene_add_light(s, white, lpos), "add_light");

  if (rt_device_count() < 1) {
    fprintf(stderr, "no GPU: %d devices\n", rt_device_count());
    rt_scene_destroy(s);
    return 3;
  }
  check(rt_scene_commit(s, 0, 0), "commit");
  rt_scene_info info;
  check(rt_scene_get_info(s, &info), "get_info");

  /* Camera::new (scene/camera.rs:33-59) with fov pi/3 and the inverse of
   * view_transform((0,0,-5), (0,0,0), (0,1,0)) (camera.rs:189-209) */
  rt_camera cam;
  memset(&cam, 0, sizeof cam);
  cam.hsize = HS;
  cam.vsize = VS;
  const double half_view = tan((3.14159265358979323846 / 3.0) / 2.0);
  const double aspect = (double)HS / (double)VS;
  cam.half_width = half_view;
  cam.half_height = half_view / aspect;
  cam.pixel_size = (cam.half_width * 2.0) / (double)HS;
  identity(cam.inv);
  cam.inv[0] = -1.0;
  cam.inv[10] = -1.0;
  cam.inv[11] = -5.0;

  static uint8_t rgb[HS * VS * 3];
  rt_stats st;
  check(rt_render(s, &cam, 5, RT_MODE_F64_PARITY, RT_PIPELINE_WAVEFRONT, NULL, rgb, &st), "render");
  printf("primitives=%lld rays=%llu crc32=%08x\n", (long long)info.n_primitives,
         (unsigned long long)(st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract),
         crc32_of(rgb, sizeof rgb));
  if (argc > 1) {
    FILE* f = fopen(argv[1], "wb");
    if (!f || fwrite(rgb, 1, sizeof rgb, f) != sizeof rgb) {
      fprintf(stderr, "cannot write %s\n", argv[1]);
      return 1;
    }
    fclose(f);
  }
  rt_scene_destroy(s);
  return 0;
}
