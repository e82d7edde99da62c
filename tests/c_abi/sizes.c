/* Prints the size of every struct of include/rt_core.h as a C compiler lays it out; the test
 * suite compares them with the ctypes mirrors and with the repr(C) mirrors of the Rust shim. */
#include <stdio.h>

#include "rt_core.h"

int main(void) {
  printf("rt_material %zu\n", sizeof(rt_material));
  printf("rt_group %zu\n", sizeof(rt_group));
  printf("rt_primitive %zu\n", sizeof(rt_primitive));
  printf("rt_mesh %zu\n", sizeof(rt_mesh));
  printf("rt_camera %zu\n", sizeof(rt_camera));
  printf("rt_stats %zu\n", sizeof(rt_stats));
  printf("rt_scene_info %zu\n", sizeof(rt_scene_info));
  return 0;
}
