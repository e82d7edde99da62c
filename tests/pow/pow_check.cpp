// Host-side check of pow_specular (raytracer_amd/csrc/rt_pow.h, the routine the kernels use for
// the specular term) against the C library's pow: prints the worst absolute error, the worst
// relative error among results above 1e-6, and how many arguments were tried.
#include <cmath>
#include <cstdio>
#include <cstdint>

#include "rt_pow.h"

static uint64_t state = 0x5EED5EEDull;
static double uniform() {  // SplitMix64
  uint64_t z = (state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

int main() {
  const double ys[] = {0.0, 0.5, 1.0, 2.0, 2.5, 3.0, 5.0, 10.0, 10.3, 50.0, 77.7, 100.0, 200.0, 200.25, 300.0, 1000.0};
  double worst_abs = 0.0, worst_rel = 0.0;
  long n = 0;
  for (long it = 0; it < 4000000; it++) {
    double x = (it & 1) ? 1.0 - uniform() * uniform() * uniform() : uniform();
    if (it % 7 == 0) x = 1.0 + (uniform() - 0.5) * 1e-9;
    if (x <= 0.0) continue;
    const double y = ys[it % 16];
    const double got = rtd::pow_specular(x, y), want = std::pow(x, y);
    const double err = std::fabs(got - want);
    if (err > worst_abs) worst_abs = err;
    if (want > 1e-6 && err / want > worst_rel) worst_rel = err / want;
    n++;
  }
  // outside the fast domain the library routine answers: equal bits expected
  const double nan = std::nan(""), inf = HUGE_VAL;
  const double outside[][2] = {{1e-320, 2.0}, {3.0, 2.0}, {0.5, 2e6}, {1e-300, 300.0}, {0.5, 1074.0}, {nan, 2.0},
                               {0.5, nan}, {inf, 2.0}, {0.5, inf}, {0.0, 2.0}, {-0.5, 2.0}};
  int edge_bad = 0;
  for (const auto& e : outside) {
    const double got = rtd::pow_specular(e[0], e[1]), want = std::pow(e[0], e[1]);
    if (!(got == want || (got != got && want != want))) edge_bad++;
  }
  // unusual arguments inside it: large results, negative exponents, exact cases
  const double inside[][2] = {{2.0, 1000.0}, {0.5, -3.0}, {1.0, 300.0}, {2.0, 10.0}, {0.25, 0.5}, {1e-300, 2.0}};
  for (const auto& e : inside) {
    const double got = rtd::pow_specular(e[0], e[1]), want = std::pow(e[0], e[1]);
    if (!(std::fabs(got - want) <= 1e-12 * std::fabs(want))) edge_bad++;
  }
  std::printf("%ld %.3e %.3e %d\n", n, worst_abs, worst_rel, edge_bad);
  return 0;
}
