// rt_pow.h — x^y of the specular term, shared by the device code and a host-side unit test.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define RT_POW_HD __host__ __device__
#else
#define RT_POW_HD
#endif

namespace rtd {
namespace powbits {
RT_POW_HD inline int32_t hi_word(double x) {
  uint64_t b;
  __builtin_memcpy(&b, &x, 8);
  return (int32_t)(b >> 32);
}
RT_POW_HD inline uint32_t lo_word(double x) {
  uint64_t b;
  __builtin_memcpy(&b, &x, 8);
  return (uint32_t)b;
}
RT_POW_HD inline double from_words(int32_t hi, uint32_t lo) {
  const uint64_t b = ((uint64_t)(uint32_t)hi << 32) | lo;
  double x;
  __builtin_memcpy(&x, &b, 8);
  return x;
}
}  // namespace powbits

// x^y of the specular term (light.rs:78), x = reflect . eye in (0, 1], y = shininess: exp(y ln x)
// with the kernels of fdlibm's log and exp (each below one ulp) in plain double arithmetic.
// The product y ln x carries |y ln x| ulps of relative error into the result, i.e. at most
// 1.2e-16 ABSOLUTE (results near 1 have y ln x near 0; measured against glibc's pow over 2e7
// arguments): the colour it scales feeds no decision.  A quarter of the instructions of the
// library's pow, which stays for everything outside the domain below.
RT_POW_HD inline double pow_specular(double x, double y) {
  // (one call of the library's pow for both ways out of the domain: it is inlined where it stands)
  const bool inside = x >= 2.2250738585072014e-308 && x <= 2.0 && fabs(y) <= 1e6;
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  int hx = powbits::hi_word(x);
  int k = (hx >> 20) - 1023;
  hx &= 0x000fffff;
  const int up = (hx + 0x95f64) & 0x100000;  // mantissa at or above sqrt(2): halve it, raise the exponent
  const double m = powbits::from_words(hx | (up ^ 0x3ff00000), powbits::lo_word(x));
  k += up >> 20;
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double dk = (double)k;
  const double z = s * s, w = z * z;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double lnx = dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  const double t = y * lnx;
  if (!(inside && fabs(t) < 700.0)) return pow(x, y);
  const double invln2 = 1.44269504088896338700e+00,
               P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  const int n = (int)(invln2 * t + (t < 0.0 ? -0.5 : 0.5));
  const double dn = (double)n;
  const double hi = t - dn * ln2_hi, lo = dn * ln2_lo;
  const double r = hi - lo;
  const double tt = r * r;
  const double c = r - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
  const double e = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  if (n >= -1000) return powbits::from_words(powbits::hi_word(e) + n * 0x100000, powbits::lo_word(e));
  return powbits::from_words(powbits::hi_word(e) + (n + 1000) * 0x100000, powbits::lo_word(e)) * 9.33263618503218878990e-302;
}

}  // namespace rtd
