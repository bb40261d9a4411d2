// tests/host/test_trig_port.cpp — the shared acos / sin / cos restatement (a-loam_amd/csrc/aloam_trig.hpp), compiled for the host with
// -ffp-contract=off, against glibc: every result within 1 ulp (FDLIBM's bound; glibc's own routines are correctly rounded in nearly
// all cases), over the ranges the DISTORTION 1 path uses and far beyond: angles up to 1e5, multiples of pi / 2 and their neighbours,
// acos arguments next to 0, 0.5 and 1.  Prints "ok <count> identical <fraction>" or the first offenders; exit code 1 on failure.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include "../../a-loam_amd/csrc/aloam_trig.hpp"

static long long ulps(double a, double b) {
  if (a == b) return 0;
  if (std::isnan(a) || std::isnan(b)) return std::isnan(a) && std::isnan(b) ? 0 : 1LL << 40;
  long long x, y;
  memcpy(&x, &a, 8); memcpy(&y, &b, 8);
  if (x < 0) x = (long long)0x8000000000000000ull - x;
  if (y < 0) y = (long long)0x8000000000000000ull - y;
  return x > y ? x - y : y - x;
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 2000000;
  std::mt19937_64 rng(777);
  std::uniform_real_distribution<double> u(-1.0, 1.0);
  long bad = 0, count = 0, same = 0;
  auto check = [&](const char* name, double x, double a, double b) {
    ++count;
    const long long d = ulps(a, b);
    if (d == 0) ++same;
    if (d > 1 && bad++ < 10) printf("%s(%a) = %a, glibc %a (%lld ulp)\n", name, x, a, b, d);
  };
  auto trig = [&](double x) { check("sin", x, aloam::sin_port(x), std::sin(x)); check("cos", x, aloam::cos_port(x), std::cos(x)); };
  auto arc = [&](double x) { check("acos", x, aloam::acos_port(x), std::acos(x)); };
  for (int k = -70000; k <= 70000; ++k) {                       // multiples of pi / 2 and their neighbours (worst cancellation)
    double x = k * 1.5707963267948966;
    for (int d = -2; d <= 2; ++d) { double y = x; for (int s = 0; s < (d < 0 ? -d : d); ++s) y = std::nextafter(y, d < 0 ? -1e300 : 1e300); trig(y); }
  }
  for (long i = 0; i < n; ++i) {
    trig(u(rng) * 0.8);                                         // inside pi / 4
    trig(u(rng) * 3.0);
    trig(u(rng) * 20.0);                                        // s * theta with the reference's relTime quirk (s up to ~10)
    trig(u(rng) * std::pow(10.0, 5.0 * std::fabs(u(rng))));     // up to 1e5
    trig(u(rng) * std::pow(10.0, -12.0 * std::fabs(u(rng))));   // tiny
    arc(u(rng));
    arc(1.0 - std::pow(10.0, -16.0 * std::fabs(u(rng))));       // next to 1: small rotations, theta = acos(|w|)
    arc(0.5 + u(rng) * 1e-3);
    arc(u(rng) * 1e-9);
  }
  const double edges[] = {0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1.0 - 2.220446049250313e-16, 1.0 - 1.1102230246251565e-16, 0x1p-57, 0x1p-27, 0x1p-28};
  for (double e : edges) { arc(e); trig(e); }
  if (!std::isnan(aloam::acos_port(1.5)) || !std::isnan(aloam::sin_port(1e9)) || !std::isnan(aloam::cos_port(INFINITY))) { printf("out-of-range arguments must give NaN\n"); return 1; }
  if (bad) { printf("MORE THAN 1 ULP: %ld of %ld\n", bad, count); return 1; }
  printf("ok %ld identical %.6f\n", count, (double)same / (double)count);
  return 0;
}
