// Host check of webaudio_modem_amd/csrc/fsk_f64math.h (tests/test_f64math_cpu.py compiles this with g++): both functions
// against long double libm over their argument ranges.  Prints the worst errors; exits 1 beyond the stated bounds.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "fsk_f64math.h"
static double ulp(double v) { v = std::fabs(v); if (v == 0) return 4.9e-324; int e; std::frexp(v, &e); return std::ldexp(1.0, e - 53); }
int main() {
  srand48(1);
  double worst_s = 0, worst_c = 0, worst_a = 0;
  for (int i = 0; i < 2000000; i++) {
    double phi = drand48() * 6.283185307179586;
    if (i < 1000) phi = (i % 5) * 1.5707963267948966 + (drand48() - 0.5) * 1e-9 * (i % 7);   // around the quadrant boundaries
    if (phi < 0) phi = 0;
    double c, s;
    fsk::sincos_0_2pi(phi, c, s);
    worst_c = std::fmax(worst_c, std::fabs((double)(c - cosl((long double)phi))) / 1.1102230246251565e-16);
    worst_s = std::fmax(worst_s, std::fabs((double)(s - sinl((long double)phi))) / 1.1102230246251565e-16);
  }
  for (int i = 0; i < 3000000; i++) {
    const double m = std::exp((drand48() - 0.5) * 200.0);
    double x = (drand48() * 2 - 1) * m, y = (drand48() * 2 - 1) * m;
    if (i % 7 == 0) x = y * (0.4375 + (drand48() - 0.5) * 1e-12);     // around the second reduction's breakpoints
    if (i % 11 == 0) y = x * (0.6875 + (drand48() - 0.5) * 1e-12);
    if (i % 13 == 0) y = x * (1 + (drand48() - 0.5) * 1e-12);
    if (i % 17 == 0) x = 0;
    if (i % 19 == 0) y = 0;
    if (x == 0 && std::signbit(x)) x = 0.0;                           // (the I/Q averages are never -0)
    const double a = fsk::atan2_lean(y, x);
    const long double al = atan2l((long double)y, (long double)x);
    worst_a = std::fmax(worst_a, std::fabs((double)(a - al)) / ulp((double)al));
  }
  const bool zeros = fsk::atan2_lean(0, 0) == 0.0 && fsk::atan2_lean(0, -1) == 3.14159265358979323846 &&
                     fsk::atan2_lean(1, 0) == 1.57079632679489661923 && fsk::atan2_lean(-1, 0) == -1.57079632679489661923;
  printf("sincos worst |error| / 2^-53: cos %.3f sin %.3f; atan2 worst %.3f ulp; conventions %s\n", worst_c, worst_s, worst_a, zeros ? "ok" : "WRONG");
  return (worst_c < 2.0 && worst_s < 2.0 && worst_a < 2.0 && zeros) ? 0 : 1;
}
