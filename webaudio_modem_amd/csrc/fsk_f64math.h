// fsk_f64math.h -- lean double-precision elementary functions for the fp64 demodulator (fsk_demod.hip):
// sin/cos of an NCO phase in [0, 2 pi] and atan2 of a finite I/Q pair, branch-free, each within ~1 ulp of the correctly
// rounded value.  The reference calls Math.cos / Math.sin / Math.atan2 (fsk.ts:229-230, 251); V8's are an fdlibm port,
// the CPU restatement's are glibc's, the device library's are ocml's -- three implementations that already differ from each other
// in the last ulp, which is why fp64 intermediates are compared at 1e-12.  What the device
// library's versions cost is their generality: argument reduction for any magnitude (Payne-Hanek loops), special cases,
// 60-120 instructions and a dozen branches per call, three calls per input sample -- 80 % of the fp64 kernel's time
// (the exact path ran at 3 % of the roofline).  Here the argument ranges are known.
//
// Plain C++ (fma through the builtin): compiled for the device by hipcc and for the host by g++
// (tests/test_f64math_cpu.py checks both functions against libm over their whole argument range).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define FSK_HD __host__ __device__ inline
#else
#define FSK_HD inline
#endif

namespace fsk {

FSK_HD double f64_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// cos and sin of phi in [0, 2 pi] (the reference's localOscPhase after its `% (2 pi)`, fsk.ts:231).  Cody-Waite
// reduction by pi/2 in two pieces (n <= 4, so n * pio2_hi is exact: 33 x 3 bits) and fdlibm's kernel polynomials on
// |r| <= pi/4 (Sun's coefficients: the published minimax fits every fdlibm descendant uses).
FSK_HD void sincos_0_2pi(double phi, double &c, double &s) {
  const double n = __builtin_rint(phi * 6.36619772367581382433e-01);           // 2/pi
  double r = f64_fma(-n, 1.57079632673412561417e+00, phi);                     // first 33 bits of pi/2
  r = f64_fma(-n, 6.07710050650619224932e-11, r);                              // pi/2 - the above
  const double z = r * r;
  double ps = 1.58969099521155010221e-10;                                      // S6
  ps = f64_fma(ps, z, -2.50507602534068634195e-08);                            // S5
  ps = f64_fma(ps, z, 2.75573137070700676789e-06);                             // S4
  ps = f64_fma(ps, z, -1.98412698298579493134e-04);                            // S3
  ps = f64_fma(ps, z, 8.33333333332248946124e-03);                             // S2
  ps = f64_fma(ps, z, -1.66666666666666324348e-01);                            // S1
  const double sr = f64_fma(r * z, ps, r);
  double pc = -1.13596475577881948265e-11;                                     // C6
  pc = f64_fma(pc, z, 2.08757232129817482790e-09);                             // C5
  pc = f64_fma(pc, z, -2.75573143513906633035e-07);                            // C4
  pc = f64_fma(pc, z, 2.48015872894767294178e-05);                             // C3
  pc = f64_fma(pc, z, -1.38888888888741095749e-03);                            // C2
  pc = f64_fma(pc, z, 4.16666666666666019037e-02);                             // C1
  const double cr = f64_fma(z * z, pc, f64_fma(-0.5, z, 1.0));
  const int q = (int)n & 3;
  const double a = (q & 1) ? sr : cr, b = (q & 1) ? cr : sr;                   // quadrant: (c, s) = (cr, sr) rotated by q * pi/2
  c = (q == 1 || q == 2) ? -a : a;
  s = (q >= 2) ? -b : b;
}

// Math.atan2(y, x) for finite arguments with x never -0 (the reference's I/Q averages: sums that start at +0,
// fsk.ts:241-252).  Octant reduction to t = min/max in [0, 1], fdlibm's second reduction atan(t) = atan(c) +
// atan((t - c) / (1 + c t)) with c in {0, 1/2, 1} folded INTO the one division ((mn - c mx) / (mx + c mn)), fdlibm's
// degree-11 polynomial pair on |t'| < 7/16.  atan2(0, 0) = 0 as Math.atan2 has it.
FSK_HD double atan2_lean(double y, double x) {
  const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
  const bool swap = ay > ax;
  const double mx = swap ? ay : ax, mn = swap ? ax : ay;
  // breakpoints on t = mn / mx without dividing: t >= 7/16, t >= 11/16
  const bool k1 = (mn >= 0.4375 * mx) & (mn > 0.0), k2 = (mn >= 0.6875 * mx) & (mn > 0.0);   // ((0, 0): no reduction, t = 0)
  const double cc = k2 ? 1.0 : (k1 ? 0.5 : 0.0);
  const double hi = k2 ? 7.85398163397448278999e-01 : (k1 ? 4.63647609000806093515e-01 : 0.0);   // atan(1), atan(1/2)
  const double lo = k2 ? 3.06161699786838301793e-17 : (k1 ? 2.26987774529616870924e-17 : 0.0);
  const double num = f64_fma(-cc, mx, mn), den = f64_fma(cc, mn, mx);
  const double t = num / (den > 0.0 ? den : 1.0);                              // ((0, 0): 0 / 1; no unused quotient for the compiler to jump around)
  const double z = t * t, w = z * z;
  double s1 = 1.62858201153657823623e-02;                                      // aT[10]
  s1 = f64_fma(s1, w, 4.97687799461593236017e-02);                             // aT[8]
  s1 = f64_fma(s1, w, 6.66107313738753120669e-02);                             // aT[6]
  s1 = f64_fma(s1, w, 9.09088713343650656196e-02);                             // aT[4]
  s1 = f64_fma(s1, w, 1.42857142725034663711e-01);                             // aT[2]
  s1 = f64_fma(s1, w, 3.33333333333329318027e-01);                             // aT[0]
  s1 *= z;
  double s2 = -3.65315727442169155270e-02;                                     // aT[9]
  s2 = f64_fma(s2, w, -5.83357013379057348645e-02);                            // aT[7]
  s2 = f64_fma(s2, w, -7.69187620504482999495e-02);                            // aT[5]
  s2 = f64_fma(s2, w, -1.11111104054623557880e-01);                            // aT[3]
  s2 = f64_fma(s2, w, -1.99999999998764832476e-01);                            // aT[1]
  s2 *= w;
  double r = hi - ((t * (s1 + s2) - lo) - t);                                  // atan(mn / mx) in [0, pi/4]
  r = swap ? 1.57079632679489661923 - r : r;                                   // (pi/2's low word, 6e-17, is below the ulp of any r >= pi/4)
  r = x < 0.0 ? 3.14159265358979323846 - r : r;
  return __builtin_copysign(r, y);
}

}  // namespace fsk
