/* v8_sin.h -- Math.sin as V8 computes it (TEST INFRASTRUCTURE, part of the oracle).
 *
 * The reference's modulator stores (float)Math.sin(phase) (fsk.ts:403).  Math.sin in V8 is base::ieee754::sin, a port
 * of fdlibm 5.3 (s_sin.c, k_sin.c, k_cos.c, e_rem_pio2.c; V8 is not part of /root/reference: it is the engine the
 * reference runs on, restated here from the published fdlibm algorithm).  glibc's sin is at least as accurate but is
 * a different algorithm, so the two can differ in the last ulp of the double; restating the fdlibm operation sequence
 * keeps the oracle's modulated Float32Array identical to the reference's at any length.  Pinned by the modulate
 * fixtures of tests/golden (tests/test_oracle_golden.py).  Built with -ffp-contract=off.
 *
 * v8_sin() is exact for |x| <= 2^19*pi/2 (fdlibm's medium range); beyond it the reference's libm switches to
 * Payne-Hanek reduction and this file falls back to the C library's sin().
 */
#ifndef V8_SIN_H
#define V8_SIN_H
#include <math.h>
#include <stdint.h>
#include <string.h>

static uint32_t v8s_hi(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)(u >> 32); }
static double v8s_make(uint32_t hi, uint32_t lo) { uint64_t u = ((uint64_t)hi << 32) | lo; double x; memcpy(&x, &u, 8); return x; }

static double v8s_ksin(double x, double y, int iy) { /* k_sin.c */
  static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                      S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  uint32_t ix = v8s_hi(x) & 0x7fffffffu;
  double z, r, v;
  if (ix < 0x3e400000u) { if ((int)x == 0) return x; }
  z = x * x;
  v = z * x;
  r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

static double v8s_kcos(double x, double y) { /* k_cos.c */
  static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                      C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  uint32_t ix = v8s_hi(x) & 0x7fffffffu;
  double z, r, qx, hz, a;
  if (ix < 0x3e400000u) { if ((int)x == 0) return 1.0; }
  z = x * x;
  r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  if (ix < 0x3FD33333u) return 1.0 - (0.5 * z - (z * r - x * y));
  if (ix > 0x3fe90000u) qx = 0.28125;
  else qx = v8s_make(ix - 0x00200000u, 0u);
  hz = 0.5 * z - qx;
  a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}

static int v8s_rem_pio2(double x, double *y) { /* e_rem_pio2.c, pi/4 < |x| <= 2^19*pi/2 */
  static const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                      pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
                      pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
                      pio2_3t = 8.47842766036889956997e-32;
  static const uint32_t npio2_hw[32] = {
      0x3FF921FB, 0x400921FB, 0x4012D97C, 0x401921FB, 0x401F6A7A, 0x4022D97C, 0x4025FDBB, 0x402921FB,
      0x402C463A, 0x402F6A7A, 0x4031475C, 0x4032D97C, 0x40346B9C, 0x4035FDBB, 0x40378FDB, 0x403921FB,
      0x403AB41B, 0x403C463A, 0x403DD85A, 0x403F6A7A, 0x40407E4C, 0x4041475C, 0x4042106C, 0x4042D97C,
      0x4043A28C, 0x40446B9C, 0x404534AC, 0x4045FDBB, 0x4046C6CB, 0x40478FDB, 0x404858EB, 0x404921FB};
  uint32_t hx = v8s_hi(x), ix = hx & 0x7fffffffu, j, i;
  int neg = (hx >> 31) != 0, n;
  double z, t, r, w, fn;
  if (ix < 0x4002d97cu) {
    if (!neg) {
      z = x - pio2_1;
      if (ix != 0x3ff921fbu) { y[0] = z - pio2_1t; y[1] = (z - y[0]) - pio2_1t; }
      else { z -= pio2_2; y[0] = z - pio2_2t; y[1] = (z - y[0]) - pio2_2t; }
      return 1;
    }
    z = x + pio2_1;
    if (ix != 0x3ff921fbu) { y[0] = z + pio2_1t; y[1] = (z - y[0]) + pio2_1t; }
    else { z += pio2_2; y[0] = z + pio2_2t; y[1] = (z - y[0]) + pio2_2t; }
    return -1;
  }
  t = neg ? -x : x;
  n = (int)(t * invpio2 + 0.5);
  fn = (double)n;
  r = t - fn * pio2_1;
  w = fn * pio2_1t;
  if (n < 32 && ix != npio2_hw[n - 1]) {
    y[0] = r - w;
  } else {
    j = ix >> 20;
    y[0] = r - w;
    i = j - ((v8s_hi(y[0]) >> 20) & 0x7ffu);
    if ((int32_t)i > 16) {
      t = r;
      w = fn * pio2_2;
      r = t - w;
      w = fn * pio2_2t - ((t - r) - w);
      y[0] = r - w;
      i = j - ((v8s_hi(y[0]) >> 20) & 0x7ffu);
      if ((int32_t)i > 49) {
        t = r;
        w = fn * pio2_3;
        r = t - w;
        w = fn * pio2_3t - ((t - r) - w);
        y[0] = r - w;
      }
    }
  }
  y[1] = (r - y[0]) - w;
  if (neg) { y[0] = -y[0]; y[1] = -y[1]; return -n; }
  return n;
}

static double v8_sin(double x) { /* s_sin.c */
  uint32_t ix = v8s_hi(x) & 0x7fffffffu;
  double y[2];
  int n;
  if (ix <= 0x3fe921fbu) return v8s_ksin(x, 0.0, 0);
  if (ix >= 0x7ff00000u) return x - x;
  if (ix > 0x413921fbu) return sin(x); /* Payne-Hanek range: not restated */
  n = v8s_rem_pio2(x, y);
  switch (n & 3) {
    case 0: return v8s_ksin(y[0], y[1], 1);
    case 1: return v8s_kcos(y[0], y[1]);
    case 2: return -v8s_ksin(y[0], y[1], 1);
    default: return -v8s_kcos(y[0], y[1]);
  }
}
#endif
