// fsk_fdlibm.h -- sin() the way V8 computes Math.sin: the fdlibm 5.3 algorithm (k_sin.c, k_cos.c, e_rem_pio2.c) that
// V8's src/base/ieee754.cc ports.  FSKCore.modulateData stores (float)Math.sin(phase) for an unwrapped phase that grows
// into the thousands of radians (fsk.ts:398-406); two correctly-working libms can differ in the last ulp of the double
// and, rarely, flip the float rounding.  Restating the same sequence of IEEE double operations (this TU is built
// -ffp-contract=off) makes the modulated Float32Array bit-identical to the reference's at any length -- and it is about
// half the instructions of the device library's sin().
//
// Covered exactly: |x| <= 2^19 * pi/2 (fdlibm's "medium" argument reduction: three-stage Cody-Waite with the
// cancellation checks).  Beyond that fdlibm switches to Payne-Hanek (__kernel_rem_pio2); such phases need more than a
// minute of continuous tone at 48 kHz, and fall through to the device library's sin().
#pragma once
#include <stdint.h>
#include <string.h>

namespace fsk {
namespace fdlibm {

__host__ __device__ inline uint32_t hi_word(double x) {
  uint64_t u;
  memcpy(&u, &x, sizeof(u));
  return (uint32_t)(u >> 32);
}
__host__ __device__ inline double from_words(uint32_t hi, uint32_t lo) {
  const uint64_t u = ((uint64_t)hi << 32) | lo;
  double x;
  memcpy(&x, &u, sizeof(x));
  return x;
}

// __kernel_sin(x, y, iy): sin(x + y) on [-pi/4, pi/4], y the tail of x (iy = 0: y is zero)
__host__ __device__ inline double k_sin(double x, double y, int iy) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const uint32_t ix = hi_word(x) & 0x7fffffffu;
  if (ix < 0x3e400000u) {  // |x| < 2^-27
    if ((int)x == 0) return x;
  }
  const double z = x * x;
  const double v = z * x;
  const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

// __kernel_cos(x, y)
__host__ __device__ inline double k_cos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const uint32_t ix = hi_word(x) & 0x7fffffffu;
  if (ix < 0x3e400000u) {
    if ((int)x == 0) return 1.0;
  }
  const double z = x * x;
  const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  // |x| < 0.3: 1 - (0.5 z - (z r - x y)), which is the general form below with qx = 0 (0.5 z - 0 and 1 - 0 are exact)
  double qx = from_words(ix - 0x00200000u, 0u);  // x/4
  qx = ix > 0x3fe90000u ? 0.28125 : qx;          // x > 0.78125
  qx = ix < 0x3FD33333u ? 0.0 : qx;
  const double hz = 0.5 * z - qx;
  const double a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}

// __ieee754_rem_pio2(x, y) for pi/4 < |x| <= 2^19*pi/2; returns n, y[0] + y[1] = x - n*pi/2
__host__ __device__ inline int rem_pio2_medium(double x, double *y0, double *y1) {
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
               pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
               pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
               pio2_3t = 8.47842766036889956997e-32;
  // fdlibm's npio2_hw[n-1] table (high words of n*pi/2, n = 1..32) is reproduced by hi_word(n * 1.5707963267948966)
  // for all 32 entries (checked against the table); computing it keeps a per-lane indexed table out of scratch
  const uint32_t hx = hi_word(x);
  const uint32_t ix = hx & 0x7fffffffu;
  const bool neg = (hx >> 31) != 0;
  if (ix < 0x4002d97cu) {  // |x| < 3pi/4: n = +-1
    if (!neg) {
      double z = x - pio2_1;
      if (ix != 0x3ff921fbu) {
        *y0 = z - pio2_1t;
        *y1 = (z - *y0) - pio2_1t;
      } else {  // near pi/2: use 33+33+53 bits of pi
        z -= pio2_2;
        *y0 = z - pio2_2t;
        *y1 = (z - *y0) - pio2_2t;
      }
      return 1;
    }
    double z = x + pio2_1;
    if (ix != 0x3ff921fbu) {
      *y0 = z + pio2_1t;
      *y1 = (z - *y0) + pio2_1t;
    } else {
      z += pio2_2;
      *y0 = z + pio2_2t;
      *y1 = (z - *y0) + pio2_2t;
    }
    return -1;
  }
  double t = neg ? -x : x;
  const int n = (int)(t * invpio2 + 0.5);
  const double fn = (double)n;
  double r = t - fn * pio2_1;
  double w = fn * pio2_1t;  // first round, good to 85 bits
  double v0;
  if (n < 32 && ix != hi_word(fn * 1.5707963267948966)) {
    v0 = r - w;  // quick check: no cancellation
  } else {
    const uint32_t j = ix >> 20;
    v0 = r - w;
    uint32_t i = j - ((hi_word(v0) >> 20) & 0x7ffu);
    if ((int32_t)i > 16) {  // second iteration, good to 118 bits
      t = r;
      w = fn * pio2_2;
      r = t - w;
      w = fn * pio2_2t - ((t - r) - w);
      v0 = r - w;
      i = j - ((hi_word(v0) >> 20) & 0x7ffu);
      if ((int32_t)i > 49) {  // third iteration, 151 bits
        t = r;
        w = fn * pio2_3;
        r = t - w;
        w = fn * pio2_3t - ((t - r) - w);
        v0 = r - w;
      }
    }
  }
  const double v1 = (r - v0) - w;
  if (neg) {
    *y0 = -v0;
    *y1 = -v1;
    return -n;
  }
  *y0 = v0;
  *y1 = v1;
  return n;
}

// sin(x) as fdlibm's s_sin.c; `exact` is cleared when |x| is beyond the medium range (caller falls back)
__host__ __device__ inline double sin_medium(double x, bool *exact) {
  const uint32_t ix = hi_word(x) & 0x7fffffffu;
  *exact = true;
  if (ix <= 0x3fe921fbu) return k_sin(x, 0.0, 0);  // |x| <= pi/4
  if (ix >= 0x7ff00000u) return x - x;             // inf / NaN
  if (ix > 0x413921fbu) {                          // |x| > 2^19 * pi/2
    *exact = false;
    return 0.0;
  }
  double y0, y1;
  const int n = rem_pio2_medium(x, &y0, &y1);
  // both kernels, then select: lanes of a wave sit in different quadrants, a switch would run all four arms anyway
  const double sv = k_sin(y0, y1, 1), cv = k_cos(y0, y1);
  const double v = (n & 1) ? cv : sv;
  return (n & 2) ? -v : v;
}


// Straight-line form of the same computation for the common case -- x >= 3pi/4, within the medium range, no second
// reduction stage, reduced argument not tiny -- so that several sines can be scheduled as one basic block and hide
// each other's latency.  `slow` is set when x is outside that case; the caller then takes sin_medium() for that value.
// Inside the case every operation and its order equal sin_medium()'s, so the result is bit-identical.
__host__ __device__ inline double sin_straight(double x, bool *slow) {
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
               pio2_1t = 6.07710050650619224932e-11;
  const uint32_t hx = hi_word(x);
  const uint32_t ix = hx & 0x7fffffffu;
  const int n = (int)(x * invpio2 + 0.5);
  const double fn = (double)n;
  const double r = x - fn * pio2_1;
  const double w = fn * pio2_1t;
  const double y0 = r - w;
  const double y1 = (r - y0) - w;
  const bool quick = n < 32 && ix != hi_word(fn * 1.5707963267948966);
  const uint32_t ey = (hi_word(y0) >> 20) & 0x7ffu;
  const bool more = !quick && (int32_t)((ix >> 20) - ey) > 16;
  const bool tiny = (hi_word(y0) & 0x7fffffffu) < 0x3e400000u;
  *slow = (hx >> 31) != 0 || ix < 0x4002d97cu || ix > 0x413921fbu || more || tiny;
  // k_sin(y0, y1, 1) and k_cos(y0, y1) without their |x| < 2^-27 early-outs (excluded above)
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double z = y0 * y0;
  const double v = z * y0;
  const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  const double sv = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
  const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  const uint32_t iy = hi_word(y0) & 0x7fffffffu;
  double qx = from_words(iy - 0x00200000u, 0u);
  qx = iy > 0x3fe90000u ? 0.28125 : qx;
  qx = iy < 0x3FD33333u ? 0.0 : qx;
  const double hz = 0.5 * z - qx;
  const double a = 1.0 - qx;
  const double cv = a - (hz - (z * rc - y0 * y1));
  const double res = (n & 1) ? cv : sv;
  return (n & 2) ? -res : res;
}

}  // namespace fdlibm
}  // namespace fsk
