// orb_math.h -- deterministic fp64 atan2 / sin / cos shared by host and gfx950 device code.
//
// Why this exists: the reference computes the keypoint orientation as
// std::atan2((double)m01,(double)m10) (src/ORB_SLAM2/src/ORBExtractor.cc:486) and rotates the BRIEF
// template with std::cos/std::sin of that angle (:434-435).  glibc's libm and ROCm's ocml are
// both "< 1 ulp" but not bit-identical, so a device kernel calling ocml could differ from a host
// run in the last bit.  Everything in this header uses only IEEE-754 +,-,*,/ on doubles in a fixed
// order (no FMA contraction: the translation units are built with -ffp-contract=off), so the host
// build and the gfx950 build return bit-identical results by construction.
//
// The algorithms are the classic Sun fdlibm ones (Cody-Waite reduction by pi/2 in three 33-bit
// pieces, minimax kernels on [-pi/4, pi/4], 4-interval atan reduction), restated here for the
// restricted domain the front end needs: atan2 of two finite doubles, sin/cos of |x| <= ~1e5.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define ORB_HD __host__ __device__ inline
#else
#define ORB_HD inline
#endif

namespace orbmath {

ORB_HD uint32_t hi_word(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  return (uint32_t)(u >> 32);
}
ORB_HD uint32_t lo_word(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  return (uint32_t)u;
}
ORB_HD double clear_lo_word(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  u &= 0xFFFFFFFF00000000ull;
  memcpy(&x, &u, 8);
  return x;
}
ORB_HD double dabs(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  u &= 0x7FFFFFFFFFFFFFFFull;
  memcpy(&x, &u, 8);
  return x;
}

// ---- kernels on [-pi/4, pi/4], argument x + y (y = tail) -----------------------------------
ORB_HD double k_sin(double x, double y, int have_tail) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  uint32_t ix = hi_word(x) & 0x7fffffffu;
  if (ix < 0x3e400000u) {  // |x| < 2^-27
    if ((int)x == 0) return x;
  }
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (!have_tail) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

ORB_HD double k_cos(double x, double y) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  uint32_t ix = hi_word(x) & 0x7fffffffu;
  if (ix < 0x3e400000u) {
    if ((int)x == 0) return 1.0;
  }
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  if (ix < 0x3FD33333u) return 1.0 - (0.5 * z - (z * r - x * y));
  double qx;
  if (ix > 0x3fe90000u) {
    qx = 0.28125;
  } else {
    uint64_t u = ((uint64_t)(ix - 0x00200000u)) << 32;  // x/4 with the low word cleared
    memcpy(&qx, &u, 8);
  }
  double hz = 0.5 * z - qx;
  double a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}

// ---- x = n*pi/2 + (y0 + y1), |y0| <= pi/4; valid for |x| < 2^19*pi/2 ------------------------
ORB_HD int rem_pio2(double x, double* y0, double* y1) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
               pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
               pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
               pio2_3t = 8.47842766036889956997e-32;
  uint32_t hx = hi_word(x);
  uint32_t ix = hx & 0x7fffffffu;
  if (ix <= 0x3fe921fbu) {  // |x| <= pi/4
    *y0 = x;
    *y1 = 0.0;
    return 0;
  }
  double t = dabs(x);
  int n = (int)(t * invpio2 + 0.5);
  double fn = (double)n;
  double r = t - fn * pio2_1;
  double w = fn * pio2_1t;
  int j = (int)(ix >> 20);
  double a = r - w;
  int i = j - (int)((hi_word(a) >> 20) & 0x7ffu);
  if (i > 16) {  // need a second 33-bit piece
    double t2 = r;
    w = fn * pio2_2;
    r = t2 - w;
    w = fn * pio2_2t - ((t2 - r) - w);
    a = r - w;
    i = j - (int)((hi_word(a) >> 20) & 0x7ffu);
    if (i > 49) {  // third piece
      t2 = r;
      w = fn * pio2_3;
      r = t2 - w;
      w = fn * pio2_3t - ((t2 - r) - w);
      a = r - w;
    }
  }
  double b = (r - a) - w;
  if ((int32_t)hx < 0) {
    *y0 = -a;
    *y1 = -b;
    return -n;
  }
  *y0 = a;
  *y1 = b;
  return n;
}

ORB_HD void det_sincos(double x, double* s, double* c) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  double y0, y1;
  int n = rem_pio2(x, &y0, &y1);
  uint32_t ix = hi_word(x) & 0x7fffffffu;
  int tail = (ix > 0x3fe921fbu);
  double ks = k_sin(y0, y1, tail);
  double kc = k_cos(y0, y1);
  switch (n & 3) {
    case 0: *s = ks; *c = kc; break;
    case 1: *s = kc; *c = -ks; break;
    case 2: *s = -ks; *c = -kc; break;
    default: *s = -kc; *c = ks; break;
  }
}

ORB_HD double det_atan(double x) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const double atanhi0 = 4.63647609000806093515e-01, atanhi1 = 7.85398163397448278999e-01,
               atanhi2 = 9.82793723247329054082e-01, atanhi3 = 1.57079632679489655800e+00;
  const double atanlo0 = 2.26987774529616870924e-17, atanlo1 = 3.06161699786838301793e-17,
               atanlo2 = 1.39033110312309984516e-17, atanlo3 = 6.12323399573676603587e-17;
  const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
               aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
               aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
               aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
               aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
               aT10 = 1.62858201153657823623e-02;
  uint32_t hx = hi_word(x);
  uint32_t ix = hx & 0x7fffffffu;
  int neg = ((int32_t)hx < 0);
  if (ix >= 0x44100000u) {  // |x| >= 2^66 (or inf; NaN excluded by the callers)
    double z = atanhi3 + atanlo3;
    return neg ? -z : z;
  }
  int id;
  double hi = 0.0, lo = 0.0;
  if (ix < 0x3fdc0000u) {  // |x| < 0.4375
    if (ix < 0x3e200000u) return x;  // |x| < 2^-29
    id = -1;
  } else {
    x = dabs(x);
    if (ix < 0x3ff30000u) {    // |x| < 1.1875
      if (ix < 0x3fe60000u) {  // 7/16 <= |x| < 11/16
        id = 0; hi = atanhi0; lo = atanlo0;
        x = (2.0 * x - 1.0) / (2.0 + x);
      } else {                 // 11/16 <= |x| < 19/16
        id = 1; hi = atanhi1; lo = atanlo1;
        x = (x - 1.0) / (x + 1.0);
      }
    } else {
      if (ix < 0x40038000u) {  // |x| < 2.4375
        id = 2; hi = atanhi2; lo = atanlo2;
        x = (x - 1.5) / (1.0 + 1.5 * x);
      } else {
        id = 3; hi = atanhi3; lo = atanlo3;
        x = -1.0 / x;
      }
    }
  }
  double z = x * x;
  double w = z * z;
  double s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  double s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0) return x - x * (s1 + s2);
  z = hi - ((x * (s1 + s2) - lo) - x);
  return neg ? -z : z;
}

// atan2 for finite (non-NaN, non-inf) arguments -- the front end only ever passes integers.
ORB_HD double det_atan2(double y, double x) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const double pi = 3.1415926535897931160E+00, pi_o_2 = 1.5707963267948965580E+00,
               pi_lo = 1.2246467991473531772E-16, tiny = 1.0e-300;
  uint32_t hx = hi_word(x), hy = hi_word(y);
  uint32_t lx = lo_word(x), ly = lo_word(y);
  uint32_t ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
  if (hx == 0x3ff00000u && lx == 0) return det_atan(y);  // x == 1.0
  int m = (int)((hy >> 31) & 1u) | (int)((hx >> 30) & 2u);   // 2*sign(x) + sign(y)
  if ((iy | ly) == 0) {  // y == 0
    switch (m) {
      case 0:
      case 1: return y;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if ((ix | lx) == 0) return ((int32_t)hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;  // x == 0
  int k = ((int)iy - (int)ix) >> 20;
  double z;
  if (k > 60) {
    z = pi_o_2 + 0.5 * pi_lo;
    m &= 1;
  } else if ((int32_t)hx < 0 && k < -60) {
    z = 0.0;
  } else {
    z = det_atan(dabs(y / x));
  }
  switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}

}  // namespace orbmath
