// detmath.hpp -- sin / cos / atan2 for the integrate kernel, in plain IEEE-754 double add, mul, div
// only (no FMA contraction, no vendor libm), so that the GPU, and any host, produce the same bits.
//
// The reference calls MATLAB's sin/cos/cart2pol (calc_trajectory.m:97,124,127-128,221,225).  A floe's
// ring is rotated by alpha_i every step and then snapped to a 2^-32 m grid by polyclip.m:66, so an ulp of
// difference in sin/cos can move a clipped vertex by one grid unit; evaluating the functions the same way
// everywhere removes that source of divergence.  Accuracy <= 1 ulp (sin, cos), <= 2 ulp (atan2).
//
// Method: Cody-Waite reduction by pi/2 in three parts and minimax kernels on [-pi/4, pi/4]; atan by
// reduction to [0, 7/16] and an odd polynomial split in even/odd halves.  |x| < 2^20.
#pragma once
#include "vatti.hpp" // SZ_HD

namespace szm {

SZ_HD double absd(double v) { return v < 0 ? -v : v; }

SZ_HD double ksin(double x, double y, int iy) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

SZ_HD double kcos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  double hz = 0.5 * z;
  double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * r - x * y));
}

SZ_HD int rem_pio2(double x, double &y0, double &y1) {
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
               pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
               pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
               pio2_3t = 8.47842766036889956997e-32;
  double ax = absd(x);
  double fn = (double)(long long)(ax * invpio2 + 0.5);
  double r = ax - fn * pio2_1;
  double w = fn * pio2_1t;
  double a = r - w;
  if (absd(a) < 1e-5 * ax || fn > 64.0) {
    double t = r;
    w = fn * pio2_2;
    r = t - w;
    w = fn * pio2_2t - ((t - r) - w);
    a = r - w;
    if (absd(a) < 1e-12 * ax) {
      t = r;
      w = fn * pio2_3;
      r = t - w;
      w = fn * pio2_3t - ((t - r) - w);
      a = r - w;
    }
  }
  double b = (r - a) - w;
  long long n = (long long)fn;
  if (x < 0) { y0 = -a; y1 = -b; return (int)((-n) & 3); }
  y0 = a; y1 = b;
  return (int)(n & 3);
}

SZ_HD void sincos(double x, double &s, double &c) {
  if (absd(x) <= 0.785398163397448278999) {
    s = ksin(x, 0.0, 0);
    c = kcos(x, 0.0);
    return;
  }
  double y0, y1;
  int n = rem_pio2(x, y0, y1);
  double ks = ksin(y0, y1, 1), kc = kcos(y0, y1);
  switch (n) {
  case 0: s = ks; c = kc; break;
  case 1: s = kc; c = -ks; break;
  case 2: s = -ks; c = -kc; break;
  default: s = -kc; c = ks; break;
  }
}

SZ_HD double atan_pos(double x) {
  const double hi0 = 4.63647609000806093515e-01, hi1 = 7.85398163397448278999e-01,
               hi2 = 9.82793723247329054082e-01, hi3 = 1.57079632679489655800e+00;
  const double lo0 = 2.26987774529616870924e-17, lo1 = 3.06161699786838301793e-17,
               lo2 = 1.39033110312309984516e-17, lo3 = 6.12323399573676603587e-17;
  const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01,
               a2 = 1.42857142725034663711e-01, a3 = -1.11111104054623557880e-01,
               a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02,
               a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02,
               a8 = 4.97687799461593236017e-02, a9 = -3.65315727442169155270e-02,
               a10 = 1.62858201153657823623e-02;
  int id;
  double hi = 0, lo = 0;
  if (x >= 1.0e19) return hi3 + lo3;
  if (x < 0.4375) {
    if (x < 1e-9) return x;
    id = -1;
  } else if (x < 1.1875) {
    if (x < 0.6875) { id = 0; hi = hi0; lo = lo0; x = (2.0 * x - 1.0) / (2.0 + x); }
    else { id = 1; hi = hi1; lo = lo1; x = (x - 1.0) / (x + 1.0); }
  } else {
    if (x < 2.4375) { id = 2; hi = hi2; lo = lo2; x = (x - 1.5) / (1.0 + 1.5 * x); }
    else { id = 3; hi = hi3; lo = lo3; x = -1.0 / x; }
  }
  double z = x * x;
  double w = z * z;
  double s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
  double s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
  if (id < 0) return x - x * (s1 + s2);
  return hi - ((x * (s1 + s2) - lo) - x);
}

SZ_HD double atan2(double y, double x) {
  const double pi = 3.14159265358979311600e+00, pi_lo = 1.2246467991473531772e-16;
  if (x != x || y != y) return x + y;
  if (y == 0) {
    if (x > 0 || (x == 0 && 1.0 / x > 0)) return y;
    return (1.0 / y < 0) ? -pi : pi;
  }
  if (x == 0) return y > 0 ? 0.5 * pi : -0.5 * pi;
  double ay = absd(y), ax = absd(x);
  double z = atan_pos(ay / ax);
  if (x < 0) z = pi - (z - pi_lo);
  return y < 0 ? -z : z;
}

// real(acos(a)) for any real a (MATLAB returns a complex value outside [-1, 1]; its real part is 0 or pi)
SZ_HD double acos_real(double a) {
  if (a != a) return a;
  if (a >= 1.0) return 0.0;
  if (a <= -1.0) return 3.14159265358979311600e+00;
  return atan2(::sqrt((1.0 - a) * (1.0 + a)), a);
}

} // namespace szm
