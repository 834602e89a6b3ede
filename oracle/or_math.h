/* TEST INFRASTRUCTURE ONLY (oracle).
 *
 * sin / cos / atan2 / hypot in plain IEEE-754 double add, mul, div, sqrt only (no FMA, no libm), so the
 * oracle's trajectories do not depend on which libm the host happens to have.  The reference calls
 * MATLAB's sin/cos/cart2pol (calc_trajectory.m:97,124,127-128,221,225); those builtins are not in the
 * repo and differ from any other libm by up to an ulp, so a self-contained <= 1 ulp evaluation is as
 * faithful a restatement as libm would be -- and it is reproducible.
 *
 * Method: Cody-Waite reduction by pi/2 in three parts + the classic minimax kernels on [-pi/4, pi/4];
 * atan by argument reduction to [0, 7/16] with a degree-11 odd/even split polynomial.
 * Valid for |x| < 2^20 (floe angles are a few radians).
 */
#ifndef OR_MATH_H
#define OR_MATH_H
#include <math.h>   /* sqrt only */

static inline double orm_abs(double v) { return v < 0 ? -v : v; }

static inline double orm_ksin(double x, double y, int iy) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

static inline double orm_kcos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  double hz = 0.5 * z;
  double w = 1.0 - hz;
  return w + (((1.0 - w) - hz) + (z * r - x * y));
}

/* x = n*(pi/2) + (y0 + y1), |y0 + y1| <= pi/4; returns n mod 4 */
static inline int orm_rem_pio2(double x, double *y0, double *y1) {
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
               pio2_1t = 6.07710050650619224932e-11, pio2_2 = 6.07710050630396597660e-11,
               pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
               pio2_3t = 8.47842766036889956997e-32;
  double ax = orm_abs(x);
  double fn = (double)(long long)(ax * invpio2 + 0.5);
  double r = ax - fn * pio2_1;
  double w = fn * pio2_1t;
  double a = r - w;
  /* second and third terms whenever the first difference lost more than a few bits */
  if (orm_abs(a) < 1e-5 * ax || fn > 64.0) {
    double t = r;
    w = fn * pio2_2;
    r = t - w;
    w = fn * pio2_2t - ((t - r) - w);
    a = r - w;
    if (orm_abs(a) < 1e-12 * ax) {
      t = r;
      w = fn * pio2_3;
      r = t - w;
      w = fn * pio2_3t - ((t - r) - w);
      a = r - w;
    }
  }
  double b = (r - a) - w;
  long long n = (long long)fn;
  if (x < 0) { *y0 = -a; *y1 = -b; return (int)((-n) & 3); }
  *y0 = a; *y1 = b;
  return (int)(n & 3);
}

static inline void orm_sincos(double x, double *s, double *c) {
  if (orm_abs(x) <= 0.785398163397448278999) {
    *s = orm_ksin(x, 0.0, 0);
    *c = orm_kcos(x, 0.0);
    return;
  }
  double y0, y1;
  int n = orm_rem_pio2(x, &y0, &y1);
  double ks = orm_ksin(y0, y1, 1), kc = orm_kcos(y0, y1);
  switch (n) {
  case 0: *s = ks; *c = kc; break;
  case 1: *s = kc; *c = -ks; break;
  case 2: *s = -ks; *c = -kc; break;
  default: *s = -kc; *c = ks; break;
  }
}
static inline double orm_sin(double x) { double s, c; orm_sincos(x, &s, &c); return s; }
static inline double orm_cos(double x) { double s, c; orm_sincos(x, &s, &c); return c; }

static inline double orm_atan_pos(double x) { /* x >= 0 */
  const double hi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01,
                        9.82793723247329054082e-01, 1.57079632679489655800e+00};
  const double lo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17,
                        1.39033110312309984516e-17, 6.12323399573676603587e-17};
  const double aT[11] = {3.33333333333329318027e-01, -1.99999999998764832476e-01,
                         1.42857142725034663711e-01, -1.11111104054623557880e-01,
                         9.09088713343650656196e-02, -7.69187620504482999495e-02,
                         6.66107313738753120669e-02, -5.83357013379057348645e-02,
                         4.97687799461593236017e-02, -3.65315727442169155270e-02,
                         1.62858201153657823623e-02};
  int id;
  if (x >= 1.0e19) return hi[3] + lo[3];
  if (x < 0.4375) {
    if (x < 1e-9) return x;
    id = -1;
  } else if (x < 1.1875) {
    if (x < 0.6875) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); }
    else { id = 1; x = (x - 1.0) / (x + 1.0); }
  } else {
    if (x < 2.4375) { id = 2; x = (x - 1.5) / (1.0 + 1.5 * x); }
    else { id = 3; x = -1.0 / x; }
  }
  double z = x * x;
  double w = z * z;
  double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  return hi[id] - ((x * (s1 + s2) - lo[id]) - x);
}

static inline double orm_atan2(double y, double x) {
  const double pi = 3.14159265358979311600e+00, pi_lo = 1.2246467991473531772e-16;
  if (x != x || y != y) return x + y;
  if (y == 0) {
    if (x > 0 || (x == 0 && 1.0 / x > 0)) return y;
    return (1.0 / y < 0) ? -pi : pi;
  }
  if (x == 0) return y > 0 ? 0.5 * pi : -0.5 * pi;
  double ay = orm_abs(y), ax = orm_abs(x);
  double z = orm_atan_pos(ay / ax);
  if (x < 0) z = pi - (z - pi_lo);
  return y < 0 ? -z : z;
}

/* real part of acos(a) for any real a (MATLAB's acos is complex outside [-1, 1]: real part 0 or pi) */
static inline double orm_acos_real(double a) {
  if (a != a) return a;
  if (a >= 1.0) return 0.0;
  if (a <= -1.0) return 3.14159265358979311600e+00;
  return orm_atan2(sqrt((1.0 - a) * (1.0 + a)), a);
}

#endif /* OR_MATH_H */
