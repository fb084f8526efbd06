/* e3d_libm.h -- bit-defined elementary functions shared by the HIP kernels, the host code and the CPU oracle.
 *
 * The reference calls the C library (std::atan2 / cos / sin / tan / atan / log2 on float: pcl::eigen33 via
 * two_pass_normal_3d.h:92-109, camera_base_impl_fisheye.h:66-153, camera_fisheye_fov.h, visibility_estimator.cc:437).
 * The device math library and glibc differ in the last ulp for these functions, which is enough to flip a border
 * observation or to turn a normal by 1e-4 and with it a correspondence count.  Here every function is
 *     evaluate in IEEE binary64 with +, -, *, / only (fixed operation order, no fused multiply-add), round ONCE to binary32.
 * x86-64 (SSE2) and gfx950 implement these four operations and the f64 -> f32 conversion identically (round to nearest
 * even, denormals kept), and every translation unit that includes this header is built with -ffp-contract=off, so the
 * results are bit-identical on the CPU and on the GPU.  The binary64 algorithms are the classic fdlibm ones (argument
 * reduction + minimax polynomial, < 1 ulp in binary64), so the binary32 result is the correctly rounded value of the exact
 * function except when the exact value lies within ~2^-29 (relative) of a rounding boundary -- glibc's own float functions
 * are within 1 ulp of the same value, i.e. they agree with these in all but last-bit cases (tests/test_libm.py measures it).
 *
 * sin / cos / tan reduce |x| < 2^20 with a three-part pi/2 (Cody-Waite, 118+ bits) and larger binary32 arguments with an
 * integer multiplication by 192 bits of 2/pi (the method of the ARM optimized routines / glibc sinf).
 *
 * Plain C99 / C++ / HIP.  Everything is `static inline`; device code gets __host__ __device__.
 */
#ifndef E3D_LIBM_H_
#define E3D_LIBM_H_

#if defined(__HIPCC__) || defined(__HIP__)
#define E3D_LIBM_FN __host__ __device__ static inline
#else
#define E3D_LIBM_FN static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

E3D_LIBM_FN unsigned e3d_libm_hi(double x) { unsigned long long u; __builtin_memcpy(&u, &x, 8); return (unsigned)(u >> 32); }
E3D_LIBM_FN unsigned e3d_libm_lo(double x) { unsigned long long u; __builtin_memcpy(&u, &x, 8); return (unsigned)u; }
E3D_LIBM_FN double e3d_libm_make(unsigned hi, unsigned lo) {
  unsigned long long u = ((unsigned long long)hi << 32) | lo; double x; __builtin_memcpy(&x, &u, 8); return x;
}
E3D_LIBM_FN double e3d_libm_abs(double x) { return e3d_libm_make(e3d_libm_hi(x) & 0x7fffffffu, e3d_libm_lo(x)); }

/* ---- atan (binary64): reduction to [0, 7/16] around 0.5, 1, 1.5, inf + odd polynomial ---- */
E3D_LIBM_FN double e3d_libm_atan(double x) {
  const double hi0 = 4.63647609000806093515e-01, hi1 = 7.85398163397448278999e-01, hi2 = 9.82793723247329054082e-01,
               hi3 = 1.57079632679489655800e+00;
  const double lo0 = 2.26987774529616870924e-17, lo1 = 3.06161699786838301793e-17, lo2 = 1.39033110312309984516e-17,
               lo3 = 6.12323399573676603587e-17;
  const double a0 = 3.33333333333329318027e-01, a1 = -1.99999999998764832476e-01, a2 = 1.42857142725034663711e-01,
               a3 = -1.11111104054623557880e-01, a4 = 9.09088713343650656196e-02, a5 = -7.69187620504482999495e-02,
               a6 = 6.66107313738753120669e-02, a7 = -5.83357013379057348645e-02, a8 = 4.97687799461593236017e-02,
               a9 = -3.65315727442169155270e-02, a10 = 1.62858201153657823623e-02;
  const unsigned hx = e3d_libm_hi(x), ix = hx & 0x7fffffffu;
  double ahi = 0.0, alo = 0.0, num, den;
  int id;
  if (ix >= 0x44100000u) {                      /* |x| >= 2^66, inf, NaN */
    if (ix > 0x7ff00000u || (ix == 0x7ff00000u && e3d_libm_lo(x) != 0u)) return x + x;
    return (hx >> 31) ? -(hi3 + lo3) : (hi3 + lo3);
  }
  if (ix < 0x3e400000u) return x;               /* |x| < 2^-27 */
  /* The reduced argument is a quotient in four of the five ranges.  Numerator and denominator are SELECTED per range and
   * divided once (the fifth range divides by one, which is exact): the lanes of a wavefront whose arguments fall into
   * different ranges then share one division instead of running four, and the operands -- hence the bits -- of every
   * operation are those of the branchy form. */
  if (ix < 0x3fdc0000u) {                       /* |x| < 0.4375 */
    id = -1; num = x; den = 1.0;
  } else {
    x = e3d_libm_abs(x);
    if (ix < 0x3ff30000u) {                     /* |x| < 1.1875 */
      if (ix < 0x3fe60000u) { id = 0; ahi = hi0; alo = lo0; num = 2.0 * x - 1.0; den = 2.0 + x; }
      else { id = 1; ahi = hi1; alo = lo1; num = x - 1.0; den = x + 1.0; }
    } else {
      if (ix < 0x40038000u) { id = 2; ahi = hi2; alo = lo2; num = x - 1.5; den = 1.0 + 1.5 * x; }
      else { id = 3; ahi = hi3; alo = lo3; num = -1.0; den = x; }
    }
  }
  x = num / den;
  {
    const double z = x * x, w = z * z;
    const double s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))));
    const double s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))));
    if (id < 0) return x - x * (s1 + s2);
    {
      const double r = ahi - ((x * (s1 + s2) - alo) - x);
      return (hx >> 31) ? -r : r;
    }
  }
}

/* ---- atan2 (binary64) ---- */
E3D_LIBM_FN double e3d_libm_atan2(double y, double x) {
  const double tiny = 1.0e-300, pi_o_4 = 7.8539816339744827900e-01, pi_o_2 = 1.5707963267948965580e+00,
               pi = 3.1415926535897931160e+00, pi_lo = 1.2246467991473531772e-16;
  const unsigned hx = e3d_libm_hi(x), lx = e3d_libm_lo(x), hy = e3d_libm_hi(y), ly = e3d_libm_lo(y);
  const unsigned ix = hx & 0x7fffffffu, iy = hy & 0x7fffffffu;
  int m, k;
  double z;
  if (ix > 0x7ff00000u || (ix == 0x7ff00000u && lx != 0u) || iy > 0x7ff00000u || (iy == 0x7ff00000u && ly != 0u)) return x + y;
  if (hx == 0x3ff00000u && lx == 0u) return e3d_libm_atan(y);                 /* x == 1 */
  m = (int)((hy >> 31) & 1u) | (int)((hx >> 30) & 2u);                         /* 2 sign(x) + sign(y) */
  if ((iy | ly) == 0u) {                                                       /* y == 0 */
    if (m < 2) return y;
    return (m == 2) ? pi + tiny : -pi - tiny;
  }
  if ((ix | lx) == 0u) return (hy >> 31) ? -pi_o_2 - tiny : pi_o_2 + tiny;     /* x == 0 */
  if (ix == 0x7ff00000u) {                                                     /* x == inf */
    if (iy == 0x7ff00000u) {
      if (m == 0) return pi_o_4 + tiny;
      if (m == 1) return -pi_o_4 - tiny;
      if (m == 2) return 3.0 * pi_o_4 + tiny;
      return -3.0 * pi_o_4 - tiny;
    }
    if (m == 0) return 0.0;
    if (m == 1) return -0.0;
    return (m == 2) ? pi + tiny : -pi - tiny;
  }
  if (iy == 0x7ff00000u) return (hy >> 31) ? -pi_o_2 - tiny : pi_o_2 + tiny;   /* y == inf */
  k = (int)(iy >> 20) - (int)(ix >> 20);
  if (k > 60) { z = pi_o_2 + 0.5 * pi_lo; m &= 1; }                            /* |y / x| > 2^60 */
  else if ((hx >> 31) && k < -60) z = 0.0;                                      /* 0 > |y| / x > -2^-60 */
  else z = e3d_libm_atan(e3d_libm_abs(y / x));
  if (m == 0) return z;
  if (m == 1) return -z;
  if (m == 2) return pi - (z - pi_lo);
  return (z - pi_lo) - pi;
}

/* ---- sin / cos kernels on [-pi/4, pi/4] (x + y = reduced argument) and the reduction ---- */
E3D_LIBM_FN double e3d_libm_ksin(double x, double y, int iy) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const unsigned ix = e3d_libm_hi(x) & 0x7fffffffu;
  double z, v, r;
  if (ix < 0x3e400000u) return x;               /* |x| < 2^-27 */
  z = x * x; v = z * x;
  r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}
E3D_LIBM_FN double e3d_libm_kcos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const unsigned ix = e3d_libm_hi(x) & 0x7fffffffu;
  double z, r, qx, hz, a;
  if (ix < 0x3e400000u) return 1.0;             /* |x| < 2^-27 */
  z = x * x;
  r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  if (ix < 0x3fd33333u) return 1.0 - (0.5 * z - (z * r - x * y));
  qx = (ix > 0x3fe90000u) ? 0.28125 : e3d_libm_make(ix - 0x00200000u, 0u);     /* x / 4 */
  hz = 0.5 * z - qx;
  a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}
/* x = n pi/2 + (y0 + y1), |y0 + y1| <= pi/4; returns n mod 4 */
E3D_LIBM_FN int e3d_libm_rem_pio2(double x, double* y0, double* y1) {
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11,
               pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
               pio2_3t = 8.47842766036889956997e-32, rnd = 6755399441055744.0;       /* 1.5 * 2^52 */
  const unsigned ix = e3d_libm_hi(x) & 0x7fffffffu;
  const double fn = (x * invpio2 + rnd) - rnd;                                  /* nearest integer (round to nearest even) */
  double r = x - fn * pio2_1, w = fn * pio2_1t, t, y;
  int j = (int)(ix >> 20), i, n;
  y = r - w;
  i = j - (int)((e3d_libm_hi(y) >> 20) & 0x7ffu);
  if (i > 16) {                                 /* second iteration, good to 118 bits */
    t = r; w = fn * pio2_2; r = t - w; w = fn * pio2_2t - ((t - r) - w); y = r - w;
    i = j - (int)((e3d_libm_hi(y) >> 20) & 0x7ffu);
    if (i > 49) {                               /* third iteration, 151 bits */
      t = r; w = fn * pio2_3; r = t - w; w = fn * pio2_3t - ((t - r) - w); y = r - w;
    }
  }
  *y0 = y;
  *y1 = (r - y) - w;
  n = (int)(((long long)fn) & 3);                 /* |x| < 2^20 here */
  return n;
}
/* |x| >= 2^20 (binary32 bits xi; valid from 2^7): |x| = n pi/2 + r with |r| <= pi/4, by multiplying the 24-bit significand with the bits of
 * 4/pi that matter for its exponent (integer arithmetic only); returns r, *np = n mod 4 */
E3D_LIBM_FN double e3d_libm_reduce_large(unsigned xi, int* np) {
  const unsigned inv_pio4[24] = {0xa2u, 0xa2f9u, 0xa2f983u, 0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u,
                                 0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu,
                                 0xf534ddc0u, 0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u,
                                 0x993c4390u, 0x3c439041u};
  const double pi63 = 3.40612158008655459330e-19;               /* pi / 2^63 */
  const unsigned* arr = &inv_pio4[(xi >> 26) & 15u];
  const int shift = (int)((xi >> 23) & 7u);
  unsigned long long n, res0, res1, res2;
  unsigned m = (xi & 0xffffffu) | 0x800000u;
  m <<= shift;
  res0 = (unsigned long long)(unsigned)(m * arr[0]);
  res1 = (unsigned long long)m * arr[4];
  res2 = (unsigned long long)m * arr[8];
  res0 = (res2 >> 32) | (res0 << 32);
  res0 += res1;
  n = (res0 + (1ull << 61)) >> 62;
  res0 -= n << 62;
  *np = (int)(n & 3ull);
  return (double)(long long)res0 * pi63;
}
/* |x| = n pi/2 + (y0 + y1) for a finite binary32 x with |x| > pi/4; returns n mod 4 */
E3D_LIBM_FN int e3d_libm_reduce(float x, double* y0, double* y1) {
  unsigned xi;
  __builtin_memcpy(&xi, &x, 4);
  xi &= 0x7fffffffu;
  if (xi >= 0x49800000u) {                      /* |x| >= 2^20: fn * pio2_1 would no longer be exact */
    int n;
    *y0 = e3d_libm_reduce_large(xi, &n);
    *y1 = 0.0;
    return n;
  }
  return e3d_libm_rem_pio2(e3d_libm_abs((double)x), y0, y1);
}
E3D_LIBM_FN double e3d_libm_sin(float x) {
  const double xd = (double)x;
  const unsigned ix = e3d_libm_hi(xd) & 0x7fffffffu;
  double y0, y1, r;
  int n;
  if (ix <= 0x3fe921fbu) return e3d_libm_ksin(xd, 0.0, 0);
  if (ix >= 0x7ff00000u) return xd - xd;
  n = e3d_libm_reduce(x, &y0, &y1);
  if (n == 0) r = e3d_libm_ksin(y0, y1, 1);
  else if (n == 1) r = e3d_libm_kcos(y0, y1);
  else if (n == 2) r = -e3d_libm_ksin(y0, y1, 1);
  else r = -e3d_libm_kcos(y0, y1);
  return (xd < 0.0) ? -r : r;
}
E3D_LIBM_FN double e3d_libm_cos(float x) {
  const double xd = (double)x;
  const unsigned ix = e3d_libm_hi(xd) & 0x7fffffffu;
  double y0, y1;
  int n;
  if (ix <= 0x3fe921fbu) return e3d_libm_kcos(xd, 0.0);
  if (ix >= 0x7ff00000u) return xd - xd;
  n = e3d_libm_reduce(x, &y0, &y1);
  if (n == 0) return e3d_libm_kcos(y0, y1);
  if (n == 1) return -e3d_libm_ksin(y0, y1, 1);
  if (n == 2) return -e3d_libm_kcos(y0, y1);
  return e3d_libm_ksin(y0, y1, 1);
}
/* tan = sin / cos of the same reduced argument (both < 1 ulp in binary64: ample for one rounding to binary32) */
E3D_LIBM_FN double e3d_libm_tan(float x) {
  const double xd = (double)x;
  const unsigned ix = e3d_libm_hi(xd) & 0x7fffffffu;
  double y0 = xd, y1 = 0.0, s, c, r;
  int n = 0;
  if (ix >= 0x7ff00000u) return xd - xd;
  if (ix < 0x3e400000u) return xd;              /* |x| < 2^-27 (keeps -0) */
  if (ix <= 0x3fe921fbu) {
    s = e3d_libm_ksin(xd, 0.0, 0);
    c = e3d_libm_kcos(xd, 0.0);
    return s / c;
  }
  n = e3d_libm_reduce(x, &y0, &y1);
  s = e3d_libm_ksin(y0, y1, 1);
  c = e3d_libm_kcos(y0, y1);
  r = (n & 1) ? -c / s : s / c;
  return (xd < 0.0) ? -r : r;
}

/* ---- log2 (binary64) ---- */
E3D_LIBM_FN double e3d_libm_log2(double x) {
  const double ivln2hi = 1.44269504072144627571e+00, ivln2lo = 1.67517131648865118353e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  unsigned hx = e3d_libm_hi(x);
  const unsigned lx = e3d_libm_lo(x);
  int k = 0;
  if ((int)hx < 0x00100000) {                   /* x < 2^-1022 (incl. negative) */
    if (((hx & 0x7fffffffu) | lx) == 0u) return -1.0 / (x * x);   /* log(+-0) = -inf */
    if ((int)hx < 0) return (x - x) / (x - x);                       /* log(-#) = NaN */
    k -= 54; x *= 1.80143985094819840000e+16;                        /* subnormal, scale up */
    hx = e3d_libm_hi(x);
  }
  if (hx >= 0x7ff00000u) return x + x;
  if (hx == 0x3ff00000u && e3d_libm_lo(x) == 0u) return 0.0;         /* log(1) = +0 */
  {
    unsigned i;
    double f, hfsq, s, z, w, t1, t2, r, hi, lo, val_hi, val_lo, y, ww;
    k += (int)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    i = (hx + 0x95f64u) & 0x100000u;
    x = e3d_libm_make(hx | (i ^ 0x3ff00000u), e3d_libm_lo(x));        /* normalize x or x/2 */
    k += (int)(i >> 20);
    y = (double)k;
    f = x - 1.0;
    hfsq = 0.5 * f * f;
    s = f / (2.0 + f); z = s * s; w = z * z;
    t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    r = s * (hfsq + (t2 + t1));
    hi = f - hfsq;
    hi = e3d_libm_make(e3d_libm_hi(hi), 0u);
    lo = (f - hi) - hfsq + r;
    val_hi = hi * ivln2hi;
    val_lo = (lo + hi) * ivln2lo + lo * ivln2hi;
    ww = y + val_hi;
    val_lo += (y - ww) + val_hi;
    val_hi = ww;
    return val_lo + val_hi;
  }
}

/* ---- the binary32 functions the pipeline calls ---- */
E3D_LIBM_FN float e3d_atanf(float x) { return (float)e3d_libm_atan((double)x); }
E3D_LIBM_FN float e3d_atan2f(float y, float x) { return (float)e3d_libm_atan2((double)y, (double)x); }
E3D_LIBM_FN float e3d_sinf(float x) { return (float)e3d_libm_sin(x); }
E3D_LIBM_FN float e3d_cosf(float x) { return (float)e3d_libm_cos(x); }
E3D_LIBM_FN float e3d_tanf(float x) { return (float)e3d_libm_tan(x); }
E3D_LIBM_FN float e3d_log2f(float x) { return (float)e3d_libm_log2((double)x); }

#endif /* E3D_LIBM_H_ */
