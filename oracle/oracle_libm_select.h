/* oracle_libm_select.h -- which elementary functions the CPU oracle calls (TEST INFRASTRUCTURE ONLY).
 *
 * Default: include/e3d_libm.h, the bit-defined functions the HIP kernels evaluate too (oracle == kernels bit for bit).
 * -DE3D_ORACLE_GLIBC (second build, libe3d_oracle_glibc.so): the C library's own atanf / atan2f / sinf / cosf / tanf / log2f --
 * what the reference calls (pcl::eigen33 in two_pass_normal_3d.h:92-109, camera_base_impl_fisheye.h:66-153,
 * visibility_estimator.cc:437).  tests/test_oracle_glibc.py runs both builds on the same scenes and bounds what the
 * substitution changes end to end (correspondence counts, observation lists, final poses); DESIGN.md section 8 quotes it.
 * Include AFTER <math.h> and e3d_libm.h. */
#ifndef E3D_ORACLE_LIBM_SELECT_H
#define E3D_ORACLE_LIBM_SELECT_H
#ifdef E3D_ORACLE_GLIBC
#define om_atanf(x) atanf(x)
#define om_atan2f(y, x) atan2f(y, x)
#define om_sinf(x) sinf(x)
#define om_cosf(x) cosf(x)
#define om_tanf(x) tanf(x)
#define om_log2f(x) log2f(x)
#else
#define om_atanf(x) e3d_atanf(x)
#define om_atan2f(y, x) e3d_atan2f(y, x)
#define om_sinf(x) e3d_sinf(x)
#define om_cosf(x) e3d_cosf(x)
#define om_tanf(x) e3d_tanf(x)
#define om_log2f(x) e3d_log2f(x)
#endif
#endif
