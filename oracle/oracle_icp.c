/*
 * oracle/oracle_icp.c -- TEST INFRASTRUCTURE ONLY (CPU oracle, path (A)).
 *
 * Restates, function by function:
 *   FindCorrespondencesFast        src/icp/icp_point_to_plane.cc:42-105
 *   PointToPlaneICP::AddPointCloud src/icp/icp_point_to_plane.cc:109-135
 *   PointToPlaneICP::Run           src/icp/icp_point_to_plane.cc:137-163
 *   PointToPlaneICP::AlignMeshes   src/icp/icp_point_to_plane.cc:169-342
 *   PointToPlaneICPImpl::Accumulate / compute
 *                                  src/icp/icp_point_to_plane_impl.h:82-113,115-293
 * Third-party pieces (PCL/FLANN/Eigen/Sophus arithmetic) are restated in
 * oracle_math.h / oracle_kdtree.c.  Quirks kept on purpose are tagged [QUIRK].
 *
 * Determinism: the reference pushes correspondence sets in OpenMP completion
 * order (icp_point_to_plane.cc:224-244, nondeterministic).  The oracle uses the
 * sequential order of the ik loop: (i,k) for i!=k, and for i==k first
 * (i -> fixed) then (fixed -> i).
 */
#include "e3d_oracle.h"
#include "oracle_kdtree.h"
#include "oracle_math.h"

#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void) {
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct { float* xyz; float* nrm; size_t n; } ocloud;

typedef struct {
  ocloud local;           /* as given by the caller */
  float T[12];            /* global_T_cloud, row-major 3x4 */
  ocloud global;          /* global_frame_point_cloud of the current AlignMeshes */
  float bmin[3], bmax[3];
  int cloud_index;        /* impl index */
} omovable;

typedef struct {
  int src, tgt;           /* impl indices */
  int64_t n;
  int32_t* iq; int32_t* im; float* d;
} ocorr;

struct oracle_icp {
  ocloud fixed; int has_fixed;
  omovable* clouds; int n_clouds, cap_clouds;
  int max_inner; int all_core;
  oracle_icp_pair_record* prec; size_t n_prec, cap_prec;
  oracle_icp_iter_record* irec; size_t n_irec, cap_irec;
};

oracle_icp* oracle_icp_create(void) {
  oracle_icp* o = (oracle_icp*)calloc(1, sizeof(*o));
  o->max_inner = 150;
  return o;
}
static void ocloud_free(ocloud* c) { free(c->xyz); free(c->nrm); c->xyz = c->nrm = NULL; c->n = 0; }
void oracle_icp_destroy(oracle_icp* o) {
  if (!o) return;
  ocloud_free(&o->fixed);
  for (int i = 0; i < o->n_clouds; ++i) { ocloud_free(&o->clouds[i].local); ocloud_free(&o->clouds[i].global); }
  free(o->clouds); free(o->prec); free(o->irec); free(o);
}
void oracle_icp_set_max_inner_iterations(oracle_icp* o, int n) { o->max_inner = n; }
void oracle_icp_set_all_core(oracle_icp* o, int on) { o->all_core = on; }
size_t oracle_icp_num_pair_records(const oracle_icp* o) { return o->n_prec; }
const oracle_icp_pair_record* oracle_icp_pair_records(const oracle_icp* o) { return o->prec; }
size_t oracle_icp_num_iter_records(const oracle_icp* o) { return o->n_irec; }
const oracle_icp_iter_record* oracle_icp_iter_records(const oracle_icp* o) { return o->irec; }

void oracle_transform_cloud(const float* xyz, const float* nrm, size_t n, const float T[12],
                            float* out_xyz, float* out_nrm, float bmin[3], float bmax[3]) {
  /* Eigen::AlignedBox::setEmpty: min = +max, max = -max (lowest) */
  for (int d = 0; d < 3; ++d) { bmin[d] = 3.402823466e+38f; bmax[d] = -3.402823466e+38f; }
  for (size_t i = 0; i < n; ++i) {
    om_pcl_se3(T, xyz + 3 * i, out_xyz + 3 * i);
    om_pcl_so3(T, nrm + 3 * i, out_nrm + 3 * i);
    for (int d = 0; d < 3; ++d) {
      float v = out_xyz[3 * i + d];
      if (v < bmin[d]) bmin[d] = v;
      if (v > bmax[d]) bmax[d] = v;
    }
  }
}

int oracle_icp_add_cloud(oracle_icp* o, const float* xyz, const float* nrm, size_t n,
                         const float T[12], int fixed) {
  if (fixed) {
    /* transform to the global frame once and concatenate (cc:110-127) */
    size_t n0 = o->fixed.n;
    o->fixed.xyz = (float*)realloc(o->fixed.xyz, sizeof(float) * 3 * (n0 + n + 1));
    o->fixed.nrm = (float*)realloc(o->fixed.nrm, sizeof(float) * 3 * (n0 + n + 1));
    float bmin[3], bmax[3];
    oracle_transform_cloud(xyz, nrm, n, T, o->fixed.xyz + 3 * n0, o->fixed.nrm + 3 * n0, bmin, bmax);
    o->fixed.n = n0 + n;
    o->has_fixed = 1;
    return -1;
  }
  if (o->n_clouds == o->cap_clouds) {
    o->cap_clouds = o->cap_clouds ? 2 * o->cap_clouds : 8;
    o->clouds = (omovable*)realloc(o->clouds, sizeof(omovable) * (size_t)o->cap_clouds);
  }
  omovable* c = &o->clouds[o->n_clouds];
  memset(c, 0, sizeof(*c));
  c->local.n = n;
  c->local.xyz = (float*)malloc(sizeof(float) * 3 * (n + 1));
  c->local.nrm = (float*)malloc(sizeof(float) * 3 * (n + 1));
  memcpy(c->local.xyz, xyz, sizeof(float) * 3 * n);
  memcpy(c->local.nrm, nrm, sizeof(float) * 3 * n);
  memcpy(c->T, T, sizeof(float) * 12);
  c->global.n = n;
  c->global.xyz = (float*)malloc(sizeof(float) * 3 * (n + 1));
  c->global.nrm = (float*)malloc(sizeof(float) * 3 * (n + 1));
  return o->n_clouds++;
}

int oracle_icp_get_pose(oracle_icp* o, int idx, float T[12]) {
  if (idx < 0 || idx >= o->n_clouds) return -1;   /* reference: clouds_.at() throws */
  memcpy(T, o->clouds[idx].T, sizeof(float) * 12);
  return 0;
}

/* ---- FindCorrespondencesFast ------------------------------------------------ */
static inline float radius_sq(float d) {
  /* pcl::KdTreeFLANN::radiusSearch: static_cast<float>(radius * radius), radius double */
  double r = (double)d;
  return (float)(r * r);
}

static int64_t find_corr_tree(const okd_tree* tree, const float* src, size_t n_src, float r2,
                              int32_t* iq, int32_t* im, float* sd, int parallel) {
  if (!parallel) {
    int64_t cnt = 0;
    for (size_t i = 0; i < n_src; ++i) {
      int32_t idx; float dist;
      if (!okd_nearest_within(tree, src + 3 * i, r2, &idx, &dist)) continue;
      iq[cnt] = (int32_t)i; im[cnt] = idx; sd[cnt] = dist; ++cnt;
    }
    return cnt;
  }
  /* all-core variant: per-query results, then an order-preserving compaction */
  int32_t* tmp_idx = (int32_t*)malloc(sizeof(int32_t) * (n_src + 1));
  float* tmp_d = (float*)malloc(sizeof(float) * (n_src + 1));
#pragma omp parallel for schedule(dynamic, 4096)
  for (long long i = 0; i < (long long)n_src; ++i) {
    int32_t idx; float dist;
    if (okd_nearest_within(tree, src + 3 * (size_t)i, r2, &idx, &dist)) { tmp_idx[i] = idx; tmp_d[i] = dist; }
    else tmp_idx[i] = -1;
  }
  int64_t cnt = 0;
  for (size_t i = 0; i < n_src; ++i)
    if (tmp_idx[i] >= 0) { iq[cnt] = (int32_t)i; im[cnt] = tmp_idx[i]; sd[cnt] = tmp_d[i]; ++cnt; }
  free(tmp_idx); free(tmp_d);
  return cnt;
}

int64_t oracle_find_correspondences(const float* src, size_t n_src, const float* tgt, size_t n_tgt,
                                    float d, int32_t* iq, int32_t* im, float* sd) {
  okd_tree* tree = okd_build(tgt, n_tgt);
  int64_t c = find_corr_tree(tree, src, n_src, radius_sq(d), iq, im, sd, 0);
  okd_free(tree);
  return c;
}

int64_t oracle_find_correspondences_brute(const float* src, size_t n_src, const float* tgt,
                                          size_t n_tgt, float d, int32_t* iq, int32_t* im, float* sd) {
  float r2 = radius_sq(d);
  int64_t cnt = 0;
  for (size_t i = 0; i < n_src; ++i) {
    int found = 0; float bd = 0.f; int32_t bi = 0;
    for (size_t j = 0; j < n_tgt; ++j) {
      float dist = om_sqdist3f(src + 3 * i, tgt + 3 * j);
      if (!found) { if (dist < r2) { found = 1; bd = dist; bi = (int32_t)j; } }
      else if (dist < bd) { bd = dist; bi = (int32_t)j; }   /* ascending j => lowest index on ties */
    }
    if (found) { iq[cnt] = (int32_t)i; im[cnt] = bi; sd[cnt] = bd; ++cnt; }
  }
  return cnt;
}

/* ---- PointToPlaneICPImpl ------------------------------------------------------ */
typedef struct { const ocloud* cloud; om_se3f pose; } icloud;

/* f32 residual/Jacobian rows of one correspondence, literally as
 * icp_point_to_plane_impl.h:144-204 (left-to-right evaluation). */
static inline void corr_rows(const float* sp, const float* sn, const float* tp, const float* tn,
                             float* r1, float* j1_t, float* j1_s,
                             float* r2, float* j2_t, float* j2_s) {
  float d[3] = {tp[0] - sp[0], tp[1] - sp[1], tp[2] - sp[2]};
  *r1 = om_dot3f(sn, d);
  j1_t[0] = sn[0]; j1_t[1] = sn[1]; j1_t[2] = sn[2];
  j1_t[3] = -sn[1] * tp[2] + sn[2] * tp[1];
  j1_t[4] = sn[0] * tp[2] - sn[2] * tp[0];
  j1_t[5] = -sn[0] * tp[1] + sn[1] * tp[0];
  j1_s[0] = -sn[0]; j1_s[1] = -sn[1]; j1_s[2] = -sn[2];
  j1_s[3] = sn[1] * sp[2] - sn[1] * (sp[2] - tp[2]) - sn[2] * sp[1] + sn[2] * (sp[1] - tp[1]);
  j1_s[4] = -sn[0] * sp[2] + sn[0] * (sp[2] - tp[2]) + sn[2] * sp[0] - sn[2] * (sp[0] - tp[0]);
  j1_s[5] = sn[0] * sp[1] - sn[0] * (sp[1] - tp[1]) - sn[1] * sp[0] + sn[1] * (sp[0] - tp[0]);
  float e[3] = {sp[0] - tp[0], sp[1] - tp[1], sp[2] - tp[2]};
  *r2 = om_dot3f(tn, e);
  j2_t[0] = -tn[0]; j2_t[1] = -tn[1]; j2_t[2] = -tn[2];
  j2_t[3] = tn[1] * tp[2] - tn[1] * (tp[2] - sp[2]) - tn[2] * tp[1] + tn[2] * (tp[1] - sp[1]);
  j2_t[4] = -tn[0] * tp[2] + tn[0] * (tp[2] - sp[2]) + tn[2] * tp[0] - tn[2] * (tp[0] - sp[0]);
  j2_t[5] = tn[0] * tp[1] - tn[0] * (tp[1] - sp[1]) - tn[1] * tp[0] + tn[1] * (tp[0] - sp[0]);
  j2_s[0] = tn[0]; j2_s[1] = tn[1]; j2_s[2] = tn[2];
  j2_s[3] = -tn[1] * sp[2] + tn[2] * sp[1];
  j2_s[4] = tn[0] * sp[2] - tn[2] * sp[0];
  j2_s[5] = -tn[0] * sp[1] + tn[1] * sp[0];
}

/* Accumulate (impl.h:82-113), weight == 1. */
static inline void accumulate(double residual, int si, const float* js, int ti, const float* jt,
                              double* H, double* b, int nv) {
  double Js[6], Jt[6];
  for (int i = 0; i < 6; ++i) { Js[i] = (double)js[i]; Jt[i] = (double)jt[i]; }
  if (si >= 0) {
    for (int i = 0; i < 6; ++i) {
      for (int j = i; j < 6; ++j) H[(size_t)(si + i) * nv + (si + j)] += Js[i] * Js[j];
      b[si + i] += residual * Js[i];
    }
    if (ti >= 0) {
      /* [QUIRK] written at (src,tgt) even when that is the lower triangle, which the
       * solver (selfadjointView<Upper>) never reads. */
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) H[(size_t)(si + i) * nv + (ti + j)] += Js[i] * Jt[j];
    }
  }
  if (ti >= 0) {
    for (int i = 0; i < 6; ++i) {
      for (int j = i; j < 6; ++j) H[(size_t)(ti + i) * nv + (ti + j)] += Jt[i] * Jt[j];
      b[ti + i] += residual * Jt[i];
    }
  }
}

static double cost_pass(const icloud* cl, const ocorr* cs, int n_cs) {
  double cost = 0.0;
  for (int s = 0; s < n_cs; ++s) {
    const icloud* sc = &cl[cs[s].src]; const icloud* tc = &cl[cs[s].tgt];
    float Rs[9], Rt[9];
    om_quat_to_R_f(sc->pose.q, Rs); om_quat_to_R_f(tc->pose.q, Rt);
    for (int64_t c = 0; c < cs[s].n; ++c) {
      float sp[3], sn[3], tp[3], tn[3];
      om_rot_trans_f(Rs, sc->pose.t, sc->cloud->xyz + 3 * (size_t)cs[s].iq[c], sp);
      om_rot_f(Rs, sc->cloud->nrm + 3 * (size_t)cs[s].iq[c], sn);
      om_rot_trans_f(Rt, tc->pose.t, tc->cloud->xyz + 3 * (size_t)cs[s].im[c], tp);
      om_rot_f(Rt, tc->cloud->nrm + 3 * (size_t)cs[s].im[c], tn);
      float d[3] = {tp[0] - sp[0], tp[1] - sp[1], tp[2] - sp[2]};
      float r1 = om_dot3f(sn, d);
      cost += r1 * r1;
      float e[3] = {sp[0] - tp[0], sp[1] - tp[1], sp[2] - tp[2]};
      float r2 = om_dot3f(tn, e);
      cost += r2 * r2;
    }
  }
  return cost;
}

/* ---- all-core variants of the two passes ("queries AND reductions parallelised"; bench.py's
 * cpu_baseline.all_core only -- the oracle proper keeps the reference's sequential sums).  Every thread sums a contiguous share of
 * each correspondence set into its own H / b / cost, the partials are added in thread order: the same terms in another order of
 * f64 additions (results agree with the sequential pass to rounding, not bit for bit). */
static double cost_pass_par(const icloud* cl, const ocorr* cs, int n_cs) {
  double cost = 0.0;
  for (int s = 0; s < n_cs; ++s) {
    const icloud* sc = &cl[cs[s].src]; const icloud* tc = &cl[cs[s].tgt];
    float Rs[9], Rt[9];
    om_quat_to_R_f(sc->pose.q, Rs); om_quat_to_R_f(tc->pose.q, Rt);
    double set_cost = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : set_cost)
    for (int64_t c = 0; c < cs[s].n; ++c) {
      float sp[3], sn[3], tp[3], tn[3];
      om_rot_trans_f(Rs, sc->pose.t, sc->cloud->xyz + 3 * (size_t)cs[s].iq[c], sp);
      om_rot_f(Rs, sc->cloud->nrm + 3 * (size_t)cs[s].iq[c], sn);
      om_rot_trans_f(Rt, tc->pose.t, tc->cloud->xyz + 3 * (size_t)cs[s].im[c], tp);
      om_rot_f(Rt, tc->cloud->nrm + 3 * (size_t)cs[s].im[c], tn);
      float d[3] = {tp[0] - sp[0], tp[1] - sp[1], tp[2] - sp[2]};
      float r1 = om_dot3f(sn, d);
      float e[3] = {sp[0] - tp[0], sp[1] - tp[1], sp[2] - tp[2]};
      float r2 = om_dot3f(tn, e);
      set_cost += (double)(r1 * r1) + (double)(r2 * r2);
    }
    cost += set_cost;
  }
  return cost;
}

static double accumulate_pass_par(const icloud* cl, const ocorr* cs, int n_cs, double* H, double* b, int nv) {
  int nt = omp_get_max_threads();
  size_t stride = (size_t)nv * nv + (size_t)nv + 1;
  double* part = (double*)calloc((size_t)nt * stride, sizeof(double));
#pragma omp parallel num_threads(nt)
  {
    int tid = omp_get_thread_num();
    double* Hp = part + (size_t)tid * stride; double* bp = Hp + (size_t)nv * nv; double* cp = bp + nv;
    for (int s = 0; s < n_cs; ++s) {
      int si = 6 * (cs[s].src - 1), ti = 6 * (cs[s].tgt - 1);
      const icloud* sc = &cl[cs[s].src]; const icloud* tc = &cl[cs[s].tgt];
      float Rs[9], Rt[9];
      om_quat_to_R_f(sc->pose.q, Rs); om_quat_to_R_f(tc->pose.q, Rt);
#pragma omp for schedule(static) nowait
      for (int64_t c = 0; c < cs[s].n; ++c) {
        float sp[3], sn[3], tp[3], tn[3];
        om_rot_trans_f(Rs, sc->pose.t, sc->cloud->xyz + 3 * (size_t)cs[s].iq[c], sp);
        om_rot_f(Rs, sc->cloud->nrm + 3 * (size_t)cs[s].iq[c], sn);
        om_rot_trans_f(Rt, tc->pose.t, tc->cloud->xyz + 3 * (size_t)cs[s].im[c], tp);
        om_rot_f(Rt, tc->cloud->nrm + 3 * (size_t)cs[s].im[c], tn);
        float r1, r2, j1t[6], j1s[6], j2t[6], j2s[6];
        corr_rows(sp, sn, tp, tn, &r1, j1t, j1s, &r2, j2t, j2s);
        *cp += r1 * r1;
        accumulate((double)r1, si, j1s, ti, j1t, Hp, bp, nv);
        *cp += r2 * r2;
        accumulate((double)r2, si, j2s, ti, j2t, Hp, bp, nv);
      }
    }
  }
  double cost = 0.0;
  for (int t = 0; t < nt; ++t) {
    const double* Hp = part + (size_t)t * stride; const double* bp = Hp + (size_t)nv * nv;
    for (size_t i = 0; i < (size_t)nv * nv; ++i) H[i] += Hp[i];
    for (int i = 0; i < nv; ++i) b[i] += bp[i];
    cost += bp[nv];
  }
  free(part);
  return cost;
}

static void impl_compute(icloud* cl, int n_cl, const ocorr* cs, int n_cs, int max_it,
                         oracle_icp_iter_record* rec, int all_core) {
  int nv = 6 * (n_cl - 1);
  double* H = (double*)malloc(sizeof(double) * (size_t)(nv > 0 ? nv * nv : 1));
  double* Hl = (double*)malloc(sizeof(double) * (size_t)(nv > 0 ? nv * nv : 1));
  double* W = (double*)malloc(sizeof(double) * (size_t)(nv > 0 ? nv * nv : 1));
  double* b = (double*)malloc(sizeof(double) * (size_t)(nv > 0 ? nv : 1));
  double* x = (double*)malloc(sizeof(double) * (size_t)(nv > 0 ? nv : 1));
  int* perm = (int*)malloc(sizeof(int) * (size_t)(nv > 0 ? nv : 1));
  icloud* upd = (icloud*)malloc(sizeof(icloud) * (size_t)n_cl);
  double lambda = 0.1;
  for (int it = 0; it < max_it; ++it) {
    memset(H, 0, sizeof(double) * (size_t)nv * nv);
    memset(b, 0, sizeof(double) * (size_t)nv);
    double cost = 0.0;
    if (all_core) cost = accumulate_pass_par(cl, cs, n_cs, H, b, nv);
    for (int s = 0; s < n_cs && !all_core; ++s) {
      int si = 6 * (cs[s].src - 1), ti = 6 * (cs[s].tgt - 1);
      const icloud* sc = &cl[cs[s].src]; const icloud* tc = &cl[cs[s].tgt];
      float Rs[9], Rt[9];
      om_quat_to_R_f(sc->pose.q, Rs); om_quat_to_R_f(tc->pose.q, Rt);
      for (int64_t c = 0; c < cs[s].n; ++c) {
        float sp[3], sn[3], tp[3], tn[3];
        om_rot_trans_f(Rs, sc->pose.t, sc->cloud->xyz + 3 * (size_t)cs[s].iq[c], sp);
        om_rot_f(Rs, sc->cloud->nrm + 3 * (size_t)cs[s].iq[c], sn);
        om_rot_trans_f(Rt, tc->pose.t, tc->cloud->xyz + 3 * (size_t)cs[s].im[c], tp);
        om_rot_f(Rt, tc->cloud->nrm + 3 * (size_t)cs[s].im[c], tn);
        float r1, r2, j1t[6], j1s[6], j2t[6], j2s[6];
        corr_rows(sp, sn, tp, tn, &r1, j1t, j1s, &r2, j2t, j2s);
        cost += r1 * r1;
        accumulate((double)r1, si, j1s, ti, j1t, H, b, nv);
        cost += r2 * r2;
        accumulate((double)r2, si, j2s, ti, j2t, H, b, nv);
      }
    }
    rec->accumulate_passes++;
    rec->inner_iterations++;
    if (it == 0) rec->initial_cost = cost;
    rec->final_cost = cost;

    int applied = 0;
    for (int lm = 0; lm < 10; ++lm) {
      memcpy(Hl, H, sizeof(double) * (size_t)nv * nv);
      for (int i = 0; i < nv; ++i) Hl[(size_t)i * nv + i] += lambda;     /* additive damping (impl.h:223) */
      memcpy(x, b, sizeof(double) * (size_t)nv);
      om_ldlt_solve_upper(Hl, nv, x, W, perm);
      upd[0] = cl[0];
      for (int ci = 1; ci < n_cl; ++ci) {
        double mx[6]; for (int i = 0; i < 6; ++i) mx[i] = -x[6 * (ci - 1) + i];
        om_se3d e; om_se3d_exp(mx, &e);
        om_se3f ef; om_se3_cast_f(&e, &ef);
        upd[ci].cloud = cl[ci].cloud;
        om_se3f_mul(&ef, &cl[ci].pose, &upd[ci].pose);
      }
      double new_cost = all_core ? cost_pass_par(upd, cs, n_cs) : cost_pass(upd, cs, n_cs);
      rec->cost_passes++;
      if (new_cost < cost) {
        memcpy(cl, upd, sizeof(icloud) * (size_t)n_cl);
        lambda = 0.5f * lambda;
        applied = 1;
        rec->final_cost = new_cost;
        break;
      } else {
        lambda = 2.f * lambda;
      }
    }
    if (!applied) break;
  }
  free(H); free(Hl); free(W); free(b); free(x); free(perm); free(upd);
}

/* ---- AlignMeshes --------------------------------------------------------------- */
static int bbox_intersects(const float* amin, const float* amax, const float* bmin, const float* bmax) {
  /* !a.intersection(b).isEmpty():  isEmpty = (min > max).any() */
  for (int d = 0; d < 3; ++d) {
    float lo = amin[d] > bmin[d] ? amin[d] : bmin[d];
    float hi = amax[d] < bmax[d] ? amax[d] : bmax[d];
    if (lo > hi) return 0;
  }
  return 1;
}

static void push_prec(oracle_icp* o, int it, int src, int tgt, int64_t count, float dsum) {
  if (o->n_prec == o->cap_prec) {
    o->cap_prec = o->cap_prec ? 2 * o->cap_prec : 64;
    o->prec = (oracle_icp_pair_record*)realloc(o->prec, sizeof(*o->prec) * o->cap_prec);
  }
  oracle_icp_pair_record r = {it, src, tgt, count, dsum};
  o->prec[o->n_prec++] = r;
}

typedef struct { ocorr c[2]; int n; } pair_slot;

static void make_corr(ocorr* c, int src, int tgt, const ocloud* s, const okd_tree* tree, float r2, int par) {
  c->src = src; c->tgt = tgt;
  c->iq = (int32_t*)malloc(sizeof(int32_t) * (s->n + 1));
  c->im = (int32_t*)malloc(sizeof(int32_t) * (s->n + 1));
  c->d = (float*)malloc(sizeof(float) * (s->n + 1));
  c->n = find_corr_tree(tree, s->xyz, s->n, r2, c->iq, c->im, c->d, par);
}

static int align_meshes(oracle_icp* o, float max_d, float thr, int print, int iteration) {
  oracle_icp_iter_record rec; memset(&rec, 0, sizeof rec); rec.iteration = iteration;
  double t0 = now_s();
  int M = o->n_clouds;
  int fixed_vertex = -1; int n_impl = 0;
  float fmin[3], fmax[3];
  icloud* icl = (icloud*)calloc((size_t)M + 1, sizeof(icloud));
  if (o->has_fixed) {
    fixed_vertex = n_impl;
    icl[n_impl].cloud = &o->fixed; om_se3f_identity(&icl[n_impl].pose); ++n_impl;
    for (int d = 0; d < 3; ++d) { fmin[d] = 3.402823466e+38f; fmax[d] = -3.402823466e+38f; }
    for (size_t i = 0; i < o->fixed.n; ++i)
      for (int d = 0; d < 3; ++d) {
        float v = o->fixed.xyz[3 * i + d];
        if (v < fmin[d]) fmin[d] = v;
        if (v > fmax[d]) fmax[d] = v;
      }
  }
  for (int i = 0; i < M; ++i) {
    omovable* c = &o->clouds[i];
    oracle_transform_cloud(c->local.xyz, c->local.nrm, c->local.n, c->T, c->global.xyz, c->global.nrm, c->bmin, c->bmax);
    c->cloud_index = n_impl;
    icl[n_impl].cloud = &c->global; om_se3f_identity(&icl[n_impl].pose); ++n_impl;
  }
  double t1 = now_s();
  rec.t_transform_s = t1 - t0;

  float r2 = radius_sq(max_d);
  /* The reference rebuilds a kd-tree per directed pair (cc:46-51).  Trees only
   * depend on the target cloud, so the oracle builds one per target that is used
   * at all -- results are identical. */
  okd_tree** trees = (okd_tree**)calloc((size_t)M + 1, sizeof(okd_tree*));   /* [M] = fixed */
  char* need = (char*)calloc((size_t)M + 1, 1);
  for (int ik = 0; ik < M * M; ++ik) {
    int i = ik / M, k = ik % M;
    if (i != k && bbox_intersects(o->clouds[i].bmin, o->clouds[i].bmax, o->clouds[k].bmin, o->clouds[k].bmax)) need[k] = 1;
    if (i == k && o->has_fixed && bbox_intersects(fmin, fmax, o->clouds[i].bmin, o->clouds[i].bmax)) { need[M] = 1; need[i] = 1; }
  }
  if (o->all_core) {
    for (int k = 0; k <= M; ++k) {                         /* one tree after the other, every core on each */
      if (!need[k]) continue;
      const ocloud* c = (k == M) ? &o->fixed : &o->clouds[k].global;
      trees[k] = okd_build_parallel(c->xyz, c->n);
    }
  }
#pragma omp parallel for schedule(dynamic, 1)
  for (int k = 0; k <= M; ++k) {
    if (!need[k] || trees[k]) continue;
    const ocloud* c = (k == M) ? &o->fixed : &o->clouds[k].global;
    trees[k] = okd_build(c->xyz, c->n);
  }
  pair_slot* slots = (pair_slot*)calloc((size_t)M * M + 1, sizeof(pair_slot));
  int par_q = o->all_core;
  if (par_q) {
    for (int ik = 0; ik < M * M; ++ik) {
      int i = ik / M, k = ik % M;
      if (i != k && need[k] && bbox_intersects(o->clouds[i].bmin, o->clouds[i].bmax, o->clouds[k].bmin, o->clouds[k].bmax)) {
        make_corr(&slots[ik].c[0], o->clouds[i].cloud_index, o->clouds[k].cloud_index, &o->clouds[i].global, trees[k], r2, 1);
        slots[ik].n = 1;
      }
      if (i == k && o->has_fixed && bbox_intersects(fmin, fmax, o->clouds[i].bmin, o->clouds[i].bmax)) {
        make_corr(&slots[ik].c[0], o->clouds[i].cloud_index, fixed_vertex, &o->clouds[i].global, trees[M], r2, 1);
        make_corr(&slots[ik].c[1], fixed_vertex, o->clouds[i].cloud_index, &o->fixed, trees[i], r2, 1);
        slots[ik].n = 2;
      }
    }
  } else {
    /* reference-faithful threading: parallel over ik only (cc:208) */
#pragma omp parallel for schedule(dynamic, 1)
    for (int ik = 0; ik < M * M; ++ik) {
      int i = ik / M, k = ik % M;
      if (i != k && bbox_intersects(o->clouds[i].bmin, o->clouds[i].bmax, o->clouds[k].bmin, o->clouds[k].bmax)) {
        make_corr(&slots[ik].c[0], o->clouds[i].cloud_index, o->clouds[k].cloud_index, &o->clouds[i].global, trees[k], r2, 0);
        slots[ik].n = 1;
      }
      if (i == k && o->has_fixed && bbox_intersects(fmin, fmax, o->clouds[i].bmin, o->clouds[i].bmax)) {
        make_corr(&slots[ik].c[0], o->clouds[i].cloud_index, fixed_vertex, &o->clouds[i].global, trees[M], r2, 0);
        make_corr(&slots[ik].c[1], fixed_vertex, o->clouds[i].cloud_index, &o->fixed, trees[i], r2, 0);
        slots[ik].n = 2;
      }
    }
  }
  for (int k = 0; k <= M; ++k) okd_free(trees[k]);
  free(trees); free(need);

  /* gather in canonical order, print, keep non-empty sets (cc:224-244) */
  ocorr* cs = (ocorr*)calloc((size_t)2 * M * M + 1, sizeof(ocorr)); int n_cs = 0;
  for (int ik = 0; ik < M * M; ++ik) {
    for (int s = 0; s < slots[ik].n; ++s) {
      ocorr* c = &slots[ik].c[s];
      float dsum = 0.f;
      for (int64_t q = 0; q < c->n; ++q) dsum += c->d[q];
      int psrc = (c->src == fixed_vertex) ? -1 : c->src;
      int ptgt = (c->tgt == fixed_vertex) ? -1 : c->tgt;
      push_prec(o, iteration, psrc, ptgt, c->n, dsum);
      if (print) {
        char avg[64] = "";
        if (c->n > 0) snprintf(avg, sizeof avg, " (avg. distance: %g)", (double)(dsum / (float)c->n));
        if (psrc >= 0 && ptgt >= 0) printf("  found correspondences from %d to %d: %lld%s\n", c->src, c->tgt, (long long)c->n, avg);
        else if (ptgt < 0) printf("  found correspondences from %d to fixed clouds: %lld%s\n", c->src, (long long)c->n, avg);
        else printf("  found correspondences from fixed clouds to %d: %lld%s\n", c->tgt, (long long)c->n, avg);
      }
      rec.correspondences += c->n;
      if (c->n > 0) cs[n_cs++] = *c;
      else { free(c->iq); free(c->im); free(c->d); }
    }
  }
  free(slots);
  double t2 = now_s();
  rec.t_nn_s = t2 - t1;

  impl_compute(icl, n_impl, cs, n_cs, o->max_inner, &rec, o->all_core);
  rec.t_lm_s = now_s() - t2;

  /* pose write-back (cc:318-341) */
  int converged = 1;
  for (int i = 0; i < M; ++i) {
    omovable* c = &o->clouds[i];
    const om_se3f* p = &icl[c->cloud_index].pose;
    float R[9]; om_quat_to_R_f(p->q, R);
    float Tn[12];
    for (int r = 0; r < 3; ++r) {
      for (int col = 0; col < 3; ++col) {
        float col_v[3] = {c->T[0 * 4 + col], c->T[1 * 4 + col], c->T[2 * 4 + col]};
        Tn[4 * r + col] = om_dot3f(R + 3 * r, col_v);
      }
      float tv[3] = {c->T[3], c->T[7], c->T[11]};
      Tn[4 * r + 3] = om_dot3f(R + 3 * r, tv) + p->t[r];
    }
    float dx = c->T[3] - Tn[3], dy = c->T[7] - Tn[7], dz = c->T[11] - Tn[11];
    float movement = sqrtf(dx * dx + (dy * dy + dz * dz));
    if (movement > thr) converged = 0;
    if (print) printf("  %d moved by %g\n", c->cloud_index, (double)movement);
    memcpy(c->T, Tn, sizeof Tn);
  }
  for (int s = 0; s < n_cs; ++s) { free(cs[s].iq); free(cs[s].im); free(cs[s].d); }
  free(cs); free(icl);

  if (o->n_irec == o->cap_irec) {
    o->cap_irec = o->cap_irec ? 2 * o->cap_irec : 64;
    o->irec = (oracle_icp_iter_record*)realloc(o->irec, sizeof(*o->irec) * o->cap_irec);
  }
  o->irec[o->n_irec++] = rec;
  return converged;
}

int oracle_icp_run(oracle_icp* o, float max_d, int initial_iteration, int max_num_iterations,
                   float thr, int print) {
  if (o->n_clouds == 0) return -1;   /* reference: CHECK(!clouds_.empty()) aborts */
  for (int i = initial_iteration; i < initial_iteration + max_num_iterations; ++i) {
    if (print) printf("-- Alignment iteration %d --\n", i);
    int converged = align_meshes(o, max_d, thr, print, i);
    if (converged) {
      if (print) printf("Convergence is assumed as the maximum movement is less than the threshold.\n");
      if (print) fflush(stdout);
      return 1;
    }
  }
  if (print) fflush(stdout);
  return 0;
}

/* ---- stand-alone pieces for unit parity tests ---------------------------------- */
void oracle_icp_pair_system(const float* sxyz, const float* snrm, const float* txyz, const float* tnrm,
                            const int32_t* iq, const int32_t* im, int64_t n,
                            const float sq[4], const float st[3], const float tq[4], const float tt[3],
                            double H[144], double b[12], double* cost) {
  float Rs[9], Rt[9];
  om_quat_to_R_f(sq, Rs); om_quat_to_R_f(tq, Rt);
  memset(H, 0, sizeof(double) * 144); memset(b, 0, sizeof(double) * 12);
  double c = 0.0;
  for (int64_t q = 0; q < n; ++q) {
    float sp[3], sn[3], tp[3], tn[3];
    om_rot_trans_f(Rs, st, sxyz + 3 * (size_t)iq[q], sp);
    om_rot_f(Rs, snrm + 3 * (size_t)iq[q], sn);
    om_rot_trans_f(Rt, tt, txyz + 3 * (size_t)im[q], tp);
    om_rot_f(Rt, tnrm + 3 * (size_t)im[q], tn);
    float r1, r2, j1t[6], j1s[6], j2t[6], j2s[6];
    corr_rows(sp, sn, tp, tn, &r1, j1t, j1s, &r2, j2t, j2s);
    c += r1 * r1; c += r2 * r2;
    double J1[12], J2[12];
    for (int i = 0; i < 6; ++i) { J1[i] = j1s[i]; J1[6 + i] = j1t[i]; J2[i] = j2s[i]; J2[6 + i] = j2t[i]; }
    for (int i = 0; i < 12; ++i) {
      for (int j = 0; j < 12; ++j) H[12 * i + j] += J1[i] * J1[j];
      b[i] += (double)r1 * J1[i];
    }
    for (int i = 0; i < 12; ++i) {
      for (int j = 0; j < 12; ++j) H[12 * i + j] += J2[i] * J2[j];
      b[i] += (double)r2 * J2[i];
    }
  }
  *cost = c;
}

void oracle_se3_update(const double x[6], const float q_in[4], const float t_in[3], float q_out[4], float t_out[3]) {
  double mx[6]; for (int i = 0; i < 6; ++i) mx[i] = -x[i];
  om_se3d e; om_se3d_exp(mx, &e);
  om_se3f ef; om_se3_cast_f(&e, &ef);
  om_se3f in, out;
  memcpy(in.q, q_in, sizeof in.q); memcpy(in.t, t_in, sizeof in.t);
  om_se3f_mul(&ef, &in, &out);
  memcpy(q_out, out.q, sizeof out.q); memcpy(t_out, out.t, sizeof out.t);
}

void oracle_quat_to_R(const float q[4], float R[9]) { om_quat_to_R_f(q, R); }

void oracle_ldlt_solve_upper(const double* A, int n, const double* b, double* x) {
  double* W = (double*)malloc(sizeof(double) * (size_t)n * n);
  int* perm = (int*)malloc(sizeof(int) * (size_t)n);
  memcpy(x, b, sizeof(double) * (size_t)n);
  om_ldlt_solve_upper(A, n, x, W, perm);
  free(W); free(perm);
}
