/*
 * oracle/oracle_kdtree.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 * See oracle_kdtree.h for what this restates and which result semantics it pins.
 *
 * Pruning is done with a per-node bounding box and a lower bound that is
 * provably <= the f32 L2_Simple distance of every point inside the box
 * (each per-axis term is <= the point's term and f32 addition is monotone), so
 * the search is exact with respect to the f32-computed distances, including
 * (dist, index) tie-breaking: sub-trees are only skipped when bound > worst.
 */
#include "oracle_kdtree.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define OKD_LEAF 15

typedef struct {
  float lo[3], hi[3];
  int32_t left, right;   /* -1 for leaves */
  int32_t begin, end;    /* range in the reordered point array */
} okd_node;

struct okd_tree {
  const float* xyz;
  size_t n;
  int32_t* perm;    /* reordered position -> original index */
  float* pts;       /* reordered xyz */
  okd_node* nodes;
  size_t n_nodes, cap_nodes;
};

static int32_t okd_new_node(okd_tree* t) {
  if (t->n_nodes == t->cap_nodes) {
    t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
    t->nodes = (okd_node*)realloc(t->nodes, t->cap_nodes * sizeof(okd_node));
  }
  return (int32_t)t->n_nodes++;
}

/* quickselect on perm[b..e) by coordinate dim so that perm[m] is the median */
static void okd_select(const float* xyz, int32_t* perm, int32_t b, int32_t e,
                       int32_t m, int dim) {
  while (e - b > 1) {
    /* median-of-three pivot */
    int32_t mid = b + (e - b) / 2;
    float a0 = xyz[3 * (size_t)perm[b] + dim];
    float a1 = xyz[3 * (size_t)perm[mid] + dim];
    float a2 = xyz[3 * (size_t)perm[e - 1] + dim];
    float pv = a0;
    if ((a0 <= a1 && a1 <= a2) || (a2 <= a1 && a1 <= a0)) pv = a1;
    else if ((a0 <= a2 && a2 <= a1) || (a1 <= a2 && a2 <= a0)) pv = a2;
    /* three-way partition */
    int32_t lt = b, i = b, gt = e;
    while (i < gt) {
      float v = xyz[3 * (size_t)perm[i] + dim];
      if (v < pv) { int32_t t = perm[lt]; perm[lt] = perm[i]; perm[i] = t; ++lt; ++i; }
      else if (v > pv) { --gt; int32_t t = perm[gt]; perm[gt] = perm[i]; perm[i] = t; }
      else ++i;
    }
    if (m < lt) e = lt;
    else if (m >= gt) b = gt;
    else return;
  }
}

static int32_t okd_build_rec(okd_tree* t, int32_t b, int32_t e) {
  int32_t id = okd_new_node(t);
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int32_t i = b; i < e; ++i) {
    const float* p = t->xyz + 3 * (size_t)t->perm[i];
    for (int d = 0; d < 3; ++d) { if (p[d] < lo[d]) lo[d] = p[d]; if (p[d] > hi[d]) hi[d] = p[d]; }
  }
  okd_node nd;
  memcpy(nd.lo, lo, sizeof lo); memcpy(nd.hi, hi, sizeof hi);
  nd.begin = b; nd.end = e; nd.left = nd.right = -1;
  if (e - b > OKD_LEAF) {
    int dim = 0; float ext = hi[0] - lo[0];
    if (hi[1] - lo[1] > ext) { dim = 1; ext = hi[1] - lo[1]; }
    if (hi[2] - lo[2] > ext) { dim = 2; ext = hi[2] - lo[2]; }
    int32_t m = b + (e - b) / 2;
    okd_select(t->xyz, t->perm, b, e, m, dim);
    int32_t l = okd_build_rec(t, b, m);
    int32_t r = okd_build_rec(t, m, e);
    nd.left = l; nd.right = r;
  }
  t->nodes[id] = nd;
  return id;
}

/* ---- all-core build (bench.py's cpu_baseline.all_core only): the same median splits, sub-trees as OpenMP tasks.  Node ids come
 * from an atomic counter over a preallocated array (leaves hold 8 .. 15 points: at most n / 4 + 2 nodes), so the numbering differs
 * from okd_build's; the boxes, the splits and therefore every search result are the same. */
static int32_t okd_build_par_rec(okd_tree* t, int32_t b, int32_t e) {
  size_t idv;
#pragma omp atomic capture
  idv = t->n_nodes++;
  const int32_t id = (int32_t)idv;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int32_t i = b; i < e; ++i) {
    const float* p = t->xyz + 3 * (size_t)t->perm[i];
    for (int d = 0; d < 3; ++d) { if (p[d] < lo[d]) lo[d] = p[d]; if (p[d] > hi[d]) hi[d] = p[d]; }
  }
  okd_node nd;
  memcpy(nd.lo, lo, sizeof lo); memcpy(nd.hi, hi, sizeof hi);
  nd.begin = b; nd.end = e; nd.left = nd.right = -1;
  if (e - b > OKD_LEAF) {
    int dim = 0; float ext = hi[0] - lo[0];
    if (hi[1] - lo[1] > ext) { dim = 1; ext = hi[1] - lo[1]; }
    if (hi[2] - lo[2] > ext) { dim = 2; ext = hi[2] - lo[2]; }
    int32_t m = b + (e - b) / 2;
    okd_select(t->xyz, t->perm, b, e, m, dim);
    int32_t l = -1, r = -1;
    if (e - b > 32768) {
#pragma omp task shared(l) firstprivate(t, b, m)
      l = okd_build_par_rec(t, b, m);
#pragma omp task shared(r) firstprivate(t, m, e)
      r = okd_build_par_rec(t, m, e);
#pragma omp taskwait
    } else {
      l = okd_build_par_rec(t, b, m);
      r = okd_build_par_rec(t, m, e);
    }
    nd.left = l; nd.right = r;
  }
  t->nodes[id] = nd;
  return id;
}

okd_tree* okd_build_parallel(const float* xyz, size_t n) {
  okd_tree* t = (okd_tree*)calloc(1, sizeof(okd_tree));
  t->xyz = xyz; t->n = n;
  t->perm = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
  t->pts = (float*)malloc(sizeof(float) * 3 * (n ? n : 1));
  t->cap_nodes = n / 4 + 64;
  t->nodes = (okd_node*)malloc(t->cap_nodes * sizeof(okd_node));
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)n; ++i) t->perm[i] = (int32_t)i;
  if (n > 0) {
#pragma omp parallel
#pragma omp single
    okd_build_par_rec(t, 0, (int32_t)n);
  }
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)n; ++i) memcpy(t->pts + 3 * (size_t)i, xyz + 3 * (size_t)t->perm[i], 3 * sizeof(float));
  return t;
}

okd_tree* okd_build(const float* xyz, size_t n) {
  okd_tree* t = (okd_tree*)calloc(1, sizeof(okd_tree));
  t->xyz = xyz; t->n = n;
  t->perm = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
  t->pts = (float*)malloc(sizeof(float) * 3 * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) t->perm[i] = (int32_t)i;
  if (n > 0) okd_build_rec(t, 0, (int32_t)n);
  for (size_t i = 0; i < n; ++i) memcpy(t->pts + 3 * i, xyz + 3 * (size_t)t->perm[i], 3 * sizeof(float));
  return t;
}

void okd_free(okd_tree* t) {
  if (!t) return;
  free(t->perm); free(t->pts); free(t->nodes); free(t);
}

static inline float okd_sqdist(const float* a, const float* b) {
  float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  float acc = dx * dx;
  acc = acc + dy * dy;
  acc = acc + dz * dz;
  return acc;
}

static inline float okd_box_bound(const okd_node* nd, const float* q) {
  float d[3];
  for (int a = 0; a < 3; ++a) {
    if (q[a] < nd->lo[a]) d[a] = q[a] - nd->lo[a];
    else if (q[a] > nd->hi[a]) d[a] = q[a] - nd->hi[a];
    else d[a] = 0.f;
  }
  float acc = d[0] * d[0];
  acc = acc + d[1] * d[1];
  acc = acc + d[2] * d[2];
  return acc;
}

/* ---- 1-NN within radius -------------------------------------------------- */
typedef struct { const okd_tree* t; const float* q; float r2; int found; float bd; int32_t bi; } okd_nn1;

static void okd_nn1_rec(okd_nn1* s, int32_t id) {
  const okd_node* nd = &s->t->nodes[id];
  if (nd->left < 0) {
    for (int32_t i = nd->begin; i < nd->end; ++i) {
      float d = okd_sqdist(s->q, s->t->pts + 3 * (size_t)i);
      int32_t oi = s->t->perm[i];
      if (!s->found) { if (d < s->r2) { s->found = 1; s->bd = d; s->bi = oi; } }
      else if (d < s->bd || (d == s->bd && oi < s->bi)) { s->bd = d; s->bi = oi; }
    }
    return;
  }
  float bl = okd_box_bound(&s->t->nodes[nd->left], s->q);
  float br = okd_box_bound(&s->t->nodes[nd->right], s->q);
  int32_t first = nd->left, second = nd->right; float b1 = bl, b2 = br;
  if (br < bl) { first = nd->right; second = nd->left; b1 = br; b2 = bl; }
  if (s->found ? (b1 <= s->bd) : (b1 < s->r2)) okd_nn1_rec(s, first);
  if (s->found ? (b2 <= s->bd) : (b2 < s->r2)) okd_nn1_rec(s, second);
}

int okd_nearest_within(const okd_tree* t, const float* q, float r2, int32_t* idx, float* dist) {
  if (t->n == 0) return 0;
  okd_nn1 s = {t, q, r2, 0, 0.f, 0};
  if (okd_box_bound(&t->nodes[0], q) < r2) okd_nn1_rec(&s, 0);
  if (!s.found) return 0;
  *idx = s.bi; *dist = s.bd;
  return 1;
}

/* ---- kNN ----------------------------------------------------------------- */
typedef struct { const okd_tree* t; const float* q; int k; int cnt; float* hd; int32_t* hi; } okd_knn_s;

static inline int okd_lt(float d1, int32_t i1, float d2, int32_t i2) {
  return d1 < d2 || (d1 == d2 && i1 < i2);
}
/* max-heap on (d, idx) */
static void okd_heap_push(okd_knn_s* s, float d, int32_t oi) {
  if (s->cnt < s->k) {
    int c = s->cnt++;
    s->hd[c] = d; s->hi[c] = oi;
    while (c > 0) {
      int p = (c - 1) / 2;
      if (okd_lt(s->hd[p], s->hi[p], s->hd[c], s->hi[c])) {
        float td = s->hd[p]; s->hd[p] = s->hd[c]; s->hd[c] = td;
        int32_t ti = s->hi[p]; s->hi[p] = s->hi[c]; s->hi[c] = ti;
        c = p;
      } else break;
    }
  } else if (okd_lt(d, oi, s->hd[0], s->hi[0])) {
    s->hd[0] = d; s->hi[0] = oi;
    int c = 0;
    for (;;) {
      int l = 2 * c + 1, r = l + 1, m = c;
      if (l < s->cnt && okd_lt(s->hd[m], s->hi[m], s->hd[l], s->hi[l])) m = l;
      if (r < s->cnt && okd_lt(s->hd[m], s->hi[m], s->hd[r], s->hi[r])) m = r;
      if (m == c) break;
      float td = s->hd[m]; s->hd[m] = s->hd[c]; s->hd[c] = td;
      int32_t ti = s->hi[m]; s->hi[m] = s->hi[c]; s->hi[c] = ti;
      c = m;
    }
  }
}

static void okd_knn_rec(okd_knn_s* s, int32_t id) {
  const okd_node* nd = &s->t->nodes[id];
  if (nd->left < 0) {
    for (int32_t i = nd->begin; i < nd->end; ++i) {
      float d = okd_sqdist(s->q, s->t->pts + 3 * (size_t)i);
      okd_heap_push(s, d, s->t->perm[i]);
    }
    return;
  }
  float bl = okd_box_bound(&s->t->nodes[nd->left], s->q);
  float br = okd_box_bound(&s->t->nodes[nd->right], s->q);
  int32_t first = nd->left, second = nd->right; float b1 = bl, b2 = br;
  if (br < bl) { first = nd->right; second = nd->left; b1 = br; b2 = bl; }
  if (s->cnt < s->k || b1 <= s->hd[0]) okd_knn_rec(s, first);
  if (s->cnt < s->k || b2 <= s->hd[0]) okd_knn_rec(s, second);
}

static void okd_sort_pairs(float* d, int32_t* idx, int n) {
  /* insertion sort (n is small: k or a radius result) would be O(n^2); use heap-free shell sort */
  for (int gap = n / 2; gap > 0; gap /= 2)
    for (int i = gap; i < n; ++i) {
      float td = d[i]; int32_t ti = idx[i]; int j = i;
      while (j >= gap && okd_lt(td, ti, d[j - gap], idx[j - gap])) { d[j] = d[j - gap]; idx[j] = idx[j - gap]; j -= gap; }
      d[j] = td; idx[j] = ti;
    }
}

int okd_knn(const okd_tree* t, const float* q, int k, int32_t* idx, float* dist) {
  if (t->n == 0 || k <= 0) return 0;
  okd_knn_s s = {t, q, k, 0, dist, idx};
  okd_knn_rec(&s, 0);
  okd_sort_pairs(dist, idx, s.cnt);
  return s.cnt;
}

/* ---- radius (all) --------------------------------------------------------- */
typedef struct { const okd_tree* t; const float* q; float r2; int cap; int cnt; float* d; int32_t* i;
                 float* od; int32_t* oi; int ocap; } okd_rad_s;

static void okd_rad_rec(okd_rad_s* s, int32_t id) {
  const okd_node* nd = &s->t->nodes[id];
  if (okd_box_bound(nd, s->q) >= s->r2) return;
  if (nd->left < 0) {
    for (int32_t i = nd->begin; i < nd->end; ++i) {
      float d = okd_sqdist(s->q, s->t->pts + 3 * (size_t)i);
      if (d < s->r2) {
        if (s->cnt == s->ocap) {
          s->ocap = s->ocap ? s->ocap * 2 : 256;
          s->od = (float*)realloc(s->od, sizeof(float) * (size_t)s->ocap);
          s->oi = (int32_t*)realloc(s->oi, sizeof(int32_t) * (size_t)s->ocap);
        }
        s->od[s->cnt] = d; s->oi[s->cnt] = s->t->perm[i]; ++s->cnt;
      }
    }
    return;
  }
  okd_rad_rec(s, nd->left);
  okd_rad_rec(s, nd->right);
}

int okd_radius(const okd_tree* t, const float* q, float r2, int cap, int32_t* idx, float* dist) {
  if (t->n == 0) return 0;
  okd_rad_s s = {t, q, r2, cap, 0, dist, idx, NULL, NULL, 0};
  okd_rad_rec(&s, 0);
  okd_sort_pairs(s.od, s.oi, s.cnt);
  int w = s.cnt < cap ? s.cnt : cap;
  for (int i = 0; i < w; ++i) { dist[i] = s.od[i]; idx[i] = s.oi[i]; }
  free(s.od); free(s.oi);
  return s.cnt;
}
