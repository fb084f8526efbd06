/*
 * oracle/oracle_kdtree.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Exact single kd-tree over xyz (leaf size 15), standing in for
 * pcl::search::KdTree -> pcl::KdTreeFLANN -> flann::KDTreeSingleIndex
 * (PCL 1.10 / FLANN 1.9.1, NOT in /root/reference; behaviour recalled, see
 * SURVEY.md Appendix C).  Reference call sites:
 *   src/icp/icp_point_to_plane.cc:46-51,65-66   radiusSearch(q, d, max_nn=1)
 *   src/exe/icp_scan_aligner.cc:325-328          nearestKSearch(k) for normals
 *
 * Result semantics pinned by this oracle (the reference leaves ties to FLANN's
 * traversal order, which is unpinned):
 *   - distance = ((dx*dx) + dy*dy) + dz*dz in f32 (L2_Simple order)
 *   - radius search accepts iff dist < (float)((double)r*(double)r)  (strict)
 *   - candidates are ordered by (dist, index) lexicographically: the nearest
 *     neighbour among equidistant points is the one with the LOWEST index.
 */
#ifndef E3D_ORACLE_KDTREE_H
#define E3D_ORACLE_KDTREE_H

#include <stddef.h>
#include <stdint.h>

typedef struct okd_tree okd_tree;

/* xyz: n x 3 floats (kept by pointer; must outlive the tree). */
okd_tree* okd_build(const float* xyz, size_t n);
/* the same tree built by all host cores (OpenMP tasks; node numbering differs, search results do not): bench.py's all-core CPU baseline */
okd_tree* okd_build_parallel(const float* xyz, size_t n);
void okd_free(okd_tree* t);

/* Exact nearest neighbour with dist < r2.  Returns 1 and fills idx/dist when
 * found, 0 otherwise. */
int okd_nearest_within(const okd_tree* t, const float* q, float r2,
                       int32_t* idx, float* dist);

/* Exact k nearest neighbours sorted ascending by (dist, index).  Returns the
 * number found (= min(k, n)). */
int okd_knn(const okd_tree* t, const float* q, int k, int32_t* idx, float* dist);

/* All neighbours with dist < r2, sorted ascending by (dist, index), at most
 * cap results written (returns the total found, which may exceed cap). */
int okd_radius(const okd_tree* t, const float* q, float r2, int cap,
               int32_t* idx, float* dist);

#endif
