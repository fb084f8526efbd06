// e3d_normals.hip -- kNN normal estimation (pcl::NormalEstimationTwoPassOMP replacement).
#include "../../include/e3d_hip.h"
#include "e3d_common.hpp"

extern "C" int e3d_normals_knn(const float*, size_t, int, const float*, float*, float*, int32_t*) {
  e3d::set_last_error("e3d_normals_knn: not implemented yet");
  return E3D_ERR_INVALID;
}
