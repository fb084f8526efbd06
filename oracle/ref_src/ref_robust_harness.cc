// oracle/_ref/libe3d_ref.so -- the parts of the reference that compile from their own sources in this image (TEST INFRASTRUCTURE ONLY).
//
// Two headers of the hot path (B) depend on nothing but the C++ standard library and are compiled where they lie under /root/reference
// (the -I of oracle/Makefile, target _ref): src/opt/robust_weighting.h (opt::RobustWeighting::CalculateWeight / CalculateRobustResidual,
// used by AccumulateHAndBAndResidualsForObservations and CostCalculator) and src/opt/descriptor.h (opt::ComputeDescriptor).  Everything else
// of the path needs Eigen / PCL / OpenCV / glog / Boost, which the image lacks.  This file only exports them with C
// linkage; it contains no restatement of their arithmetic.  tests/test_oracle_ref.py compares the oracle's restatement with it bit for bit.
#include "opt/descriptor.h"
#include "opt/robust_weighting.h"

extern "C" {

// type: 0 none, 1 Huber, 2 Tukey (opt::RobustWeighting::Type)
float e3d_ref_robust_weight(int type, float parameter, float residual) {
  opt::RobustWeighting w(static_cast<opt::RobustWeighting::Type>(type));
  w.set_parameter(parameter);
  return w.CalculateWeight(residual);
}
float e3d_ref_robust_residual(int type, float parameter, float residual) {
  opt::RobustWeighting w(static_cast<opt::RobustWeighting::Type>(type));
  w.set_parameter(parameter);
  return w.CalculateRobustResidual(residual);
}
void e3d_ref_robust_many(int type, float parameter, const float* residuals, long n, float* weights, float* robust_residuals) {
  opt::RobustWeighting w(static_cast<opt::RobustWeighting::Type>(type));
  w.set_parameter(parameter);
  for (long i = 0; i < n; ++i) { weights[i] = w.CalculateWeight(residuals[i]); robust_residuals[i] = w.CalculateRobustResidual(residuals[i]); }
}
float e3d_ref_compute_descriptor(float center_intensity, float neighbor_intensity) {
  return opt::ComputeDescriptor(center_intensity, neighbor_intensity);
}

}
