// PointCloudCleaner -- drop-in replacement of the reference tool (src/exe/point_cloud_cleaner.cc:44-136): applies one or
// more pcl::LocalStatisticalOutlierRemoval passes (--filter <knn,factor>, in the order given) to a PLY cloud and writes
// <in>.inliers.ply and <in>.outliers.ply (binary, x y z + red green blue).  Each pass runs on the MI355X behind
// e3d_local_outlier_removal; outliers accumulate over the passes in removal order.
#include <exception>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "e3d_loader.h"
#include "io_ply.h"
#include "util.h"

using namespace e3d_host;

// pcl::console::parse_multiple_2x_arguments: every occurrence of `name` followed by "a,b"
static bool parse_multiple_2x_arguments(int argc, char** argv, const char* name, std::vector<double>& a, std::vector<double>& b) {
  for (int i = 1; i < argc; ++i) {
    if (strcmp(argv[i], name) != 0 || ++i >= argc) continue;
    std::vector<double> values;
    std::string token;
    for (const char* c = argv[i];; ++c) {
      if (*c == ',' || *c == 0) { values.push_back(atof(token.c_str())); token.clear(); if (*c == 0) break; }
      else token += *c;
    }
    if (values.size() != 2) {
      std::cerr << "[parse_multiple_2x_arguments] Number of values for " << name << " (" << values.size() << ") different than 2!" << std::endl;
      return false;
    }
    a.push_back(values[0]);
    b.push_back(values[1]);
  }
  return !a.empty();
}

static int run_tool(int argc, char** argv) {
  int dummy;
  if (argc <= 1 || parse_argument(argc, argv, "-h", dummy) >= 0 || parse_argument(argc, argv, "--help", dummy) >= 0) {
    std::cerr << "Usage: " << argv[0] << " --in <file.ply> --filter <knn,factor> [--filter <knn2,factor2>, ...]" << std::endl;
    return EXIT_FAILURE;
  }
  std::string point_cloud_file_path;
  parse_argument(argc, argv, "--in", point_cloud_file_path);
  std::vector<double> knn_parameters, factor_parameters;
  parse_multiple_2x_arguments(argc, argv, "--filter", knn_parameters, factor_parameters);
  if (knn_parameters.size() != factor_parameters.size()) return EXIT_FAILURE;
  if (knn_parameters.empty()) {
    std::cerr << "One or more --filter knn,factor parameter values must be given." << std::endl;
    return EXIT_FAILURE;
  }

  PointCloud current;
  if (loadPLYFile(point_cloud_file_path, current, /*want_rgb=*/true) < 0) {
    std::cerr << "Cannot read " << point_cloud_file_path << std::endl;
    return EXIT_FAILURE;
  }
  const size_t total_point_count = current.size();
  std::vector<float> outlier_xyz;
  std::vector<uint8_t> outlier_rgb;

  for (size_t iteration = 0; iteration < knn_parameters.size(); ++iteration) {
    const int knn = (int)(knn_parameters[iteration] + 0.5);
    const double factor = factor_parameters[iteration];
    std::cerr << "Applying filter with knn = " << knn << ", factor = " << factor << " ..." << std::endl;
    const size_t n = current.size();
    std::vector<uint8_t> inlier(n);
    if (api().e3d_local_outlier_removal(current.xyz.data(), n, knn, factor, /*negative*/ 0, inlier.data(), nullptr) < 0) {
      std::cerr << "filter failed: " << api().e3d_last_error() << std::endl;
      return EXIT_FAILURE;
    }
    PointCloud filtered;
    for (size_t i = 0; i < n; ++i) {
      std::vector<float>& xyz = inlier[i] ? filtered.xyz : outlier_xyz;
      std::vector<uint8_t>& rgb = inlier[i] ? filtered.rgb : outlier_rgb;
      xyz.insert(xyz.end(), current.xyz.begin() + 3 * i, current.xyz.begin() + 3 * i + 3);
      rgb.insert(rgb.end(), current.rgb.begin() + 3 * i, current.rgb.begin() + 3 * i + 3);
    }
    if (filtered.size() + outlier_xyz.size() / 3 != total_point_count) {      // CHECK_EQ (:115)
      std::cerr << "Check failed: inliers + outliers != total point count" << std::endl;
      return EXIT_FAILURE;
    }
    current.xyz.swap(filtered.xyz);
    current.rgb.swap(filtered.rgb);
  }

  if (savePLYFileBinaryXYZRGB(point_cloud_file_path + ".inliers.ply", current.xyz, current.rgb) < 0 ||
      savePLYFileBinaryXYZRGB(point_cloud_file_path + ".outliers.ply", outlier_xyz, outlier_rgb) < 0)
    return EXIT_FAILURE;
  return EXIT_SUCCESS;
}

// library errors (no device, out of memory, ...) arrive as exceptions of the host classes: report, EXIT_FAILURE
int main(int argc, char** argv) {
  try {
    return run_tool(argc, argv);
  } catch (const std::exception& e) {
    std::cerr << "PointCloudCleaner: " << e.what() << std::endl;
    return EXIT_FAILURE;
  }
}
