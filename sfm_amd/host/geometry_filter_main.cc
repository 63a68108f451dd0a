// The geometric verification of a matched data set as a program:
//
//   geometry_filter_main <sfm_data.bin|.json> [--seed S] [--devices N] [--quiet]
//
// Reads the archive, replaces its matches by the ones GeometryFilter keeps (src/sfm_main.cc:259-282:
// per pair a fundamental matrix by RANSAC, the matches that agree with it, pairs with 30 inliers or
// fewer dropped), and writes the archive back in place.  In the reference this step is the first
// thing its reconstruction program does (src/sfm_main.cc:296-305); here it stands alone.
// Exit status: 0 done, 1 usage, 2 the GPU path failed, 3 the archive could not be written,
// 4 the archive could not be read.
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>

#include "hip_geometry_filter.hpp"
#include "sfm_data_io.hpp"

int main(int argc, char** argv) {
  std::string path;
  HipGeometryFilterOptions opt;
  for (int i = 1; i < argc; i++) {
    const std::string arg = argv[i];
    if (arg == "--seed" && i + 1 < argc) opt.seed = std::strtoull(argv[++i], nullptr, 10);
    else if (arg == "--devices" && i + 1 < argc) opt.n_devices = std::atoi(argv[++i]);
    else if (arg == "--quiet") opt.print_progress = false;
    else if (!arg.empty() && arg[0] != '-' && path.empty()) path = arg;
    else path.clear(), i = argc;
  }
  if (path.empty()) {
    std::fprintf(stderr, "usage: %s <sfm_data.bin|.json> [--seed S] [--devices N] [--quiet]\n", argv[0]);
    return 1;
  }
  SfMData scene;
  if (!Load(scene, path)) {
    std::fprintf(stderr, "Load Sfm Data From %s Fails\n", path.c_str());
    return 4;
  }
  try {
    HipGeometryFilter(scene, opt);
  } catch (const std::exception& failure) {
    std::fprintf(stderr, "%s\n", failure.what());
    return 2;
  }
  return Save(scene, path) ? 0 : 3;
}
