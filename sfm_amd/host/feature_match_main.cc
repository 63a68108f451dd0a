// The matching stage as a program, with the GPU brute-force matcher:
//
//   feature_match_main <sfm_data.bin> [--matcher brute|cascade] [--no-cross-check]
//                      [--engine f32|f16x3] [--devices N] [--quiet]
//
// --matcher cascade runs the GPU cascade-hash matcher, the counterpart of the matcher the
// reference's driver is shipped with (src/feature_match_main.cc:26); the default is the GPU
// brute-force matcher (the alternative at src/feature_match_main.cc:27).
//
// What it does is what src/feature_match_main.cc:7-33 does -- read the archive, pair up all views,
// match, write the archive back in place -- and it prints the same three kinds of lines.  As in
// the reference a failed read is reported and the stage carries on with an empty data set.
// Exit status: 0 done, 1 usage, 2 the GPU matcher failed, 3 the archive could not be written.
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <string>

#include <memory>

#include "hip_brute_force_matcher.hpp"
#include "hip_cascade_hash_matcher.hpp"
#include "matcher_interfaces.hpp"
#include "sfm_data_io.hpp"
#include "sfm_hip.h"

namespace {

struct Options {
  std::string path;
  bool cross_check = true;
  bool progress = true;
  int devices = 0;     // 0: every visible GPU
  int engine = -1;     // -1: SFM_HIP_ENGINE or the library default
  bool cascade = false;
};

bool parse(int argc, char** argv, Options& opt) {
  for (int i = 1; i < argc; i++) {
    const std::string arg = argv[i];
    if (arg == "--no-cross-check") {
      opt.cross_check = false;
    } else if (arg == "--quiet") {
      opt.progress = false;
    } else if (arg == "--devices" && i + 1 < argc) {
      opt.devices = std::atoi(argv[++i]);
    } else if (arg == "--matcher" && i + 1 < argc) {
      const std::string name = argv[++i];
      if (name == "cascade") opt.cascade = true;
      else if (name == "brute") opt.cascade = false;
      else return false;
    } else if (arg == "--engine" && i + 1 < argc) {
      const std::string name = argv[++i];
      if (name == "f32") opt.engine = SFM_HIP_ENGINE_F32;
      else if (name == "f16x3") opt.engine = SFM_HIP_ENGINE_F16X3;
      else return false;
    } else if (!arg.empty() && arg[0] != '-' && opt.path.empty()) {
      opt.path = arg;
    } else {
      return false;
    }
  }
  return !opt.path.empty();
}

}  // namespace

int main(int argc, char** argv) {
  Options opt;
  if (!parse(argc, argv, opt)) {
    std::fprintf(stderr,
                 "usage: %s <sfm_data.bin> [--matcher brute|cascade] [--no-cross-check] [--engine f32|f16x3] "
                 "[--devices N] [--quiet]\n",
                 argv[0]);
    return 1;
  }

  SfMData scene;
  const bool loaded = Load(scene, opt.path);
  if (loaded) std::printf("Load Sfm Data Finish\n");
  else std::printf("Load Sfm Data From %s Fails\n", opt.path.c_str());

  std::set<Pair> todo;
  ExhaustivePairBuilder().BuildPair(scene, todo);
  std::printf("Need To Match %lu Pairs.\n", (unsigned long)todo.size());

  std::shared_ptr<FeatureMatcherInterface> matcher;
  if (opt.cascade) {
    matcher = std::make_shared<HipCascadeHashMatcher>(opt.devices, opt.progress);
  } else {
    auto brute = std::make_shared<HipBruteForceMatcher>(opt.devices, 0.8, opt.progress);
    brute->set_engine(opt.engine);
    matcher = brute;
  }
  try {
    matcher->Match(scene, todo, opt.cross_check);
  } catch (const std::exception& failure) {
    std::fprintf(stderr, "%s\n", failure.what());
    return 2;
  }
  return Save(scene, opt.path) ? 0 : 3;
}
