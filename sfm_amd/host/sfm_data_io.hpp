// src/sfm_data_io.hpp:5-6.
#ifndef SFM_AMD_HOST_SFM_DATA_IO_HPP_
#define SFM_AMD_HOST_SFM_DATA_IO_HPP_
#include <string>

#include "sfm_data.hpp"

// ".bin": the byte stream cereal's BinaryOutputArchive produces for the reference's serialize()
// functions (src/sfm_data_io.cc:13-47): native little-endian, no names, no versions.
// ".json": the document cereal's JSONOutputArchive produces for the same functions (layout in
// sfm_data_json.cc).
// Any other extension, or a path that cannot be opened: false, as the reference.
bool Save(const SfMData& sfm_data, const std::string path);
bool Load(SfMData& sfm_data, const std::string path);
#endif
