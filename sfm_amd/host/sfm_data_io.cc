// Reader/writer for the reference's ".bin" SfMData files without cereal (".json": sfm_data_json.cc).
//
// cereal's binary archive writes, for the containers used in src/sfm_data_io.cc:40-47:
//   arithmetic value      raw bytes, native endianness
//   size tag              uint64
//   std::string           size tag, bytes
//   std::vector<float>    size tag, raw block                  (Descriptors: .cc:24-26)
//   std::vector<T>        size tag, each T through its serialize()
//   std::map<K, V>        size tag, then key, value per item
//   std::pair<A, B>       first, second
// and the field order of SfMData is image_width, image_height, views, key_points,
// descriptors, matches (.cc:41-46); KeyPoint is x, y (.cc:19-20); Match is
// lhs x, lhs y, rhs x, rhs y, lhs_idx, rhs_idx (.cc:32-37).
// No cereal is available to cross-check against, so byte compatibility with files written by
// the reference is derived from cereal's documented behaviour, not verified.
#include "sfm_data_io.hpp"

#include <cstdio>
#include <cstring>
#include <fstream>

namespace sfm_data_json {     // sfm_data_json.cc
bool Save(const SfMData& d, const std::string& path);
bool Load(SfMData& d, const std::string& path);
}  // namespace sfm_data_json

namespace {

std::string extension_of(const std::string& path) {
  const size_t dot = path.find_last_of('.');   // npos + 1 == 0: whole path, as the reference
  return path.substr(dot + 1);
}

struct Writer {
  std::ofstream& os;
  template <typename T>
  void raw(const T& v) { os.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
  void size(uint64_t n) { raw(n); }
};

struct Reader {
  std::ifstream& is;
  bool ok = true;
  template <typename T>
  T raw() {
    T v{};
    if (!is.read(reinterpret_cast<char*>(&v), sizeof(T))) ok = false;
    return v;
  }
  uint64_t size(uint64_t elem_bytes) {
    const uint64_t n = raw<uint64_t>();
    // a corrupt size must not turn into a huge allocation
    if (ok && elem_bytes && n > (uint64_t(1) << 40) / elem_bytes) ok = false;
    return ok ? n : 0;
  }
};

}  // namespace

bool Save(const SfMData& d, const std::string path) {
  const std::string ext = extension_of(path);
  if (ext == "json") return sfm_data_json::Save(d, path);     // sfm_data_io.cc:73-74
  if (ext != "bin") return false;
  std::ofstream os(path, std::ios::trunc | std::ios::binary);
  if (!os.is_open()) {
    std::printf("Path %s can't open.\n", path.c_str());
    return false;
  }
  Writer w{os};
  w.raw<uint64_t>(d.image_width);
  w.raw<uint64_t>(d.image_height);
  w.size(d.views.size());
  for (const auto& kv : d.views) {
    w.raw<int64_t>(kv.first);
    w.size(kv.second.image_path.size());
    os.write(kv.second.image_path.data(), (std::streamsize)kv.second.image_path.size());
  }
  w.size(d.key_points.size());
  for (const auto& kv : d.key_points) {
    w.raw<int64_t>(kv.first);
    w.size(kv.second.size());
    for (const KeyPoint& k : kv.second) {
      w.raw<double>(k.x);
      w.raw<double>(k.y);
    }
  }
  w.size(d.descriptors.size());
  for (const auto& kv : d.descriptors) {
    w.raw<int64_t>(kv.first);
    const std::vector<float>& raw = kv.second.getRaw_Data();
    w.size(raw.size());
    os.write(reinterpret_cast<const char*>(raw.data()), (std::streamsize)(raw.size() * sizeof(float)));
  }
  w.size(d.matches.size());
  for (const auto& kv : d.matches) {
    w.raw<int64_t>(kv.first.first);
    w.raw<int64_t>(kv.first.second);
    w.size(kv.second.size());
    for (const Match& m : kv.second) {
      w.raw<double>(m.lhs_observation.x());
      w.raw<double>(m.lhs_observation.y());
      w.raw<double>(m.rhs_observation.x());
      w.raw<double>(m.rhs_observation.y());
      w.raw<int64_t>(m.lhs_idx);
      w.raw<int64_t>(m.rhs_idx);
    }
  }
  return os.good();
}

bool Load(SfMData& d, const std::string path) {
  const std::string ext = extension_of(path);
  if (ext == "json") return sfm_data_json::Load(d, path);     // sfm_data_io.cc:101-102
  if (ext != "bin") return false;
  std::ifstream is(path, std::ios::in | std::ios::binary);
  if (!is.is_open()) {
    std::printf("Path %s can't open\n", path.c_str());
    return false;
  }
  Reader r{is};
  SfMData t;
  t.image_width = (std::size_t)r.raw<uint64_t>();
  t.image_height = (std::size_t)r.raw<uint64_t>();
  for (uint64_t n = r.size(16), i = 0; i < n && r.ok; i++) {
    const int64_t id = r.raw<int64_t>();
    std::string s((size_t)r.size(1), '\0');
    if (r.ok && !s.empty() && !is.read(&s[0], (std::streamsize)s.size())) r.ok = false;
    t.views[id].image_path = std::move(s);
  }
  for (uint64_t n = r.size(16), i = 0; i < n && r.ok; i++) {
    const int64_t id = r.raw<int64_t>();
    std::vector<KeyPoint>& v = t.key_points[id];
    v.resize((size_t)r.size(16));
    for (KeyPoint& k : v) {
      k.x = r.raw<double>();
      k.y = r.raw<double>();
    }
  }
  for (uint64_t n = r.size(16), i = 0; i < n && r.ok; i++) {
    const int64_t id = r.raw<int64_t>();
    std::vector<float>& raw = t.descriptors[id].getRaw_Data();
    raw.resize((size_t)r.size(4));
    if (r.ok && !raw.empty() && !is.read(reinterpret_cast<char*>(raw.data()), (std::streamsize)(raw.size() * 4)))
      r.ok = false;
  }
  for (uint64_t n = r.size(24), i = 0; i < n && r.ok; i++) {
    Pair p;
    p.first = r.raw<int64_t>();
    p.second = r.raw<int64_t>();
    Matches& v = t.matches[p];
    v.resize((size_t)r.size(48));
    for (Match& m : v) {
      const double lx = r.raw<double>(), ly = r.raw<double>(), rx = r.raw<double>(), ry = r.raw<double>();
      m.lhs_observation = Observation(lx, ly);
      m.rhs_observation = Observation(rx, ry);
      m.lhs_idx = r.raw<int64_t>();
      m.rhs_idx = r.raw<int64_t>();
    }
  }
  if (!r.ok) {
    std::printf("Path %s is truncated or not an SfMData binary archive\n", path.c_str());
    return false;
  }
  d = std::move(t);
  return true;
}
