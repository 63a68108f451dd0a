// ".json" SfMData archives without cereal.
//
// cereal's JSONOutputArchive turns the reference's serialize() functions (src/sfm_data_io.cc:13-47)
// into this document: the archived object itself is unnamed, hence "value0"; named fields keep
// their names; unnamed fields are "value0", "value1", ... in order; a std::map is an array of
// {"key": ..., "value": ...} objects; a std::pair is {"first": ..., "second": ...}; a std::vector is
// an array:
//
//   { "value0": {
//       "Image_Width": u64, "Image_Height": u64,
//       "Views":       [ { "key": id, "value": { "value0": "path" } }, ... ],
//       "KeyPoint":    [ { "key": id, "value": [ { "x": f64, "y": f64 }, ... ] }, ... ],
//       "Descriptors": [ { "key": id, "value": { "value0": [ f32, ... ] } }, ... ],
//       "Matches":     [ { "key": { "first": id, "second": id },
//                          "value": [ { "value0": lx, "value1": ly, "value2": rx, "value3": ry,
//                                       "value4": lhs_idx, "value5": rhs_idx }, ... ] }, ... ] } }
//
// Numbers are written in the shortest form that reads back to the same value (floats as the double
// they convert to, as cereal does).  The reader takes the members of an object in any order and
// skips members it does not know.  No cereal is available to cross-check against: the layout follows
// cereal's documented behaviour; the exact spelling of numbers and white space may differ from what
// cereal writes, which a JSON reader does not see.
#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>

#include "sfm_data_io.hpp"

namespace {

// ---------------------------------------------------------------- writing
struct Out {
  std::string buf;
  std::ofstream& os;
  explicit Out(std::ofstream& o) : os(o) { buf.reserve(1 << 20); }
  void flush_if_large() {
    if (buf.size() > (1 << 20)) {
      os.write(buf.data(), (std::streamsize)buf.size());
      buf.clear();
    }
  }
  void text(const char* s) { buf += s; }
  void u64(uint64_t v) { buf += std::to_string(v); }
  void i64(int64_t v) { buf += std::to_string(v); }
  void real(double v) {
    char tmp[40];
    auto res = std::to_chars(tmp, tmp + sizeof(tmp), v);   // shortest round-trip form
    std::string s(tmp, res.ptr);
    // keep it a JSON real, as cereal's writer does ("1" -> "1.0")
    if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
    buf += s;
    flush_if_large();
  }
  void str(const std::string& s) {
    buf += '"';
    for (unsigned char c : s) {
      switch (c) {
        case '"': buf += "\\\""; break;
        case '\\': buf += "\\\\"; break;
        case '\n': buf += "\\n"; break;
        case '\r': buf += "\\r"; break;
        case '\t': buf += "\\t"; break;
        default:
          if (c < 0x20) {
            char esc[8];
            std::snprintf(esc, sizeof(esc), "\\u%04x", c);
            buf += esc;
          } else {
            buf += (char)c;
          }
      }
    }
    buf += '"';
  }
};

// ---------------------------------------------------------------- reading
struct In {
  const char* p;
  const char* end;
  bool ok = true;
  void fail() { ok = false; p = end; }
  void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
  bool take(char c) {
    ws();
    if (p < end && *p == c) { p++; return true; }
    return false;
  }
  void expect(char c) { if (!take(c)) fail(); }
  bool peek(char c) { ws(); return p < end && *p == c; }
  std::string string() {
    std::string s;
    expect('"');
    while (ok && p < end && *p != '"') {
      char c = *p++;
      if (c == '\\' && p < end) {
        c = *p++;
        switch (c) {
          case 'n': s += '\n'; break;
          case 't': s += '\t'; break;
          case 'r': s += '\r'; break;
          case 'b': s += '\b'; break;
          case 'f': s += '\f'; break;
          case 'u': {
            if (end - p < 4) { fail(); break; }
            unsigned cp = (unsigned)std::strtoul(std::string(p, p + 4).c_str(), nullptr, 16);
            p += 4;
            if (cp < 0x80) s += (char)cp;
            else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
            else { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
            break;
          }
          default: s += c;   // \" \\ \/
        }
      } else {
        s += c;
      }
    }
    expect('"');
    return s;
  }
  const char* number_end() {
    const char* q = p;
    while (q < end && (std::strchr("+-0123456789.eE", *q) != nullptr)) q++;
    return q;
  }
  double real() {
    ws();
    const char* q = number_end();
    double v = 0;
    if (q == p || std::from_chars(p, q, v).ec != std::errc()) fail();
    else p = q;
    return v;
  }
  int64_t integer() {
    ws();
    const char* q = number_end();
    int64_t v = 0;
    auto res = std::from_chars(p, q, v);
    if (q == p || res.ec != std::errc()) { fail(); return 0; }
    if (res.ptr != q) {            // written as a real ("3.0"): accept when integral
      double d = 0;
      if (std::from_chars(p, q, d).ec != std::errc() || d != (double)(int64_t)d) { fail(); return 0; }
      v = (int64_t)d;
    }
    p = q;
    return v;
  }
  void skip_value() {
    ws();
    if (p >= end) return fail();
    if (*p == '"') { (void)string(); return; }
    if (*p == '{' || *p == '[') {
      const char close = *p == '{' ? '}' : ']';
      p++;
      if (take(close)) return;
      do {
        if (close == '}') { (void)string(); expect(':'); }
        skip_value();
      } while (ok && take(','));
      expect(close);
      return;
    }
    if (std::strncmp(p, "true", 4) == 0 || std::strncmp(p, "null", 4) == 0) { p += 4; return; }
    if (std::strncmp(p, "false", 5) == 0) { p += 5; return; }
    (void)real();
  }
  // calls member(name) for every member of an object; member must consume the value
  template <typename F>
  void object(F&& member) {
    expect('{');
    if (take('}')) return;
    do {
      const std::string name = string();
      expect(':');
      if (ok) member(name);
    } while (ok && take(','));
    expect('}');
  }
  template <typename F>
  void array(F&& element) {
    expect('[');
    if (take(']')) return;
    do {
      if (ok) element();
    } while (ok && take(','));
    expect(']');
  }
};

}  // namespace

namespace sfm_data_json {

bool Save(const SfMData& d, const std::string& path) {
  std::ofstream os(path, std::ios::trunc);
  if (!os.is_open()) {
    std::printf("Path %s can't open.\n", path.c_str());
    return false;
  }
  Out o(os);
  o.text("{\n    \"value0\": {\n        \"Image_Width\": ");
  o.u64(d.image_width);
  o.text(",\n        \"Image_Height\": ");
  o.u64(d.image_height);
  o.text(",\n        \"Views\": [");
  bool first = true;
  for (const auto& kv : d.views) {
    o.text(first ? "\n" : ",\n");
    first = false;
    o.text("            { \"key\": ");
    o.i64(kv.first);
    o.text(", \"value\": { \"value0\": ");
    o.str(kv.second.image_path);
    o.text(" } }");
  }
  o.text(first ? "],\n        \"KeyPoint\": [" : "\n        ],\n        \"KeyPoint\": [");
  first = true;
  for (const auto& kv : d.key_points) {
    o.text(first ? "\n" : ",\n");
    first = false;
    o.text("            { \"key\": ");
    o.i64(kv.first);
    o.text(", \"value\": [");
    for (size_t k = 0; k < kv.second.size(); k++) {
      o.text(k ? ", { \"x\": " : " { \"x\": ");
      o.real(kv.second[k].x);
      o.text(", \"y\": ");
      o.real(kv.second[k].y);
      o.text(" }");
    }
    o.text(" ] }");
  }
  o.text(first ? "],\n        \"Descriptors\": [" : "\n        ],\n        \"Descriptors\": [");
  first = true;
  for (const auto& kv : d.descriptors) {
    o.text(first ? "\n" : ",\n");
    first = false;
    o.text("            { \"key\": ");
    o.i64(kv.first);
    o.text(", \"value\": { \"value0\": [");
    const std::vector<float>& raw = kv.second.getRaw_Data();
    for (size_t k = 0; k < raw.size(); k++) {
      o.text(k ? ", " : " ");
      o.real((double)raw[k]);
    }
    o.text(" ] } }");
  }
  o.text(first ? "],\n        \"Matches\": [" : "\n        ],\n        \"Matches\": [");
  first = true;
  for (const auto& kv : d.matches) {
    o.text(first ? "\n" : ",\n");
    first = false;
    o.text("            { \"key\": { \"first\": ");
    o.i64(kv.first.first);
    o.text(", \"second\": ");
    o.i64(kv.first.second);
    o.text(" }, \"value\": [");
    for (size_t k = 0; k < kv.second.size(); k++) {
      const Match& m = kv.second[k];
      o.text(k ? ", { \"value0\": " : " { \"value0\": ");
      o.real(m.lhs_observation.x());
      o.text(", \"value1\": ");
      o.real(m.lhs_observation.y());
      o.text(", \"value2\": ");
      o.real(m.rhs_observation.x());
      o.text(", \"value3\": ");
      o.real(m.rhs_observation.y());
      o.text(", \"value4\": ");
      o.i64(m.lhs_idx);
      o.text(", \"value5\": ");
      o.i64(m.rhs_idx);
      o.text(" }");
    }
    o.text(" ] }");
  }
  o.text(first ? "]\n    }\n}\n" : "\n        ]\n    }\n}\n");
  os.write(o.buf.data(), (std::streamsize)o.buf.size());
  return os.good();
}

bool Load(SfMData& out, const std::string& path) {
  std::ifstream is(path, std::ios::in);
  if (!is.is_open()) {
    std::printf("Path %s can't open\n", path.c_str());
    return false;
  }
  std::stringstream ss;
  ss << is.rdbuf();
  const std::string text = ss.str();
  In in{text.data(), text.data() + text.size()};
  SfMData t;
  bool seen = false;
  in.object([&](const std::string& top) {
    if (top != "value0") return in.skip_value();
    seen = true;
    in.object([&](const std::string& field) {
      if (field == "Image_Width") {
        t.image_width = (std::size_t)in.integer();
      } else if (field == "Image_Height") {
        t.image_height = (std::size_t)in.integer();
      } else if (field == "Views") {
        in.array([&] {
          IndexT id = 0;
          View v;
          in.object([&](const std::string& m) {
            if (m == "key") id = in.integer();
            else if (m == "value") in.object([&](const std::string& n) { if (n == "value0") v.image_path = in.string(); else in.skip_value(); });
            else in.skip_value();
          });
          t.views[id] = v;
        });
      } else if (field == "KeyPoint") {
        in.array([&] {
          IndexT id = 0;
          std::vector<KeyPoint> pts;
          in.object([&](const std::string& m) {
            if (m == "key") id = in.integer();
            else if (m == "value") in.array([&] {
              KeyPoint k{};
              in.object([&](const std::string& n) {
                if (n == "x") k.x = in.real();
                else if (n == "y") k.y = in.real();
                else in.skip_value();
              });
              pts.push_back(k);
            });
            else in.skip_value();
          });
          t.key_points[id] = std::move(pts);
        });
      } else if (field == "Descriptors") {
        in.array([&] {
          IndexT id = 0;
          std::vector<float> raw;
          in.object([&](const std::string& m) {
            if (m == "key") id = in.integer();
            else if (m == "value") in.object([&](const std::string& n) {
              if (n == "value0") in.array([&] { raw.push_back((float)in.real()); });
              else in.skip_value();
            });
            else in.skip_value();
          });
          t.descriptors[id].getRaw_Data() = std::move(raw);
        });
      } else if (field == "Matches") {
        in.array([&] {
          Pair key{0, 0};
          Matches list;
          in.object([&](const std::string& m) {
            if (m == "key") in.object([&](const std::string& n) {
              if (n == "first") key.first = in.integer();
              else if (n == "second") key.second = in.integer();
              else in.skip_value();
            });
            else if (m == "value") in.array([&] {
              double v[4] = {0, 0, 0, 0};
              Match rec;
              in.object([&](const std::string& n) {
                if (n.size() == 6 && n.compare(0, 5, "value") == 0 && n[5] >= '0' && n[5] <= '3') v[n[5] - '0'] = in.real();
                else if (n == "value4") rec.lhs_idx = in.integer();
                else if (n == "value5") rec.rhs_idx = in.integer();
                else in.skip_value();
              });
              rec.lhs_observation = Observation(v[0], v[1]);
              rec.rhs_observation = Observation(v[2], v[3]);
              list.push_back(rec);
            });
            else in.skip_value();
          });
          t.matches[key] = std::move(list);
        });
      } else {
        in.skip_value();
      }
    });
  });
  in.ws();
  if (!in.ok || !seen || in.p != in.end) {
    std::printf("Path %s is not an SfMData JSON archive\n", path.c_str());
    return false;
  }
  out = std::move(t);
  return true;
}

}  // namespace sfm_data_json
