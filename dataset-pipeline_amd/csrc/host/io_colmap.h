// io_colmap.h -- COLMAP text model and rigs.json I/O with the reference's semantics (src/io/colmap_model.cc):
//   ReadColmapCameras :53-78, ReadColmapImages :100-153 (translation scaled by the global scale_factor, second line
//   skipped), ReadColmapRigs :190-233, WriteColmapRigs :235-281 (rapidjson PrettyWriter layout: 4-space indent).
// Export of an optimised state (ExportProblemToColmap :286-483 with write_points = write_images = write_project = false
// and ExportRigs) lives in opt_problem.h because it needs the problem state.
#pragma once

#include <cctype>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "io_scans.h"

namespace e3d_host {

struct ColmapCamera {
  int camera_id = 0;
  std::string model_name;
  int width = 0, height = 0;
  std::vector<double> parameters;
};
struct ColmapImage {
  int image_id = 0;
  float q[4] = {1, 0, 0, 0};   // w x y z, as read (no normalisation)
  float t[3] = {0, 0, 0};      // already multiplied by scale_factor
  int camera_id = 0;
  std::string file_path;
};
struct ColmapRigCamera { int camera_id = 0; std::string image_prefix; };
struct ColmapRig { int ref_camera_id = 0; std::vector<ColmapRigCamera> cameras; };

inline bool ReadColmapCameras(const std::string& path, std::map<int, ColmapCamera>* cameras) {
  std::ifstream f(path);
  if (!f) return false;
  std::string line;
  while (std::getline(f, line)) {
    if (line.empty() || line[0] == '#') continue;
    ColmapCamera c;
    std::istringstream s(line);
    s >> c.camera_id >> c.model_name >> c.width >> c.height;
    double v;
    while (s >> v) c.parameters.push_back(v);
    cameras->insert(std::make_pair(c.camera_id, c));      // a repeated id keeps the FIRST entry (std::map::insert, like the reference)
  }
  return true;
}

// images keep the file's order of first appearance per id (std::map: ascending id, the order the tools iterate in)
inline bool ReadColmapImages(const std::string& path, std::map<int, ColmapImage>* images) {
  std::ifstream f(path);
  if (!f) return false;
  std::string line;
  while (std::getline(f, line)) {
    if (line.empty() || line[0] == '#') continue;
    ColmapImage im;
    std::istringstream s(line);
    s >> im.image_id >> im.q[0] >> im.q[1] >> im.q[2] >> im.q[3] >> im.t[0] >> im.t[1] >> im.t[2] >> im.camera_id >> im.file_path;
    if (global_scale_factor() != 0) {
      for (int i = 0; i < 3; ++i) im.t[i] *= global_scale_factor();
    } else {
      std::cerr << "Please load point clouds before images" << std::endl;
    }
    std::getline(f, line);      // feature observations line (not needed)
    images->insert(std::make_pair(im.image_id, im));      // likewise
  }
  return true;
}

// ---- rigs.json: [ { "ref_camera_id": int, "cameras": [ { "camera_id": int, "image_prefix": string }, ... ] }, ... ] --------
namespace json_detail {
struct Cursor {
  const std::string& s; size_t p = 0; bool ok = true;
  explicit Cursor(const std::string& str) : s(str) {}
  void ws() { while (p < s.size() && isspace((unsigned char)s[p])) ++p; }
  bool eat(char c) { ws(); if (p < s.size() && s[p] == c) { ++p; return true; } return false; }
  std::string str() {
    ws(); std::string o;
    if (p >= s.size() || s[p] != '"') { ok = false; return o; }
    ++p;
    while (p < s.size() && s[p] != '"') {
      if (s[p] == '\\' && p + 1 < s.size()) { ++p; o += (s[p] == 'n') ? '\n' : (s[p] == 't' ? '\t' : s[p]); }
      else o += s[p];
      ++p;
    }
    if (p >= s.size()) ok = false; else ++p;
    return o;
  }
  long integer() {
    ws(); size_t q = p;
    if (q < s.size() && (s[q] == '-' || s[q] == '+')) ++q;
    while (q < s.size() && isdigit((unsigned char)s[q])) ++q;
    if (q == p) { ok = false; return 0; }
    const long v = atol(s.substr(p, q - p).c_str()); p = q; return v;
  }
  void skip_value() {       // unknown members
    ws();
    if (p >= s.size()) { ok = false; return; }
    if (s[p] == '"') { str(); return; }
    if (s[p] == '{' || s[p] == '[') {
      const char open = s[p], close = open == '{' ? '}' : ']';
      int depth = 0;
      for (; p < s.size(); ++p) {
        if (s[p] == '"') { str(); --p; continue; }
        if (s[p] == open) ++depth;
        else if (s[p] == close && --depth == 0) { ++p; return; }
      }
      ok = false; return;
    }
    while (p < s.size() && s[p] != ',' && s[p] != '}' && s[p] != ']') ++p;
  }
};
}  // namespace json_detail

inline bool ReadColmapRigs(const std::string& path, std::vector<ColmapRig>* rigs) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  const std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  json_detail::Cursor c(text);
  if (!c.eat('[')) return false;
  if (c.eat(']')) return true;
  do {
    if (!c.eat('{')) return false;
    ColmapRig rig;
    if (!c.eat('}')) {
      do {
        const std::string key = c.str();
        if (!c.eat(':')) return false;
        if (key == "ref_camera_id") rig.ref_camera_id = (int)c.integer();
        else if (key == "cameras") {
          if (!c.eat('[')) return false;
          if (!c.eat(']')) {
            do {
              if (!c.eat('{')) return false;
              ColmapRigCamera cam;
              if (!c.eat('}')) {
                do {
                  const std::string k2 = c.str();
                  if (!c.eat(':')) return false;
                  if (k2 == "camera_id") cam.camera_id = (int)c.integer();
                  else if (k2 == "image_prefix") cam.image_prefix = c.str();
                  else c.skip_value();
                } while (c.ok && c.eat(','));
                if (!c.eat('}')) return false;
              }
              rig.cameras.push_back(cam);
            } while (c.ok && c.eat(','));
            if (!c.eat(']')) return false;
          }
        } else c.skip_value();
      } while (c.ok && c.eat(','));
      if (!c.eat('}')) return false;
    }
    rigs->push_back(rig);
  } while (c.ok && c.eat(','));
  return c.ok && c.eat(']');
}

inline bool WriteColmapRigs(const std::string& path, const std::vector<ColmapRig>& rigs) {
  std::ofstream f(path);
  if (!f) return false;
  if (rigs.empty()) { f << "[]"; return true; }
  f << "[\n";
  for (size_t i = 0; i < rigs.size(); ++i) {
    f << "    {\n        \"ref_camera_id\": " << rigs[i].ref_camera_id << ",\n        \"cameras\": [";
    for (size_t c = 0; c < rigs[i].cameras.size(); ++c) {
      f << "\n            {\n                \"camera_id\": " << rigs[i].cameras[c].camera_id
        << ",\n                \"image_prefix\": \"" << rigs[i].cameras[c].image_prefix << "\"\n            }"
        << (c + 1 < rigs[i].cameras.size() ? "," : "");
    }
    f << (rigs[i].cameras.empty() ? "]" : "\n        ]") << "\n    }" << (i + 1 < rigs.size() ? "," : "") << "\n";
  }
  f << "]";
  return true;
}

}  // namespace e3d_host
