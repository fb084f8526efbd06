// io_mlp.h -- MeshLab project (.mlp) reading / writing for the tools.  Two flavours exist in the reference and both
// are mirrored: the private reader/writer of ICPScanAligner (src/exe/icp_scan_aligner.cc:72-136; plain R, T in
// double, no scale handling) and io::ReadMeshLabProject (src/io/meshlab_project.cc:39-87; Sim3 with the global
// scale_factor side effect [QUIRK]).  XML handling is a minimal hand-written reader for the MeshLabProject subset
// (tinyxml2 is a reference third-party source and is not copied).
#pragma once

#include <cmath>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace e3d_host {

struct MlpMesh {
  std::string label, filename;
  bool has_matrix = false;
  std::string matrix_text;
};

namespace mlp_detail {
inline std::string unescape(const std::string& s) {
  std::string o;
  for (size_t i = 0; i < s.size(); ++i) {
    if (s[i] == '&') {
      if (!s.compare(i, 5, "&amp;")) { o += '&'; i += 4; continue; }
      if (!s.compare(i, 4, "&lt;")) { o += '<'; i += 3; continue; }
      if (!s.compare(i, 4, "&gt;")) { o += '>'; i += 3; continue; }
      if (!s.compare(i, 6, "&quot;")) { o += '"'; i += 5; continue; }
      if (!s.compare(i, 6, "&apos;")) { o += '\''; i += 5; continue; }
    }
    o += s[i];
  }
  return o;
}
inline std::string escape(const std::string& s) {
  std::string o;
  for (char c : s) {
    if (c == '&') o += "&amp;"; else if (c == '<') o += "&lt;"; else if (c == '>') o += "&gt;"; else if (c == '"') o += "&quot;"; else o += c;
  }
  return o;
}
inline bool attribute(const std::string& tag, const std::string& name, std::string* out) {
  size_t p = 0;
  while ((p = tag.find(name, p)) != std::string::npos) {
    const bool left_ok = (p == 0) || isspace((unsigned char)tag[p - 1]);
    size_t q = p + name.size();
    while (q < tag.size() && isspace((unsigned char)tag[q])) ++q;
    if (left_ok && q < tag.size() && tag[q] == '=') {
      ++q;
      while (q < tag.size() && isspace((unsigned char)tag[q])) ++q;
      if (q < tag.size() && (tag[q] == '"' || tag[q] == '\'')) {
        const char quote = tag[q];
        const size_t e = tag.find(quote, q + 1);
        if (e == std::string::npos) return false;
        *out = unescape(tag.substr(q + 1, e - q - 1));
        return true;
      }
    }
    p += name.size();
  }
  return false;
}
}  // namespace mlp_detail

// Parses <MeshLabProject><MeshGroup><MLMesh label filename><MLMatrix44>text (first MeshGroup only).
inline bool ParseMeshLabProject(const std::string& path, std::vector<MlpMesh>* meshes) {
  std::ifstream f(path);
  if (!f) return false;
  std::stringstream ss; ss << f.rdbuf();
  const std::string doc = ss.str();
  const size_t proj = doc.find("<MeshLabProject");
  if (proj == std::string::npos) return false;
  const size_t grp = doc.find("<MeshGroup", proj);
  if (grp == std::string::npos) return false;
  size_t grp_end = doc.find("</MeshGroup>", grp);
  if (grp_end == std::string::npos) grp_end = doc.size();
  size_t p = grp;
  while (true) {
    const size_t m = doc.find("<MLMesh", p);
    if (m == std::string::npos || m >= grp_end) break;
    const size_t tag_end = doc.find('>', m);
    if (tag_end == std::string::npos) return false;
    const std::string tag = doc.substr(m, tag_end - m + 1);
    MlpMesh mesh;
    mlp_detail::attribute(tag, "label", &mesh.label);
    mlp_detail::attribute(tag, "filename", &mesh.filename);
    size_t next = tag_end + 1;
    if (tag.size() < 2 || tag[tag.size() - 2] != '/') {   // not self-closing: look for the matrix child
      size_t close = doc.find("</MLMesh>", tag_end);
      if (close == std::string::npos) close = grp_end;
      const size_t mm = doc.find("<MLMatrix44", tag_end);
      if (mm != std::string::npos && mm < close) {
        const size_t ts = doc.find('>', mm);
        const size_t te = doc.find("</MLMatrix44>", ts);
        if (ts != std::string::npos && te != std::string::npos) { mesh.has_matrix = true; mesh.matrix_text = mlp_detail::unescape(doc.substr(ts + 1, te - ts - 1)); }
      }
      next = close;
    }
    meshes->push_back(mesh);
    p = next;
  }
  return true;
}

// tinyxml2-style pretty printing (4-space indentation; element text printed verbatim).
inline bool WriteMeshLabProjectXml(const std::string& path, const std::vector<MlpMesh>& meshes) {
  std::ofstream f(path);
  if (!f) return false;
  f << "<MeshLabProject>\n    <MeshGroup>\n";
  for (const MlpMesh& m : meshes) {
    f << "        <MLMesh label=\"" << mlp_detail::escape(m.label) << "\" filename=\"" << mlp_detail::escape(m.filename) << "\">\n";
    f << "            <MLMatrix44>" << mlp_detail::escape(m.matrix_text) << "</MLMatrix44>\n";
    f << "        </MLMesh>\n";
  }
  f << "    </MeshGroup>\n</MeshLabProject>\n";
  f.close();
  return (bool)f;
}

inline std::string parent_path(const std::string& p) {
  const size_t s = p.find_last_of('/');
  if (s == std::string::npos) return "";
  return s == 0 ? "/" : p.substr(0, s);
}
inline std::string join_path(const std::string& dir, const std::string& file) {
  if (dir.empty()) return file;
  return (dir.back() == '/') ? dir + file : dir + "/" + file;
}

}  // namespace e3d_host
