// io_ply.h -- PLY reader / writer for the tools (stand-in for pcl::io::loadPLYFile / savePLYFile, which the
// reference uses at src/exe/icp_scan_aligner.cc:144 and src/exe/normal_estimator.cc:86,225).
// Reader: ascii and binary_little_endian / binary_big_endian, any element order, vertex properties x y z
// [red green blue | r g b] [nx ny nz] of any scalar type (converted to f32 / u8), list properties skipped.
// Writer: the binary layout NormalEstimator emits (x y z nx ny nz f32 + red green blue u8 = 27 B/vertex) with the
// "element camera" block PCL's PLYWriter appends.
#pragma once

#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "host_types.h"

namespace e3d_host {

namespace ply_detail {
struct Prop { std::string name, type, count_type; bool is_list = false; };
struct Elem { std::string name; size_t count = 0; std::vector<Prop> props; };

inline int type_size(const std::string& t) {
  if (t == "char" || t == "uchar" || t == "int8" || t == "uint8") return 1;
  if (t == "short" || t == "ushort" || t == "int16" || t == "uint16") return 2;
  if (t == "int" || t == "uint" || t == "float" || t == "int32" || t == "uint32" || t == "float32") return 4;
  if (t == "double" || t == "float64" || t == "int64" || t == "uint64") return 8;
  return 0;
}

inline double read_scalar(const unsigned char* p, const std::string& t, bool swap) {
  unsigned char b[8];
  const int n = type_size(t);
  for (int i = 0; i < n; ++i) b[i] = swap ? p[n - 1 - i] : p[i];
  if (t == "char" || t == "int8") { int8_t v; memcpy(&v, b, 1); return v; }
  if (t == "uchar" || t == "uint8") { uint8_t v; memcpy(&v, b, 1); return v; }
  if (t == "short" || t == "int16") { int16_t v; memcpy(&v, b, 2); return v; }
  if (t == "ushort" || t == "uint16") { uint16_t v; memcpy(&v, b, 2); return v; }
  if (t == "int" || t == "int32") { int32_t v; memcpy(&v, b, 4); return v; }
  if (t == "uint" || t == "uint32") { uint32_t v; memcpy(&v, b, 4); return v; }
  if (t == "float" || t == "float32") { float v; memcpy(&v, b, 4); return v; }
  if (t == "double" || t == "float64") { double v; memcpy(&v, b, 8); return v; }
  if (t == "int64") { int64_t v; memcpy(&v, b, 8); return (double)v; }
  if (t == "uint64") { uint64_t v; memcpy(&v, b, 8); return (double)v; }
  return 0;
}
// type name -> code, resolved once per property (the per-value string comparisons of read_scalar cost 0.5 us per point)
enum TypeCode { kI8, kU8, kI16, kU16, kI32, kU32, kF32, kF64, kI64, kU64, kUnknownType };
inline TypeCode type_code(const std::string& t) {
  if (t == "char" || t == "int8") return kI8;
  if (t == "uchar" || t == "uint8") return kU8;
  if (t == "short" || t == "int16") return kI16;
  if (t == "ushort" || t == "uint16") return kU16;
  if (t == "int" || t == "int32") return kI32;
  if (t == "uint" || t == "uint32") return kU32;
  if (t == "float" || t == "float32") return kF32;
  if (t == "double" || t == "float64") return kF64;
  if (t == "int64") return kI64;
  if (t == "uint64") return kU64;
  return kUnknownType;
}
inline double read_scalar_code(const unsigned char* p, TypeCode c, bool swap) {
  static const int size_of[] = {1, 1, 2, 2, 4, 4, 4, 8, 8, 8, 0};
  unsigned char b[8];
  const int n = size_of[c];
  if (swap) { for (int i = 0; i < n; ++i) b[i] = p[n - 1 - i]; } else { memcpy(b, p, (size_t)n); }
  switch (c) {
    case kI8: { int8_t v; memcpy(&v, b, 1); return v; }
    case kU8: { uint8_t v; memcpy(&v, b, 1); return v; }
    case kI16: { int16_t v; memcpy(&v, b, 2); return v; }
    case kU16: { uint16_t v; memcpy(&v, b, 2); return v; }
    case kI32: { int32_t v; memcpy(&v, b, 4); return v; }
    case kU32: { uint32_t v; memcpy(&v, b, 4); return v; }
    case kF32: { float v; memcpy(&v, b, 4); return v; }
    case kF64: { double v; memcpy(&v, b, 8); return v; }
    case kI64: { int64_t v; memcpy(&v, b, 8); return (double)v; }
    case kU64: { uint64_t v; memcpy(&v, b, 8); return (double)v; }
    default: return 0;
  }
}
// sequential reader over an ifstream with a 4 MB window: get(n) returns n contiguous bytes or nullptr at the end of the file
struct BufReader {
  std::ifstream& f;
  std::vector<unsigned char> buf;
  size_t pos = 0, end = 0;
  explicit BufReader(std::ifstream& file) : f(file), buf((size_t)4 << 20) {}
  const unsigned char* get(size_t n) {
    if (end - pos < n) {
      if (n > buf.size()) buf.resize(n);
      memmove(buf.data(), buf.data() + pos, end - pos);
      end -= pos; pos = 0;
      f.read(reinterpret_cast<char*>(buf.data() + end), (std::streamsize)(buf.size() - end));
      end += (size_t)f.gcount();
      if (end < n) return nullptr;
    }
    const unsigned char* r = buf.data() + pos;
    pos += n;
    return r;
  }
};
// bytes between the current read position and the end of the file (0 if the stream cannot tell)
inline size_t bytes_left(std::ifstream& f) {
  const std::streampos here = f.tellg();
  if (here < 0) return 0;
  f.seekg(0, std::ios::end);
  const std::streampos end = f.tellg();
  f.seekg(here);
  return (end < here) ? 0 : (size_t)(end - here);
}
// Every record of an element takes at least one byte, so a count beyond the rest of the file (a negative count read into
// size_t, a damaged header) cannot be honest: refuse it before anything is sized after it.
inline bool counts_fit(const std::vector<Elem>& elems, size_t left) {
  for (const Elem& e : elems)
    if (e.count > left) return false;
  return true;
}
// a list count as read from the file: non-negative, and the list must fit in what is left of the file
inline bool list_count_ok(double v, size_t item_size, size_t left, size_t* c) {
  if (!(v >= 0) || v > (double)left) return false;
  *c = (size_t)v;
  return *c <= left / (item_size ? item_size : 1);
}
}  // namespace ply_detail

// Returns 0 on success, < 0 on failure (like pcl::io::loadPLYFile).
inline int loadPLYFile(const std::string& path, PointCloud& cloud, bool want_rgb = false) {
  using namespace ply_detail;
  std::ifstream f(path, std::ios::binary);
  if (!f) { std::cerr << "[loadPLYFile] cannot open " << path << std::endl; return -1; }
  std::string line;
  if (!std::getline(f, line) || line.substr(0, 3) != "ply") { std::cerr << "[loadPLYFile] not a PLY file: " << path << std::endl; return -1; }
  std::string format;
  std::vector<Elem> elems;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::istringstream ls(line);
    std::string tok;
    ls >> tok;
    if (tok == "format") ls >> format;
    else if (tok == "element") { Elem e; ls >> e.name >> e.count; elems.push_back(e); }
    else if (tok == "property" && !elems.empty()) {
      Prop p; std::string t; ls >> t;
      if (t == "list") { p.is_list = true; ls >> p.count_type >> p.type >> p.name; }
      else { p.type = t; ls >> p.name; }
      elems.back().props.push_back(p);
    } else if (tok == "end_header") break;
  }
  const bool ascii = (format == "ascii");
  const bool swap = (format == "binary_big_endian");
  if (!ascii && format != "binary_little_endian" && !swap) { std::cerr << "[loadPLYFile] unsupported format '" << format << "'" << std::endl; return -1; }
  const size_t body_bytes = bytes_left(f);
  if (!counts_fit(elems, body_bytes)) { std::cerr << "[loadPLYFile] element count beyond the end of " << path << std::endl; return -1; }
  for (const Elem& e : elems) {
    const bool is_vertex = (e.name == "vertex");
    int ix = -1, iy = -1, iz = -1, ir = -1, ig = -1, ib = -1, inx = -1, iny = -1, inz = -1, iint = -1;
    bool has_list = false;
    for (size_t i = 0; i < e.props.size(); ++i) {
      const std::string& nm = e.props[i].name;
      if (e.props[i].is_list) has_list = true;
      if (nm == "x") ix = (int)i; else if (nm == "y") iy = (int)i; else if (nm == "z") iz = (int)i;
      else if (nm == "red" || nm == "r" || nm == "diffuse_red") ir = (int)i;
      else if (nm == "green" || nm == "g" || nm == "diffuse_green") ig = (int)i;
      else if (nm == "blue" || nm == "b" || nm == "diffuse_blue") ib = (int)i;
      else if (nm == "nx" || nm == "normal_x") inx = (int)i; else if (nm == "ny" || nm == "normal_y") iny = (int)i;
      else if (nm == "nz" || nm == "normal_z") inz = (int)i;
      else if (nm == "intensity") iint = (int)i;
    }
    if (is_vertex) {
      if (ix < 0 || iy < 0 || iz < 0) { std::cerr << "[loadPLYFile] vertex element without x/y/z in " << path << std::endl; return -1; }
      cloud.xyz.resize(3 * e.count);
      if (want_rgb) cloud.rgb.assign(3 * e.count, 0);
      const bool nrm = inx >= 0 && iny >= 0 && inz >= 0;
      if (nrm) cloud.normals.resize(3 * e.count);
      if (iint >= 0) cloud.intensity.resize(e.count);
    }
    std::vector<double> vals(e.props.size());
    if (ascii) {
      for (size_t i = 0; i < e.count; ++i) {
        if (!std::getline(f, line)) { std::cerr << "[loadPLYFile] truncated file " << path << std::endl; return -1; }
        if (!is_vertex) continue;
        std::istringstream ls(line);
        for (size_t p = 0; p < e.props.size(); ++p) {
          if (e.props[p].is_list) { double cv = -1; size_t c = 0; ls >> cv; if (!ls || !list_count_ok(cv, 1, line.size(), &c)) { std::cerr << "[loadPLYFile] bad list count in " << path << std::endl; return -1; } double d; for (size_t k = 0; k < c; ++k) ls >> d; vals[p] = 0; }
          else ls >> vals[p];
        }
        cloud.xyz[3 * i] = (float)vals[ix]; cloud.xyz[3 * i + 1] = (float)vals[iy]; cloud.xyz[3 * i + 2] = (float)vals[iz];
        if (!cloud.rgb.empty() && ir >= 0 && ig >= 0 && ib >= 0) { cloud.rgb[3 * i] = (uint8_t)vals[ir]; cloud.rgb[3 * i + 1] = (uint8_t)vals[ig]; cloud.rgb[3 * i + 2] = (uint8_t)vals[ib]; }
        if (!cloud.normals.empty()) { cloud.normals[3 * i] = (float)vals[inx]; cloud.normals[3 * i + 1] = (float)vals[iny]; cloud.normals[3 * i + 2] = (float)vals[inz]; }
        if (iint >= 0) cloud.intensity[i] = (float)vals[iint];
      }
    } else if (!has_list) {
      size_t stride = 0;
      std::vector<size_t> off(e.props.size());
      for (size_t p = 0; p < e.props.size(); ++p) { off[p] = stride; const int ts = type_size(e.props[p].type); if (!ts) { std::cerr << "[loadPLYFile] unknown type " << e.props[p].type << std::endl; return -1; } stride += ts; }
      if (stride * e.count > bytes_left(f)) { std::cerr << "[loadPLYFile] truncated file " << path << std::endl; return -1; }
      if (!is_vertex) { f.seekg((std::streamoff)(stride * e.count), std::ios::cur); continue; }
      std::vector<TypeCode> code(e.props.size());
      for (size_t p = 0; p < e.props.size(); ++p) code[p] = type_code(e.props[p].type);
      const size_t chunk = 1 << 20;
      std::vector<unsigned char> buf(stride * std::min(chunk, e.count ? e.count : 1));
      for (size_t i0 = 0; i0 < e.count; i0 += chunk) {
        const size_t m = std::min(chunk, e.count - i0);
        f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)(stride * m));
        if ((size_t)f.gcount() != stride * m) { std::cerr << "[loadPLYFile] truncated file " << path << std::endl; return -1; }
        for (size_t j = 0; j < m; ++j) {
          const unsigned char* r = buf.data() + stride * j;
          const size_t i = i0 + j;
          cloud.xyz[3 * i] = (float)read_scalar_code(r + off[ix], code[ix], swap);
          cloud.xyz[3 * i + 1] = (float)read_scalar_code(r + off[iy], code[iy], swap);
          cloud.xyz[3 * i + 2] = (float)read_scalar_code(r + off[iz], code[iz], swap);
          if (!cloud.rgb.empty() && ir >= 0 && ig >= 0 && ib >= 0) {
            cloud.rgb[3 * i] = (uint8_t)read_scalar_code(r + off[ir], code[ir], swap);
            cloud.rgb[3 * i + 1] = (uint8_t)read_scalar_code(r + off[ig], code[ig], swap);
            cloud.rgb[3 * i + 2] = (uint8_t)read_scalar_code(r + off[ib], code[ib], swap);
          }
          if (!cloud.normals.empty()) {
            cloud.normals[3 * i] = (float)read_scalar_code(r + off[inx], code[inx], swap);
            cloud.normals[3 * i + 1] = (float)read_scalar_code(r + off[iny], code[iny], swap);
            cloud.normals[3 * i + 2] = (float)read_scalar_code(r + off[inz], code[inz], swap);
          }
          if (iint >= 0) cloud.intensity[i] = (float)read_scalar_code(r + off[iint], code[iint], swap);
        }
      }
    } else {
      // binary element with list properties (faces): walk it record by record
      size_t left = bytes_left(f);
      for (size_t i = 0; i < e.count; ++i)
        for (size_t p = 0; p < e.props.size(); ++p) {
          unsigned char b[8];
          if (e.props[p].is_list) {
            const int cs = type_size(e.props[p].count_type), ts = type_size(e.props[p].type);
            if (!cs || !ts) { std::cerr << "[loadPLYFile] unknown list type in " << path << std::endl; return -1; }
            f.read(reinterpret_cast<char*>(b), cs);
            size_t c = 0;
            if (!f || (int)f.gcount() != cs || !list_count_ok(read_scalar(b, e.props[p].count_type, swap), (size_t)ts, left - (size_t)cs, &c)) {
              std::cerr << "[loadPLYFile] truncated file or bad list count in " << path << std::endl; return -1;
            }
            f.seekg((std::streamoff)(c * (size_t)ts), std::ios::cur);
            left -= (size_t)cs + c * (size_t)ts;
          } else {
            const int ts = type_size(e.props[p].type);
            if (!ts || (size_t)ts > left) { std::cerr << "[loadPLYFile] truncated file " << path << std::endl; return -1; }
            f.seekg(ts, std::ios::cur);
            left -= (size_t)ts;
          }
          if (!f) { std::cerr << "[loadPLYFile] truncated file " << path << std::endl; return -1; }
        }
      if (is_vertex) { std::cerr << "[loadPLYFile] list properties in the vertex element are not supported" << std::endl; return -1; }
    }
    if (is_vertex) return 0;   // everything after the vertices (faces, camera) is irrelevant for the tools
  }
  std::cerr << "[loadPLYFile] no vertex element in " << path << std::endl;
  return -1;
}

// Triangle mesh: vertices (x y z) + faces (list property, 3 indices each) -- pcl::io::loadPLYFile(path, pcl::PolygonMesh&) as
// OcclusionGeometry::AddMeshPLY uses it (src/opt/occlusion_geometry.cc:118-139); non-triangular faces are an error like the
// reference's CHECK_EQ(face_vertices.size(), 3) (:508).
inline int loadPLYMesh(const std::string& path, std::vector<float>& xyz, std::vector<uint32_t>& triangles) {
  using namespace ply_detail;
  std::ifstream f(path, std::ios::binary);
  if (!f) { std::cerr << "[loadPLYMesh] cannot open " << path << std::endl; return -1; }
  std::string line;
  if (!std::getline(f, line) || line.substr(0, 3) != "ply") { std::cerr << "[loadPLYMesh] not a PLY file: " << path << std::endl; return -1; }
  std::string format;
  std::vector<Elem> elems;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    std::istringstream ls(line);
    std::string tok;
    ls >> tok;
    if (tok == "format") ls >> format;
    else if (tok == "element") { Elem e; ls >> e.name >> e.count; elems.push_back(e); }
    else if (tok == "property" && !elems.empty()) {
      Prop p; std::string t; ls >> t;
      if (t == "list") { p.is_list = true; ls >> p.count_type >> p.type >> p.name; }
      else { p.type = t; ls >> p.name; }
      elems.back().props.push_back(p);
    } else if (tok == "end_header") break;
  }
  const bool ascii = (format == "ascii"), swap = (format == "binary_big_endian");
  if (!ascii && format != "binary_little_endian" && !swap) { std::cerr << "[loadPLYMesh] unsupported format '" << format << "'" << std::endl; return -1; }
  xyz.clear(); triangles.clear();
  const size_t body_bytes = bytes_left(f);
  if (!counts_fit(elems, body_bytes)) { std::cerr << "[loadPLYMesh] element count beyond the end of " << path << std::endl; return -1; }
  size_t consumed = 0;              // binary body bytes handed out so far
  BufReader reader(f);               // binary files: everything after the header goes through it
  for (const Elem& e : elems) {
    const bool is_vertex = (e.name == "vertex"), is_face = (e.name == "face");
    int ix = -1, iy = -1, iz = -1, il = -1;
    for (size_t i = 0; i < e.props.size(); ++i) {
      const std::string& nm = e.props[i].name;
      if (nm == "x") ix = (int)i; else if (nm == "y") iy = (int)i; else if (nm == "z") iz = (int)i;
      if (e.props[i].is_list && (nm == "vertex_indices" || nm == "vertex_index")) il = (int)i;
    }
    if (is_vertex) { if (ix < 0 || iy < 0 || iz < 0) return -1; xyz.resize(3 * e.count); }
    if (is_face) triangles.reserve(3 * e.count);
    if (!ascii) {
      // binary: one buffered pass with the property types resolved up front (a mesh of 10^8 faces is 10^9 values)
      std::vector<TypeCode> code(e.props.size()), ccode(e.props.size());
      std::vector<int> ts(e.props.size()), cs(e.props.size());
      for (size_t p = 0; p < e.props.size(); ++p) {
        code[p] = type_code(e.props[p].type); ts[p] = type_size(e.props[p].type);
        ccode[p] = e.props[p].is_list ? type_code(e.props[p].count_type) : kU8; cs[p] = e.props[p].is_list ? type_size(e.props[p].count_type) : 0;
        if (!ts[p] || (e.props[p].is_list && !cs[p])) { std::cerr << "[loadPLYMesh] unknown property type in " << path << std::endl; return -1; }
      }
      for (size_t i = 0; i < e.count; ++i)
        for (size_t p = 0; p < e.props.size(); ++p) {
          if (e.props[p].is_list) {
            const unsigned char* cb = reader.get((size_t)cs[p]);
            if (!cb) { std::cerr << "[loadPLYMesh] truncated file " << path << std::endl; return -1; }
            consumed += (size_t)cs[p];
            size_t c = 0;
            if (!list_count_ok(read_scalar_code(cb, ccode[p], swap), (size_t)ts[p], body_bytes - std::min(consumed, body_bytes), &c)) {
              std::cerr << "[loadPLYMesh] truncated file or bad list count in " << path << std::endl; return -1;
            }
            const unsigned char* d = reader.get(c * (size_t)ts[p]);
            if (!d && c) { std::cerr << "[loadPLYMesh] truncated file " << path << std::endl; return -1; }
            consumed += c * (size_t)ts[p];
            if (is_face && (int)p == il) {
              if (c != 3) { std::cerr << "[loadPLYMesh] only triangle meshes are supported: " << path << std::endl; return -1; }
              for (size_t k = 0; k < 3; ++k) triangles.push_back((uint32_t)read_scalar_code(d + k * (size_t)ts[p], code[p], swap));
            }
          } else {
            const unsigned char* d = reader.get((size_t)ts[p]);
            if (!d) { std::cerr << "[loadPLYMesh] truncated file " << path << std::endl; return -1; }
            consumed += (size_t)ts[p];
            if (is_vertex && ((int)p == ix || (int)p == iy || (int)p == iz))
              xyz[3 * i + ((int)p == ix ? 0 : ((int)p == iy ? 1 : 2))] = (float)read_scalar_code(d, code[p], swap);
          }
        }
      continue;
    }
    for (size_t i = 0; i < e.count; ++i) {
      std::istringstream ls;
      if (ascii) { if (!std::getline(f, line)) return -1; ls.str(line); }
      for (size_t p = 0; p < e.props.size(); ++p) {
        const Prop& pr = e.props[p];
        if (pr.is_list) {
          size_t c = 0;
          std::vector<double> idx;
          if (ascii) {
            double cv = -1;
            ls >> cv;
            if (!ls || !list_count_ok(cv, 1, line.size(), &c)) { std::cerr << "[loadPLYMesh] bad list count in " << path << std::endl; return -1; }
            idx.resize(c);
            for (size_t k = 0; k < c; ++k) ls >> idx[k];
          }
          else {
            unsigned char b[8];
            f.read(reinterpret_cast<char*>(b), type_size(pr.count_type));
            c = (size_t)read_scalar(b, pr.count_type, swap);
            idx.resize(c);
            for (size_t k = 0; k < c; ++k) { f.read(reinterpret_cast<char*>(b), type_size(pr.type)); idx[k] = read_scalar(b, pr.type, swap); }
          }
          if (is_face && (int)p == il) {
            if (c != 3) { std::cerr << "[loadPLYMesh] only triangle meshes are supported: " << path << std::endl; return -1; }
            for (size_t k = 0; k < 3; ++k) triangles.push_back((uint32_t)idx[k]);
          }
        } else {
          double v = 0;
          if (ascii) ls >> v;
          else { unsigned char b[8]; f.read(reinterpret_cast<char*>(b), type_size(pr.type)); v = read_scalar(b, pr.type, swap); }
          if (is_vertex) { if ((int)p == ix) xyz[3 * i] = (float)v; else if ((int)p == iy) xyz[3 * i + 1] = (float)v; else if ((int)p == iz) xyz[3 * i + 2] = (float)v; }
        }
      }
      if (!f && !ascii) { std::cerr << "[loadPLYMesh] truncated file " << path << std::endl; return -1; }
    }
  }
  if (xyz.empty() || triangles.empty()) { std::cerr << "[loadPLYMesh] no vertices or faces in " << path << std::endl; return -1; }
  return 0;
}

// x y z nx ny nz (f32) + red green blue (u8), binary little endian, PCL-style header incl. the camera element.
inline int savePLYFileBinaryXYZNormalRGB(const std::string& path, const std::vector<unsigned char>& data, size_t n) {
  std::ofstream f(path, std::ios::binary);
  if (!f) { std::cerr << "[savePLYFile] cannot open " << path << std::endl; return -1; }
  f << "ply\nformat binary_little_endian 1.0\ncomment PCL generated\nelement vertex " << n << "\n"
    << "property float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n"
    << "property uchar red\nproperty uchar green\nproperty uchar blue\n"
    << "element camera 1\nproperty float view_px\nproperty float view_py\nproperty float view_pz\n"
    << "property float x_axisx\nproperty float x_axisy\nproperty float x_axisz\n"
    << "property float y_axisx\nproperty float y_axisy\nproperty float y_axisz\n"
    << "property float z_axisx\nproperty float z_axisy\nproperty float z_axisz\n"
    << "property float focal\nproperty float scalex\nproperty float scaley\nproperty float centerx\nproperty float centery\n"
    << "property int viewportx\nproperty int viewporty\nproperty float k1\nproperty float k2\nend_header\n";
  f.write(reinterpret_cast<const char*>(data.data()), (std::streamsize)data.size());
  const float cam_f[12] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1};   // origin, identity orientation
  f.write(reinterpret_cast<const char*>(cam_f), sizeof cam_f);
  const float zeros5[5] = {0, 0, 0, 0, 0};
  f.write(reinterpret_cast<const char*>(zeros5), sizeof zeros5);
  const int32_t vp[2] = {(int32_t)n, 1};
  f.write(reinterpret_cast<const char*>(vp), sizeof vp);
  const float zeros2[2] = {0, 0};
  f.write(reinterpret_cast<const char*>(zeros2), sizeof zeros2);
  return f ? 0 : -1;
}

// pcl::io::savePLYFileBinary of a pcl::PointXYZRGB cloud (point_cloud_cleaner.cc:132-133): x y z f32 + red green blue u8
// (15 B/vertex) and PCL's camera element.  Clouds without colours get 0 0 0 (the packed rgb field of points loaded from a
// colourless file).
inline int savePLYFileBinaryXYZRGB(const std::string& path, const std::vector<float>& xyz, const std::vector<uint8_t>& rgb) {
  std::ofstream f(path, std::ios::binary);
  if (!f) { std::cerr << "[savePLYFile] cannot open " << path << std::endl; return -1; }
  const size_t n = xyz.size() / 3;
  f << "ply\nformat binary_little_endian 1.0\ncomment PCL generated\nelement vertex " << n << "\n"
    << "property float x\nproperty float y\nproperty float z\n"
    << "property uchar red\nproperty uchar green\nproperty uchar blue\n"
    << "element camera 1\nproperty float view_px\nproperty float view_py\nproperty float view_pz\n"
    << "property float x_axisx\nproperty float x_axisy\nproperty float x_axisz\n"
    << "property float y_axisx\nproperty float y_axisy\nproperty float y_axisz\n"
    << "property float z_axisx\nproperty float z_axisy\nproperty float z_axisz\n"
    << "property float focal\nproperty float scalex\nproperty float scaley\nproperty float centerx\nproperty float centery\n"
    << "property int viewportx\nproperty int viewporty\nproperty float k1\nproperty float k2\nend_header\n";
  std::vector<unsigned char> row(15 * n);
  const bool colours = rgb.size() == 3 * n;
  for (size_t i = 0; i < n; ++i) {
    memcpy(&row[15 * i], &xyz[3 * i], 12);
    for (int c = 0; c < 3; ++c) row[15 * i + 12 + c] = colours ? rgb[3 * i + c] : 0;
  }
  f.write(reinterpret_cast<const char*>(row.data()), (std::streamsize)row.size());
  const float cam_f[12] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1};
  f.write(reinterpret_cast<const char*>(cam_f), sizeof cam_f);
  const float zeros5[5] = {0, 0, 0, 0, 0};
  f.write(reinterpret_cast<const char*>(zeros5), sizeof zeros5);
  const int32_t vp[2] = {(int32_t)n, 1};
  f.write(reinterpret_cast<const char*>(vp), sizeof vp);
  const float zeros2[2] = {0, 0};
  f.write(reinterpret_cast<const char*>(zeros2), sizeof zeros2);
  return f ? 0 : -1;
}

// x y z intensity (f32) binary little endian: the pcl::PointXYZI files of the multi-resolution cloud cache (problem.cc:364-411)
inline int savePLYFileBinaryXYZI(const std::string& path, const std::vector<float>& xyz, const std::vector<float>& intensity) {
  std::ofstream f(path, std::ios::binary);
  if (!f) return -1;
  const size_t n = xyz.size() / 3;
  f << "ply\nformat binary_little_endian 1.0\ncomment PCL generated\nelement vertex " << n
    << "\nproperty float x\nproperty float y\nproperty float z\nproperty float intensity\nend_header\n";
  for (size_t i = 0; i < n; ++i) {
    const float v[4] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], intensity[i]};
    f.write(reinterpret_cast<const char*>(v), sizeof v);
  }
  return f ? 0 : -1;
}

// pcl::io::savePLYFileBinary of a pcl::PointXYZ cloud (ground_truth_creator.cc:414): x y z f32 and PCL's camera element
inline int savePLYFileBinaryXYZPcl(const std::string& path, const std::vector<float>& xyz) {
  std::ofstream f(path, std::ios::binary);
  if (!f) { std::cerr << "[savePLYFile] cannot open " << path << std::endl; return -1; }
  const size_t n = xyz.size() / 3;
  f << "ply\nformat binary_little_endian 1.0\ncomment PCL generated\nelement vertex " << n << "\n"
    << "property float x\nproperty float y\nproperty float z\n"
    << "element camera 1\nproperty float view_px\nproperty float view_py\nproperty float view_pz\n"
    << "property float x_axisx\nproperty float x_axisy\nproperty float x_axisz\n"
    << "property float y_axisx\nproperty float y_axisy\nproperty float y_axisz\n"
    << "property float z_axisx\nproperty float z_axisy\nproperty float z_axisz\n"
    << "property float focal\nproperty float scalex\nproperty float scaley\nproperty float centerx\nproperty float centery\n"
    << "property int viewportx\nproperty int viewporty\nproperty float k1\nproperty float k2\nend_header\n";
  f.write(reinterpret_cast<const char*>(xyz.data()), (std::streamsize)(xyz.size() * sizeof(float)));
  const float cam_f[12] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1};
  f.write(reinterpret_cast<const char*>(cam_f), sizeof cam_f);
  const float zeros5[5] = {0, 0, 0, 0, 0};
  f.write(reinterpret_cast<const char*>(zeros5), sizeof zeros5);
  const int32_t vp[2] = {(int32_t)n, 1};
  f.write(reinterpret_cast<const char*>(vp), sizeof vp);
  const float zeros2[2] = {0, 0};
  f.write(reinterpret_cast<const char*>(zeros2), sizeof zeros2);
  return f ? 0 : -1;
}

// minimal writer used by tests / examples: x y z as binary f32
inline int savePLYFileBinaryXYZ(const std::string& path, const std::vector<float>& xyz) {
  std::ofstream f(path, std::ios::binary);
  if (!f) return -1;
  f << "ply\nformat binary_little_endian 1.0\nelement vertex " << xyz.size() / 3 << "\nproperty float x\nproperty float y\nproperty float z\nend_header\n";
  f.write(reinterpret_cast<const char*>(xyz.data()), (std::streamsize)(xyz.size() * sizeof(float)));
  return f ? 0 : -1;
}

}  // namespace e3d_host
