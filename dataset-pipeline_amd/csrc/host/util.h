// util.h -- command-line and string helpers with the semantics the reference's tools rely on:
// pcl::console::parse_argument ("--flag value" pairs in any order, first occurrence wins, unknown flags ignored)
// and util::SplitStringIntoSet (src/base/util.cc:85-105).
#pragma once

#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_set>

namespace e3d_host {

inline int find_argument(int argc, char** argv, const char* name) {
  for (int i = 1; i < argc; ++i)
    if (strcmp(argv[i], name) == 0) return i;
  return -1;
}
inline int parse_argument(int argc, char** argv, const char* name, std::string& val) {
  const int i = find_argument(argc, argv, name) + 1;
  if (i > 0 && i < argc) val = argv[i];
  return i - 1;
}
inline int parse_argument(int argc, char** argv, const char* name, int& val) {
  const int i = find_argument(argc, argv, name) + 1;
  if (i > 0 && i < argc) val = atoi(argv[i]);
  return i - 1;
}
// pcl::console::parse_argument(bool&): true iff atoi(value) == 1
inline int parse_argument(int argc, char** argv, const char* name, bool& val) {
  const int i = find_argument(argc, argv, name) + 1;
  if (i > 0 && i < argc) val = atoi(argv[i]) == 1;
  return i - 1;
}
inline int parse_argument(int argc, char** argv, const char* name, float& val) {
  const int i = find_argument(argc, argv, name) + 1;
  if (i > 0 && i < argc) val = static_cast<float>(atof(argv[i]));
  return i - 1;
}

inline std::unordered_set<std::string> SplitStringIntoSet(char character, const std::string& input) {
  std::unordered_set<std::string> result;
  if (input.empty()) return result;
  std::size_t index = 0;
  while (true) {
    const std::size_t new_index = input.find(character, index);
    result.insert(input.substr(index, (new_index == std::string::npos) ? std::string::npos : (new_index - index)));
    index = new_index + 1;
    if (new_index == std::string::npos || index >= input.size()) break;
  }
  return result;
}

}  // namespace e3d_host
