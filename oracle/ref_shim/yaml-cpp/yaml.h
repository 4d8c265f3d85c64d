// -*- C++ -*-
// oracle/_ref build shim (TEST INFRASTRUCTURE): reads the calibration files CameraSystem::loadCalibInfo
// opens (scalars, one level of nested maps, flow sequences of numbers that may span lines) -- no more.
#ifndef ESVO_REF_SHIM_YAML
#define ESVO_REF_SHIM_YAML
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
namespace YAML {
class Node {
 public:
  std::string scalar;
  std::map<std::string, Node> children;
  Node operator[](const std::string& k) const {
    auto it = children.find(k);
    return it == children.end() ? Node() : it->second;
  }
  template <class T> T as() const { return conv((T*)nullptr); }
 private:
  int conv(int*) const { return std::atoi(scalar.c_str()); }
  double conv(double*) const { return std::strtod(scalar.c_str(), nullptr); }
  std::string conv(std::string*) const { return scalar; }
  std::vector<double> conv(std::vector<double>*) const {
    std::vector<double> out;
    std::string s = scalar;
    for (char& c : s) if (c == '[' || c == ']' || c == ',') c = ' ';
    std::istringstream is(s);
    std::string tok;
    while (is >> tok) out.push_back(std::strtod(tok.c_str(), nullptr));
    return out;
  }
};
inline std::string shim_trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
inline Node LoadFile(const std::string& path) {
  std::ifstream f(path);
  if (!f) std::abort();
  Node root;
  Node* parent = &root;     // where "  key: value" lines with indentation go
  std::string line, open_key;
  Node* open_node = nullptr;  // flow sequence still waiting for ']'
  while (std::getline(f, line)) {
    size_t hash = line.find('#');
    if (hash != std::string::npos) line = line.substr(0, hash);
    if (shim_trim(line).empty()) continue;
    if (open_node) {
      open_node->scalar += " " + shim_trim(line);
      if (line.find(']') != std::string::npos) open_node = nullptr;
      continue;
    }
    const size_t indent = line.find_first_not_of(' ');
    const size_t colon = line.find(':');
    if (colon == std::string::npos) continue;
    const std::string key = shim_trim(line.substr(0, colon));
    std::string val = shim_trim(line.substr(colon + 1));
    if (val.size() >= 2 && val.front() == '"' && val.back() == '"') val = val.substr(1, val.size() - 2);
    Node* where = indent == 0 ? &root : parent;
    Node& n = where->children[key];
    n.scalar = val;
    if (indent == 0 && val.empty()) parent = &n;
    if (!val.empty() && val.front() == '[' && val.find(']') == std::string::npos) open_node = &n;
  }
  return root;
}
}  // namespace YAML
#endif
