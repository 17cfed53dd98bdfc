// gtn/utils.h -- the fixtures of the parity suite from reference gtn/utils.h:23-153:
// equal, isomorphic, loadTxt, saveTxt, operator<<.
#pragma once

#include <istream>
#include <ostream>
#include <sstream>
#include <string>
#include <unordered_set>

#include "gtn/graph.h"

namespace gtn {
inline bool equal(const Graph& a, const Graph& b) {
  int v;
  detail::check(gtnx_equal(a.handle(), b.handle(), &v));
  return v != 0;
}
inline bool isomorphic(const Graph& a, const Graph& b) {
  int v;
  detail::check(gtnx_isomorphic(a.handle(), b.handle(), &v));
  return v != 0;
}

/** text format of reference utils.cpp:283-345: start ids / accept ids / arcs */
inline Graph loadTxt(std::istream& in) {
  auto ints = [](const std::string& line) {
    std::vector<std::string> cols;
    std::istringstream ss(line);
    std::string tok;
    while (ss >> tok) cols.push_back(tok);
    return cols;
  };
  std::string line;
  if (!std::getline(in, line)) throw std::invalid_argument("Must specify start node(s).");
  std::vector<int> start;
  for (auto& c : ints(line)) start.push_back(std::stoi(c));
  if (!std::getline(in, line)) throw std::invalid_argument("Must specify accept node(s).");
  std::vector<int> accept;
  for (auto& c : ints(line)) accept.push_back(std::stoi(c));
  std::unordered_set<int> sset(start.begin(), start.end()), aset(accept.begin(), accept.end());
  if (sset.size() != start.size()) throw std::invalid_argument("Repeat start node detected.");
  if (aset.size() != accept.size()) throw std::invalid_argument("Repeat accept node detected.");
  int maxNode = -1;
  for (int s : start) maxNode = std::max(maxNode, s);
  for (int a : accept) maxNode = std::max(maxNode, a);
  Graph g;
  for (int i = 0; i <= maxNode; ++i) g.addNode(sset.count(i) != 0, aset.count(i) != 0);
  while (std::getline(in, line)) {
    auto cols = ints(line);
    if (cols.size() < 3 || cols.size() > 5) throw std::invalid_argument("Bad line for loading arc.");
    const int src = std::stoi(cols[0]), dst = std::stoi(cols[1]);
    for (; maxNode < std::max(src, dst); ++maxNode) g.addNode();
    if (cols.size() == 5)
      g.addArc(src, dst, std::stoi(cols[2]), std::stoi(cols[3]), std::stof(cols[4]));
    else if (cols.size() == 4)
      g.addArc(src, dst, std::stoi(cols[2]), std::stoi(cols[3]));
    else
      g.addArc(src, dst, std::stoi(cols[2]));
  }
  return g;
}
inline Graph loadTxt(std::istream&& in) { return loadTxt(in); }

inline void saveTxt(std::ostream& out, const Graph& g) {
  auto list = [&](const std::vector<int>& v) {
    for (size_t i = 0; i < v.size(); ++i) out << (i ? " " : "") << v[i];
    out << "\n";
  };
  list(g.start());
  list(g.accept());
  for (size_t a = 0; a < g.numArcs(); ++a)
    out << g.srcNode(a) << " " << g.dstNode(a) << " " << g.ilabel(a) << " " << g.olabel(a) << " " << g.weight(a)
        << "\n";
}
inline std::ostream& operator<<(std::ostream& out, const Graph& g) {
  saveTxt(out, g);
  return out;
}
} // namespace gtn
