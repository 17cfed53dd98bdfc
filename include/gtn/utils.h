// gtn/utils.h -- the fixtures of the parity suite from reference gtn/utils.h:23-153:
// equal, isomorphic, load/save (binary), loadTxt/saveTxt, operator<<, draw.
#pragma once

#include <cstring>
#include <fstream>
#include <iostream>  // as utils.h:10-12 of the reference: examples print graphs with std::cout
#include <istream>
#include <unordered_map>
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

inline Graph loadTxt(const std::string& fileName) {
  std::ifstream in(fileName);
  if (!in) throw std::invalid_argument("Couldn't find graph file to load. '" + fileName + "'");
  return loadTxt(in);
}

namespace detail {
/** text writer shared by saveTxt and operator<<; `cap` < 0 prints everything,
 *  otherwise at most `cap` entries per section (reference utils.cpp:226-262) */
inline void writeTxt(std::ostream& out, const Graph& g, int cap) {
  auto ids = [&](const std::vector<int>& v) {
    for (size_t i = 0; i < v.size(); ++i) {
      if (cap >= 0 && (int)i >= cap) {
        out << " ...";
        break;
      }
      out << (i ? " " : "") << v[i];
    }
    out << "\n";
  };
  ids(g.start());
  ids(g.accept());
  for (size_t a = 0; a < g.numArcs(); ++a) {
    if (cap >= 0 && (int)a >= cap) {
      out << "...\n";
      break;
    }
    out << g.srcNode(a) << " " << g.dstNode(a) << " " << g.ilabel(a) << " " << g.olabel(a) << " " << g.weight(a)
        << "\n";
  }
}
} // namespace detail

inline void saveTxt(std::ostream& out, const Graph& g) { detail::writeTxt(out, g, -1); }
inline void saveTxt(const std::string& fileName, const Graph& g) {
  std::ofstream out(fileName);
  saveTxt(out, g);
}
/** graphs with more than 20 nodes or arcs print a 10-entry summary (utils.cpp:391-394) */
inline std::ostream& operator<<(std::ostream& out, const Graph& g) {
  detail::writeTxt(out, g, std::max(g.numArcs(), g.numNodes()) > 20 ? 10 : -1);
  return out;
}

/** binary format of reference utils.cpp:150-224: int32 {numNodes, numArcs, numStart,
 *  numAccept}, start ids, accept ids, numArcs x {src, dst, ilabel, olabel}, weights */
inline void save(std::ostream& out, const Graph& g) {
  auto put = [&](const void* p, size_t bytes) { out.write(static_cast<const char*>(p), (std::streamsize)bytes); };
  const int head[4] = {(int)g.numNodes(), (int)g.numArcs(), (int)g.numStart(), (int)g.numAccept()};
  put(head, sizeof(head));
  put(g.start().data(), sizeof(int) * g.start().size());
  put(g.accept().data(), sizeof(int) * g.accept().size());
  // the arc table in one call (not four accessor calls per arc)
  const size_t A = g.numArcs();
  std::vector<int> src(A), dst(A), il(A), ol(A), rows(size_t(4) * A);
  if (A) detail::check(gtnx_graph_get_arcs(g.handle(), src.data(), dst.data(), il.data(), ol.data()));
  for (size_t a = 0; a < A; ++a) {
    rows[4 * a] = src[a];
    rows[4 * a + 1] = dst[a];
    rows[4 * a + 2] = il[a];
    rows[4 * a + 3] = ol[a];
  }
  put(rows.data(), sizeof(int) * rows.size());
  if (g.numArcs()) put(g.weights(), sizeof(float) * g.numArcs());
}
inline void save(const std::string& fileName, const Graph& g) {
  std::ofstream out(fileName, std::ios::binary);
  save(out, g);
}
inline Graph load(std::istream& in) {
  // utils.cpp:185-225.  The file image goes to the engine in one piece (gtnx_graph_load_buffer): the arc table and the
  // weights are copied to the device once and split into the graph's arrays there -- no call per node or arc.
  int head[4] = {0, 0, 0, 0};
  in.read(reinterpret_cast<char*>(head), sizeof(head));
  if (!in || head[0] < 0 || head[1] < 0 || head[2] < 0 || head[3] < 0) throw std::invalid_argument("[gtn::load] not a graph file");
  const size_t rest = sizeof(int) * (size_t(head[2]) + size_t(head[3])) + size_t(20) * size_t(head[1]);
  std::vector<char> image(sizeof(head) + rest);
  std::memcpy(image.data(), head, sizeof(head));
  in.read(image.data() + sizeof(head), (std::streamsize)rest);
  if (size_t(in.gcount()) != rest) throw std::invalid_argument("[gtn::load] truncated graph file");
  gtnx_graph_t h;
  detail::check(gtnx_graph_load_buffer(image.data(), image.size(), &h));
  return Graph::fromHandle(h);
}
inline Graph load(std::istream&& in) { return load(in); }
inline Graph load(const std::string& fileName) {
  std::ifstream in(fileName, std::ios::binary);
  if (!in) throw std::invalid_argument("Couldn't find graph file to load. '" + fileName + "'");
  return load(in);
}

using SymbolMap = std::unordered_map<int, std::string>;

/** graphviz dot writer (reference utils.cpp:396-461): start nodes first, accept
 *  nodes last; labels through the symbol maps, epsilon as "ε" */
inline void draw(const Graph& g, std::ostream& out, const SymbolMap& isymbols = SymbolMap(),
                 const SymbolMap& osymbols = SymbolMap()) {
  auto sym = [](const SymbolMap& m, int label) -> std::string {
    if (label == epsilon) return "ε";
    return m.empty() ? std::to_string(label) : m.at(label);
  };
  auto node = [&](int n) {
    out << "  " << n << " [label = \"" << n << "\", shape = " << (g.isAccept(n) ? "doublecircle" : "circle")
        << ", penwidth = " << (g.isStart(n) ? "2.0" : "1.0") << ", fontsize = 14];\n";
    for (int a : g.out(n)) {
      out << "  " << g.srcNode(a) << " -> " << g.dstNode(a) << " [label = \"" << sym(isymbols, g.ilabel(a));
      if (!osymbols.empty()) out << ":" << sym(osymbols, g.olabel(a));
      out << "/" << g.weight(a) << "\", fontsize = 14];\n";
    }
  };
  out << "digraph FST {\n  margin = 0;\n  rankdir = LR;\n  label = \"\";\n"
      << "  center = 1;\n  ranksep = \"0.4\";\n  nodesep = \"0.25\";\n";
  for (int n : g.start()) node(n);
  for (size_t n = 0; n < g.numNodes(); ++n)
    if (!g.isStart(n) && !g.isAccept(n)) node((int)n);
  for (int n : g.accept())
    if (!g.isStart(n)) node(n);
  out << "}";
}
inline void draw(const Graph& g, const std::string& filename, const SymbolMap& isymbols = SymbolMap(),
                 const SymbolMap& osymbols = SymbolMap()) {
  std::ofstream out(filename);
  if (!out.is_open()) throw std::runtime_error("Could not open file [" + filename + "]");
  draw(g, out, isymbols, osymbols);
}
} // namespace gtn
