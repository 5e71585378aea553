// ingest.cpp -- host side of the path in C++ (the reference's host code is C++
// too): GFA + GAF text -> ChainBatch.
//
// The hot path's results are only comparable with the reference if chains,
// bubbles and alleles carry the SAME ids, and those ids are a by-product of
// how the reference walks its graph:
//   * Graph::ReadGraph            src/graph.cpp:188-249
//   * Graph::findBubbles/findBubble   src/graph.cpp:333-501 -- chains are
//     discovered while iterating a std::unordered_map<int,Node>, so chain ids
//     follow libstdc++'s bucket order; a chain's bubbles are numbered in walk
//     order (direction 0 from the start node first, then direction 1).
//   * ChainsToReadsetDetailed     src/chainstoreadset.cpp:161-203 -- allele
//     order: inner nodes in visit order for 2-inner-node bubbles, DFS from the
//     sink otherwise (paths then run sink -> source, quirk Q7).
//   * AlignmentReader::readAlignmentfile   src/alignmentreader.cpp:69-186 --
//     16th whitespace token is id:f:<float>; a line is filed under every chain
//     one of its nodes belongs to (nodes of no chain count as chain 0).
// This file re-implements that behaviour on its own data structures; it keeps
// an unordered_map<int,...> keyed and filled exactly like the reference's so
// that the iteration order -- and with it every id -- comes out the same under
// the same libstdc++.  tests/test_ingest_golden.py pins it against the
// reference's own compiled parsers (oracle/_ref/ref_dump) on committed
// fixtures.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <new>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "host_batch.h"

namespace {

struct Link { int to; int plus; };  // child id and whether the child is entered on its '+' strand

struct GNode {
  int id = 0;
  int bubble = 0, chain = 0;
  bool visited = false;
  std::vector<Link> out_plus;   // links leaving the '+' strand  (reference: childrenleft)
  std::vector<Link> out_minus;  // links leaving the '-' strand  (reference: childrenright)
};

struct GBubble {
  int id = 0;
  int source = 0, sink = 0;
  std::vector<int> inner;
  bool has(int nid) const {
    return nid == source || nid == sink || std::find(inner.begin(), inner.end(), nid) != inner.end();
  }
};

struct GChain {
  int id = 0;
  std::vector<GBubble> bubbles;
};

struct Oriented {
  int id;
  bool plus;
  bool operator==(const Oriented& o) const { return id == o.id && plus == o.plus; }
};
struct OrientedHash {
  size_t operator()(const Oriented& x) const { return std::hash<int>()(x.id) ^ std::hash<bool>()(x.plus); }
};

int digits_of(const std::string& s, bool* ok) {
  std::string d;
  for (char c : s)
    if (c >= '0' && c <= '9') d.push_back(c);
  if (d.empty() || d.size() > 9) { *ok = false; return 0; }
  return std::atoi(d.c_str());
}

struct Graph {
  std::unordered_map<int, GNode> nodes;  // same key type, same insertion sequence as the reference
  std::unordered_map<Oriented, std::vector<Oriented>, OrientedHash> edges;
  std::vector<GChain> chains;

  int read_gfa(const char* path) {
    std::ifstream f(path);
    if (!f) return AHSOKA_E_IO;
    std::string line;
    while (std::getline(f, line)) {
      if (!f.good()) break;  // a last line without '\n' is dropped, as in the reference (:193)
      if (line.empty()) continue;
      if (line[0] == 'S') {
        std::stringstream ss(line);
        std::string tag, name, seq;
        ss >> tag >> name >> seq;
        bool ok = true;
        int id = digits_of(name, &ok);
        if (!ok || tag != "S" || seq.empty()) return AHSOKA_E_PARSE;
        GNode nd;
        nd.id = id;
        nodes[id] = nd;
      } else if (line[0] == 'L') {
        std::stringstream ss(line);
        std::string tag, a, ao, b, bo;
        int ov = -1;
        char unit = 0;
        ss >> tag >> a >> ao >> b >> bo >> ov >> unit;
        bool ok = true;
        int ida = digits_of(a, &ok), idb = digits_of(b, &ok);
        if (!ok || tag != "L" || (ao != "+" && ao != "-") || (bo != "+" && bo != "-") || ov < 0) return AHSOKA_E_PARSE;
        if (!(unit == 'M' || (unit == 'S' && ov == 0))) return AHSOKA_E_PARSE;
        Oriented from{ida, ao == "+"}, to{idb, bo == "+"};
        edges[from].push_back(to);
        // source node first, then the target: both lookups may create a default
        // node (id 0) when an L line precedes its S line (quirk Q15)
        GNode& src = nodes[ida];
        const int child_id = nodes[idb].id;
        if (from.plus) src.out_plus.push_back(Link{child_id, to.plus ? 1 : 0});
        else src.out_minus.push_back(Link{child_id, to.plus ? 1 : 0});
      }
    }
    return AHSOKA_OK;
  }

  // One superbubble walk starting at `start` (a snapshot of the node) along
  // `dir` (false: follow out_plus, true: follow out_minus); appends every bubble
  // found to `chain` and continues from each bubble's sink.
  void walk(const GNode start, bool dir, GChain* chain) {
    std::unordered_set<Oriented, OrientedHash> seen;
    std::unordered_set<int> done;
    std::vector<GNode> inside;
    seen.insert(Oriented{start.id, dir});
    // frontier ordered by (node id, direction): the reference keeps a std::set of (Node,bool)
    std::map<std::pair<int, bool>, GNode> frontier;
    frontier.emplace(std::make_pair(start.id, dir), start);
    while (!frontier.empty()) {
      auto first = frontier.begin();
      const GNode v = first->second;
      const bool vdir = first->first.second;
      frontier.erase(first);
      done.insert(v.id);
      nodes[v.id].visited = true;
      inside.push_back(v);
      seen.erase(Oriented{v.id, vdir});
      const std::vector<Link>& next = vdir ? v.out_minus : v.out_plus;
      if (next.empty()) break;  // tip
      for (const Link& u : next) {
        const bool udir = (u.plus == 0);
        const std::vector<Link> parents = udir ? nodes[u.to].out_plus : nodes[u.to].out_minus;
        if (u.to == start.id) {  // walked back into the start: a loop, give up
          frontier.clear();
          break;
        }
        seen.insert(Oriented{u.to, udir});
        bool ready = true;
        for (const Link& p : parents)
          if (!done.count(p.to)) ready = false;
        if (ready) frontier.emplace(std::make_pair(nodes[u.to].id, udir), nodes[u.to]);
      }
      if (frontier.size() == 1 && seen.size() == 1) {
        auto tit = frontier.begin();
        const GNode t = tit->second;
        const bool tdir = tit->first.second;
        frontier.erase(tit);
        inside.push_back(t);
        if (inside.size() == 2) break;
        auto drop = [&](int nid) {
          for (size_t i = 0; i < inside.size(); i++)
            if (inside[i].id == nid) { inside.erase(inside.begin() + (long)i); break; }
        };
        drop(start.id);
        drop(t.id);
        GBubble b;
        b.source = start.id;
        b.sink = t.id;
        for (const GNode& x : inside) b.inner.push_back(x.id);
        chain->bubbles.push_back(b);
        walk(t, tdir, chain);
      }
    }
  }

  void find_chains() {
    for (auto entry : nodes) {  // by value: the visited flag is the one at this moment
      GChain ch;
      if (!entry.second.visited) {
        walk(entry.second, false, &ch);
        walk(entry.second, true, &ch);
        if (!ch.bubbles.empty()) chains.push_back(ch);
      }
    }
    int cid = 0;
    for (GChain& ch : chains) {
      ch.id = cid;
      int bid = 0;
      for (GBubble& b : ch.bubbles) {
        b.id = bid;
        auto mark = [&](int nid) { nodes[nid].chain = cid; nodes[nid].bubble = bid; };
        mark(b.source);
        for (int x : b.inner) mark(x);
        mark(b.sink);
        bid++;
      }
      cid++;
    }
  }

  // allele paths of a bubble with != 2 inner nodes: depth-first from the sink
  void dfs_paths(const GBubble& b, int nid, int dirflag, std::vector<int>* seq, std::vector<std::vector<int>>* out) {
    if (std::find(seq->begin(), seq->end(), nid) == seq->end()) seq->push_back(nid);
    const GNode& nd = nodes[nid];
    const std::vector<Link>& next = (dirflag == 0) ? nd.out_minus : nd.out_plus;
    bool within = true;
    for (const Link& ch : next)
      if (!b.has(ch.to)) within = false;
    if (!next.empty() && within) {
      for (const Link& ch : next) {
        size_t idx = (size_t)(std::find(seq->begin(), seq->end(), nid) - seq->begin());
        std::vector<int> prefix(seq->begin(), seq->begin() + (long)idx + 1);
        dfs_paths(b, ch.to, ch.plus, &prefix, out);
      }
    } else {
      out->push_back(*seq);
    }
  }

  std::vector<std::vector<int>> allele_paths(const GBubble& b) {
    std::vector<std::vector<int>> paths;
    if (b.inner.size() == 2) {
      for (int x : b.inner) paths.push_back({b.source, x, b.sink});
      return paths;
    }
    bool minus_within = true;
    for (const Link& ch : nodes[b.sink].out_minus)
      if (!b.has(ch.to)) minus_within = false;
    const int dirflag = minus_within ? 0 : 1;
    std::vector<int> seq;
    dfs_paths(b, b.sink, dirflag, &seq, &paths);
    return paths;
  }

  // orientation printed for `a` when followed by `b` (src/alignmentstoreadset.cpp:441-446)
  char step_sign(int a, int b) {
    for (bool plus : {true, false}) {
      auto it = edges.find(Oriented{a, plus});
      if (it == edges.end()) continue;
      for (const Oriented& to : it->second)
        if (to.id == b) return plus ? '+' : '-';
    }
    return '?';  // the reference's getEdge falls off its end here (undefined)
  }
};

struct GafLine {
  std::string name;
  float identity;
  std::vector<int> node_ids;
};

int read_gaf(const char* path, std::vector<GafLine>* out) {
  std::ifstream f(path);
  if (!f) return AHSOKA_E_IO;
  std::string line;
  while (std::getline(f, line)) {
    if (!f.good()) break;
    if (line.empty()) continue;
    std::stringstream ss(line);
    std::string tok[16];
    int nt = 0;
    while (nt < 16 && (ss >> tok[nt])) nt++;
    if (nt < 16) return AHSOKA_E_PARSE;
    // token 16: "id:f:<float>" -- split on ':', first field must be "id", value is the last
    const std::string& idtok = tok[15];
    size_t c1 = idtok.find(':');
    if (c1 == std::string::npos || idtok.substr(0, c1) != "id") return AHSOKA_E_PARSE;
    size_t cl = idtok.rfind(':');
    GafLine gl;
    try {
      gl.identity = std::stof(idtok.substr(cl + 1));
      (void)std::stoi(tok[7]);
      (void)std::stoi(tok[8]);
    } catch (...) {
      return AHSOKA_E_PARSE;
    }
    gl.name = tok[0];
    const std::string& p = tok[5];
    size_t pos = 0;
    while (pos < p.size()) {
      while (pos < p.size() && (p[pos] == '<' || p[pos] == '>')) pos++;
      size_t beg = pos;
      while (pos < p.size() && p[pos] != '<' && p[pos] != '>') pos++;
      if (pos > beg) {
        bool ok = true;
        int id = digits_of(p.substr(beg, pos - beg), &ok);
        if (!ok) return AHSOKA_E_PARSE;
        gl.node_ids.push_back(id);
      }
    }
    out->push_back(std::move(gl));
  }
  return AHSOKA_OK;
}

}  // namespace

// Text dump of everything the ingest derived, in the line format of
// oracle/ref_harness.cpp (CHAIN / BUBBLE / ALLELE / ALN / ORDER), so the two can
// be compared byte for byte; also the source of <o>-bubbleinfo.txt.
extern "C" int ahsoka_ingest_dump(const char* gfa_path, const char* gaf_path, const char* out_path) {
  if (!gfa_path || !out_path) return AHSOKA_E_ARG;
  try {
    Graph g;
    if (int e = g.read_gfa(gfa_path)) return e;
    g.find_chains();
    std::vector<GafLine> gaf;
    if (gaf_path && *gaf_path)
      if (int e = read_gaf(gaf_path, &gaf)) return e;
    FILE* f = std::fopen(out_path, "w");
    if (!f) return AHSOKA_E_IO;
    for (const GChain& ch : g.chains) {
      std::fprintf(f, "CHAIN %d %zu\n", ch.id, ch.bubbles.size());
      for (const GBubble& b : ch.bubbles) {
        std::fprintf(f, "BUBBLE %d %d %d %d %zu", ch.id, b.id, b.source, b.sink, b.inner.size());
        for (int x : b.inner) std::fprintf(f, " %d", x);
        std::fprintf(f, "\n");
      }
    }
    std::vector<std::pair<int, int>> order;
    for (const GChain& ch : g.chains) order.emplace_back((int)ch.bubbles.size(), ch.id);
    std::sort(order.begin(), order.end(), std::greater<>());
    for (auto& oc : order) {
      const GChain& ch = g.chains[(size_t)oc.second];
      for (const GBubble& b : ch.bubbles) {
        std::vector<std::vector<int>> paths = g.allele_paths(b);
        for (size_t a = 0; a < paths.size(); a++) {
          std::fprintf(f, "ALLELE %d %d %zu %zu", ch.id, b.id, a, paths[a].size());
          for (int nd : paths[a]) std::fprintf(f, " %d", nd);
          std::fprintf(f, " |");
          for (size_t i = 0; i + 1 < paths[a].size(); i++) std::fprintf(f, " %c", g.step_sign(paths[a][i], paths[a][i + 1]));
          std::fprintf(f, "\n");
        }
      }
    }
    // per chain (ascending id): one ALN line per GAF line, with the number of copies
    // the reference stores (one per path node that maps to the chain)
    auto chain_of = [&](int nid) {
      auto it = g.nodes.find(nid);
      return it == g.nodes.end() ? 0 : it->second.chain;
    };
    std::map<int, std::vector<std::pair<int, int>>> per_chain;  // chain -> (line, copies)
    for (size_t li = 0; li < gaf.size(); li++) {
      std::map<int, int> cnt;
      for (int nid : gaf[li].node_ids) cnt[chain_of(nid)]++;
      for (auto& kv : cnt) per_chain[kv.first].emplace_back((int)li, kv.second);
    }
    for (auto& kv : per_chain)
      for (auto& lc : kv.second) {
        const GafLine& gl = gaf[(size_t)lc.first];
        const float scaled = gl.identity * 100;
        std::fprintf(f, "ALN %d %s %d %d %zu", kv.first, gl.name.c_str(), (int)scaled, lc.second, gl.node_ids.size());
        for (int nd : gl.node_ids) std::fprintf(f, " %d", nd);
        std::fprintf(f, "\n");
      }
    std::fprintf(f, "ORDER");
    for (auto& oc : order) std::fprintf(f, " %d", oc.second);
    std::fprintf(f, "\n");
    std::fclose(f);
  } catch (const std::bad_alloc&) {
    return AHSOKA_E_NOMEM;
  } catch (...) {
    return AHSOKA_E_PARSE;
  }
  return AHSOKA_OK;
}

extern "C" int ahsoka_ingest(const char* gfa_path, const char* gaf_path, ahsoka_chain_batch** out) {
  if (!gfa_path || !out) return AHSOKA_E_ARG;
  try {
    Graph g;
    if (int e = g.read_gfa(gfa_path)) return e;
    g.find_chains();
    std::vector<GafLine> gaf;
    if (gaf_path && *gaf_path)
      if (int e = read_gaf(gaf_path, &gaf)) return e;

    // size_sorting, src/polyassembly.cpp:136-140: (bubbles, chain id) descending
    std::vector<std::pair<int, int>> order;
    for (const GChain& ch : g.chains) order.emplace_back((int)ch.bubbles.size(), ch.id);
    std::sort(order.begin(), order.end(), std::greater<>());

    // chain of every GAF node: nodes of no chain read as chain 0 (default chain_id)
    auto chain_of = [&](int nid) {
      auto it = g.nodes.find(nid);
      return it == g.nodes.end() ? 0 : it->second.chain;
    };
    std::vector<std::vector<int>> lines_of_chain(g.chains.size());
    for (size_t li = 0; li < gaf.size(); li++) {
      std::vector<int> touched;
      for (int nid : gaf[li].node_ids) {
        int c = chain_of(nid);
        if (c >= 0 && c < (int)g.chains.size() && std::find(touched.begin(), touched.end(), c) == touched.end())
          touched.push_back(c);
      }
      for (int c : touched) lines_of_chain[c].push_back((int)li);
    }

    ahk::HostBatch* hb = new ahk::HostBatch();
    hb->has_sign = true;
    hb->bubble_off.push_back(0);
    hb->allele_off.push_back(0);
    hb->apath_off.push_back(0);
    hb->node_off.push_back(0);
    hb->read_off.push_back(0);
    hb->rnode_off.push_back(0);
    for (auto& oc : order) {
      const GChain& ch = g.chains[(size_t)oc.second];
      hb->chain_id.push_back(ch.id);
      std::unordered_map<int, int> local;
      auto local_of = [&](int nid) {
        auto it = local.find(nid);
        if (it != local.end()) return it->second;
        int idx = (int)local.size();
        local[nid] = idx;
        hb->node_id.push_back(nid);
        return idx;
      };
      for (const GBubble& b : ch.bubbles) {
        local_of(b.source);
        for (int x : b.inner) local_of(x);
        local_of(b.sink);
      }
      for (const GBubble& b : ch.bubbles) {
        for (const std::vector<int>& p : g.allele_paths(b)) {
          for (size_t i = 0; i < p.size(); i++) {
            hb->apath_node.push_back(local_of(p[i]));
            hb->apath_sign.push_back(i + 1 < p.size() ? (uint8_t)g.step_sign(p[i], p[i + 1]) : (uint8_t)'+');
          }
          hb->apath_off.push_back((int64_t)hb->apath_node.size());
        }
        hb->allele_off.push_back((int64_t)hb->apath_off.size() - 1);
      }
      hb->bubble_off.push_back((int64_t)hb->allele_off.size() - 1);
      hb->node_off.push_back((int64_t)hb->node_id.size());
      std::unordered_map<std::string, int> name_id;
      for (int li : lines_of_chain[(size_t)ch.id]) {
        const GafLine& gl = gaf[(size_t)li];
        for (int nid : gl.node_ids) {
          auto it = local.find(nid);
          hb->rnode.push_back(it == local.end() ? -1 : it->second);
        }
        hb->rnode_off.push_back((int64_t)hb->rnode.size());
        const float scaled = gl.identity * 100;  // float multiply, then the implicit float->int of the Read ctor
        hb->read_mapq.push_back((int32_t)scaled);
        auto ins = name_id.emplace(gl.name, (int)name_id.size());
        hb->read_name.push_back(ins.first->second);
      }
      hb->read_off.push_back((int64_t)hb->read_mapq.size());
    }
    hb->finish();
    *out = &hb->view;
  } catch (const std::bad_alloc&) {
    return AHSOKA_E_NOMEM;
  } catch (...) {
    return AHSOKA_E_PARSE;
  }
  return AHSOKA_OK;
}
