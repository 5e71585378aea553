// ref_bridge.cpp -- the reference-side binding of INTEGRATION.md section 2 as a compilable,
// tested file: it holds the reference's OWN objects (Graph, AlignmentReader, the map that
// ChainsToReadsetDetailed returns, size_sorting -- built by the very call sequence of
// src/polyassembly.cpp:82-140) and flattens them into the ChainBatch of include/ahsoka_phase.h,
// i.e. it is the code a maintainer would put in place of the alignmentsToReadset call at
// src/polyassembly.cpp:167-172.
//
// It contains no reference text: it only calls the reference's classes.  It is compiled
// together with the reference's translation units where they lie (oracle/Makefile, target
// _ref/ref_bridge; output under oracle/_ref/, git-ignored) and checked by
// tests/test_ref_bridge.py: the ChainBatch it builds equals the one ahsoka_ingest derives from
// the same files, and on a GPU its rendered result equals the library's own CLI path.
//
//   ref_bridge --dump  <gfa> <gaf>     canonical text of the flattened ChainBatch
//   ref_bridge --phase <gfa> <gaf>     ahsoka_phase_batch + ahsoka_render_chain: the chain blocks of <o>-result.txt
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <iostream>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "alignmentreader.hpp"
#include "graph.hpp"

#include "ahsoka_phase.h"

using namespace std;

// defined in the reference's chainstoreadset.cpp (no header declares it)
unordered_map<int, unordered_map<int, vector<vector<int>>>> ChainsToReadsetDetailed(Graph graph);

struct FlatBatch {
  vector<int32_t> chain_id, apath_node, node_id, rnode, read_mapq, read_name, s1_order;
  vector<uint8_t> apath_sign, read_flags;
  vector<int64_t> bubble_off{0}, allele_off{0}, apath_off{0}, node_off{0}, read_off{0}, rnode_off{0};
  ahsoka_chain_batch view() const {
    return ahsoka_chain_batch{(int32_t)chain_id.size(), chain_id.data(), bubble_off.data(), allele_off.data(), apath_off.data(),
                              apath_node.data(), apath_sign.data(), /*genotype*/ nullptr, node_off.data(), node_id.data(),
                              read_off.data(), rnode_off.data(), rnode.data(), read_mapq.data(), read_name.data(),
                              read_flags.data(), s1_order.data()};
  }
};

// does the graph hold an edge first -> sec in either orientation of first?  (getEdge falls off its end otherwise)
static bool has_edge(Graph& graph, const Node& first, const Node& sec) {
  for (bool val : {true, false}) {
    auto it = graph.edges.find(DirectedNode(first.node_id, val));
    if (it == graph.edges.end()) continue;
    for (auto& to : it->second)
      if (to.id == sec.node_id) return true;
  }
  return false;
}

static void flatten(AlignmentReader& reader, Graph& graph, unordered_map<int, unordered_map<int, vector<vector<int>>>>& pathToAlleles,
                    vector<pair<int, int>>& size_sorting, FlatBatch& fb) {
  for (auto& s : size_sorting) {  // chains in the reference's own order (src/polyassembly.cpp:136-140)
    const int chain = s.second;
    fb.chain_id.push_back(chain);
    unordered_map<int, int> local;  // node id -> chain-local index
    const size_t node_base = fb.node_id.size();
    auto local_of = [&](int nid) {
      auto it = local.find(nid);
      if (it != local.end()) return it->second;
      const int idx = (int)(fb.node_id.size() - node_base);
      local[nid] = idx;
      fb.node_id.push_back(nid);
      return idx;
    };
    auto& bubbles = pathToAlleles[chain];
    {  // the order S1 will visit the bubbles in: the iteration order of the COPY the hot path makes (:76,:90)
      auto chainmap = make_pair(chain, bubbles);
      for (auto& bubblemap : chainmap.second) fb.s1_order.push_back(bubblemap.first);
    }
    for (int b = 0; b < (int)bubbles.size(); b++) {  // bubble ids are dense 0..B-1 (src/graph.cpp:351-365)
      for (auto& path : bubbles[b]) {
        for (size_t i = 0; i < path.size(); i++) {
          fb.apath_node.push_back(local_of(path[i]));
          char sign = '+';
          if (i + 1 < path.size()) {  // orientation exactly as src/alignmentstoreadset.cpp:441-446
            Node first = graph.getNode(path[i]), sec = graph.getNode(path[i + 1]);
            if (has_edge(graph, first, sec)) {
              auto tup = graph.getEdge(first, sec);
              sign = tup.first.end ? '+' : '-';
            } else {
              sign = '?';
            }
          }
          fb.apath_sign.push_back((uint8_t)sign);
        }
        fb.apath_off.push_back((int64_t)fb.apath_node.size());
      }
      fb.allele_off.push_back((int64_t)fb.apath_off.size() - 1);
    }
    fb.bubble_off.push_back((int64_t)fb.allele_off.size() - 1);
    fb.node_off.push_back((int64_t)fb.node_id.size());
    unordered_map<string, int> names;
    const AlignmentPath* prev = nullptr;
    for (auto& al : reader.alignments[chain]) {  // one copy per path node (src/alignmentreader.cpp:176-183): keep the first of a run
      if (prev && prev->name == al.name && prev->nodes == al.nodes) continue;
      prev = &al;
      for (int nid : al.getRawIds()) {
        auto it = local.find(nid);
        fb.rnode.push_back(it == local.end() ? -1 : it->second);
      }
      fb.rnode_off.push_back((int64_t)fb.rnode.size());
      fb.read_mapq.push_back((int)(al.id * 100));             // the implicit float->int of the Read ctor (:117)
      fb.read_flags.push_back((al.id * 100) > 90 ? 1 : 0);    // the FLOAT compare of :245, not the truncated value
      fb.read_name.push_back(names.emplace(al.name, (int)names.size()).first->second);
    }
    fb.read_off.push_back((int64_t)fb.read_mapq.size());
  }
}

static void dump(const FlatBatch& fb) {
  const ahsoka_chain_batch in = fb.view();
  for (int c = 0; c < in.n_chains; c++) {
    const int64_t b0 = in.bubble_off[c], b1 = in.bubble_off[c + 1], r0 = in.read_off[c], r1 = in.read_off[c + 1], n0 = in.node_off[c];
    printf("CHAIN %d B=%lld R=%lld\n", in.chain_id[c], (long long)(b1 - b0), (long long)(r1 - r0));
    for (int64_t g = b0; g < b1; g++)
      for (int64_t a = in.allele_off[g]; a < in.allele_off[g + 1]; a++) {
        printf("A %lld %lld :", (long long)(g - b0), (long long)(a - in.allele_off[g]));
        for (int64_t e = in.apath_off[a]; e < in.apath_off[a + 1]; e++) printf(" %d", in.node_id[n0 + in.apath_node[e]]);
        printf(" |");
        for (int64_t e = in.apath_off[a]; e + 1 < in.apath_off[a + 1]; e++) printf(" %c", (char)in.apath_sign[e]);
        printf("\n");
      }
    printf("S1");
    for (int64_t g = b0; g < b1; g++) printf(" %d", in.s1_order[g]);
    printf("\n");
    for (int64_t r = r0; r < r1; r++) {
      printf("R %d %d %d :", in.read_name[r], in.read_mapq[r], (int)in.read_flags[r]);
      for (int64_t e = in.rnode_off[r]; e < in.rnode_off[r + 1]; e++) printf(" %d", in.rnode[e] < 0 ? -1 : in.node_id[n0 + in.rnode[e]]);
      printf("\n");
    }
  }
}

int main(int argc, char** argv) {
  if (argc < 4 || (strcmp(argv[1], "--dump") != 0 && strcmp(argv[1], "--phase") != 0)) {
    fprintf(stderr, "usage: ref_bridge --dump|--phase <gfa> <gaf>\n");
    return 2;
  }
  // same sequence as src/polyassembly.cpp:82-140
  Graph graph = Graph::ReadGraph(argv[2]);
  graph.findBubbles();
  AlignmentReader reader;
  reader.readAlignmentfile(argv[3], graph);
  auto pathToAlleles = ChainsToReadsetDetailed(graph);
  vector<pair<int, int>> size_sorting;
  for (auto& chainmap : pathToAlleles) size_sorting.emplace_back(chainmap.second.size(), chainmap.first);
  sort(begin(size_sorting), end(size_sorting), greater<>());

  FlatBatch fb;
  flatten(reader, graph, pathToAlleles, size_sorting, fb);
  if (strcmp(argv[1], "--dump") == 0) {
    dump(fb);
    return 0;
  }
  const ahsoka_chain_batch in = fb.view();
  ahsoka_phase_params prm;
  ahsoka_default_params(&prm);
  ahsoka_phase_result res;
  if (int rc = ahsoka_phase_batch(&in, &prm, &res)) {
    fprintf(stderr, "ref_bridge: ahsoka_phase_batch failed: %d\n", rc);
    return 1;
  }
  for (int c = 0; c < in.n_chains; c++) {  // what the hot path appends to <o>-result.txt (:72,:82-83,:415-458)
    vector<char> buf((size_t)ahsoka_render_chain(&in, &res, c, nullptr, 0) + 1);
    ahsoka_render_chain(&in, &res, c, buf.data(), (int64_t)buf.size());
    fputs(buf.data(), stdout);
  }
  ahsoka_result_free(&res);
  return 0;
}
