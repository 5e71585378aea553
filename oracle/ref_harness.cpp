/*
 * ref_harness.cpp -- driver around the reference's OWN translation units.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/Makefile, target _ref).  This file is
 * linked with /root/reference/src/{graph,alignmentreader,chainstoreadset}.cpp,
 * compiled where they lie, into oracle/_ref/ (git-ignored).  It contains no
 * reference text: it only calls the reference's classes and dumps what they
 * computed, so that the repository's own ingest can be pinned against it:
 *
 *   ref_dump <gfa> <gaf>   ->  stdout, line oriented:
 *     CHAIN <id> <nbubbles>
 *     BUBBLE <chain> <bubble id> <source> <sink> <ninner> <inner...>
 *     ALLELE <chain> <bubble id> <allele> <n> <node...> | <sign...>
 *     ALN <chain> <name> <mapq> <id*100 > 90> <copies> <n> <node...>
 *     S1ORDER <chain> <bubble ids in the order S1 visits them>
 *     ORDER <chain ids in size_sorting order>
 *
 * and, through the extracted-at-build-time slice of is_subset
 * (src/alignmentstoreadset.cpp:495-548; the rest of that file needs WhatsHap
 * headers and is unbuildable here), a C entry point ref_is_subset().
 */
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <functional>
#include <iostream>
#include <string>
#include <unordered_map>
#include <vector>

#include "alignmentreader.hpp"
#include "graph.hpp"

using namespace std;

// defined in the reference's chainstoreadset.cpp (no header declares it)
unordered_map<int, unordered_map<int, vector<vector<int>>>> ChainsToReadsetDetailed(Graph graph);

// the slice of the reference's alignmentstoreadset.cpp holding is_subset
bool is_subset(vector<int> allelepath, AlignmentPath alignment, bool take_partial);
#include "is_subset_slice.inc"

extern "C" int ref_is_subset(const int* path, int npath, const int* aln, int naln, int take_partial) {
  vector<int> p(path, path + npath);
  vector<string> nodes;
  for (int i = 0; i < naln; i++) nodes.push_back("n" + to_string(aln[i]));
  AlignmentPath ap(nodes);
  return is_subset(p, ap, take_partial != 0) ? 1 : 0;
}

#ifdef REF_MAIN
int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: ref_dump <gfa> <gaf>\n");
    return 2;
  }
  // same sequence as src/polyassembly.cpp:82-140; REF_TIME=1: only time the four steps
  // (the CPU baseline of the ingest rows f1/f2/f4, tools/ingest_bench.py) and exit
  const bool timing = getenv("REF_TIME") != nullptr;
  auto now = [] { return chrono::duration<double>(chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  Graph graph = Graph::ReadGraph(argv[1]);
  const double t1 = now();
  graph.findBubbles();
  const double t2 = now();
  AlignmentReader reader;
  reader.readAlignmentfile(argv[2], graph);
  const double t3 = now();
  auto pathToAlleles = ChainsToReadsetDetailed(graph);
  const double t4 = now();
  if (timing) {
    printf("{\"read_graph_s\": %.4f, \"find_bubbles_s\": %.4f, \"read_alignments_s\": %.4f, \"allele_paths_s\": %.4f, \"total_s\": %.4f}\n", t1 - t0, t2 - t1,
           t3 - t2, t4 - t3, t4 - t0);
    return 0;
  }
  vector<pair<int, int>> size_sorting;
  for (auto& chainmap : pathToAlleles) size_sorting.emplace_back(chainmap.second.size(), chainmap.first);
  sort(begin(size_sorting), end(size_sorting), greater<>());

  for (auto& chain : graph.chains) {
    printf("CHAIN %d %zu\n", chain.id, chain.bubbles.size());
    for (auto& b : chain.bubbles) {
      printf("BUBBLE %d %d %d %d %zu", chain.id, b.id, b.source.node_id, b.sink.node_id, b.innerNodes.size());
      for (auto& nd : b.innerNodes) printf(" %d", nd.node_id);
      printf("\n");
    }
  }
  for (auto& s : size_sorting) {
    int chain = s.second;
    auto& bm = pathToAlleles[chain];
    vector<int> bids;
    for (auto& kv : bm) bids.push_back(kv.first);
    sort(bids.begin(), bids.end());
    for (int bid : bids) {
      auto& paths = bm[bid];
      for (size_t a = 0; a < paths.size(); a++) {
        printf("ALLELE %d %d %zu %zu", chain, bid, a, paths[a].size());
        for (int nd : paths[a]) printf(" %d", nd);
        printf(" |");
        // orientation exactly as src/alignmentstoreadset.cpp:441-446 derives it
        for (size_t i = 0; i + 1 < paths[a].size(); i++) {
          Node first = graph.getNode(paths[a][i]);
          Node sec = graph.getNode(paths[a][i + 1]);
          bool found = false;
          for (bool val : {true, false}) {
            auto it = graph.edges.find(DirectedNode(first.node_id, val));
            if (it == graph.edges.end()) continue;
            for (auto& to : it->second)
              if (to.id == sec.node_id) found = true;
          }
          if (!found) { printf(" ?"); continue; }  // getEdge would fall off its end (UB)
          auto tup = graph.getEdge(first, sec);
          printf(" %c", tup.first.end ? '+' : '-');
        }
        printf("\n");
      }
    }
  }
  vector<int> chains;
  for (auto& kv : reader.alignments) chains.push_back(kv.first);
  sort(chains.begin(), chains.end());
  for (int chain : chains) {
    auto& v = reader.alignments[chain];
    size_t i = 0;
    while (i < v.size()) {
      size_t j = i + 1;
      while (j < v.size() && v[j].name == v[i].name && v[j].nodes == v[i].nodes) j++;
      int mapq = v[i].id * 100;  // the implicit float->int of the Read ctor, :117
      vector<int> raw = v[i].getRawIds();
      int gt90 = (v[i].id * 100) > 90;  // the float compare of :245 on the reference's own field
      printf("ALN %d %s %d %d %zu %zu", chain, v[i].name.c_str(), mapq, gt90, j - i, raw.size());
      for (int nd : raw) printf(" %d", nd);
      printf("\n");
      i = j;
    }
  }
  // S1 walks a COPY of the chain's bubble map (src/alignmentstoreadset.cpp:76,90):
  // same copy expression here, then the map's own iteration order
  for (auto& size : size_sorting) {
    auto chainmap = make_pair(size.second, pathToAlleles[size.second]);
    printf("S1ORDER %d", chainmap.first);
    for (auto& bubblemap : chainmap.second) printf(" %d", bubblemap.first);
    printf("\n");
  }
  printf("ORDER");
  for (auto& s : size_sorting) printf(" %d", s.second);
  printf("\n");
  return 0;
}
#endif
