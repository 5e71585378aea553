/*
 * whatshap_ref_main.cpp -- driver for a REAL run of the reference's hot path.
 *
 * TEST INFRASTRUCTURE ONLY, and UNEXERCISED in this repository's build
 * container: the reference's alignmentsToReadset (src/alignmentstoreadset.cpp:55)
 * calls WhatsHap (ReadScoring, ClusterEditingSolver, HaploThreader, Read/ReadSet),
 * pinned by container/ahsoka.def:16-18 to whatshap/whatshap @
 * 8f4c0c070d0b5d8e6d2c03e363965d0efb50a960, which is neither in /root/reference
 * nor on this machine and cannot be fetched.  No stand-in headers are written:
 * this file only compiles where a checkout of that revision is supplied,
 *
 *     make -C oracle _whatshap WHATSHAP_SRC=/path/to/whatshap/src
 *
 * which builds oracle/_ref/ahsoka_ref_phase from the reference's own sources,
 * compiled where they lie (the two .cpp files below are included as text, exactly
 * as src/polyassembly.cpp:11-12 does; jellyfish, which the reference includes but
 * never uses, is left out).  tests/test_reference_run.py then compares its
 * <prefix>-result.txt -- the haplotype node lists, the parity contract -- with
 * this repository's result on every golden fixture, which would turn "parity
 * unpinned" for S3/S4/S5/S7 into a pin.
 *
 *     ahsoka_ref_phase <gfa> <gaf> <output prefix>
 */
#include <algorithm>
#include <functional>
#include <iostream>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "graph.hpp"
#include "alignmentreader.hpp"
#include "chainstoreadset.cpp"
#include "alignmentstoreadset.cpp"

int main(int argc, char** argv) {
  if (argc < 4) {
    std::cerr << "usage: ahsoka_ref_phase <gfa> <gaf> <output prefix>\n";
    return 2;
  }
  // the sequence of src/polyassembly.cpp:82-171, single-thread branch
  Graph graph = Graph::ReadGraph(argv[1]);
  graph.findBubbles();
  AlignmentReader alignmentreader;
  alignmentreader.readAlignmentfile(argv[2], graph);
  unordered_map<int, unordered_map<int, vector<vector<int>>>> chainpathToAlleles = ChainsToReadsetDetailed(graph);
  vector<pair<int, int>> size_sorting;
  for (auto& chainmap : chainpathToAlleles) size_sorting.emplace_back(chainmap.second.size(), chainmap.first);
  sort(begin(size_sorting), end(size_sorting), greater<>());
  std::mutex g_display_mutex;
  alignmentsToReadset(alignmentreader, graph, chainpathToAlleles, argv[3], false, size_sorting, g_display_mutex);
  return 0;
}
