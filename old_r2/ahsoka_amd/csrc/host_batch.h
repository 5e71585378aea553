// host_batch.h -- owning container behind an ahsoka_chain_batch returned by the
// library (ahsoka_ingest / ahsoka_synth_batch).
#pragma once
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "../../include/ahsoka_phase.h"

namespace ahk {

// Order in which S1 visits the bubbles of a chain with B bubbles
// (src/alignmentstoreadset.cpp:76,90): the iteration order of an
// unordered_map<int, ...> that ChainsToReadsetDetailed filled with the bubble
// ids 0..B-1 in ascending order (src/chainstoreadset.cpp:171-198; the later
// copy-assignment and copy keep a source's node order).  Not re-stated: a real
// libstdc++ unordered_map<int,...> is filled the same way and walked, so the
// order is whatever this toolchain's hash table produces.
// Pinned on the reference's own compiled code by tests/test_ingest_golden.py.
inline void ref_s1_order(int B, std::vector<int32_t>* out) {
  std::unordered_map<int, char> m;
  for (int k = 0; k < B; k++) m[k];
  for (auto& kv : m) out->push_back((int32_t)kv.first);
}

// `alignment.id*100 > 90` of src/alignmentstoreadset.cpp:245: float times int in
// single precision, compared as float.
inline uint8_t ref_gt90_flag(float identity) { return (identity * 100) > 90 ? 1 : 0; }

// structural checks of a caller-supplied batch (offsets monotone, indices in range); defined in phase_api.hip
int validate_batch(const ahsoka_chain_batch* in);

struct HostBatch {
  ahsoka_chain_batch view;  // MUST stay the first member (handle == &view)
  std::vector<int32_t> chain_id;
  std::vector<int64_t> bubble_off, allele_off, apath_off, node_off, read_off, rnode_off;
  std::vector<int32_t> apath_node, genotype, node_id, rnode, read_mapq, read_name;
  std::vector<uint8_t> apath_sign, read_flags;
  std::vector<int32_t> s1_order;
  bool has_sign = false, has_genotype = false;

  void finish() {
    view.n_chains = (int32_t)chain_id.size();
    view.chain_id = chain_id.data();
    view.bubble_off = bubble_off.data();
    view.allele_off = allele_off.data();
    view.apath_off = apath_off.data();
    view.apath_node = apath_node.data();
    view.apath_sign = has_sign ? apath_sign.data() : nullptr;
    view.genotype = has_genotype ? genotype.data() : nullptr;
    view.node_off = node_off.data();
    view.node_id = node_id.data();
    view.read_off = read_off.data();
    view.rnode_off = rnode_off.data();
    view.rnode = rnode.data();
    view.read_mapq = read_mapq.data();
    view.read_name = read_name.data();
    view.read_flags = read_flags.data();
    view.s1_order = s1_order.data();
  }
};

}  // namespace ahk
