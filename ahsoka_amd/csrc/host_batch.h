// host_batch.h -- owning container behind an ahsoka_chain_batch returned by the
// library (ahsoka_ingest / ahsoka_synth_batch).
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/ahsoka_phase.h"

namespace ahk {

struct HostBatch {
  ahsoka_chain_batch view;  // MUST stay the first member (handle == &view)
  std::vector<int32_t> chain_id;
  std::vector<int64_t> bubble_off, allele_off, apath_off, node_off, read_off, rnode_off;
  std::vector<int32_t> apath_node, genotype, node_id, rnode, read_mapq, read_name;
  std::vector<uint8_t> apath_sign;
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
  }
};

}  // namespace ahk
