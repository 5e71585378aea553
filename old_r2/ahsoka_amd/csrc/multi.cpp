// multi.cpp -- ahsoka_phase_batch_multi: one ChainBatch over several HIP devices
// of one node, behind the C-ABI (include/ahsoka_phase.h).
//
// Chains are independent (every iteration of the chain loop of
// alignmentsToReadset, src/alignmentstoreadset.cpp:75-490, reads only its own
// alignments and allele paths), so the shard step needs no data-path
// collective: a static, cost-weighted partition of the chain list (longest
// processing time first over the cost model below), one host thread + one
// library-owned device handle and HIP stream per device, each running the very
// same upload / k_readset / k_phase / download sequence as ahsoka_phase_batch,
// and a merge of the per-device results back into the caller's chain order --
// the rendered output therefore does not depend on the device list.
// (The multi-PROCESS form of the same partition, one rank per GPU with a single
// RCCL gather of the packed result blocks, is ahsoka_amd/shard.py + bench.py.)
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <thread>
#include <vector>

#include "host_batch.h"

namespace {

// work model of one chain: the read-pair matrix dominates (R^2 per position word),
// plus the R x B match matrices (same model as shard.chain_cost)
double chain_cost(const ahsoka_chain_batch* in, int c) {
  const double r = (double)(in->read_off[c + 1] - in->read_off[c]);
  const double b = (double)(in->bubble_off[c + 1] - in->bubble_off[c]);
  return r * r * std::max(1.0, b / 64.0) + r * b;
}

// the chains `idx` (ascending) of `in` as a batch of their own
void gather_chains(const ahsoka_chain_batch* in, const std::vector<int>& idx, ahk::HostBatch* hb) {
  hb->has_sign = in->apath_sign != nullptr;
  hb->has_genotype = in->genotype != nullptr;
  const bool has_flags = in->read_flags != nullptr, has_s1 = in->s1_order != nullptr;
  hb->bubble_off.push_back(0); hb->allele_off.push_back(0); hb->apath_off.push_back(0);
  hb->node_off.push_back(0); hb->read_off.push_back(0); hb->rnode_off.push_back(0);
  for (int c : idx) {
    hb->chain_id.push_back(in->chain_id ? in->chain_id[c] : c);
    const int64_t b0 = in->bubble_off[c], b1 = in->bubble_off[c + 1];
    const int64_t a0 = in->allele_off[b0], a1 = in->allele_off[b1];
    const int64_t p0 = in->apath_off[a0], p1 = in->apath_off[a1];
    const int64_t abase = (int64_t)hb->apath_off.size() - 1, pbase = (int64_t)hb->apath_node.size();
    for (int64_t g = b0; g < b1; g++) hb->allele_off.push_back(abase + (in->allele_off[g + 1] - a0));
    for (int64_t a = a0; a < a1; a++) hb->apath_off.push_back(pbase + (in->apath_off[a + 1] - p0));
    hb->apath_node.insert(hb->apath_node.end(), in->apath_node + p0, in->apath_node + p1);
    if (hb->has_sign) hb->apath_sign.insert(hb->apath_sign.end(), in->apath_sign + p0, in->apath_sign + p1);
    if (hb->has_genotype) hb->genotype.insert(hb->genotype.end(), in->genotype + a0, in->genotype + a1);
    if (has_s1) hb->s1_order.insert(hb->s1_order.end(), in->s1_order + b0, in->s1_order + b1);
    hb->bubble_off.push_back((int64_t)hb->allele_off.size() - 1);
    const int64_t n0 = in->node_off[c], n1 = in->node_off[c + 1];
    if (in->node_id) hb->node_id.insert(hb->node_id.end(), in->node_id + n0, in->node_id + n1);
    else hb->node_id.insert(hb->node_id.end(), (size_t)(n1 - n0), 0);
    hb->node_off.push_back((int64_t)hb->node_id.size());
    const int64_t r0 = in->read_off[c], r1 = in->read_off[c + 1];
    const int64_t e0 = in->rnode_off[r0], e1 = in->rnode_off[r1];
    const int64_t ebase = (int64_t)hb->rnode.size();
    for (int64_t r = r0; r < r1; r++) hb->rnode_off.push_back(ebase + (in->rnode_off[r + 1] - e0));
    hb->rnode.insert(hb->rnode.end(), in->rnode + e0, in->rnode + e1);
    hb->read_mapq.insert(hb->read_mapq.end(), in->read_mapq + r0, in->read_mapq + r1);
    hb->read_name.insert(hb->read_name.end(), in->read_name + r0, in->read_name + r1);
    if (has_flags) hb->read_flags.insert(hb->read_flags.end(), in->read_flags + r0, in->read_flags + r1);
    hb->read_off.push_back((int64_t)hb->read_mapq.size());
  }
  hb->finish();
  if (!has_flags) hb->view.read_flags = nullptr;
  if (!has_s1) hb->view.s1_order = nullptr;
}

template <class T>
T* hostdup(const std::vector<T>& v) {
  T* p = (T*)std::malloc(std::max<size_t>(1, v.size()) * sizeof(T));
  if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}

}  // namespace

extern "C" int ahsoka_phase_batch_multi(const ahsoka_chain_batch* in, const ahsoka_phase_params* prm, const int32_t* devices,
                                        int32_t n_devices, ahsoka_phase_result* out) {
  if (!in || !prm || !out || !devices || n_devices < 1 || n_devices > 64 || in->n_chains < 0) return AHSOKA_E_ARG;
  for (int d = 0; d < n_devices; d++)
    if (devices[d] < 0) return AHSOKA_E_ARG;
  if (n_devices == 1) {
    ahsoka_phase_params p1 = *prm;
    p1.device = devices[0];
    return ahsoka_phase_batch(in, &p1, out);
  }
  const int C = in->n_chains;
  if (int e = ahk::validate_batch(in)) return e;  // the shards are cut out of the arrays: check them first
  try {
    // ---- static partition: longest processing time first, ties to the lower shard ----
    std::vector<int> order((size_t)C);
    std::iota(order.begin(), order.end(), 0);
    std::vector<double> cost((size_t)C);
    for (int c = 0; c < C; c++) cost[(size_t)c] = chain_cost(in, c);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[(size_t)a] > cost[(size_t)b]; });
    std::vector<double> load((size_t)n_devices, 0.0);
    std::vector<int> owner((size_t)C, 0);
    for (int c : order) {
      const int d = (int)(std::min_element(load.begin(), load.end()) - load.begin());
      owner[(size_t)c] = d;
      load[(size_t)d] += cost[(size_t)c];
    }
    std::vector<std::vector<int>> mine((size_t)n_devices);
    for (int c = 0; c < C; c++) mine[(size_t)owner[(size_t)c]].push_back(c);  // ascending within a shard

    // ---- one thread, handle and stream per device -----------------------------------
    std::vector<ahk::HostBatch> shards((size_t)n_devices);
    std::vector<ahsoka_phase_result> res((size_t)n_devices);
    std::vector<int> rcs((size_t)n_devices, AHSOKA_OK);
    for (int d = 0; d < n_devices; d++) {
      std::memset(&res[(size_t)d], 0, sizeof(ahsoka_phase_result));
      gather_chains(in, mine[(size_t)d], &shards[(size_t)d]);
    }
    {
      std::vector<std::thread> th;
      for (int d = 0; d < n_devices; d++)
        th.emplace_back([&, d]() {
          ahsoka_phase_params pd = *prm;
          pd.device = devices[d];
          rcs[(size_t)d] = ahsoka_phase_batch(&shards[(size_t)d].view, &pd, &res[(size_t)d]);
        });
      for (auto& t : th) t.join();
    }
    int rc = AHSOKA_OK;
    for (int d = 0; d < n_devices; d++)
      if (rcs[(size_t)d] && !rc) rc = rcs[(size_t)d];
    if (rc) {
      for (int d = 0; d < n_devices; d++)
        if (!rcs[(size_t)d]) ahsoka_result_free(&res[(size_t)d]);
      return rc;
    }

    // ---- merge into the caller's chain order ------------------------------------------
    const int p = prm->ploidy;
    const bool stages = prm->want_stages != 0;
    std::vector<int32_t> status((size_t)C), nreads((size_t)C), nclus((size_t)C), pos, ha, hc, fname, fclus, score;
    std::vector<int64_t> niter((size_t)C), pos_off((size_t)C + 1, 0), vmat_off((size_t)C + 1, 0), fread_off((size_t)C + 1, 0),
        score_off((size_t)C + 1, 0);
    std::vector<int8_t> vfull, vpart;
    std::vector<int> at((size_t)n_devices, 0);  // next local chain of every shard
    double kernel_ms = 0, elapsed_ms = 0;
    for (int d = 0; d < n_devices; d++) {
      kernel_ms = std::max(kernel_ms, res[(size_t)d].kernel_ms);
      elapsed_ms = std::max(elapsed_ms, res[(size_t)d].elapsed_ms);
    }
    for (int c = 0; c < C; c++) {
      const int d = owner[(size_t)c];
      const ahsoka_phase_result& r = res[(size_t)d];
      const int lc = at[(size_t)d]++;
      status[(size_t)c] = r.status[lc];
      nreads[(size_t)c] = r.n_reads[lc];
      nclus[(size_t)c] = r.n_clusters[lc];
      niter[(size_t)c] = r.n_iterations[lc];
      const int64_t q0 = r.pos_off[lc], q1 = r.pos_off[lc + 1];
      pos.insert(pos.end(), r.pos + q0, r.pos + q1);
      ha.insert(ha.end(), r.hap_allele + p * q0, r.hap_allele + p * q1);
      hc.insert(hc.end(), r.hap_cluster + p * q0, r.hap_cluster + p * q1);
      pos_off[(size_t)c + 1] = (int64_t)pos.size();
      if (stages) {
        vfull.insert(vfull.end(), r.vfull + r.vmat_off[lc], r.vfull + r.vmat_off[lc + 1]);
        vpart.insert(vpart.end(), r.vpart + r.vmat_off[lc], r.vpart + r.vmat_off[lc + 1]);
        fname.insert(fname.end(), r.fread_name + r.fread_off[lc], r.fread_name + r.fread_off[lc + 1]);
        fclus.insert(fclus.end(), r.fread_cluster + r.fread_off[lc], r.fread_cluster + r.fread_off[lc + 1]);
        score.insert(score.end(), r.score_q + r.score_off[lc], r.score_q + r.score_off[lc + 1]);
      }
      vmat_off[(size_t)c + 1] = (int64_t)vfull.size();
      fread_off[(size_t)c + 1] = (int64_t)fname.size();
      score_off[(size_t)c + 1] = (int64_t)score.size();
    }
    for (int d = 0; d < n_devices; d++) ahsoka_result_free(&res[(size_t)d]);
    std::memset(out, 0, sizeof(*out));
    out->n_chains = C;
    out->ploidy = p;
    out->status = hostdup(status);
    out->pos_off = hostdup(pos_off);
    out->pos = hostdup(pos);
    out->hap_allele = hostdup(ha);
    out->hap_cluster = hostdup(hc);
    out->n_reads = hostdup(nreads);
    out->n_clusters = hostdup(nclus);
    out->n_iterations = hostdup(niter);
    if (stages) {
      out->vmat_off = hostdup(vmat_off);
      out->vfull = hostdup(vfull);
      out->vpart = hostdup(vpart);
      out->fread_off = hostdup(fread_off);
      out->fread_name = hostdup(fname);
      out->fread_cluster = hostdup(fclus);
      out->score_off = hostdup(score_off);
      out->score_q = hostdup(score);
    }
    out->kernel_ms = kernel_ms;
    out->elapsed_ms = elapsed_ms;
  } catch (const std::bad_alloc&) {
    return AHSOKA_E_NOMEM;
  }
  return AHSOKA_OK;
}
