// multi.cpp -- ahsoka_phase_batch_multi: one ChainBatch over several HIP devices
// of one node, behind the C-ABI (include/ahsoka_phase.h).
//
// Chains are independent (every iteration of the chain loop of
// alignmentsToReadset, src/alignmentstoreadset.cpp:75-490, reads only its own
// alignments and allele paths), so the shard step needs no data-path
// collective: a static, cost-weighted partition of the chain list, one host
// thread + one library-owned device handle and HIP stream per device, each
// running the very same upload / k_readset / k_phase / download sequence as
// ahsoka_phase_batch, and a merge of the per-device results back into the
// caller's chain order -- the rendered output therefore does not depend on the
// device list.
//
// The partition is CONTIGUOUS BLOCKS of the caller's chain order whenever cuts at
// the cost quantiles balance the shards (always the case for batches of similar
// chains: BASELINE configs 1-4): a block is a view of the caller's arrays -- the
// data arrays by pointer offset, only the six offset arrays re-based -- so nothing
// of the 0.8 GB batch is copied on the host.  Otherwise (the power-law config 5)
// it is longest-processing-time-first, whose shards are not ranges and are packed
// by the device's own thread, concurrently with the other devices' uploads and
// kernels.  Either way the serial prologue is the cost model and the partition:
// one pass over read_mapq.
// (The multi-PROCESS form of the same partition, one rank per GPU with a single
// RCCL gather of the packed result blocks, is ahsoka_amd/shard.py + bench.py; the
// reference's own attempt at parallelism is the two-thread split of
// src/polyassembly.cpp:190-215.)
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <thread>
#include <vector>

#include "host_batch.h"

namespace {

// Work model of one chain.  k_phase's time follows the FILTERED readset (n^2 pair state, ~n
// sequential iterations), not the raw alignment count: an alignment below the identity filter
// (src/alignmentstoreadset.cpp:158,270) never reaches it.  n is estimated by the alignments that
// pass the filter; the R x B term is the readset kernels' share.
double chain_cost(const ahsoka_chain_batch* in, int min_mapq, int c) {
  const int64_t r0 = in->read_off[c], r1 = in->read_off[c + 1];
  int64_t pass = 0;
  for (int64_t r = r0; r < r1; r++) pass += in->read_mapq[r] >= min_mapq ? 1 : 0;
  const double n = (double)pass, r = (double)(r1 - r0);
  const double b = (double)(in->bubble_off[c + 1] - in->bubble_off[c]);
  return n * n * std::max(1.0, b / 64.0) + r * b;
}

struct Plan {
  bool blocks = false;
  std::vector<int> owner;              // [C]
  std::vector<int> first;              // blocks: shard d = chains [first[d], first[d+1])
  std::vector<std::vector<int>> mine;  // LPT: ascending chain indices per shard
  double prologue_ms = 0;
};

template <class F>
void parallel_chunks(int n, F fn) {
  unsigned T = std::thread::hardware_concurrency();
  T = std::max(1u, std::min(T, 16u));
  if (n < 4096 || T == 1) {
    fn(0, n);
    return;
  }
  std::vector<std::thread> th;
  th.reserve(T);
  for (unsigned t = 0; t < T; t++) th.emplace_back([=] { fn((int)((long long)n * t / T), (int)((long long)n * (t + 1) / T)); });
  for (auto& x : th) x.join();
}

void make_plan(const ahsoka_chain_batch* in, int min_mapq, int n_devices, Plan* pl) {
  const auto t0 = std::chrono::steady_clock::now();
  const int C = in->n_chains;
  std::vector<double> cost((size_t)C);
  parallel_chunks(C, [&](int lo, int hi) {
    for (int c = lo; c < hi; c++) cost[(size_t)c] = chain_cost(in, min_mapq, c);
  });
  double total = 0;
  for (double v : cost) total += v;
  pl->owner.assign((size_t)C, 0);
  // contiguous blocks: cut where the running cost passes the d-th quantile
  std::vector<int> first((size_t)n_devices + 1, C);
  first[0] = 0;
  {
    double run = 0;
    int d = 1;
    for (int c = 0; c < C && d < n_devices; c++) {
      // (cut before chain c when more than half of it lies beyond the quantile)
      while (d < n_devices && run + 0.5 * cost[(size_t)c] > total * d / n_devices) first[(size_t)d++] = c;
      run += cost[(size_t)c];
    }
  }
  double worst = 0;
  for (int d = 0; d < n_devices; d++) {
    double l = 0;
    for (int c = first[(size_t)d]; c < first[(size_t)d + 1]; c++) l += cost[(size_t)c];
    worst = std::max(worst, l);
  }
  // longest processing time first, ties to the lower shard
  std::vector<int> lpt_owner((size_t)C, 0);
  double lpt_worst = 0;
  {
    std::vector<int> order((size_t)C);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[(size_t)a] > cost[(size_t)b]; });
    std::vector<double> load((size_t)n_devices, 0.0);
    for (int c : order) {
      const int d = (int)(std::min_element(load.begin(), load.end()) - load.begin());
      lpt_owner[(size_t)c] = d;
      load[(size_t)d] += cost[(size_t)c];
    }
    lpt_worst = *std::max_element(load.begin(), load.end());
  }
  // blocks are taken whenever their most loaded shard is within 2 % of LPT's
  if (worst <= lpt_worst * 1.02) {
    pl->blocks = true;
    pl->first = first;
    for (int d = 0; d < n_devices; d++)
      for (int c = first[(size_t)d]; c < first[(size_t)d + 1]; c++) pl->owner[(size_t)c] = d;
  } else {
    pl->owner = lpt_owner;
    pl->mine.assign((size_t)n_devices, std::vector<int>());
    for (int c = 0; c < C; c++) pl->mine[(size_t)pl->owner[(size_t)c]].push_back(c);  // ascending within a shard
  }
  pl->prologue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// chains [c0, c1) of `in` as a batch of their own WITHOUT copying their data: the six offset
// arrays re-based to zero (their entries index the next array), every data array a pointer
// into the caller's
struct BlockView {
  ahsoka_chain_batch view;
  std::vector<int64_t> bubble_off, allele_off, apath_off, node_off, read_off, rnode_off;
};
void rebase(std::vector<int64_t>* dst, const int64_t* src, int64_t lo, int64_t hi) {
  dst->resize((size_t)(hi - lo) + 1);
  const int64_t base = src[lo];
  int64_t* o = dst->data();
  for (int64_t i = lo; i <= hi; i++) o[i - lo] = src[i] - base;
}
void make_block_view(const ahsoka_chain_batch* in, int c0, int c1, BlockView* bv) {
  const int64_t b0 = in->bubble_off[c0], b1 = in->bubble_off[c1];
  const int64_t a0 = in->allele_off[b0], a1 = in->allele_off[b1];
  const int64_t p0 = in->apath_off[a0];
  const int64_t n0 = in->node_off[c0];
  const int64_t r0 = in->read_off[c0], r1 = in->read_off[c1];
  const int64_t e0 = in->rnode_off[r0];
  rebase(&bv->bubble_off, in->bubble_off, c0, c1);
  rebase(&bv->allele_off, in->allele_off, b0, b1);
  rebase(&bv->apath_off, in->apath_off, a0, a1);
  rebase(&bv->node_off, in->node_off, c0, c1);
  rebase(&bv->read_off, in->read_off, c0, c1);
  rebase(&bv->rnode_off, in->rnode_off, r0, r1);
  ahsoka_chain_batch& v = bv->view;
  std::memset(&v, 0, sizeof(v));
  v.n_chains = c1 - c0;
  v.chain_id = in->chain_id ? in->chain_id + c0 : nullptr;
  v.bubble_off = bv->bubble_off.data();
  v.allele_off = bv->allele_off.data();
  v.apath_off = bv->apath_off.data();
  v.apath_node = in->apath_node ? in->apath_node + p0 : nullptr;
  v.apath_sign = in->apath_sign ? in->apath_sign + p0 : nullptr;
  v.genotype = in->genotype ? in->genotype + a0 : nullptr;
  v.node_off = bv->node_off.data();
  v.node_id = in->node_id ? in->node_id + n0 : nullptr;
  v.read_off = bv->read_off.data();
  v.rnode_off = bv->rnode_off.data();
  v.rnode = in->rnode ? in->rnode + e0 : nullptr;
  v.read_mapq = in->read_mapq ? in->read_mapq + r0 : nullptr;
  v.read_name = in->read_name ? in->read_name + r0 : nullptr;
  v.read_flags = in->read_flags ? in->read_flags + r0 : nullptr;
  v.s1_order = in->s1_order ? in->s1_order + b0 : nullptr;
}

// the chains `idx` (ascending) of `in` as a batch of their own (a packed copy)
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

// joins whatever threads were started, on every way out (a joinable std::thread that is
// destroyed calls std::terminate -- e.g. when starting the next one throws)
struct JoinAll {
  std::vector<std::thread>& th;
  ~JoinAll() {
    for (auto& t : th)
      if (t.joinable()) t.join();
  }
};

int phase_multi(const ahsoka_chain_batch* in, const ahsoka_phase_params* prm, const int32_t* devices, int32_t n_devices,
                ahsoka_phase_result* out) {
  const int C = in->n_chains;
  Plan pl;
  make_plan(in, prm->min_mapq, n_devices, &pl);

  // ---- one thread, handle and stream per device; a thread first cuts its own shard ----
  std::vector<BlockView> views((size_t)n_devices);
  std::vector<ahk::HostBatch> shards((size_t)n_devices);
  std::vector<ahsoka_phase_result> res((size_t)n_devices);
  std::vector<int> rcs((size_t)n_devices, AHSOKA_OK);
  for (int d = 0; d < n_devices; d++) std::memset(&res[(size_t)d], 0, sizeof(ahsoka_phase_result));
  {
    std::vector<std::thread> th;
    th.reserve((size_t)n_devices);
    JoinAll guard{th};
    for (int d = 0; d < n_devices; d++)
      th.emplace_back([&, d]() {
        try {  // (an exception that left this lambda would terminate the process)
          const ahsoka_chain_batch* mine;
          if (pl.blocks) {
            make_block_view(in, pl.first[(size_t)d], pl.first[(size_t)d + 1], &views[(size_t)d]);
            mine = &views[(size_t)d].view;
          } else {
            gather_chains(in, pl.mine[(size_t)d], &shards[(size_t)d]);
            mine = &shards[(size_t)d].view;
          }
          ahsoka_phase_params pd = *prm;
          pd.device = devices[d];
          rcs[(size_t)d] = ahsoka_phase_batch(mine, &pd, &res[(size_t)d]);
        } catch (const std::bad_alloc&) {
          rcs[(size_t)d] = AHSOKA_E_NOMEM;
        } catch (...) {
          rcs[(size_t)d] = AHSOKA_E_HIP;
        }
      });
  }
  int rc = AHSOKA_OK;
  for (int d = 0; d < n_devices; d++)
    if (rcs[(size_t)d] && !rc) rc = rcs[(size_t)d];
  struct FreeAll {  // the per-device results are released on every way out, a throwing merge included
    std::vector<ahsoka_phase_result>& res;
    std::vector<int>& rcs;
    ~FreeAll() {
      for (size_t d = 0; d < res.size(); d++)
        if (!rcs[d]) ahsoka_result_free(&res[d]);
    }
  } free_all{res, rcs};
  if (rc) return rc;

  // ---- merge into the caller's chain order ------------------------------------------
  const int p = prm->ploidy;
  const bool stages = prm->want_stages != 0;
  std::vector<int32_t> status((size_t)C), nreads((size_t)C), nclus((size_t)C), pos, ha, hc, fname, fclus, score;
  std::vector<int64_t> niter((size_t)C), pos_off((size_t)C + 1, 0), vmat_off((size_t)C + 1, 0), fread_off((size_t)C + 1, 0),
      score_off((size_t)C + 1, 0);
  std::vector<int8_t> vfull, vpart;
  std::vector<int> at((size_t)n_devices, 0);  // next local chain of every shard
  double kernel_ms = 0, elapsed_ms = 0;
  {
    size_t npos = 0;
    for (int d = 0; d < n_devices; d++) {
      kernel_ms = std::max(kernel_ms, res[(size_t)d].kernel_ms);
      elapsed_ms = std::max(elapsed_ms, res[(size_t)d].elapsed_ms);
      npos += (size_t)res[(size_t)d].pos_off[res[(size_t)d].n_chains];
    }
    pos.reserve(npos);
    ha.reserve(npos * (size_t)p);
    hc.reserve(npos * (size_t)p);
  }
  for (int c = 0; c < C; c++) {
    const int d = pl.owner[(size_t)c];
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
  bool ok = out->status && out->pos_off && out->pos && out->hap_allele && out->hap_cluster && out->n_reads && out->n_clusters &&
            out->n_iterations;
  if (stages) {
    out->vmat_off = hostdup(vmat_off);
    out->vfull = hostdup(vfull);
    out->vpart = hostdup(vpart);
    out->fread_off = hostdup(fread_off);
    out->fread_name = hostdup(fname);
    out->fread_cluster = hostdup(fclus);
    out->score_off = hostdup(score_off);
    out->score_q = hostdup(score);
    ok = ok && out->vmat_off && out->vfull && out->vpart && out->fread_off && out->fread_name && out->fread_cluster && out->score_off &&
         out->score_q;
  }
  if (!ok) {  // a failed malloc must not come back as AHSOKA_OK with a NULL array
    ahsoka_result_free(out);
    return AHSOKA_E_NOMEM;
  }
  out->kernel_ms = kernel_ms;
  out->elapsed_ms = elapsed_ms;
  return AHSOKA_OK;
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
  if (int e = ahk::validate_batch(in)) return e;  // the shards are cut out of the arrays: check them first
  try {
    return phase_multi(in, prm, devices, n_devices, out);
  } catch (const std::bad_alloc&) {
    return AHSOKA_E_NOMEM;
  } catch (...) {  // e.g. std::system_error: a host thread could not be started
    return AHSOKA_E_HIP;
  }
}

// Test / measurement hook (host only, no GPU): the static partition of ahsoka_phase_batch_multi
// and the per-shard host work up to the point where a device thread would start its upload.
// owner_out [C] (may be NULL) receives the shard of every chain; *blocks_out 1 = contiguous
// blocks (zero-copy views), 0 = LPT (packed copies); *serial_ms the serial prologue (cost model +
// partition), *shard_ms_max the longest per-shard view/pack time (these run concurrently, one per
// device thread).  With check != 0 every shard view is validated like a caller's batch.
extern "C" int ahsoka_test_multi_plan(const ahsoka_chain_batch* in, const ahsoka_phase_params* prm, int32_t n_devices, int32_t* owner_out,
                                      int32_t* blocks_out, double* serial_ms, double* shard_ms_max, int32_t check) {
  if (!in || !prm || n_devices < 1 || n_devices > 64 || in->n_chains < 0) return AHSOKA_E_ARG;
  try {
    Plan pl;
    make_plan(in, prm->min_mapq, n_devices, &pl);
    double worst = 0;
    for (int d = 0; d < n_devices; d++) {
      const auto t0 = std::chrono::steady_clock::now();
      BlockView bv;
      ahk::HostBatch hb;
      const ahsoka_chain_batch* mine;
      if (pl.blocks) {
        make_block_view(in, pl.first[(size_t)d], pl.first[(size_t)d + 1], &bv);
        mine = &bv.view;
      } else {
        gather_chains(in, pl.mine[(size_t)d], &hb);
        mine = &hb.view;
      }
      worst = std::max(worst, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      if (check)
        if (int e = ahk::validate_batch(mine)) return e;
    }
    if (owner_out)
      for (int c = 0; c < in->n_chains; c++) owner_out[c] = pl.owner[(size_t)c];
    if (blocks_out) *blocks_out = pl.blocks ? 1 : 0;
    if (serial_ms) *serial_ms = pl.prologue_ms;
    if (shard_ms_max) *shard_ms_max = worst;
    return AHSOKA_OK;
  } catch (const std::bad_alloc&) {
    return AHSOKA_E_NOMEM;
  } catch (...) {
    return AHSOKA_E_HIP;
  }
}
