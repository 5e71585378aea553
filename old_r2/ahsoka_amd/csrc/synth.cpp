// synth.cpp -- deterministic synthetic bubble-chain + read-path workload
// (BASELINE.json configs c1..c5).
//
// The reference ships no example data, so every input is
// generated: per chain B simple biallelic bubbles in series
//     s0 -{x0a,x0b}- s1 -{x1a,x1b}- ... - sB          (3B+1 nodes)
// which is what Graph::findBubble (src/graph.cpp:370-501) recognises and what
// findPathsSimple (src/chainstoreadset.cpp:17-30) turns into the two allele
// paths [src, inner_i, sink].  Reads are contiguous bubble windows of one
// truth haplotype with per-bubble allele errors; identities 0.900..0.999 give
// mapq = (int)(id*100) in 90..99 (src/alignmentstoreadset.cpp:117).
//
// Every chain draws from its own splitmix64 stream seeded by (seed, chain
// index), so a rank can generate exactly its shard of the global workload.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "host_batch.h"

namespace {

struct SplitMix64 {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint32_t below(uint32_t n) { return (uint32_t)(next() % n); }
  double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

SplitMix64 chain_rng(uint64_t seed, int64_t chain) {
  SplitMix64 m{seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(chain + 1))};
  m.next();
  return SplitMix64{m.next()};
}

struct ChainShape { int B, R; };

ChainShape chain_shape(const ahsoka_synth_spec& sp, SplitMix64& rng) {
  if (sp.b_mode == 0) return ChainShape{sp.n_bubbles, sp.n_reads};
  // config 5: B = floor(x), x ~ Pareto(alpha = 1.8, x_m = 1) clipped to [1,500]
  double u = rng.unit();
  double x = std::pow(1.0 - u, -1.0 / 1.8);
  int B = (int)x;
  if (B < 1) B = 1;
  if (B > 500) B = 500;
  int R = (40 * B + 10) / 11;  // ceil(40*B/11): about 40x depth
  return ChainShape{B, R};
}

struct SynthChain {
  int B, R;
  std::vector<uint8_t> hap;              // [ploidy][B]
  std::vector<std::vector<int>> reads;   // chain-local node paths
  std::vector<int> idnum;                // identity * 1000 (900..999)
};

void make_chain(const ahsoka_synth_spec& sp, int64_t ci, SynthChain& ch) {
  SplitMix64 rng = chain_rng(sp.seed, ci);
  ChainShape shp = chain_shape(sp, rng);
  ch.B = shp.B;
  ch.R = shp.R;
  const int p = sp.ploidy;
  ch.hap.assign((size_t)p * ch.B, 0);
  for (int k = 0; k < ch.B; k++) {
    if (p == 2) {
      uint8_t h0 = (uint8_t)(rng.next() & 1);
      ch.hap[k] = h0;
      ch.hap[ch.B + k] = (uint8_t)(1 - h0);
    } else {
      // uniform over the 2^p - 2 non-constant assignments
      uint32_t code = 1 + rng.below((1u << p) - 2);
      for (int i = 0; i < p; i++) ch.hap[(size_t)i * ch.B + k] = (uint8_t)((code >> i) & 1);
    }
  }
  const double pstop = 1.0 / (sp.mean_span + 1.0);  // Geom on {0,1,..} with mean mean_span
  ch.reads.assign(ch.R, {});
  ch.idnum.assign(ch.R, 0);
  for (int r = 0; r < ch.R; r++) {
    int u = (int)rng.below((uint32_t)ch.B);
    int g = 0;
    while (rng.unit() >= pstop && g < 100000) g++;
    int s = 1 + g;
    if (s > ch.B - u) s = ch.B - u;
    int h = (int)rng.below((uint32_t)p);
    std::vector<int>& nodes = ch.reads[r];
    nodes.push_back(3 * u);
    for (int k = u; k < u + s; k++) {
      int a = ch.hap[(size_t)h * ch.B + k];
      if (rng.unit() < sp.allele_err) a = 1 - a;
      nodes.push_back(3 * k + 1 + a);
      nodes.push_back(3 * k + 3);
    }
    ch.idnum[r] = 900 + (int)rng.below(100);
  }
}

int check_spec(const ahsoka_synth_spec* sp) {
  if (!sp || sp->n_chains < 0 || sp->ploidy < 2 || sp->ploidy > 8 || sp->first_chain < 0) return AHSOKA_E_ARG;
  if (sp->b_mode == 0 && (sp->n_bubbles < 1 || sp->n_reads < 0)) return AHSOKA_E_ARG;
  if (sp->mean_span < 0 || sp->allele_err < 0 || sp->allele_err > 1) return AHSOKA_E_ARG;
  return AHSOKA_OK;
}

}  // namespace

namespace {

// the chains `list` (global indices, any order) of the workload `sp` as one batch
int build_batch(const ahsoka_synth_spec* sp, const std::vector<int64_t>& list, ahsoka_chain_batch** out) {
  ahk::HostBatch* hb = new (std::nothrow) ahk::HostBatch();
  if (!hb) return AHSOKA_E_NOMEM;
  try {
    // global node ids: chain ci's nodes follow those of chains 0..ci-1
    int64_t maxc = -1;
    for (int64_t ci : list) maxc = std::max(maxc, ci);
    std::vector<int64_t> base_of;  // only for non-contiguous lists: node-id base of every chain up to maxc
    int64_t base = 0;
    const bool contiguous = [&]() {
      for (size_t i = 1; i < list.size(); i++)
        if (list[i] != list[i - 1] + 1) return false;
      return true;
    }();
    if (contiguous) {
      for (int64_t ci = 0; ci < (list.empty() ? 0 : list[0]); ci++) {
        SplitMix64 rng = chain_rng(sp->seed, ci);
        base += 3 * (int64_t)chain_shape(*sp, rng).B + 1;
      }
    } else {
      base_of.resize((size_t)maxc + 1);
      int64_t acc = 0;
      for (int64_t ci = 0; ci <= maxc; ci++) {
        base_of[(size_t)ci] = acc;
        SplitMix64 rng = chain_rng(sp->seed, ci);
        acc += 3 * (int64_t)chain_shape(*sp, rng).B + 1;
      }
    }
    hb->bubble_off.push_back(0);
    hb->allele_off.push_back(0);
    hb->apath_off.push_back(0);
    hb->node_off.push_back(0);
    hb->read_off.push_back(0);
    hb->rnode_off.push_back(0);
    hb->has_genotype = sp->ploidy != 2;
    SynthChain ch;
    for (int64_t ci : list) {
      if (!contiguous) base = base_of[(size_t)ci];
      make_chain(*sp, ci, ch);
      hb->chain_id.push_back((int32_t)ci);
      for (int k = 0; k < ch.B; k++) {
        int dosage1 = 0;
        for (int i = 0; i < sp->ploidy; i++) dosage1 += ch.hap[(size_t)i * ch.B + k];
        for (int a = 0; a < 2; a++) {
          hb->apath_node.push_back(3 * k);
          hb->apath_node.push_back(3 * k + 1 + a);
          hb->apath_node.push_back(3 * k + 3);
          hb->apath_off.push_back((int64_t)hb->apath_node.size());
          hb->genotype.push_back(a == 0 ? sp->ploidy - dosage1 : dosage1);
        }
        hb->allele_off.push_back((int64_t)hb->apath_off.size() - 1);
      }
      hb->bubble_off.push_back((int64_t)hb->allele_off.size() - 1);
      ahk::ref_s1_order(ch.B, &hb->s1_order);
      for (int x = 0; x < 3 * ch.B + 1; x++) hb->node_id.push_back((int32_t)(base + x + 1));
      hb->node_off.push_back((int64_t)hb->node_id.size());
      base += 3 * (int64_t)ch.B + 1;
      for (int r = 0; r < ch.R; r++) {
        for (int nd : ch.reads[r]) hb->rnode.push_back(nd);
        hb->rnode_off.push_back((int64_t)hb->rnode.size());
        // what the reference derives from the token "id:f:0.NNN" the text form carries:
        // std::stof, then float*100 truncated (:117) resp. compared with 90 (:245)
        char tok[16];
        std::snprintf(tok, sizeof tok, "0.%03d", ch.idnum[r]);
        const float ident = std::strtof(tok, nullptr);
        hb->read_mapq.push_back((int32_t)(ident * 100));
        hb->read_flags.push_back(ahk::ref_gt90_flag(ident));
        hb->read_name.push_back(r);
      }
      hb->read_off.push_back((int64_t)hb->read_mapq.size());
    }
    hb->finish();
  } catch (const std::bad_alloc&) {
    delete hb;
    return AHSOKA_E_NOMEM;
  }
  *out = &hb->view;
  return AHSOKA_OK;
}

}  // namespace

extern "C" int ahsoka_synth_batch(const ahsoka_synth_spec* sp, ahsoka_chain_batch** out) {
  if (int e = check_spec(sp)) return e;
  if (!out) return AHSOKA_E_ARG;
  std::vector<int64_t> list((size_t)sp->n_chains);
  for (int c = 0; c < sp->n_chains; c++) list[(size_t)c] = (int64_t)sp->first_chain + c;
  return build_batch(sp, list, out);
}

// chains picked by global index (a rank's shard of a cost-weighted partition is not a range)
extern "C" int ahsoka_synth_select(const ahsoka_synth_spec* sp, const int32_t* chains, int32_t n, ahsoka_chain_batch** out) {
  if (int e = check_spec(sp)) return e;
  if (!out || n < 0 || (n > 0 && !chains)) return AHSOKA_E_ARG;
  std::vector<int64_t> list((size_t)n);
  for (int i = 0; i < n; i++) {
    if (chains[i] < 0) return AHSOKA_E_ARG;
    list[(size_t)i] = chains[i];
  }
  return build_batch(sp, list, out);
}

// bubbles and reads of every chain of the workload, without generating it (every rank can
// compute the same cost-weighted partition from this, without communication)
extern "C" int ahsoka_synth_shapes(const ahsoka_synth_spec* sp, int32_t* n_bubbles, int32_t* n_reads) {
  if (int e = check_spec(sp)) return e;
  if (!n_bubbles || !n_reads) return AHSOKA_E_ARG;
  for (int c = 0; c < sp->n_chains; c++) {
    SplitMix64 rng = chain_rng(sp->seed, (int64_t)sp->first_chain + c);
    const ChainShape shp = chain_shape(*sp, rng);
    n_bubbles[c] = shp.B;
    n_reads[c] = shp.R;
  }
  return AHSOKA_OK;
}

extern "C" void ahsoka_batch_host_free(ahsoka_chain_batch* b) {
  if (!b) return;
  delete reinterpret_cast<ahk::HostBatch*>(b);
}

// Truth haplotypes of the same workload: for every chain, ploidy rows of B alleles
// (row-major, chains concatenated).  Lets tests measure how well the phasing
// recovers what the reads were drawn from.  Returns the number of bytes needed.
extern "C" int64_t ahsoka_synth_truth(const ahsoka_synth_spec* sp, uint8_t* out, int64_t cap) {
  if (check_spec(sp)) return AHSOKA_E_ARG;
  int64_t need = 0;
  SynthChain ch;
  for (int c = 0; c < sp->n_chains; c++) {
    make_chain(*sp, (int64_t)sp->first_chain + c, ch);
    const int64_t bytes = (int64_t)sp->ploidy * ch.B;
    if (out && need + bytes <= cap) std::memcpy(out + need, ch.hap.data(), (size_t)bytes);
    need += bytes;
  }
  return need;
}

// GFA + GAF text of the same workload (small configs; used to pin the ingest
// against the reference's own parsers).  All S lines precede all L lines and
// both directions of every link are written.
extern "C" int ahsoka_synth_write_text(const ahsoka_synth_spec* sp, const char* gfa_path, const char* gaf_path) {
  if (int e = check_spec(sp)) return e;
  if (!gfa_path || !gaf_path) return AHSOKA_E_ARG;
  FILE* gfa = std::fopen(gfa_path, "w");
  if (!gfa) return AHSOKA_E_IO;
  FILE* gaf = std::fopen(gaf_path, "w");
  if (!gaf) { std::fclose(gfa); return AHSOKA_E_IO; }
  std::vector<SynthChain> chains(sp->n_chains);
  std::vector<int64_t> bases(sp->n_chains);
  int64_t base = 0;
  for (int64_t ci = 0; ci < sp->first_chain; ci++) {
    SplitMix64 rng = chain_rng(sp->seed, ci);
    base += 3 * (int64_t)chain_shape(*sp, rng).B + 1;
  }
  for (int c = 0; c < sp->n_chains; c++) {
    make_chain(*sp, (int64_t)sp->first_chain + c, chains[c]);
    bases[c] = base;
    base += 3 * (int64_t)chains[c].B + 1;
  }
  for (int c = 0; c < sp->n_chains; c++)
    for (int x = 0; x < 3 * chains[c].B + 1; x++) std::fprintf(gfa, "S\tutg%06lldl\tA\n", (long long)(bases[c] + x + 1));
  auto link = [&](long long a, long long b) {
    std::fprintf(gfa, "L\tutg%06lldl\t+\tutg%06lldl\t+\t0M\n", a, b);
    std::fprintf(gfa, "L\tutg%06lldl\t-\tutg%06lldl\t-\t0M\n", b, a);
  };
  for (int c = 0; c < sp->n_chains; c++)
    for (int k = 0; k < chains[c].B; k++) {
      long long s = bases[c] + 3 * k + 1;
      link(s, s + 1);
      link(s, s + 2);
      link(s + 1, s + 3);
      link(s + 2, s + 3);
    }
  for (int c = 0; c < sp->n_chains; c++)
    for (int r = 0; r < chains[c].R; r++) {
      std::string path;
      char tmp[32];
      for (int nd : chains[c].reads[r]) {
        std::snprintf(tmp, sizeof tmp, ">utg%06lldl", (long long)(bases[c] + nd + 1));
        path += tmp;
      }
      std::fprintf(gaf, "c%lld_r%d\t1000\t0\t1000\t+\t%s\t1000\t0\t1000\t1000\t1000\t60\tNM:i:0\tAS:f:0\tdv:f:0\tid:f:0.%03d\n",
                   (long long)sp->first_chain + c, r, path.c_str(), chains[c].idnum[r]);
    }
  std::fclose(gfa);
  std::fclose(gaf);
  return AHSOKA_OK;
}
