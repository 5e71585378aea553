/*
 * oracle.cpp -- CPU re-statement of Ahsoka's per-chain phasing path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ahsoka_amd/ (the product) may
 * include, link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * PARITY STATUS
 *   - Stages owned by Ahsoka (S1, S1b, S1c, S2, S2b, S6, S8; citations below
 *     are src/alignmentstoreadset.cpp line numbers of /root/reference) follow
 *     the reference text statement by statement.  `is_subset` is additionally
 *     pinned against the reference's own compiled function (oracle/_ref, see
 *     oracle/Makefile and tests/test_oracle_pin.py).
 *   - Stages S3 (ReadSet::sort), S4 (ReadScoring::scoreReadsetLocal), S5
 *     (ClusterEditingSolver::run) and S7 (HaploThreader::computePaths) live in
 *     WhatsHap @ 8f4c0c070d0b5d8e6d2c03e363965d0efb50a960, which is NOT in
 *     /root/reference and not on this machine; the reference ships no tests
 *     or golden vectors for them.  They are re-stated from the published
 *     algorithms (Schrinner et al., Genome Biology 2020; Boecker et al.'s
 *     induced-cost heuristic) with every tie-break written down in DESIGN.md
 *     section "Oracle specification".  For these stages:  PARITY UNPINNED.
 *
 * The structure mirrors the reference: one function per reference function,
 * plain containers, no cleverness.  It is deliberately independent of the
 * product sources (it shares only the C-ABI header, which is the interface).
 */
#include "../include/ahsoka_phase.h"

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// data model (what WhatsHap's Read / ReadSet give the reference)
// ---------------------------------------------------------------------------
struct ORead {
  int name;                               // chain-local name id
  int mapq;                               // Read ctor arg `int mapq` (:117,234)
  std::vector<std::pair<int, int>> var;   // (position=bubble id, allele), sorted by position
  bool has_pos(int p) const {
    for (auto& v : var)
      if (v.first == p) return true;
    return false;
  }
  void add_variant(int p, int a) {  // addVariant + sortVariants
    var.emplace_back(p, a);
    std::sort(var.begin(), var.end());
  }
  int first_pos() const { return var.front().first; }
  int last_pos() const { return var.back().first; }
};

struct OReadSet {
  std::vector<ORead> reads;
  std::unordered_map<int, int> by_name;  // getByName(name, 0)
  int find(int name) const {
    auto it = by_name.find(name);
    return it == by_name.end() ? -1 : it->second;
  }
  void add(ORead r) {
    by_name[r.name] = (int)reads.size();
    reads.push_back(std::move(r));
  }
  std::vector<int> positions() const {  // ReadSet::get_positions(): sorted distinct
    std::set<int> s;
    for (auto& r : reads)
      for (auto& v : r.var) s.insert(v.first);
    return std::vector<int>(s.begin(), s.end());
  }
};

struct ChainView {
  const ahsoka_chain_batch* b;
  int c;
  int64_t b0, b1;  // bubbles
  int64_t r0, r1;  // alignments
  int nbubbles() const { return (int)(b1 - b0); }
  int nalign() const { return (int)(r1 - r0); }
  int nalleles(int k) const { return (int)(b->allele_off[b0 + k + 1] - b->allele_off[b0 + k]); }
  std::vector<int> apath(int k, int a) const {
    int64_t ai = b->allele_off[b0 + k] + a;
    return std::vector<int>(b->apath_node + b->apath_off[ai], b->apath_node + b->apath_off[ai + 1]);
  }
};

// is_subset, src/alignmentstoreadset.cpp:495-548.  `aln_sorted` is
// alignment.getRawIds() after the std::sort of :496-497 (node ids replaced by
// chain-local indices: an injective relabelling, -1 for foreign nodes).
bool is_subset(std::vector<int> allelepath, const std::vector<int>& aln_sorted, bool take_partial) {
  if (!take_partial) {
    std::sort(allelepath.begin(), allelepath.end());  // :505
    return std::includes(aln_sorted.begin(), aln_sorted.end(), allelepath.begin(), allelepath.end());
  }
  // :510-512 pop_back + erase(begin).  A path of < 2 nodes is UB in the
  // reference; defined here as "empty middle".
  if (allelepath.size() >= 2) {
    allelepath.pop_back();
    allelepath.erase(allelepath.begin(), allelepath.begin() + 1);
  } else {
    allelepath.clear();
  }
  std::sort(allelepath.begin(), allelepath.end());
  return std::includes(aln_sorted.begin(), aln_sorted.end(), allelepath.begin(), allelepath.end());  // :545
}

// ---------------------------------------------------------------------------
// deterministic natural logarithm (shared SPEC with the device code: only
// IEEE-754 +,-,*,/ in a fixed order, so CPU and GPU agree bit for bit)
// ---------------------------------------------------------------------------
double det_log(double x) {
  uint64_t bits;
  std::memcpy(&bits, &x, 8);
  int e = (int)((bits >> 52) & 0x7ff) - 1023;
  uint64_t mb = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
  double m;
  std::memcpy(&m, &mb, 8);
  if (m > 1.4142135623730951) {
    m = m * 0.5;
    e += 1;
  }
  double s = (m - 1.0) / (m + 1.0);
  double s2 = s * s;
  double p = 1.0 / 23.0;
  p = p * s2 + 1.0 / 21.0;
  p = p * s2 + 1.0 / 19.0;
  p = p * s2 + 1.0 / 17.0;
  p = p * s2 + 1.0 / 15.0;
  p = p * s2 + 1.0 / 13.0;
  p = p * s2 + 1.0 / 11.0;
  p = p * s2 + 1.0 / 9.0;
  p = p * s2 + 1.0 / 7.0;
  p = p * s2 + 1.0 / 5.0;
  p = p * s2 + 1.0 / 3.0;
  p = p * s2 + 1.0;
  double r = (2.0 * s) * p;
  return r + (double)e * 0.6931471805599453;
}

// ---------------------------------------------------------------------------
// S4: ReadScoring::scoreReadsetLocal(&sim, testset, minOverlap, ploidy)
// (call site :308-311).  Re-statement, see DESIGN.md "S4".
// Output: dense symmetric n x n matrix of quantised scores (0 = no entry).
// ---------------------------------------------------------------------------
constexpr int32_t W_PERM = INT32_MAX;
constexpr int32_t W_FORB = INT32_MIN;
constexpr int32_t W_CLAMP = 1 << 30;

int32_t quantise_score(double score) {
  float f = (float)score;  // TriangleSparseMatrix stores float
  double x = (double)f * 65536.0;
  double r = std::floor(x + 0.5);
  if (r > (double)W_CLAMP) r = (double)W_CLAMP;
  if (r < -(double)W_CLAMP) r = -(double)W_CLAMP;
  return (int32_t)r;
}

void score_readset_local(const std::vector<ORead>& reads, int min_overlap, int ploidy,
                         std::vector<int32_t>& W) {
  const int n = (int)reads.size();
  W.assign((size_t)n * n, 0);
  std::vector<int32_t> ov((size_t)n * n, 0), df((size_t)n * n, 0);
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++) {
      // shared positions / differing alleles (both variant lists sorted)
      int o = 0, d = 0;
      size_t a = 0, b = 0;
      const auto& va = reads[i].var;
      const auto& vb = reads[j].var;
      while (a < va.size() && b < vb.size()) {
        if (va[a].first < vb[b].first) a++;
        else if (va[a].first > vb[b].first) b++;
        else {
          o++;
          if (va[a].second != vb[b].second) d++;
          a++;
          b++;
        }
      }
      ov[(size_t)i * n + j] = ov[(size_t)j * n + i] = o;
      df[(size_t)i * n + j] = df[(size_t)j * n + i] = d;
    }
  // local (per-read) estimates of the Hamming rate between reads of the same
  // haplotype and of different haplotypes: pooled rate of the lowest
  // |N_i|/ploidy neighbours vs pooled rate of the rest.
  std::vector<double> same(n), diff(n);
  for (int i = 0; i < n; i++) {
    std::vector<uint32_t> keys;
    for (int j = 0; j < n; j++) {
      if (j == i) continue;
      uint32_t o = (uint32_t)ov[(size_t)i * n + j], d = (uint32_t)df[(size_t)i * n + j];
      if ((int)o < min_overlap || o == 0) continue;
      // (rate, overlap): exact order for o <= 4095.  d == o gives rate 2^20, the only value above
      // 2^20 - 2^20/o: it is stored as 2^20 - 2 (same place in the order, d still recovered below) so
      // that the key fits 32 bits with a 12-bit overlap field
      uint32_t rate = (d << 20) / o;
      if (rate > (1u << 20) - 2u) rate = (1u << 20) - 2u;
      keys.push_back((rate << 12) | o);
    }
    std::sort(keys.begin(), keys.end());
    size_t m = keys.size() / (size_t)ploidy;
    if (m == 0 && !keys.empty()) m = 1;
    int64_t ds = 0, os = 0, dr = 0, orr = 0;
    for (size_t k = 0; k < keys.size(); k++) {
      uint32_t o = keys[k] & 4095u;
      uint32_t rate = keys[k] >> 12;
      // recover d: d = ceil(rate*o / 2^20) is exact because rate = floor(d*2^20/o)
      uint64_t dd = ((uint64_t)rate * o + ((1u << 20) - 1)) >> 20;
      if (k < m) { ds += (int64_t)dd; os += o; }
      else { dr += (int64_t)dd; orr += o; }
    }
    double s = os > 0 ? (double)ds / (double)os : 0.05;
    double f = orr > 0 ? (double)dr / (double)orr : 0.5;
    if (s < 0.001) s = 0.001;
    if (s > 0.999) s = 0.999;
    if (f < 0.001) f = 0.001;
    if (f > 0.999) f = 0.999;
    same[i] = s;
    diff[i] = f;
  }
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++) {
      int o = ov[(size_t)i * n + j], d = df[(size_t)i * n + j];
      if (o < min_overlap || o == 0) continue;
      double S = (same[i] + same[j]) * 0.5;
      double D = (diff[i] + diff[j]) * 0.5;
      double t1 = (double)d * (det_log(S) - det_log(D));
      double t2 = (double)(o - d) * (det_log(1.0 - S) - det_log(1.0 - D));
      int32_t q = quantise_score(t1 + t2);
      W[(size_t)i * n + j] = W[(size_t)j * n + i] = q;
    }
}

// ---------------------------------------------------------------------------
// S5: ClusterEditingSolver(sim,false).run()  (call site :312-314)
// induced-cost heuristic on exact integer weights; DESIGN.md "S5".
// ---------------------------------------------------------------------------
constexpr int64_t BIG = (int64_t)1 << 40;
inline int64_t wval(int32_t w) { return w == W_PERM ? BIG : (w == W_FORB ? -BIG : (int64_t)w); }
inline int64_t icf_term(int64_t p, int64_t q) { return (p > 0 && q > 0) ? std::min(p, q) : 0; }
inline int64_t icp_term(int64_t p, int64_t q) {
  if (p < 0 && q > 0) return std::min(q, -p);
  if (p > 0 && q < 0) return std::min(-q, p);
  return 0;
}
inline bool is_live(int32_t w) { return w != 0 && w != W_PERM && w != W_FORB; }

struct ClusterEditing {
  int n;
  std::vector<int32_t> W;
  std::vector<int64_t> ICF, ICP;
  std::vector<int> clique;
  std::vector<std::vector<int>> members;
  // row caches
  struct Best { int64_t val; int col; };
  std::vector<Best> bestF, bestP;
  std::vector<char> dirty;
  int64_t iterations = 0;
  bool brute = false;  // recompute everything from the definition each iteration

  size_t ix(int a, int b) const { return (size_t)a * n + b; }

  void full_recompute() {
    for (int a = 0; a < n; a++)
      for (int b = a + 1; b < n; b++) {
        int32_t w = W[ix(a, b)];
        if (!is_live(w)) continue;
        int64_t f = w > 0 ? w : 0, p = w < 0 ? -(int64_t)w : 0;
        for (int x = 0; x < n; x++) {
          if (x == a || x == b) continue;
          int64_t wa = wval(W[ix(a, x)]), wb = wval(W[ix(b, x)]);
          f += icf_term(wa, wb);
          p += icp_term(wa, wb);
        }
        ICF[ix(a, b)] = ICF[ix(b, a)] = f;
        ICP[ix(a, b)] = ICP[ix(b, a)] = p;
      }
  }

  void rescan(int u) {
    Best f{-1, -1}, p{-1, -1};
    for (int c = 0; c < n; c++) {
      if (c == u || !is_live(W[ix(u, c)])) continue;
      if (ICF[ix(u, c)] > f.val) f = Best{ICF[ix(u, c)], c};
      if (ICP[ix(u, c)] > p.val) p = Best{ICP[ix(u, c)], c};
    }
    bestF[u] = f;
    bestP[u] = p;
    dirty[u] = 0;
  }

  void touch(int row, int col) {
    if (dirty[row]) return;
    if (bestF[row].col == col || bestP[row].col == col) { dirty[row] = 1; return; }
    if (!is_live(W[ix(row, col)])) return;
    int64_t f = ICF[ix(row, col)], p = ICP[ix(row, col)];
    if (f > bestF[row].val || (f == bestF[row].val && col < bestF[row].col)) bestF[row] = Best{f, col};
    if (p > bestP[row].val || (p == bestP[row].val && col < bestP[row].col)) bestP[row] = Best{p, col};
  }

  // one edge changes weight o -> N (N is W_PERM or W_FORB): update the induced
  // costs of the two other edges of every triangle through it.
  void set_pair(int x, int y, int32_t N) {
    int32_t o = W[ix(x, y)];
    if (o == N) return;
    int64_t ov = wval(o), nv = wval(N);
    for (int w = 0; w < n; w++) {
      if (w == x || w == y) continue;
      int64_t a = wval(W[ix(x, w)]), b = wval(W[ix(y, w)]);
      if (is_live(W[ix(y, w)])) {
        int64_t df = icf_term(nv, a) - icf_term(ov, a), dp = icp_term(nv, a) - icp_term(ov, a);
        if (df != 0 || dp != 0) {
          ICF[ix(y, w)] += df; ICF[ix(w, y)] += df;
          ICP[ix(y, w)] += dp; ICP[ix(w, y)] += dp;
          dirty[y] = 1; dirty[w] = 1;
        }
      }
      if (is_live(W[ix(x, w)])) {
        int64_t df = icf_term(nv, b) - icf_term(ov, b), dp = icp_term(nv, b) - icp_term(ov, b);
        if (df != 0 || dp != 0) {
          ICF[ix(x, w)] += df; ICF[ix(w, x)] += df;
          ICP[ix(x, w)] += dp; ICP[ix(w, x)] += dp;
          dirty[x] = 1; dirty[w] = 1;
        }
      }
    }
    W[ix(x, y)] = W[ix(y, x)] = N;
    dirty[x] = 1;
    dirty[y] = 1;
  }

#ifdef ORACLE_CE_TRACE
  int trace_fa = 0, trace_fb = 0;
#endif
  void merge(int u, int v) {
    std::vector<int> A = members[clique[u]], B = members[clique[v]];
    std::vector<char> inAB(n, 0);
    for (int x : A) inAB[x] = 1;
    for (int y : B) inAB[y] = 1;
    // forbidden neighbours must be read BEFORE A x B is rewritten
    std::vector<int> fa, fb;
    for (int z = 0; z < n; z++) {
      if (inAB[z]) continue;
      bool za = W[ix(z, A[0])] == W_FORB, zb = W[ix(z, B[0])] == W_FORB;
      if (za && !zb) fa.push_back(z);
      if (zb && !za) fb.push_back(z);
    }
#ifdef ORACLE_CE_TRACE
    trace_fa = (int)fa.size();
    trace_fb = (int)fb.size();
#endif
    for (int x : A)
      for (int y : B) set_pair(x, y, W_PERM);
    for (int z : fa)
      for (int y : B) set_pair(y, z, W_FORB);
    for (int z : fb)
      for (int x : A) set_pair(x, z, W_FORB);
    int ca = clique[u], cb = clique[v];
    for (int y : B) { clique[y] = ca; members[ca].push_back(y); }
    std::sort(members[ca].begin(), members[ca].end());
    members[cb].clear();
  }

  void forbid(int u, int v) {
    std::vector<int> A = members[clique[u]], B = members[clique[v]];
    for (int x : A)
      for (int y : B) set_pair(x, y, W_FORB);
  }

  std::vector<int> run(const std::vector<int32_t>& W0, int n_) {
    n = n_;
    W = W0;
    ICF.assign((size_t)n * n, 0);
    ICP.assign((size_t)n * n, 0);
    clique.resize(n);
    members.assign(n, {});
    for (int u = 0; u < n; u++) { clique[u] = u; members[u] = {u}; }
    bestF.assign(n, Best{-1, -1});
    bestP.assign(n, Best{-1, -1});
    dirty.assign(n, 1);
    full_recompute();
    for (;;) {
      if (brute) {
        full_recompute();
        std::fill(dirty.begin(), dirty.end(), 1);
      }
      int64_t vf = -1, vp = -1;
      int fu = -1, fv = -1, pu = -1, pv = -1;
      for (int u = 0; u < n; u++) {
        if (dirty[u]) rescan(u);
        if (bestF[u].col >= 0) {
          int a = std::min(u, bestF[u].col), b = std::max(u, bestF[u].col);
          if (bestF[u].val > vf || (bestF[u].val == vf && (a < fu || (a == fu && b < fv)))) {
            vf = bestF[u].val; fu = a; fv = b;
          }
        }
        if (bestP[u].col >= 0) {
          int a = std::min(u, bestP[u].col), b = std::max(u, bestP[u].col);
          if (bestP[u].val > vp || (bestP[u].val == vp && (a < pu || (a == pu && b < pv)))) {
            vp = bestP[u].val; pu = a; pv = b;
          }
        }
      }
      if (fu < 0 || pu < 0) break;  // no live edge left
      iterations++;
#ifdef ORACLE_CE_TRACE
      {  // experiment support (tools/ce_pair_stats.py): the decision and the runner-up among the icp row maxima
        static FILE* tf = getenv("ORACLE_CE_TRACE") ? fopen(getenv("ORACLE_CE_TRACE"), "w") : nullptr;
        if (tf) {
          int64_t v2 = -1; int a2 = -1, b2 = -1;
          for (int u = 0; u < n; u++) {
            if (bestP[u].col < 0) continue;
            int a = std::min(u, bestP[u].col), b = std::max(u, bestP[u].col);
            if (a == pu && b == pv) continue;
            if (bestP[u].val > v2 || (bestP[u].val == v2 && (a < a2 || (a == a2 && b < b2)))) { v2 = bestP[u].val; a2 = a; b2 = b; }
          }
          const bool mg = vf >= vp;
          const int u = mg ? fu : pu, v = mg ? fv : pv;
          fprintf(tf, "%d %d %d %d %d %lld %lld %d %d %lld\n", mg ? 1 : 0, u, v, (int)members[clique[u]].size(), (int)members[clique[v]].size(),
                  (long long)vf, (long long)vp, a2, b2, (long long)v2);
        }
      }
#endif
      if (vf >= vp) merge(fu, fv);
      else forbid(pu, pv);
#ifdef ORACLE_CE_TRACE
      if (getenv("ORACLE_CE_TRACE_F") && vf >= vp) fprintf(stderr, "MERGE_F %d %d\n", trace_fa, trace_fb);
#endif
    }
    // clusters ordered by smallest member, members ascending
    std::vector<int> cl(n, -1);
    int k = 0;
    for (int u = 0; u < n; u++) {
      if (cl[u] != -1) continue;
      for (int x : members[clique[u]]) cl[x] = k;
      k++;
    }
    return cl;
  }
};

// ---------------------------------------------------------------------------
// S6 helpers, src/alignmentstoreadset.cpp:550-779
// ---------------------------------------------------------------------------
// sort(), :707-727: clusters (ascending id) sorted by coverage DESC with std::sort
std::vector<int> sort_desc(const std::map<int, double>& M) {
  std::vector<std::pair<double, double>> A;
  for (auto& it : M) A.emplace_back((double)it.first, it.second);
  std::sort(A.begin(), A.end(),
            [](const std::pair<double, double>& a, const std::pair<double, double>& b) { return a.second > b.second; });
  std::vector<int> r;
  for (auto& it : A) r.push_back((int)it.first);
  return r;
}
// sort_asc(), :729-749
std::vector<int> sort_asc(const std::map<int, double>& M) {
  std::vector<std::pair<double, double>> A;
  for (auto& it : M) A.emplace_back((double)it.first, it.second);
  std::sort(A.begin(), A.end(),
            [](const std::pair<double, double>& a, const std::pair<double, double>& b) { return a.second < b.second; });
  std::vector<int> r;
  for (auto& it : A) r.push_back((int)it.first);
  return r;
}

// ---------------------------------------------------------------------------
// S7: HaploThreader(ploidy, switchCost, affineSwitchCost, false, 0)
//       .computePaths(0, P, covMap, coverage, consensus, genotypes)   (:320,408)
// ---------------------------------------------------------------------------
struct Threader {
  int ploidy;
  double switch_cost, affine;
  // tuples are ordered ploidy-tuples of LOCAL cluster indices (into covMap[j]);
  // code = sum_i local_i * k^(ploidy-1-i); enumerated in ascending code.
  static void decode(int code, int k, int p, int* out) {
    for (int i = p - 1; i >= 0; i--) { out[i] = code % k; code /= k; }
  }
  std::vector<int> conform_tuples(int k, const std::vector<uint32_t>& consensus,
                                  const std::map<int, int>& genotype) const {
    int T = 1;
    for (int i = 0; i < ploidy; i++) T *= k;
    std::vector<int> all, conf;
    std::vector<int> t(ploidy);
    for (int code = 0; code < T; code++) {
      all.push_back(code);
      decode(code, k, ploidy, t.data());
      std::map<int, int> g;
      for (int i = 0; i < ploidy; i++) g[(int)consensus[t[i]]]++;
      std::map<int, int> want;
      for (auto& kv : genotype)
        if (kv.second != 0) want[kv.first] = kv.second;
      if (g == want) conf.push_back(code);
    }
    return conf.empty() ? all : conf;
  }
  double coverage_cost(const int* t, const std::vector<double>& coverage) const {
    double cost = 0.0;
    for (int i = 0; i < ploidy; i++) {
      int cn = 0;
      for (int j = 0; j < ploidy; j++)
        if (t[j] == t[i]) cn++;
      double cov = coverage[t[i]];
      double lo = ((double)cn - 0.5) / (double)ploidy;
      double hi = ((double)cn + 0.5) / (double)ploidy;
      if (cov < lo || cov > hi) cost += 1.0;
    }
    return cost;
  }
  std::vector<std::vector<int>> compute_paths(const std::vector<std::vector<int>>& covMap,
                                              const std::vector<std::vector<double>>& coverage,
                                              const std::vector<std::vector<uint32_t>>& consensus,
                                              const std::vector<std::map<int, int>>& genotypes) const {
    const int P = (int)covMap.size();
    std::vector<std::vector<int>> tuples(P);
    std::vector<std::vector<double>> score(P);
    std::vector<std::vector<int>> pred(P);
    std::vector<int> t(ploidy), tp(ploidy);
    for (int j = 0; j < P; j++) {
      int k = (int)covMap[j].size();
      tuples[j] = conform_tuples(k, consensus[j], genotypes[j]);
      score[j].resize(tuples[j].size());
      pred[j].assign(tuples[j].size(), -1);
      for (size_t r = 0; r < tuples[j].size(); r++) {
        decode(tuples[j][r], k, ploidy, t.data());
        double cc = coverage_cost(t.data(), coverage[j]);
        if (j == 0) { score[j][r] = cc; continue; }
        int kp = (int)covMap[j - 1].size();
        double best = INFINITY;
        int bi = -1;
        for (size_t q = 0; q < tuples[j - 1].size(); q++) {
          decode(tuples[j - 1][q], kp, ploidy, tp.data());
          int sw = 0;
          for (int i = 0; i < ploidy; i++)
            if (covMap[j - 1][tp[i]] != covMap[j][t[i]]) sw++;
          double s = score[j - 1][q] + (switch_cost * (double)sw + (sw > 0 ? affine : 0.0));
          if (s < best) { best = s; bi = (int)q; }
        }
        score[j][r] = cc + best;
        pred[j][r] = bi;
      }
    }
    int cur = 0;
    for (size_t r = 1; r < score[P - 1].size(); r++)
      if (score[P - 1][r] < score[P - 1][cur]) cur = (int)r;
    std::vector<std::vector<int>> path(P, std::vector<int>(ploidy));
    for (int j = P - 1; j >= 0; j--) {
      decode(tuples[j][cur], (int)covMap[j].size(), ploidy, t.data());
      for (int i = 0; i < ploidy; i++) path[j][i] = covMap[j][t[i]];
      cur = pred[j][cur];
    }
    return path;
  }
};

// ---------------------------------------------------------------------------
// per-chain result
// ---------------------------------------------------------------------------
struct ChainOut {
  int status = AHSOKA_ST_PHASED;
  std::vector<int> pos;
  std::vector<int> hap_allele, hap_cluster;  // [ploidy][P]
  int n_reads = 0, n_clusters = 0;
  int64_t iterations = 0;
  std::vector<int8_t> vfull, vpart;
  std::vector<int> fread_name, fread_cluster;
  std::vector<int32_t> score_q;
};

// the loop body of alignmentsToReadset, :75-490
void phase_chain(const ChainView& cv, const ahsoka_phase_params& prm, bool stages, bool brute_ce,
                 ChainOut& out) {
  const ahsoka_chain_batch* bt = cv.b;
  const int B = cv.nbubbles(), R = cv.nalign();
  if (stages) {
    out.vfull.assign((size_t)R * B, -1);
    out.vpart.assign((size_t)R * B, -1);
  }
  if (B <= 1) { out.status = AHSOKA_ST_SKIPPED_LE1; return; }  // :86

  // alignment.getRawIds() sorted once (the reference re-sorts per call, :496-497)
  std::vector<std::vector<int>> aln(R);
  for (int r = 0; r < R; r++) {
    int64_t g = cv.r0 + r;
    aln[r].assign(bt->rnode + bt->rnode_off[g], bt->rnode + bt->rnode_off[g + 1]);
    std::sort(aln[r].begin(), aln[r].end());
  }
  if (stages) {
    // stage dump: the raw is_subset outcome for every (alignment, bubble), full
    // and partial, independent of which positions S2 goes on to visit
    for (int k = 0; k < B; k++)
      for (int a = 0; a < cv.nalleles(k); a++) {
        std::vector<int> path = cv.apath(k, a);
        for (int r = 0; r < R; r++) {
          if (out.vfull[(size_t)r * B + k] < 0 && is_subset(path, aln[r], false)) out.vfull[(size_t)r * B + k] = (int8_t)a;
          if (out.vpart[(size_t)r * B + k] < 0 && is_subset(path, aln[r], true)) out.vpart[(size_t)r * B + k] = (int8_t)a;
        }
      }
  }
  auto name_of = [&](int r) { return bt->read_name[cv.r0 + r]; };
  auto mapq_of = [&](int r) { return bt->read_mapq[cv.r0 + r]; };
  // `(alignment.id*100) > 90` of :245 is a FLOAT compare on the identity, not on
  // the truncated mapq: the batch carries its outcome (bit 0 of read_flags)
  auto id_gt90 = [&](int r) { return bt->read_flags ? (bt->read_flags[cv.r0 + r] & 1) != 0 : mapq_of(r) > 90; };

  // ---- S1 (:90-135) ------------------------------------------------------
  // The reference iterates a copy of the chain's unordered_map (:76,:90, quirk
  // Q2).  The order decides which of several same-name alignments creates the
  // Read (and so the Read's mapq): the batch carries it (s1_order, pinned on the
  // compiled reference by tests/test_ingest_golden.py); without it, the order a
  // libstdc++ map filled with 0..B-1 gives (src/chainstoreadset.cpp:171-198).
  std::vector<int> s1_order;
  if (bt->s1_order) s1_order.assign(bt->s1_order + cv.b0, bt->s1_order + cv.b1);
  else {
    std::unordered_map<int, std::vector<std::vector<int>>> bubblemap;
    for (int k = 0; k < B; k++) bubblemap[k];
    auto chainmap = std::make_pair(cv.c, bubblemap);
    for (auto& kv : chainmap.second) s1_order.push_back(kv.first);
  }
  OReadSet readset;
  for (int k : s1_order)
    for (int a = 0; a < cv.nalleles(k); a++) {
      std::vector<int> path = cv.apath(k, a);
      for (int r = 0; r < R; r++) {
        if (!is_subset(path, aln[r], false)) continue;
        int idx = readset.find(name_of(r));
        if (idx < 0) {
          ORead rd{name_of(r), mapq_of(r), {}};
          rd.add_variant(k, a);
          readset.add(rd);
        } else if (!readset.reads[idx].has_pos(k)) {
          readset.reads[idx].add_variant(k, a);
        }
      }
    }
  // ---- S1b (:146-163) ----------------------------------------------------
  OReadSet testset;
  for (auto& rd : readset.reads)
    if ((int)rd.var.size() >= prm.min_variants && rd.mapq >= prm.min_mapq) testset.add(rd);
  // ---- S1c (:174-208) ----------------------------------------------------
  std::unordered_set<int> firsts, lasts;
  for (auto& rd : testset.reads) { firsts.insert(rd.first_pos()); lasts.insert(rd.last_pos()); }
  std::set<int> to_be_added;
  for (int el : lasts)
    if (!firsts.count(el)) { to_be_added.insert(el); to_be_added.insert(el + 1); }
  std::vector<int> fullpos = readset.positions();
  if (fullpos.empty()) { out.status = AHSOKA_ST_NO_FULL_MATCH; return; }  // UB in the reference (Q4)
  const int maxpos = fullpos.back();
  for (int i = 0; i < maxpos; i++) to_be_added.insert(i);  // gaps are a subset of [0,maxpos)

  // ---- S2 (:210-254) -----------------------------------------------------
  OReadSet partial;
  for (int boundary : to_be_added) {
    if (boundary < 0 || boundary >= B) continue;  // pathToAlleles[chain][boundary] auto-inserted empty (:216)
    for (int a = 0; a < cv.nalleles(boundary); a++) {
      std::vector<int> path = cv.apath(boundary, a);
      for (int r = 0; r < R; r++) {
        if (!is_subset(path, aln[r], true)) continue;
        int idx = partial.find(name_of(r));
        if (idx < 0) {
          ORead rd{name_of(r), mapq_of(r), {}};
          rd.add_variant(boundary, a);
          partial.add(rd);
        } else if (!partial.reads[idx].has_pos(boundary) && id_gt90(r)) {  // :245
          partial.reads[idx].add_variant(boundary, a);
        }
      }
    }
  }
  // ---- S2b (:262-282) ----------------------------------------------------
  std::vector<ORead> fin;
  for (auto& rd : partial.reads)
    if ((int)rd.var.size() >= prm.min_variants && rd.mapq >= prm.min_mapq) fin.push_back(rd);
  if (fin.empty()) { out.status = AHSOKA_ST_EMPTY_READSET; return; }
  // ---- S3 (:296-297) ReadSet::sort(): std::sort by firstPosition only ------
  std::sort(fin.begin(), fin.end(), [](const ORead& a, const ORead& b) { return a.first_pos() < b.first_pos(); });
  const int n = (int)fin.size();
  out.n_reads = n;

  // ---- S4 (:308-311) -----------------------------------------------------
  std::vector<int32_t> W;
  score_readset_local(fin, prm.min_overlap, prm.ploidy, W);
  if (stages) out.score_q = W;
  // ---- S5 (:312-314) -----------------------------------------------------
  ClusterEditing ce;
  ce.brute = brute_ce;
  std::vector<int> cl = ce.run(W, n);
  out.iterations = ce.iterations;
  int K = 0;
  for (int x : cl) K = std::max(K, x + 1);
  out.n_clusters = K;
  if (stages) {
    for (auto& rd : fin) out.fread_name.push_back(rd.name);
    out.fread_cluster = cl;
  }
  std::vector<std::vector<int>> cluster(K);
  for (int i = 0; i < n; i++) cluster[cl[i]].push_back(i);

  // ---- S6 (:316-404) -----------------------------------------------------
  std::vector<int> pos;
  {
    std::set<int> s;
    for (auto& rd : fin)
      for (auto& v : rd.var) s.insert(v.first);
    pos.assign(s.begin(), s.end());
  }
  const int P = (int)pos.size();
  const int lastpos = pos.back();
  std::vector<int> index_of(lastpos + 1, -1);
  for (int j = 0; j < P; j++) index_of[pos[j]] = j;
  // get_coverage, :660-697
  std::vector<std::map<int, double>> cov(lastpos + 1);
  {
    std::vector<double> sum(lastpos + 1, 0.0);
    for (int c = 0; c < K; c++)
      for (int rid : cluster[c])
        for (auto& v : fin[rid].var) { cov[v.first][c] += 1; sum[v.first] += 1; }
    for (int q = 0; q <= lastpos; q++)
      for (auto& e : cov[q]) e.second = e.second / sum[q];
  }
  // get_pos_to_clusters_map, :751-779
  std::vector<std::vector<int>> covMap;
  const uint32_t ploidy = (uint32_t)prm.ploidy;
  for (int q = 0; q <= lastpos; q++) {
    if (index_of[q] < 0) continue;
    std::vector<int> sorted_cids = sort_desc(cov[q]);
    size_t cut_off = std::min((uint32_t)sorted_cids.size(), 2 * ploidy);
    for (uint32_t i = ploidy; i < std::min((uint32_t)sorted_cids.size(), 2 * ploidy); i++)
      if (cov[q][sorted_cids[i]] < (1.0 / (8.0 * ploidy))) { cut_off = i; break; }
    covMap.emplace_back(sorted_cids.begin(), sorted_cids.begin() + cut_off);
  }
  // get_local_cluster_consensus / get_single_cluster_consensus_frac, :550-655
  std::vector<std::map<int, int>> consensus(P);  // compact pos -> cluster -> allele
  {
    std::map<int, std::vector<int>> relevant;
    for (int j = 0; j < P; j++)
      for (int c : covMap[j]) relevant[c].push_back(j);
    for (int c = 0; c < K; c++) {
      std::vector<std::map<int, double>> cnt(P);
      for (int rid : cluster[c])
        for (auto& v : fin[rid].var) cnt[index_of[v.first]][v.second] += 1;
      for (int j : relevant[c]) {
        int max_allele = 0;
        if (!cnt[j].empty()) {
          double max_count = 0;
          for (int allele : sort_asc(cnt[j])) {
            double cur = cnt[j][allele];
            if (cur > max_count) { max_allele = allele; max_count = cur; }
          }
        }
        consensus[j][c] = max_allele;
      }
    }
  }
  // compaction to vectors, :379-402 (quirk Q12)
  std::vector<std::vector<double>> coverage_vec(P);
  std::vector<std::vector<uint32_t>> consensus_vec(P);
  for (int j = 0; j < P; j++) {
    if (prm.compaction_mode == 0) {
      for (auto& e : cov[pos[j]]) coverage_vec[j].push_back(e.second);
      for (auto& e : consensus[j]) consensus_vec[j].push_back((uint32_t)e.second);
    } else {
      for (int c : covMap[j]) {
        coverage_vec[j].push_back(cov[pos[j]][c]);
        consensus_vec[j].push_back((uint32_t)consensus[j][c]);
      }
    }
  }
  // genotypes, :340-344 (reference literal) or supplied dosage
  std::vector<std::map<int, int>> genotypes(P);
  for (int j = 0; j < P; j++) {
    if (bt->genotype) {
      int64_t a0 = bt->allele_off[cv.b0 + pos[j]], a1 = bt->allele_off[cv.b0 + pos[j] + 1];
      for (int64_t a = a0; a < a1; a++) genotypes[j][(int)(a - a0)] = bt->genotype[a];
    } else {
      genotypes[j] = {{0, 1}, {1, 1}};
    }
  }
  // ---- S7 (:408) -----------------------------------------------------------
  Threader th{prm.ploidy, prm.switch_cost, prm.affine_switch_cost};
  std::vector<std::vector<int>> path = th.compute_paths(covMap, coverage_vec, consensus_vec, genotypes);
  // ---- S8 (:415-424): cons = new_consensus[j][c_id] ------------------------
  out.pos = pos;
  out.hap_allele.assign((size_t)prm.ploidy * P, 0);
  out.hap_cluster.assign((size_t)prm.ploidy * P, 0);
  for (int i = 0; i < prm.ploidy; i++)
    for (int j = 0; j < P; j++) {
      int c = path[j][i];
      auto it = consensus[j].find(c);
      out.hap_cluster[(size_t)i * P + j] = c;
      out.hap_allele[(size_t)i * P + j] = it == consensus[j].end() ? 0 : it->second;  // map operator[] default
    }
}

template <class T>
T* dup(const std::vector<T>& v) {
  T* p = (T*)std::malloc(std::max<size_t>(1, v.size()) * sizeof(T));
  if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}

}  // namespace

extern "C" {

// brute_ce != 0: cluster editing recomputes all induced costs from their
// definition every iteration (slow; cross-checks the incremental updates).
int oracle_phase_batch(const ahsoka_chain_batch* in, const ahsoka_phase_params* prm,
                       ahsoka_phase_result* res, int n_threads, int brute_ce) {
  if (!in || !prm || !res) return AHSOKA_E_ARG;
  auto t0 = std::chrono::steady_clock::now();
  const int C = in->n_chains;
  std::vector<ChainOut> outs(C);
  const bool stages = prm->want_stages != 0;
  auto work = [&](int tid, int nt) {
    for (int c = tid; c < C; c += nt) {
      ChainView cv{in, c, in->bubble_off[c], in->bubble_off[c + 1], in->read_off[c], in->read_off[c + 1]};
      phase_chain(cv, *prm, stages, brute_ce != 0, outs[c]);
    }
  };
  if (n_threads <= 1) work(0, 1);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) th.emplace_back(work, t, n_threads);
    for (auto& t : th) t.join();
  }
  std::memset(res, 0, sizeof(*res));
  res->n_chains = C;
  res->ploidy = prm->ploidy;
  std::vector<int32_t> status(C), nreads(C), nclus(C);
  std::vector<int64_t> iters(C), pos_off(C + 1, 0), vmat_off(C + 1, 0), fread_off(C + 1, 0), score_off(C + 1, 0);
  std::vector<int32_t> pos, ha, hc, fname, fclus, score;
  std::vector<int8_t> vfull, vpart;
  for (int c = 0; c < C; c++) {
    ChainOut& o = outs[c];
    status[c] = o.status; nreads[c] = o.n_reads; nclus[c] = o.n_clusters; iters[c] = o.iterations;
    pos.insert(pos.end(), o.pos.begin(), o.pos.end());
    ha.insert(ha.end(), o.hap_allele.begin(), o.hap_allele.end());
    hc.insert(hc.end(), o.hap_cluster.begin(), o.hap_cluster.end());
    pos_off[c + 1] = (int64_t)pos.size();
    if (stages) {
      vfull.insert(vfull.end(), o.vfull.begin(), o.vfull.end());
      vpart.insert(vpart.end(), o.vpart.begin(), o.vpart.end());
      fname.insert(fname.end(), o.fread_name.begin(), o.fread_name.end());
      fclus.insert(fclus.end(), o.fread_cluster.begin(), o.fread_cluster.end());
      score.insert(score.end(), o.score_q.begin(), o.score_q.end());
    }
    vmat_off[c + 1] = (int64_t)vfull.size();
    fread_off[c + 1] = (int64_t)fname.size();
    score_off[c + 1] = (int64_t)score.size();
  }
  res->status = dup(status); res->n_reads = dup(nreads); res->n_clusters = dup(nclus);
  res->n_iterations = dup(iters); res->pos_off = dup(pos_off); res->pos = dup(pos);
  res->hap_allele = dup(ha); res->hap_cluster = dup(hc);
  if (stages) {
    res->vmat_off = dup(vmat_off); res->vfull = dup(vfull); res->vpart = dup(vpart);
    res->fread_off = dup(fread_off); res->fread_name = dup(fname); res->fread_cluster = dup(fclus);
    res->score_off = dup(score_off); res->score_q = dup(score);
  }
  res->elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  res->kernel_ms = 0;
  return AHSOKA_OK;
}

void oracle_result_free(ahsoka_phase_result* r) {
  if (!r) return;
  std::free(r->status); std::free(r->n_reads); std::free(r->n_clusters); std::free(r->n_iterations);
  std::free(r->pos_off); std::free(r->pos); std::free(r->hap_allele); std::free(r->hap_cluster);
  std::free(r->vmat_off); std::free(r->vfull); std::free(r->vpart);
  std::free(r->fread_off); std::free(r->fread_name); std::free(r->fread_cluster);
  std::free(r->score_off); std::free(r->score_q);
  std::memset(r, 0, sizeof(*r));
}

// single-function hooks for the pinning tests
int oracle_is_subset(const int32_t* path, int32_t npath, const int32_t* aln, int32_t naln, int32_t take_partial) {
  std::vector<int> p(path, path + npath), a(aln, aln + naln);
  std::sort(a.begin(), a.end());
  return is_subset(p, a, take_partial != 0) ? 1 : 0;
}
double oracle_det_log(double x) { return det_log(x); }
// std::sort by the high 16 bits only (ReadSet::sort's comparator shape): the
// real libstdc++ introsort, used to pin the product's re-statement of it.
void oracle_std_sort_hi16(uint32_t* v, int32_t n) {
  std::sort(v, v + n, [](uint32_t a, uint32_t b) { return (a >> 16) < (b >> 16); });
}

}  // extern "C"
