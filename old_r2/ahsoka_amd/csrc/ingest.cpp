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
#include <cerrno>
#include <chrono>
#include <sstream>
#include <string_view>
#include <thread>
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

  // The file is read once and tokenised in place (no per-line stream objects, the
  // segment sequences are never copied); lines are applied in file order, which is
  // what fixes the iteration order of `nodes` and with it every chain / bubble id.
  int read_gfa(const char* path) {
    std::string buf;
    {
      std::ifstream f(path, std::ios::binary);
      if (!f) return AHSOKA_E_IO;
      f.seekg(0, std::ios::end);
      const std::streamoff sz = f.tellg();
      f.seekg(0, std::ios::beg);
      buf.resize((size_t)std::max<std::streamoff>(sz, 0));
      if (sz > 0) f.read(&buf[0], sz);
    }
    auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; };
    const char* p = buf.data();
    const char* end = p + buf.size();
    while (p < end) {
      const char* nl = (const char*)std::memchr(p, '\n', (size_t)(end - p));
      if (!nl) break;  // a last line without '\n' is dropped, as in the reference (:193)
      const char* lb = p;
      const char* le = nl;
      p = nl + 1;
      if (le == lb || (lb[0] != 'S' && lb[0] != 'L')) continue;
      const char* tb[7];
      const char* te[7];
      int nt = 0;
      const char* q = lb;
      while (nt < 7) {
        while (q < le && ws(*q)) q++;
        if (q >= le) break;
        tb[nt] = q;
        while (q < le && !ws(*q)) q++;
        te[nt] = q;
        nt++;
      }
      auto digits = [&](int t, bool* ok) {
        long long v = 0;
        int nd = 0;
        for (const char* c = tb[t]; c < te[t]; c++)
          if (*c >= '0' && *c <= '9') { v = v * 10 + (*c - '0'); nd++; }
        if (nd == 0 || nd > 9) *ok = false;
        return (int)v;
      };
      if (lb[0] == 'S') {
        bool ok = true;
        if (nt < 3 || te[0] - tb[0] != 1) return AHSOKA_E_PARSE;  // tag must be "S", a sequence must follow
        const int id = digits(1, &ok);
        if (!ok) return AHSOKA_E_PARSE;
        GNode nd;
        nd.id = id;
        nodes[id] = nd;
      } else {
        bool ok = true;
        if (nt < 6 || te[0] - tb[0] != 1) return AHSOKA_E_PARSE;
        const int ida = digits(1, &ok), idb = digits(3, &ok);
        const bool a1 = te[2] - tb[2] == 1 && (tb[2][0] == '+' || tb[2][0] == '-');
        const bool b1 = te[4] - tb[4] == 1 && (tb[4][0] == '+' || tb[4][0] == '-');
        if (!ok || !a1 || !b1) return AHSOKA_E_PARSE;
        // overlap "<int><unit>": the unit may also be the next token
        const char* c = tb[5];
        long long ov = 0;
        int nd = 0;
        bool neg = false;
        if (c < te[5] && (*c == '+' || *c == '-')) { neg = *c == '-'; c++; }
        while (c < te[5] && *c >= '0' && *c <= '9') { ov = ov * 10 + (*c - '0'); nd++; c++; }
        char unit = 0;
        if (c < te[5]) unit = *c;
        else if (nt >= 7) unit = tb[6][0];
        if (nd == 0 || neg || nd > 9) return AHSOKA_E_PARSE;
        if (!(unit == 'M' || (unit == 'S' && ov == 0))) return AHSOKA_E_PARSE;
        Oriented from{ida, tb[2][0] == '+'}, to{idb, tb[4][0] == '+'};
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
  void walk(const GNode* start, bool dir, GChain* chain) {
    const int start_id = start->id;
    std::unordered_set<Oriented, OrientedHash> seen;
    std::unordered_set<int> done;
    std::vector<int> inside;
    seen.insert(Oriented{start_id, dir});
    // frontier ordered by (node id, direction): the reference keeps a std::set of (Node,bool);
    // node records are referenced, not copied -- their link lists never change after parsing
    std::map<std::pair<int, bool>, const GNode*> frontier;
    frontier.emplace(std::make_pair(start_id, dir), start);
    while (!frontier.empty()) {
      auto first = frontier.begin();
      const GNode* v = first->second;
      const bool vdir = first->first.second;
      frontier.erase(first);
      done.insert(v->id);
      nodes[v->id].visited = true;
      inside.push_back(v->id);
      seen.erase(Oriented{v->id, vdir});
      const std::vector<Link>& next = vdir ? v->out_minus : v->out_plus;
      if (next.empty()) break;  // tip
      for (const Link& u : next) {
        const bool udir = (u.plus == 0);
        const GNode& un = nodes[u.to];
        const std::vector<Link>& parents = udir ? un.out_plus : un.out_minus;
        if (u.to == start_id) {  // walked back into the start: a loop, give up
          frontier.clear();
          break;
        }
        seen.insert(Oriented{u.to, udir});
        bool ready = true;
        for (const Link& p : parents)
          if (!done.count(p.to)) ready = false;
        if (ready) frontier.emplace(std::make_pair(un.id, udir), &un);
      }
      if (frontier.size() == 1 && seen.size() == 1) {
        auto tit = frontier.begin();
        const GNode* t = tit->second;
        const bool tdir = tit->first.second;
        frontier.erase(tit);
        inside.push_back(t->id);
        if (inside.size() == 2) break;
        auto drop = [&](int nid) {
          for (size_t i = 0; i < inside.size(); i++)
            if (inside[i] == nid) { inside.erase(inside.begin() + (long)i); break; }
        };
        drop(start_id);
        drop(t->id);
        GBubble b;
        b.source = start_id;
        b.sink = t->id;
        b.inner = inside;
        chain->bubbles.push_back(b);
        walk(t, tdir, chain);
      }
    }
  }

  void find_chains() {
    for (auto& entry : nodes) {  // the visited flag is read at this moment, as the reference's copy does
      if (!entry.second.visited) {
        GChain ch;
        walk(&entry.second, false, &ch);
        walk(&entry.second, true, &ch);
        if (!ch.bubbles.empty()) chains.push_back(std::move(ch));
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

  // Allele paths of a bubble with != 2 inner nodes: depth-first from the sink, children in
  // link order (src/chainstoreadset.cpp:44-116).  The listing is exponential in the number of
  // stacked sub-bubbles (2^k paths), and the reference walks all of it.  Here the number of
  // paths is first COUNTED with a memo per (node, direction) -- linear in the bubble's links --
  // and the walk itself stops after kMaxAllelePaths paths: a bubble with more alleles than the
  // device handles (16, phase_kernels.h) makes its chain AHSOKA_ST_UNSUPPORTED anyway, so the
  // hot path never sees a truncated list as if it were complete.  Up to the bound the listing
  // is the reference's, path for path, in order (tests/golden/sbub.ref.txt).
  static constexpr size_t kMaxAllelePaths = 1024;

  size_t count_paths(const GBubble& b, int nid, int dirflag, std::unordered_map<long long, size_t>* memo,
                     std::unordered_set<int>* onwalk) {
    const long long key = ((long long)nid << 1) | (dirflag ? 1 : 0);
    auto it = memo->find(key);
    if (it != memo->end()) return it->second;
    if (onwalk->count(nid)) return kMaxAllelePaths + 1;  // a cycle: the walk's revisit rule decides, no closed form
    const GNode& nd = nodes[nid];
    const std::vector<Link>& next = (dirflag == 0) ? nd.out_minus : nd.out_plus;
    bool within = true;
    for (const Link& ch : next)
      if (!b.has(ch.to)) within = false;
    size_t cnt = 1;
    if (!next.empty() && within) {
      cnt = 0;
      onwalk->insert(nid);
      for (const Link& ch : next) {
        cnt += count_paths(b, ch.to, ch.plus, memo, onwalk);
        if (cnt > kMaxAllelePaths) { cnt = kMaxAllelePaths + 1; break; }
      }
      onwalk->erase(nid);
    }
    (*memo)[key] = cnt;
    return cnt;
  }

  void dfs_paths(const GBubble& b, int nid, int dirflag, std::vector<int>* seq, std::vector<std::vector<int>>* out) {
    if (out->size() >= kMaxAllelePaths) return;
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
    {
      std::unordered_map<long long, size_t> memo;
      std::unordered_set<int> onwalk;
      const size_t cnt = count_paths(b, b.sink, dirflag, &memo, &onwalk);
      paths.reserve(std::min(cnt, kMaxAllelePaths));
    }
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

// All GAF lines, parsed.  The file is read into memory once and tokenised in
// place by several threads (chunks cut at line ends, results concatenated in
// file order); names are views into the buffer, node ids go to one flat pool.
// Semantics follow AlignmentReader::readAlignmentfile (src/alignmentreader.cpp:
// 69-186): a line is what getline returns before EOF is hit (an unterminated
// last line is dropped, :80), empty lines are skipped, fields are the first 16
// whitespace-separated tokens, token 16 is id:<..>:<float>.
struct GafTable {
  std::string buf;
  std::vector<uint64_t> name_off;
  std::vector<uint32_t> name_len;
  std::vector<float> identity;
  std::vector<uint64_t> node_off{0};
  std::vector<int> nodes;
  size_t size() const { return identity.size(); }
  std::string_view name(size_t i) const { return std::string_view(buf.data() + name_off[i], name_len[i]); }
};

inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

// std::stoi accepts optional white space, a sign and a digit prefix, and throws
// when nothing converts or the value does not fit an int
inline bool parses_as_int(const char* b, const char* e) {
  if (b < e && (*b == '+' || *b == '-')) b++;
  long long v = 0;
  int nd = 0;
  while (b < e && *b >= '0' && *b <= '9' && nd < 12) { v = v * 10 + (*b - '0'); nd++; b++; }
  return nd > 0 && nd < 12 && v <= 2147483648ll;
}

struct GafChunk {
  std::vector<uint64_t> name_off;
  std::vector<uint32_t> name_len;
  std::vector<float> identity;
  std::vector<uint32_t> node_cnt;
  std::vector<int> nodes;
  int err = AHSOKA_OK;
};

void parse_gaf_chunk(const std::string& buf, size_t beg, size_t end, GafChunk* out) {
  const char* base = buf.data();
  const size_t guess = (end - beg) / 150 + 16;  // a GAF line is rarely shorter
  out->name_off.reserve(guess); out->name_len.reserve(guess); out->identity.reserve(guess);
  out->node_cnt.reserve(guess); out->nodes.reserve(guess * 16);
  size_t pos = beg;
  while (pos < end) {
    const char* nl = (const char*)std::memchr(base + pos, '\n', end - pos);
    if (!nl) break;  // unterminated tail: dropped, like the reference
    const char* lb = base + pos;
    const char* le = nl;
    pos = (size_t)(nl - base) + 1;
    if (le == lb) continue;
    const char* tb[16];
    const char* te[16];
    int nt = 0;
    const char* q = lb;
    while (nt < 16) {
      while (q < le && is_ws(*q)) q++;
      if (q >= le) break;
      tb[nt] = q;
      while (q < le && !is_ws(*q)) q++;
      te[nt] = q;
      nt++;
    }
    if (nt < 16) { out->err = AHSOKA_E_PARSE; return; }
    // token 16: first ':'-field must be "id", the value follows the last ':'
    const char* c1 = (const char*)std::memchr(tb[15], ':', (size_t)(te[15] - tb[15]));
    if (!c1 || c1 - tb[15] != 2 || tb[15][0] != 'i' || tb[15][1] != 'd') { out->err = AHSOKA_E_PARSE; return; }
    const char* cl = te[15];
    while (cl > tb[15] && cl[-1] != ':') cl--;
    {
      // std::stof == strtof on the token (needs a terminator: copy to the stack)
      char tmp[64];
      const size_t len = std::min<size_t>((size_t)(te[15] - cl), sizeof(tmp) - 1);
      std::memcpy(tmp, cl, len);
      tmp[len] = 0;
      char* endp = nullptr;
      errno = 0;
      const float v = std::strtof(tmp, &endp);
      if (endp == tmp || errno == ERANGE) { out->err = AHSOKA_E_PARSE; return; }
      out->identity.push_back(v);
    }
    if (!parses_as_int(tb[7], te[7]) || !parses_as_int(tb[8], te[8])) { out->err = AHSOKA_E_PARSE; return; }
    out->name_off.push_back((uint64_t)(tb[0] - base));
    out->name_len.push_back((uint32_t)(te[0] - tb[0]));
    // path: maximal runs without '<' / '>'; a node id is the digits of the run
    uint32_t cnt = 0;
    const char* r = tb[5];
    while (r < te[5]) {
      while (r < te[5] && (*r == '<' || *r == '>')) r++;
      const char* rb = r;
      while (r < te[5] && *r != '<' && *r != '>') r++;
      if (r > rb) {
        long long id = 0;
        int nd = 0;
        for (const char* d = rb; d < r; d++)
          if (*d >= '0' && *d <= '9') { id = id * 10 + (*d - '0'); nd++; }
        if (nd == 0 || nd > 9) { out->err = AHSOKA_E_PARSE; return; }
        out->nodes.push_back((int)id);
        cnt++;
      }
    }
    out->node_cnt.push_back(cnt);
  }
}

int read_gaf(const char* path, GafTable* out) {
  {
    std::ifstream f(path, std::ios::binary);
    if (!f) return AHSOKA_E_IO;
    f.seekg(0, std::ios::end);
    const std::streamoff sz = f.tellg();
    f.seekg(0, std::ios::beg);
    out->buf.resize((size_t)std::max<std::streamoff>(sz, 0));
    if (sz > 0) f.read(&out->buf[0], sz);
  }
  const bool timing = std::getenv("AHSOKA_INGEST_TIMING") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "  read_gaf %-10s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  };
  lap("file");
  const std::string& buf = out->buf;
  unsigned nthr = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (buf.size() < (1u << 20)) nthr = 1;
  std::vector<size_t> cut(nthr + 1, buf.size());
  cut[0] = 0;
  for (unsigned t = 1; t < nthr; t++) {
    size_t p = buf.size() / nthr * t;
    const char* nl = (const char*)std::memchr(buf.data() + p, '\n', buf.size() - p);
    cut[t] = nl ? (size_t)(nl - buf.data()) + 1 : buf.size();
  }
  for (unsigned t = 1; t <= nthr; t++) cut[t] = std::max(cut[t], cut[t - 1]);
  std::vector<GafChunk> chunks(nthr);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nthr; t++) th.emplace_back(parse_gaf_chunk, std::cref(buf), cut[t], cut[t + 1], &chunks[t]);
  for (auto& x : th) x.join();
  lap("parse");
  for (auto& c : chunks)
    if (c.err) return c.err;  // the reference aborts at the first malformed line
  size_t nl = 0, nn = 0;
  for (auto& c : chunks) { nl += c.identity.size(); nn += c.nodes.size(); }
  out->name_off.reserve(nl); out->name_len.reserve(nl); out->identity.reserve(nl);
  out->node_off.reserve(nl + 1); out->nodes.reserve(nn);
  for (auto& c : chunks) {
    out->name_off.insert(out->name_off.end(), c.name_off.begin(), c.name_off.end());
    out->name_len.insert(out->name_len.end(), c.name_len.begin(), c.name_len.end());
    out->identity.insert(out->identity.end(), c.identity.begin(), c.identity.end());
    for (uint32_t k : c.node_cnt) out->node_off.push_back(out->node_off.back() + k);
    out->nodes.insert(out->nodes.end(), c.nodes.begin(), c.nodes.end());
  }
  lap("merge");
  return AHSOKA_OK;
}

}  // namespace

// Text dump of everything the ingest derived, in the line format of
// oracle/ref_harness.cpp (CHAIN / BUBBLE / ALLELE / ALN / S1ORDER / ORDER), so the two can
// be compared byte for byte; also the source of <o>-bubbleinfo.txt.
extern "C" int ahsoka_ingest_dump(const char* gfa_path, const char* gaf_path, const char* out_path) {
  if (!gfa_path || !out_path) return AHSOKA_E_ARG;
  try {
    Graph g;
    if (int e = g.read_gfa(gfa_path)) return e;
    g.find_chains();
    GafTable gaf;
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
      for (uint64_t e = gaf.node_off[li]; e < gaf.node_off[li + 1]; e++) cnt[chain_of(gaf.nodes[e])]++;
      for (auto& kv : cnt) per_chain[kv.first].emplace_back((int)li, kv.second);
    }
    for (auto& kv : per_chain)
      for (auto& lc : kv.second) {
        const size_t li = (size_t)lc.first;
        const float scaled = gaf.identity[li] * 100;
        const std::string nm(gaf.name(li));
        std::fprintf(f, "ALN %d %s %d %d %d %zu", kv.first, nm.c_str(), (int)scaled, (int)ahk::ref_gt90_flag(gaf.identity[li]),
                     lc.second, (size_t)(gaf.node_off[li + 1] - gaf.node_off[li]));
        for (uint64_t e = gaf.node_off[li]; e < gaf.node_off[li + 1]; e++) std::fprintf(f, " %d", gaf.nodes[e]);
        std::fprintf(f, "\n");
      }
    // the order in which S1 walks each chain's bubbles (src/alignmentstoreadset.cpp:76,90)
    for (auto& oc : order) {
      std::vector<int32_t> s1;
      ahk::ref_s1_order((int)g.chains[(size_t)oc.second].bubbles.size(), &s1);
      std::fprintf(f, "S1ORDER %d", oc.second);
      for (int32_t k : s1) std::fprintf(f, " %d", k);
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
  const bool timing = std::getenv("AHSOKA_INGEST_TIMING") != nullptr;
  auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "ingest %-14s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  };
  try {
    Graph g;
    if (int e = g.read_gfa(gfa_path)) return e;
    lap("read_gfa");
    g.find_chains();
    lap("find_chains");
    GafTable gaf;
    if (gaf_path && *gaf_path)
      if (int e = read_gaf(gaf_path, &gaf)) return e;
    lap("read_gaf");

    // size_sorting, src/polyassembly.cpp:136-140: (bubbles, chain id) descending
    std::vector<std::pair<int, int>> order;
    for (const GChain& ch : g.chains) order.emplace_back((int)ch.bubbles.size(), ch.id);
    std::sort(order.begin(), order.end(), std::greater<>());

    // chain of every node id as a dense table: nodes of no chain (and ids the graph
    // does not know) read as chain 0, the reference's default chain_id
    int max_id = 0;
    for (auto& kv : g.nodes) max_id = std::max(max_id, kv.first);
    for (int nid : gaf.nodes) max_id = std::max(max_id, nid);
    std::vector<int> chain_tab((size_t)max_id + 1, 0);
    for (auto& kv : g.nodes)
      if (kv.first >= 0) chain_tab[(size_t)kv.first] = kv.second.chain;
    const int nchains = (int)g.chains.size();
    // lines of every chain, in file order (a line is filed once per chain it touches)
    std::vector<std::vector<uint32_t>> lines_of_chain((size_t)nchains);
    {
      std::vector<int> touched;
      for (size_t li = 0; li < gaf.size(); li++) {
        touched.clear();
        for (uint64_t e = gaf.node_off[li]; e < gaf.node_off[li + 1]; e++) {
          const int c = chain_tab[(size_t)gaf.nodes[e]];
          if (c >= 0 && c < nchains && std::find(touched.begin(), touched.end(), c) == touched.end()) touched.push_back(c);
        }
        for (int c : touched) lines_of_chain[(size_t)c].push_back((uint32_t)li);
      }
    }

    ahk::HostBatch* hb = new ahk::HostBatch();
    hb->has_sign = true;
    hb->bubble_off.push_back(0);
    hb->allele_off.push_back(0);
    hb->apath_off.push_back(0);
    hb->node_off.push_back(0);
    hb->read_off.push_back(0);
    hb->rnode_off.push_back(0);
    // chain-local index of a node id: dense table + "which chain wrote it" stamp
    std::vector<int> loc_idx((size_t)max_id + 1, -1), loc_stamp((size_t)max_id + 1, -1);
    int stamp = 0;
    for (auto& oc : order) {
      const GChain& ch = g.chains[(size_t)oc.second];
      hb->chain_id.push_back(ch.id);
      int nlocal = 0;
      auto local_of = [&](int nid) {
        if (loc_stamp[(size_t)nid] == stamp) return loc_idx[(size_t)nid];
        loc_stamp[(size_t)nid] = stamp;
        loc_idx[(size_t)nid] = nlocal;
        hb->node_id.push_back(nid);
        return nlocal++;
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
      // S1 visits the bubbles in the iteration order of the chain's unordered_map
      // (src/alignmentstoreadset.cpp:76,90); bubble ids are 0..B-1 in ch.bubbles order
      ahk::ref_s1_order((int)ch.bubbles.size(), &hb->s1_order);
      hb->node_off.push_back((int64_t)hb->node_id.size());
      std::unordered_map<std::string_view, int> name_id;
      name_id.reserve(lines_of_chain[(size_t)ch.id].size() * 2);
      for (uint32_t li : lines_of_chain[(size_t)ch.id]) {
        for (uint64_t e = gaf.node_off[li]; e < gaf.node_off[li + 1]; e++) {
          const int nid = gaf.nodes[e];
          hb->rnode.push_back(loc_stamp[(size_t)nid] == stamp ? loc_idx[(size_t)nid] : -1);
        }
        hb->rnode_off.push_back((int64_t)hb->rnode.size());
        const float scaled = gaf.identity[li] * 100;  // float multiply, then the implicit float->int of the Read ctor
        hb->read_mapq.push_back((int32_t)scaled);
        hb->read_flags.push_back(ahk::ref_gt90_flag(gaf.identity[li]));  // :245, on the float
        auto ins = name_id.emplace(gaf.name(li), (int)name_id.size());
        hb->read_name.push_back(ins.first->second);
      }
      hb->read_off.push_back((int64_t)hb->read_mapq.size());
      stamp++;
    }
    hb->finish();
    lap("assemble");
    *out = &hb->view;
  } catch (const std::bad_alloc&) {
    return AHSOKA_E_NOMEM;
  } catch (...) {
    return AHSOKA_E_PARSE;
  }
  return AHSOKA_OK;
}
