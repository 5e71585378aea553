// phase_api.hip -- host side of the C-ABI (include/ahsoka_phase.h): validates
// a ChainBatch, lays it out in HBM, launches the two kernels on one HIP
// stream and copies the per-chain haplotype assignment back.
//
// Replaces one call of alignmentsToReadset (src/alignmentstoreadset.cpp:55).
// There is NO CPU implementation of the path in this library: without a HIP
// device every compute entry point returns AHSOKA_E_NODEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "detmath.h"
#include "gnusort.h"
#include "host_batch.h"
#include "phase_kernels.h"
#include "../../include/ahsoka_phase.h"

using namespace ahk;

namespace {

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      std::fprintf(stderr, "ahsoka: HIP error %s at %s:%d (%s)\n", hipGetErrorString(_e), __FILE__, __LINE__, #expr); \
      return AHSOKA_E_HIP;                                                             \
    }                                                                                  \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

// Host loops over the batch (validation, per-chain metadata, the name CSR) walk every
// alignment node once: ~2*10^8 entries at config 3.  fn(lo, hi) runs on contiguous index
// ranges of [0, n) on up to 16 host threads; ranges are independent by construction.
template <class F>
void parallel_ranges(long long n, long long grain, F fn) {
  unsigned T = std::thread::hardware_concurrency();
  if (T == 0) T = 1;
  if (T > 16) T = 16;
  const long long want = grain > 0 ? (n + grain - 1) / grain : 1;
  if ((long long)T > want) T = (unsigned)(want < 1 ? 1 : want);
  if (T <= 1) {
    fn(0LL, n);
    return;
  }
  std::vector<std::thread> th;
  th.reserve(T);
  for (unsigned t = 0; t < T; t++) {
    const long long lo = n * t / T, hi = n * (t + 1) / T;
    th.emplace_back([=] { fn(lo, hi); });
  }
  for (auto& x : th) x.join();
}

}  // namespace

// One size class of k_phase: chains whose final readset has at most nmax reads share
// a pool of `slots` workgroups (one slot of HBM working state each) and one launch of
// the persistent grid with LDS sized for nmax.  A batch of similar chains (configs
// 1-4) is one class; the power-law config 5 gets up to four, so that its few large
// chains do not dictate the footprint (and LDS occupancy) of the many small ones.
struct PhaseClass {
  int nmax = 0, slots = 0, count = 0, Pmax = 1;
  int per_cu = 0;  // workgroups of k_phase per CU the class was sized for
  PhaseLds pl{};
  SlotWorkspace ws{};
  int* queue = nullptr;  // device: this class's chains, most expensive first
  int* qhead = nullptr;  // device: three ints -- cursor of the queue, cursor of the deferred list, length of the deferred list
  int* deferred = nullptr;  // device [count]: chains the narrow (32-bit induced cost) instance of k_phase leaves to the wide one
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
};

struct ahsoka_device_batch {
  ahsoka_phase_params prm;
  int C = 0;
  int device = 0;
  std::vector<ChainMeta> meta;
  std::vector<void*> allocs;
  std::vector<void*> ws_allocs;  // slot workspaces (released as a whole if their allocation fails)
  DeviceBatch db;
  ReadsetLds rl;
  // k_readset_lds runs once per LDS class of its chains (by the chain's own LDS need: three, two or one
  // workgroup per CU), so that a few large chains do not dictate the occupancy of the many small ones
  struct RsClass {
    size_t lds_bytes = 0;
    int maxG = 1, first = 0, count = 0;
  };
  RsClass rs_class[3];
  int* rs_list = nullptr;  // device: chains of k_readset_lds (class by class) first, then those of k_readset
  int n_fit = 0, n_big = 0, n_fit_real = 0;
  std::vector<PhaseClass> classes;
  int* qheads = nullptr;
  int maxG = 1, maxB = 1, n_cu = 1;
  size_t lds_per_block = 0;
  bool ws_ready = false;
  int stop_after = 0;
  int* packed = nullptr;  // [status | npos | pos | hap_allele] for ahsoka_batch_result_device
  int slots = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  float k_ms[2] = {0, 0};
  const char* k_names[2] = {"k_readset", "k_phase"};
  long long sumB = 0, sumR = 0, sumG = 0, sumVm = 0, sumCov = 0, sumDump = 0;
  size_t device_bytes = 0;
  double last_ms = 0;

  template <class T>
  int dalloc(T** out, size_t count) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (hipMalloc(&p, bytes) != hipSuccess) return AHSOKA_E_NOMEM;
    allocs.push_back(p);
    device_bytes += bytes;
    *out = (T*)p;
    return AHSOKA_OK;
  }
  template <class T>
  int dupload(const T** out, const T* src, size_t count) {
    T* p = nullptr;
    if (int e = dalloc(&p, count)) return e;
    if (count && hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return AHSOKA_E_HIP;
    *out = p;
    return AHSOKA_OK;
  }
};

extern "C" void ahsoka_default_params(ahsoka_phase_params* prm) {
  if (!prm) return;
  prm->ploidy = 2;
  prm->min_mapq = 93;
  prm->min_variants = 2;
  prm->min_overlap = 1;
  prm->switch_cost = 32.0;
  prm->affine_switch_cost = 8.0;
  prm->compaction_mode = 0;
  prm->device = 0;
  prm->want_stages = 0;
  prm->max_slots = 0;
  prm->symmetry_opt = 0;  // HaploThreader(ploidy, 32.0, 8.0, false, 0), :320
  prm->row_limit = 0;
  prm->wave_mode = 0;
}

extern "C" int ahsoka_abi_version(void) { return AHSOKA_ABI_VERSION; }
extern "C" const char* ahsoka_kernel_revision(void) { return AHK_KERNEL_REV; }

extern "C" int ahsoka_backend_info(char* buf, size_t buflen) {
  if (!buf || buflen == 0) return AHSOKA_E_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    std::snprintf(buf, buflen, "hip:no-device");
    return AHSOKA_E_NODEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) {
    std::snprintf(buf, buflen, "hip:no-device");
    return AHSOKA_E_NODEVICE;
  }
  std::snprintf(buf, buflen, "hip:%s:%d:%d", prop.gcnArchName, prop.multiProcessorCount, ndev);
  return AHSOKA_OK;
}

int ahk::validate_batch(const ahsoka_chain_batch* in) {
  if (!in || in->n_chains < 0) return AHSOKA_E_ARG;
  if (in->n_chains == 0) return AHSOKA_OK;
  if (!in->bubble_off || !in->allele_off || !in->apath_off || !in->node_off || !in->read_off || !in->rnode_off)
    return AHSOKA_E_ARG;
  const int C = in->n_chains;
  if (in->bubble_off[0] != 0 || in->read_off[0] != 0 || in->node_off[0] != 0 || in->allele_off[0] != 0 ||
      in->apath_off[0] != 0 || in->rnode_off[0] != 0)
    return AHSOKA_E_ARG;
  for (int c = 0; c < C; c++)
    if (in->bubble_off[c + 1] < in->bubble_off[c] || in->read_off[c + 1] < in->read_off[c] ||
        in->node_off[c + 1] < in->node_off[c])
      return AHSOKA_E_ARG;
  const long long SB = in->bubble_off[C], SR = in->read_off[C];
  std::atomic<int> bad{0};
  parallel_ranges(SB, 1 << 16, [&](long long lo, long long hi) {
    for (long long g = lo; g < hi; g++)
      if (in->allele_off[g + 1] < in->allele_off[g]) { bad = 1; return; }
  });
  if (bad) return AHSOKA_E_ARG;
  const long long SA = in->allele_off[SB];
  parallel_ranges(SA, 1 << 16, [&](long long lo, long long hi) {
    for (long long a = lo; a < hi; a++)
      if (in->apath_off[a + 1] < in->apath_off[a]) { bad = 1; return; }
  });
  parallel_ranges(SR, 1 << 16, [&](long long lo, long long hi) {
    for (long long r = lo; r < hi; r++)
      if (in->rnode_off[r + 1] < in->rnode_off[r]) { bad = 1; return; }
  });
  if (bad) return AHSOKA_E_ARG;
  if (in->apath_off[SA] > 0 && !in->apath_node) return AHSOKA_E_ARG;
  if (SR > 0 && (!in->read_mapq || !in->read_name)) return AHSOKA_E_ARG;
  if (in->rnode_off[SR] > 0 && !in->rnode) return AHSOKA_E_ARG;
  // node indices must be chain-local and in range: the kernels index LDS with them
  parallel_ranges(C, 8, [&](long long c0, long long c1) {
    for (long long c = c0; c < c1; c++) {
      const long long N = in->node_off[c + 1] - in->node_off[c];
      for (long long g = in->bubble_off[c]; g < in->bubble_off[c + 1]; g++)
        for (long long a = in->allele_off[g]; a < in->allele_off[g + 1]; a++)
          for (long long e = in->apath_off[a]; e < in->apath_off[a + 1]; e++)
            if (in->apath_node[e] < 0 || in->apath_node[e] >= N) { bad = 1; return; }
      for (long long r = in->read_off[c]; r < in->read_off[c + 1]; r++) {
        for (long long e = in->rnode_off[r]; e < in->rnode_off[r + 1]; e++)
          if (in->rnode[e] < -1 || in->rnode[e] >= N) { bad = 1; return; }
        if (in->read_name[r] < 0 || in->read_name[r] >= in->read_off[c + 1] - in->read_off[c]) { bad = 1; return; }
      }
      if (in->s1_order) {  // must be a permutation of the chain's bubble ids
        const long long b0 = in->bubble_off[c], B = in->bubble_off[c + 1] - b0;
        std::vector<unsigned char> seen((size_t)B, 0);
        for (long long k = 0; k < B; k++) {
          const int v = in->s1_order[b0 + k];
          if (v < 0 || v >= B || seen[(size_t)v]) { bad = 1; return; }
          seen[(size_t)v] = 1;
        }
      }
    }
  });
  return bad ? AHSOKA_E_ARG : AHSOKA_OK;
}

namespace {
// joins a host thread on every way out of a scope, exceptions included (a joinable std::thread
// that is destroyed calls std::terminate)
struct JoinGuard {
  std::thread& t;
  ~JoinGuard() {
    if (t.joinable()) t.join();
  }
};
}  // namespace

// (*hp is set as soon as the handle exists, so that the extern "C" wrapper can release it when a
// host allocation throws)
static int upload_impl(const ahsoka_chain_batch* in, const ahsoka_phase_params* prm, ahsoka_device_batch** handle,
                       ahsoka_device_batch** hp) {
  if (!in || !prm || !handle) return AHSOKA_E_ARG;
  if (prm->ploidy < 2 || prm->ploidy > kMaxPloidy || prm->min_overlap < 0 || prm->max_slots < 0) return AHSOKA_E_ARG;
  if (prm->wave_mode != 0 && prm->wave_mode != 4 && prm->wave_mode != 8 && prm->wave_mode != 16) return AHSOKA_E_ARG;
  // (threading scores must be non-negative: the device argmin orders them by their bit patterns)
  if (!(prm->switch_cost >= 0.0) || !(prm->affine_switch_cost >= 0.0)) return AHSOKA_E_ARG;
  // the threader's symmetryOptimization / rowLimit exist at the boundary so that it mirrors :320
  // completely; only the reference's literals (false, 0) are implemented
  if (prm->symmetry_opt != 0 || prm->row_limit != 0) return AHSOKA_E_UNSUPPORTED;
  if (int e = ahk::validate_batch(in)) return e;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return AHSOKA_E_NODEVICE;
  if (prm->device < 0 || prm->device >= ndev) return AHSOKA_E_ARG;
  if (hipSetDevice(prm->device) != hipSuccess) return AHSOKA_E_NODEVICE;

  ahsoka_device_batch* h = new (std::nothrow) ahsoka_device_batch();
  if (!h) return AHSOKA_E_NOMEM;
  *hp = h;
  h->prm = *prm;
  h->device = prm->device;
  const int C = h->C = in->n_chains;
  std::memset(&h->db, 0, sizeof(h->db));
  int rc = AHSOKA_OK;
  struct RawCopy {
    void* dst;
    const void* src;
    size_t bytes;
  };
  // (declared before the copier and its guard: the thread reads them until it is joined)
  std::vector<RawCopy> raw;
  std::atomic<int> copy_rc{AHSOKA_OK};
  std::thread copier;
  JoinGuard copier_guard{copier};
  auto fail = [&](int code) {
    if (copier.joinable()) copier.join();
    ahsoka_batch_free(h);
    *hp = nullptr;
    return code;
  };
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, prm->device) != hipSuccess) return fail(AHSOKA_E_NODEVICE);
  h->n_cu = prop.multiProcessorCount;
  h->lds_per_block = (size_t)prop.sharedMemPerBlock;

  // ---- inputs to HBM: the caller's arrays go as they are, copied by a second host thread while
  // this one derives the per-chain metadata (0.8 GB at config 3: the copy and the host loops take
  // about as long as each other)
  DeviceBatch& db = h->db;
  const long long SB = in->bubble_off[C], SA = in->allele_off[SB], SP = in->apath_off[SA], SR = in->read_off[C],
                  SL = in->rnode_off[SR];
  auto stage = [&](auto** out, const auto* src, size_t count) -> int {
    typename std::remove_const<typename std::remove_pointer<decltype(src)>::type>::type* p = nullptr;
    if (int e = h->dalloc(&p, count)) return e;
    raw.push_back(RawCopy{p, src, count * sizeof(*p)});
    *out = p;
    return AHSOKA_OK;
  };
  if ((rc = stage(&db.allele_off, (const long long*)in->allele_off, (size_t)SB + 1))) return fail(rc);
  if ((rc = stage(&db.apath_off, (const long long*)in->apath_off, (size_t)SA + 1))) return fail(rc);
  if ((rc = stage(&db.apath_node, (const int*)in->apath_node, (size_t)SP))) return fail(rc);
  if (in->genotype) {
    if ((rc = stage(&db.genotype, (const int*)in->genotype, (size_t)SA))) return fail(rc);
  } else {
    db.genotype = nullptr;
  }
  if ((rc = stage(&db.rnode_off, (const long long*)in->rnode_off, (size_t)SR + 1))) return fail(rc);
  if ((rc = stage(&db.rnode, (const int*)in->rnode, (size_t)SL))) return fail(rc);
  if ((rc = stage(&db.read_mapq, (const int*)in->read_mapq, (size_t)SR))) return fail(rc);
  if (in->read_flags)
    if ((rc = stage(&db.read_flags, (const unsigned char*)in->read_flags, (size_t)SR))) return fail(rc);
  if (in->s1_order)
    if ((rc = stage(&db.s1_order, (const int*)in->s1_order, (size_t)SB))) return fail(rc);
  copier = std::thread([&raw, &copy_rc, dev = prm->device] {
    if (hipSetDevice(dev) != hipSuccess) {
      copy_rc = AHSOKA_E_NODEVICE;
      return;
    }
    for (const RawCopy& r : raw)
      if (r.bytes && hipMemcpy(r.dst, r.src, r.bytes, hipMemcpyHostToDevice) != hipSuccess) {
        copy_rc = AHSOKA_E_HIP;
        return;
      }
  });

  // ---- per-chain metadata, name CSR, limits --------------------------------
  // Three steps: (1) per chain, in parallel: sizes, limits, repeated-node flags, which readset
  // kernel fits; (2) serial prefix sums of the per-chain array offsets; (3) per chain, in
  // parallel: the name CSR (alignments of one read name side by side).
  h->meta.resize(C);
  std::vector<int> name_aln((size_t)in->read_off[C]);
  std::vector<unsigned char> apath_dup((size_t)in->allele_off[in->bubble_off[C]] + 1, 0);
  std::vector<ReadsetLds2> need2v((size_t)C);
  std::vector<int> maxAv((size_t)C, 2);
  int maxA = 2, maxG = 1, maxB = 1, maxN = 1;
  long long vm = 0, fr = 0, cov = 0, dump = 0;
  std::vector<int> fit_list[3], big_list;
  // LDS need up to which three / two workgroups of k_readset_lds share a CU (160 KB, some static LDS)
  const size_t rs_limit[2] = {52 * 1024, 78 * 1024};
  h->maxB = 1;
  const size_t lds_per_block = h->lds_per_block;
  parallel_ranges(C, 8, [&](long long c0, long long c1) {
    for (long long c = c0; c < c1; c++) {
      ChainMeta& m = h->meta[(size_t)c];
      std::memset(&m, 0, sizeof(m));
      m.b0 = in->bubble_off[c];
      m.r0 = in->read_off[c];
      m.n0 = in->node_off[c];
      m.B = (int)(in->bubble_off[c + 1] - m.b0);
      m.R = (int)(in->read_off[c + 1] - m.r0);
      m.N = (int)(in->node_off[c + 1] - m.n0);
      int G = 0;
      for (int r = 0; r < m.R; r++) G = std::max(G, in->read_name[m.r0 + r] + 1);
      m.G = G;
      m.Wb = (m.B + 63) / 64;
      bool unsupported = m.B > kMaxBubbles || m.N > kMaxNodes || G > kMaxNames || m.R > kMaxAlns;
      // a chain whose k_readset working set does not fit this device's LDS is reported
      // per chain, like the compiled-in limits, instead of failing the whole batch
      if (!unsupported && readset_lds_bytes(ReadsetLds{std::min((m.N + 63) / 64, kWinWords), std::max(G, 1), std::max(m.B, 1)}) > lds_per_block)
        unsupported = true;
      int mA = 2;
      for (long long g = m.b0; g < m.b0 + m.B; g++) {
        const int na = (int)(in->allele_off[g + 1] - in->allele_off[g]);
        if (na > kMaxAlleles) unsupported = true;
        mA = std::max(mA, std::min(na, kMaxAlleles));
        for (long long a = in->allele_off[g]; a < in->allele_off[g + 1]; a++) {
          // std::includes is a multiset test (quirk Q5); the bit test equals it only
          // for paths without repeated nodes: flag the others for the counting path
          const long long p0 = in->apath_off[a], p1 = in->apath_off[a + 1];
          for (long long e = p0; e < p1 && !apath_dup[(size_t)a]; e++)
            for (long long e2 = e + 1; e2 < p1; e2++)
              if (in->apath_node[e] == in->apath_node[e2]) { apath_dup[(size_t)a] = 1; break; }
        }
      }
      maxAv[(size_t)c] = mA;
      // chains whose alignment bitsets fit the LDS as a whole take k_readset_lds: no R x B match
      // matrices in HBM unless the stage dumps are wanted
      const long long SAc = in->allele_off[m.b0 + m.B] - in->allele_off[m.b0];
      const ReadsetLds2 need2{(int)std::min<long long>((long long)((m.N + 63) / 64) * m.R, 1 << 28),
                              (int)std::min<long long>((long long)((m.N + 63) / 64) * SAc, 1 << 28),
                              m.R, std::max(G, 1), std::max(m.B, 1), (int)std::min<long long>(SAc, 65535)};
      need2v[(size_t)c] = need2;
      const bool fit = unsupported || m.B <= 1 ||
                       (SAc < 65535 && readset_lds2_bytes(need2) <= std::min<size_t>(lds_per_block, kReadsetFitBytes));
      m.flags = (unsupported ? 1 : 0) | (fit ? 2 : 0);
    }
  });
  long long names_total = 0;
  for (int c = 0; c < C; c++) {
    ChainMeta& m = h->meta[c];
    const int G = m.G;
    const bool unsupported = (m.flags & 1) != 0, fit = (m.flags & 2) != 0;
    const ReadsetLds2& need2 = need2v[(size_t)c];
    maxA = std::max(maxA, maxAv[(size_t)c]);
    m.name0 = names_total;
    names_total += (long long)G + 1;
    m.vmat0 = (fit && !prm->want_stages) ? -1 : vm;
    m.fr0 = fr;
    m.cov0 = cov;
    m.pos0 = m.b0;
    m.dump0 = -1;
    if (m.vmat0 >= 0) vm += (long long)m.R * m.B;
    if (fit) {
      const bool real = !unsupported && m.B > 1;
      const size_t need = real ? readset_lds2_bytes(need2) : 0;
      const int k = need <= rs_limit[0] ? 0 : need <= rs_limit[1] ? 1 : 2;
      fit_list[k].push_back(c);
      if (real) {
        h->n_fit_real++;
        h->rs_class[k].lds_bytes = std::max(h->rs_class[k].lds_bytes, need);
        h->rs_class[k].maxG = std::max(h->rs_class[k].maxG, std::max(G, 1));
      }
    } else {
      big_list.push_back(c);
    }
    fr += G;
    cov += (long long)G * m.Wb;
    if (prm->want_stages) {
      m.dump0 = dump;
      dump += (long long)std::min(G, kMaxReads) * std::min(G, kMaxReads);
    }
    if (!unsupported && m.B > 1) {
      h->maxB = std::max(h->maxB, m.B);
      if (!fit) {
        maxG = std::max(maxG, G);
        maxB = std::max(maxB, m.B);
        maxN = std::max(maxN, m.N);
      }
    }
  }
  std::vector<long long> name_off((size_t)names_total);
  parallel_ranges(C, 8, [&](long long c0, long long c1) {
    std::vector<long long> cur;
    for (long long c = c0; c < c1; c++) {
      const ChainMeta& m = h->meta[(size_t)c];
      const int G = m.G;
      // counting sort of the chain's alignments by name id (stable: ascending alignment index)
      long long* cnt = name_off.data() + m.name0;  // [G+1]
      for (int g = 0; g <= G; g++) cnt[g] = 0;
      for (int r = 0; r < m.R; r++) cnt[in->read_name[m.r0 + r] + 1]++;
      for (int g = 0; g < G; g++) cnt[g + 1] += cnt[g];
      cur.assign(cnt, cnt + G);
      for (int r = 0; r < m.R; r++) name_aln[(size_t)m.r0 + cur[in->read_name[m.r0 + r]]++] = r;
    }
  });
  h->sumB = in->bubble_off[C];
  h->sumR = in->read_off[C];
  h->sumG = fr;
  h->sumVm = vm;
  h->sumCov = cov;
  h->sumDump = dump;
  int planes = 1;
  while ((1 << planes) < maxA) planes++;

  // ---- derived inputs ---------------------------------------------------------------
  if ((rc = h->dupload(&db.apath_dup, (const unsigned char*)apath_dup.data(), (size_t)SA + 1))) return fail(rc);
  if (!in->read_flags) {  // documented fallback: the guard of :245 on the truncated identity
    std::vector<unsigned char> fl((size_t)SR);
    parallel_ranges(SR, 1 << 18, [&](long long lo, long long hi) {
      for (long long r = lo; r < hi; r++) fl[(size_t)r] = in->read_mapq[r] > 90 ? 1 : 0;
    });
    if ((rc = h->dupload(&db.read_flags, (const unsigned char*)fl.data(), (size_t)SR))) return fail(rc);
  }
  if (!in->s1_order) {  // the order the reference's own map construction gives (host_batch.h)
    std::vector<int32_t> ord;
    ord.reserve((size_t)SB);
    std::vector<std::vector<int32_t>> memo;  // by B: chains of equal size share the order
    for (int c = 0; c < C; c++) {
      const size_t B = (size_t)h->meta[c].B;
      if (memo.size() <= B) memo.resize(B + 1);
      if (B > 0 && memo[B].empty()) ahk::ref_s1_order((int)B, &memo[B]);
      ord.insert(ord.end(), memo[B].begin(), memo[B].end());
    }
    if ((rc = h->dupload(&db.s1_order, (const int*)ord.data(), (size_t)SB))) return fail(rc);
  }
  if ((rc = h->dupload(&db.meta, h->meta.data(), (size_t)C))) return fail(rc);
  if ((rc = h->dupload(&db.name_off, name_off.data(), name_off.size()))) return fail(rc);
  if ((rc = h->dupload(&db.name_aln, name_aln.data(), name_aln.size()))) return fail(rc);

  // ---- per-chain outputs / k_readset workspace --------------------------------
  if ((rc = h->dalloc(&db.vfull, (size_t)vm))) return fail(rc);
  if ((rc = h->dalloc(&db.vpart, (size_t)vm))) return fail(rc);
  if ((rc = h->dalloc(&db.status, (size_t)C))) return fail(rc);
  if ((rc = h->dalloc(&db.nfin, (size_t)C))) return fail(rc);
  if ((rc = h->dalloc(&db.npos, (size_t)C))) return fail(rc);
  if ((rc = h->dalloc(&db.pos, (size_t)SB))) return fail(rc);
  if ((rc = h->dalloc(&db.fr_name, (size_t)fr))) return fail(rc);
  if ((rc = h->dalloc(&db.fr_cluster, (size_t)fr))) return fail(rc);
  if ((rc = h->dalloc(&db.cover, (size_t)cov))) return fail(rc);
  if ((rc = h->dalloc(&db.alle, (size_t)cov * planes))) return fail(rc);
  if (h->n_fit_real > 0) {  // creation-ordered staging of k_readset_lds (k_sortperm writes the final arrays)
    if ((rc = h->dalloc(&db.cover_tmp, (size_t)cov))) return fail(rc);
    if ((rc = h->dalloc(&db.alle_tmp, (size_t)cov * planes))) return fail(rc);
    if ((rc = h->dalloc(&db.sortkey, (size_t)fr))) return fail(rc);
    if ((rc = h->dalloc(&db.name_tmp, (size_t)fr))) return fail(rc);
  }
  db.alle_plane_stride = cov;
  db.planes = planes;
  if ((rc = h->dalloc(&db.hap_allele, (size_t)SB * prm->ploidy))) return fail(rc);
  if ((rc = h->dalloc(&db.hap_cluster, (size_t)SB * prm->ploidy))) return fail(rc);
  if ((rc = h->dalloc(&db.nclusters, (size_t)C))) return fail(rc);
  if ((rc = h->dalloc(&db.niter, (size_t)C))) return fail(rc);
  if (prm->want_stages) {
    if ((rc = h->dalloc(&db.score_dump, (size_t)dump))) return fail(rc);
  } else {
    db.score_dump = nullptr;
  }
  db.ploidy = prm->ploidy;
  db.min_mapq = prm->min_mapq;
  db.min_variants = prm->min_variants;
  db.min_overlap = prm->min_overlap;
  db.compaction_mode = prm->compaction_mode;
  db.switch_cost = prm->switch_cost;
  db.affine_cost = prm->affine_switch_cost;

  // ---- kernel geometry ----------------------------------------------------------
  h->rl.wordsN = std::min((maxN + 63) / 64, kWinWords);  // (k_readset's per-alignment node window)
  h->rl.maxG = maxG;
  h->rl.maxB = maxB;
  h->maxG = maxG;
  h->n_big = (int)big_list.size();
  {
    std::vector<int> both;
    for (int k = 0; k < 3; k++) {
      h->rs_class[k].first = (int)both.size();
      h->rs_class[k].count = (int)fit_list[k].size();
      if (h->rs_class[k].lds_bytes < 1024) h->rs_class[k].lds_bytes = 1024;  // (a class of skipped chains only)
      both.insert(both.end(), fit_list[k].begin(), fit_list[k].end());
    }
    h->n_fit = (int)both.size();
    both.insert(both.end(), big_list.begin(), big_list.end());
    const int* q = nullptr;
    if ((rc = h->dupload(&q, both.data(), both.size()))) return fail(rc);
    h->rs_list = const_cast<int*>(q);
  }
  // (the k_phase queues and slot workspaces are built after the first k_readset, when
  // the sizes of the final readsets are known: see build_classes)
#ifdef AHK_DIAG
  if (const char* sa = std::getenv("AHSOKA_STOP_AFTER")) h->stop_after = std::atoi(sa);  // diagnostic builds only
#endif
  if (readset_lds_bytes(h->rl) > h->lds_per_block) return fail(AHSOKA_E_UNSUPPORTED);  // (cannot happen: such chains were flagged)
#ifdef AHK_DIAG
  // diagnostic builds only: AHSOKA_CU_STRIDE=k runs the kernels on every k-th CU (a CU-masked stream): tells a
  // per-CU bound from a chip-wide one (same workgroups per CU, a fraction of the chip)
  if (const char* cs = std::getenv("AHSOKA_CU_STRIDE")) {
    const int stride = std::max(1, std::atoi(cs));
    std::vector<uint32_t> mask((size_t)(h->n_cu + 31) / 32, 0u);
    for (int cu = 0; cu < h->n_cu; cu += stride) mask[(size_t)cu / 32] |= 1u << (cu % 32);
    if (hipExtStreamCreateWithCUMask(&h->stream, (uint32_t)mask.size(), mask.data()) != hipSuccess) return fail(AHSOKA_E_HIP);
    h->n_cu = (h->n_cu + stride - 1) / stride;
  } else
#endif
  if (hipStreamCreate(&h->stream) != hipSuccess) return fail(AHSOKA_E_HIP);
  for (int i = 0; i < 3; i++)
    if (hipEventCreate(&h->ev[i]) != hipSuccess) return fail(AHSOKA_E_HIP);
  copier.join();
  if (copy_rc != AHSOKA_OK) return fail(copy_rc);
  *handle = h;
  return AHSOKA_OK;
}

// No exception crosses the C ABI: a failed host allocation (or a thread that cannot be started)
// becomes an error code, and whatever the call had built is released.
#define AHSOKA_GUARDED(call, cleanup)          \
  try {                                        \
    return call;                               \
  } catch (const std::bad_alloc&) {            \
    cleanup;                                   \
    return AHSOKA_E_NOMEM;                     \
  } catch (...) {                              \
    cleanup;                                   \
    return AHSOKA_E_HIP;                       \
  }

extern "C" int ahsoka_batch_upload(const ahsoka_chain_batch* in, const ahsoka_phase_params* prm,
                                   ahsoka_device_batch** handle) {
  ahsoka_device_batch* h = nullptr;
  AHSOKA_GUARDED(upload_impl(in, prm, handle, &h), if (h) ahsoka_batch_free(h));
}

// The per-slot state of k_phase (W, IC, bitmaps: ~24 B per read pair) is sized by
// the FINAL readsets, which are only known after k_readset has run (reads are
// filtered by identity and variant count: ~600 of 1000 survive at config 3).
// Chains are binned into size classes by their final read count n: the largest
// class holds chains up to top = max n (rounded up to 64), the next ones up to
// top/2, top/4, top/8.  Every non-empty class gets its own pool of slots, its own
// work queue (most expensive chain first: the reference orders chains by size too,
// src/polyassembly.cpp:136-140; here it balances the persistent grid) and its own
// launch of k_phase with LDS sized for the class -- all launches run concurrently.
static size_t class_slot_bytes(const PhaseClass& pc, long long Tmax) {
  const double nn = (double)pc.nmax * pc.nmax;
  const double words = (double)pc.nmax * ((pc.nmax + 63) / 64);
  return (size_t)(nn * (4 + 16 + 4) + words * 16 + (double)pc.nmax * (16 + 16) + (double)pc.Pmax * (4 + 8 * (4 + 8 + 4 + 4) + 4) +
                  (double)pc.Pmax * (double)Tmax * (4 + 8 + 2) + 4096);
}

static void free_classes(ahsoka_device_batch* h) {
  for (void* p : h->ws_allocs) (void)hipFree(p);
  h->ws_allocs.clear();
  for (PhaseClass& pc : h->classes) {
    if (pc.done) (void)hipEventDestroy(pc.done);
    if (pc.stream) (void)hipStreamDestroy(pc.stream);
  }
  h->classes.clear();
  h->qheads = nullptr;
  h->slots = 0;
}

static int build_classes(ahsoka_device_batch* h) {
  const int C = h->C;
  std::vector<int> nfin((size_t)C), status((size_t)C);
  if (hipMemcpy(nfin.data(), h->db.nfin, (size_t)C * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return AHSOKA_E_HIP;
  if (hipMemcpy(status.data(), h->db.status, (size_t)C * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return AHSOKA_E_HIP;
  // largest readset whose per-read LDS arrays fit a workgroup of this device
  int ncap = 64;
  for (int v = 64; v <= ((kMaxReads + 63) & ~63); v += 64) {
    PhaseLds t{v, 0, h->prm.ploidy, 0, 0, 4};
    if (phase_lds_bytes(t) <= h->lds_per_block) ncap = v;
  }
  bool demoted = false;
  int top = 0;
  for (int c = 0; c < C; c++) {
    if (status[c] != AHSOKA_ST_PHASED) continue;
    if (nfin[c] > ncap) {  // per chain, like the compiled-in limits
      status[c] = AHSOKA_ST_UNSUPPORTED;
      demoted = true;
      continue;
    }
    top = std::max(top, nfin[c]);
  }
  if (demoted && hipMemcpy(h->db.status, status.data(), (size_t)C * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return AHSOKA_E_HIP;
  if (top == 0) {  // nothing to phase: no class, no launch
    h->ws_ready = true;
    return AHSOKA_OK;
  }
  std::vector<int> caps;
  for (int i = 0; i < 4; i++) {
    const int v = std::max(64, ((((top + (1 << i) - 1) >> i)) + 63) & ~63);
    if (!caps.empty() && v >= caps.back()) break;
    caps.push_back(v);
  }
  auto class_of = [&](int n) {
    int k = 0;
    while (k + 1 < (int)caps.size() && n <= caps[k + 1]) k++;
    return k;
  };
  std::vector<std::vector<int>> members(caps.size());
  for (int c = 0; c < C; c++)
    if (status[c] == AHSOKA_ST_PHASED) members[(size_t)class_of(nfin[c])].push_back(c);
  long long Tmax = 1;
  for (int i = 0; i < h->prm.ploidy; i++) Tmax *= 2 * h->prm.ploidy;
  int per_cu_cap4 = 5;  // launch bounds of k_phase: five workgroups of four waves per CU
#ifdef AHK_DIAG
  if (const char* e = std::getenv("AHSOKA_WG_PER_CU")) per_cu_cap4 = std::max(1, std::atoi(e));  // diagnostic builds only
#endif
  int per_cu_cap = per_cu_cap4;
  for (size_t k = 0; k < caps.size(); k++) {
    if (members[k].empty()) continue;
    PhaseClass pc;
    pc.nmax = caps[k];
    pc.count = (int)members[k].size();
    pc.pl.nmax = pc.nmax;
    pc.pl.stop_after = h->stop_after;
    pc.pl.ploidy = h->prm.ploidy;
    // Wavefronts per workgroup.  A class whose chains all get half a CU to themselves is bound by the latency of
    // its largest chain, not by throughput: it takes the 8-wave form of the kernel, whose trip-count-bound passes
    // shorten with the lane count (measured, tools/wave_modes.sh: config 5's k_phase 62.9 -> 51.5 ms, 256 chains
    // of config 3 125.8 -> 103.7 ms, 512 chains 134.8 -> 117.2 ms; the 16-wave form is no faster than four waves:
    // its barriers and the sequential single-read separations outweigh the shorter passes).  Everything else
    // takes the four-wave form, five workgroups per CU.  prm.wave_mode forces a form (tests and experiments;
    // results do not depend on it).
    pc.pl.nwv = 4;
    if (h->prm.wave_mode != 0) pc.pl.nwv = h->prm.wave_mode;
    else if (pc.count <= 2 * h->n_cu) pc.pl.nwv = 8;
    {  // (a wide form whose per-wave scratch does not fit the LDS even without the extras below: the next narrower one)
      PhaseLds base = pc.pl;
      base.items_lds = 0;
      base.stage_row = 0;
      while (base.nwv > 4 && phase_lds_bytes(base) > h->lds_per_block) base.nwv /= 2;
      pc.pl.nwv = base.nwv;
    }
    per_cu_cap = pc.pl.nwv == 16 ? 1 : pc.pl.nwv == 8 ? 2 : per_cu_cap4;
    for (int c : members[k]) pc.Pmax = std::max(pc.Pmax, h->meta[(size_t)c].B);
    // LDS extras of S5 by occupancy: the LDS part of the candidate-edge list (8 B per read) is
    // dropped where that buys a workgroup per CU; the staged weight row (4 B per read) is added
    // where it does not cost one
    pc.pl.stage_row = 0;
    pc.pl.items_lds = 1;
    int per_cu = std::min(phase_max_blocks_per_cu(pc.pl), per_cu_cap);
    {
      PhaseLds lean = pc.pl;
      lean.items_lds = 0;
      const int per_cu_lean = std::min(phase_max_blocks_per_cu(lean), per_cu_cap);
      // (only where there are chains to fill the extra workgroups: a class that is in flight as a
      // whole is bound by its largest chain, which more co-resident workgroups only slow down)
      if ((per_cu_lean > per_cu && pc.count > h->n_cu * per_cu) || phase_lds_bytes(pc.pl) > h->lds_per_block) {
        pc.pl = lean;
        per_cu = per_cu_lean;
      }
    }
    {
      PhaseLds with = pc.pl;
      with.stage_row = 1;
      if (phase_lds_bytes(with) <= h->lds_per_block && std::min(phase_max_blocks_per_cu(with), per_cu_cap) >= per_cu) pc.pl = with;
    }
#ifdef AHK_DIAG
    std::fprintf(stderr, "ahsoka: class nmax=%d chains=%d waves=%d workgroups/CU=%d lds=%zu B items_lds=%d stage_row=%d\n", pc.nmax, pc.count,
                 pc.pl.nwv, per_cu, phase_lds_bytes(pc.pl), pc.pl.items_lds, pc.pl.stage_row);
#endif
    pc.per_cu = per_cu;
    pc.slots = std::max(1, std::min(pc.count, h->n_cu * per_cu));
    if (h->prm.max_slots > 0) pc.slots = std::min(pc.slots, h->prm.max_slots);
    h->classes.push_back(pc);
  }
  {  // fit the slot pools into 60 % of the free HBM: shrink the largest pool first
    size_t freeb = 0, totalb = 0;
    if (hipMemGetInfo(&freeb, &totalb) != hipSuccess) return AHSOKA_E_HIP;
    const double budget = (double)freeb * 0.6;
    for (;;) {
      double sum = 0;
      size_t big = 0;
      double bigb = -1;
      for (size_t k = 0; k < h->classes.size(); k++) {
        const double b = (double)class_slot_bytes(h->classes[k], Tmax) * h->classes[k].slots;
        sum += b;
        if (h->classes[k].slots > 1 && b > bigb) { bigb = b; big = k; }
      }
      if (sum <= budget) break;
      if (bigb < 0) { free_classes(h); return AHSOKA_E_NOMEM; }
      h->classes[big].slots = std::max(1, h->classes[big].slots * 3 / 4);
    }
  }
  auto wsalloc = [&](auto** out, size_t count) -> int {
    using T = std::remove_pointer_t<std::remove_pointer_t<decltype(out)>>;
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (hipMalloc(&p, bytes) != hipSuccess) return AHSOKA_E_NOMEM;
    h->ws_allocs.push_back(p);
    h->device_bytes += bytes;
    *out = (T*)p;
    return AHSOKA_OK;
  };
  int rc = AHSOKA_OK;
  if ((rc = wsalloc(&h->qheads, 3 * h->classes.size()))) { free_classes(h); return rc; }
  size_t ki = 0, mi = 0;
  for (PhaseClass& pc : h->classes) {
    while (members[mi].empty()) mi++;
    std::vector<int>& q = members[mi++];
    std::stable_sort(q.begin(), q.end(), [&](int a, int b) {
      const double ca = (double)nfin[a] * nfin[a] * h->meta[(size_t)a].B, cb = (double)nfin[b] * nfin[b] * h->meta[(size_t)b].B;
      return ca > cb;
    });
    pc.qhead = h->qheads + 3 * ki++;
    SlotWorkspace& ws = pc.ws;
    ws.nmax = pc.nmax;
    ws.Pmax = pc.Pmax;
    ws.Tmax = Tmax;
    const size_t S = (size_t)pc.slots, nn = (size_t)ws.nmax * ws.nmax, words = (size_t)ws.nmax * ((ws.nmax + 63) / 64);
    if ((rc = wsalloc(&pc.queue, q.size())) || (rc = wsalloc(&pc.deferred, q.size())) || (rc = wsalloc(&ws.W, S * nn)) || (rc = wsalloc(&ws.IC, S * nn)) ||
        (rc = wsalloc(&ws.live, S * words)) || (rc = wsalloc(&ws.nz, S * words)) || (rc = wsalloc(&ws.items, S * nn)) ||
        (rc = wsalloc(&ws.rates, S * 2 * ws.nmax)) || (rc = wsalloc(&ws.tp_k, S * ws.Pmax)) ||
        (rc = wsalloc(&ws.tp_gcid, S * ws.Pmax * 8)) || (rc = wsalloc(&ws.tp_cov, S * ws.Pmax * 8)) ||
        (rc = wsalloc(&ws.tp_cons, S * ws.Pmax * 8)) || (rc = wsalloc(&ws.tp_consmap, S * ws.Pmax * 8)) ||
        (rc = wsalloc(&ws.tp_ntup, S * ws.Pmax)) || (rc = wsalloc(&ws.tp_tuples, S * ws.Pmax * ws.Tmax)) ||
        (rc = wsalloc(&ws.tp_score, S * ws.Pmax * ws.Tmax)) || (rc = wsalloc(&ws.tp_pred, S * ws.Pmax * ws.Tmax)) ||
        (rc = wsalloc(&ws.tp_best, S * ws.Tmax)) ||
        (rc = wsalloc(&ws.sortbuf, S * 4 * ws.nmax))) {
      free_classes(h);
      return rc;
    }
    if (hipMemcpy(pc.queue, q.data(), q.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) { free_classes(h); return AHSOKA_E_HIP; }
    h->slots += pc.slots;
  }
  if (h->classes.size() > 1)
    for (PhaseClass& pc : h->classes)
      if (hipStreamCreateWithFlags(&pc.stream, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&pc.done, hipEventDisableTiming) != hipSuccess) {
        free_classes(h);
        return AHSOKA_E_HIP;
      }
  h->ws_ready = true;
  return AHSOKA_OK;
}

static int phase_device_impl(ahsoka_device_batch* h) {
  if (!h) return AHSOKA_E_ARG;
  HIP_TRY(hipSetDevice(h->device));
  if (h->C == 0) return AHSOKA_OK;
  auto t0 = std::chrono::steady_clock::now();
  HIP_TRY(hipEventRecord(h->ev[0], h->stream));
  for (const ahsoka_device_batch::RsClass& rc : h->rs_class)
    launch_readset_lds(h->db, h->rs_list + rc.first, rc.count, rc.lds_bytes, rc.maxG, h->stream);
  launch_readset(h->db, h->rs_list + h->n_fit, h->n_big, h->rl, h->stream);
  HIP_TRY(hipGetLastError());
  if (!h->ws_ready) {  // first call only: size classes and slot pools from the real readsets
    HIP_TRY(hipStreamSynchronize(h->stream));
    // (shards that share a device ordinal -- a device list may repeat one -- size their slot pools
    // one after the other: each takes its 60 % of what the previous one left)
    static std::mutex ws_mutex[64];
    std::lock_guard<std::mutex> lk(ws_mutex[h->device & 63]);
    if (int rc = build_classes(h)) return rc;
  }
  if (!h->classes.empty()) HIP_TRY(hipMemsetAsync(h->qheads, 0, 3 * h->classes.size() * sizeof(int), h->stream));
  HIP_TRY(hipEventRecord(h->ev[1], h->stream));
  if (h->classes.size() == 1) {
    PhaseClass& pc = h->classes[0];
    launch_phase(h->db, pc.ws, pc.pl, pc.queue, pc.count, pc.qhead, pc.deferred, pc.slots, h->stream);
    HIP_TRY(hipGetLastError());
  } else {
    // one launch per size class, each on its own stream: they share the CUs, the
    // class of the largest chains first in line
    for (PhaseClass& pc : h->classes) {
      HIP_TRY(hipStreamWaitEvent(pc.stream, h->ev[1], 0));
      launch_phase(h->db, pc.ws, pc.pl, pc.queue, pc.count, pc.qhead, pc.deferred, pc.slots, pc.stream);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(pc.done, pc.stream));
      HIP_TRY(hipStreamWaitEvent(h->stream, pc.done, 0));
    }
  }
  HIP_TRY(hipEventRecord(h->ev[2], h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipEventElapsedTime(&h->k_ms[0], h->ev[0], h->ev[1]));
  HIP_TRY(hipEventElapsedTime(&h->k_ms[1], h->ev[1], h->ev[2]));
  h->last_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return AHSOKA_OK;
}

extern "C" int ahsoka_phase_device(ahsoka_device_batch* h) {
  // (a slot workspace half built when a host allocation throws is released; the next call rebuilds it)
  AHSOKA_GUARDED(phase_device_impl(h), if (h && !h->ws_ready) free_classes(h));
}

extern "C" int ahsoka_batch_result_device(ahsoka_device_batch* h, void** dev_ptr, int64_t* n_int32) {
  if (!h || !dev_ptr || !n_int32) return AHSOKA_E_ARG;
  HIP_TRY(hipSetDevice(h->device));
  const size_t C = (size_t)h->C, SB = (size_t)h->sumB, p = (size_t)h->prm.ploidy;
  const size_t total = 2 * C + SB + p * SB;
  if (!h->packed)
    if (int rc = h->dalloc(&h->packed, total)) return rc;
  HIP_TRY(hipMemcpyAsync(h->packed, h->db.status, C * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->packed + C, h->db.npos, C * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->packed + 2 * C, h->db.pos, SB * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->packed + 2 * C + SB, h->db.hap_allele, p * SB * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  *dev_ptr = h->packed;
  *n_int32 = (int64_t)total;
  return AHSOKA_OK;
}

// (see the header: how the size classes of the last ahsoka_phase_device call were laid out)
extern "C" int ahsoka_batch_class_info(ahsoka_device_batch* h, int32_t* n_classes, int32_t* out, int32_t cap) {
  if (!h || !n_classes) return AHSOKA_E_ARG;
  *n_classes = (int32_t)h->classes.size();
  for (size_t k = 0; k < h->classes.size() && out && (int64_t)(k + 1) * 8 <= cap; k++) {
    const PhaseClass& pc = h->classes[k];
    int32_t* o = out + 8 * k;
    o[0] = pc.nmax;
    o[1] = pc.count;
    o[2] = pc.slots;
    o[3] = pc.pl.nwv;
    o[4] = pc.per_cu;
    o[5] = (int32_t)phase_lds_bytes(pc.pl);
    o[6] = pc.pl.items_lds;
    o[7] = pc.pl.stage_row;
  }
  return AHSOKA_OK;
}

extern "C" int ahsoka_batch_kernel_times(ahsoka_device_batch* h, int32_t* n, const char*** names, const float** ms) {
  if (!h || !n || !names || !ms) return AHSOKA_E_ARG;
  *n = 2;
  *names = h->k_names;
  *ms = h->k_ms;
  return AHSOKA_OK;
}

namespace {
template <class T>
T* hostdup(const std::vector<T>& v) {
  T* p = (T*)std::malloc(std::max<size_t>(1, v.size()) * sizeof(T));
  if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}
template <class T>
int fetch(std::vector<T>& dst, const T* src, size_t count) {
  dst.resize(count);
  if (count && hipMemcpy(dst.data(), src, count * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) return AHSOKA_E_HIP;
  return AHSOKA_OK;
}
}  // namespace

static int download_impl(ahsoka_device_batch* h, ahsoka_phase_result* out) {
  if (!h || !out) return AHSOKA_E_ARG;
  HIP_TRY(hipSetDevice(h->device));
  std::memset(out, 0, sizeof(*out));
  const int C = h->C, p = h->prm.ploidy;
  std::vector<int> status, nfin, npos, pos, ha, hc, ncl, frn, frc;
  std::vector<int64_t> niter;
  int rc;
  if ((rc = fetch(status, (const int*)h->db.status, (size_t)C))) return rc;
  if ((rc = fetch(nfin, (const int*)h->db.nfin, (size_t)C))) return rc;
  if ((rc = fetch(npos, (const int*)h->db.npos, (size_t)C))) return rc;
  if ((rc = fetch(pos, (const int*)h->db.pos, (size_t)h->sumB))) return rc;
  if ((rc = fetch(ha, (const int*)h->db.hap_allele, (size_t)h->sumB * p))) return rc;
  if ((rc = fetch(hc, (const int*)h->db.hap_cluster, (size_t)h->sumB * p))) return rc;
  if ((rc = fetch(ncl, (const int*)h->db.nclusters, (size_t)C))) return rc;
  if ((rc = fetch(niter, (const int64_t*)h->db.niter, (size_t)C))) return rc;
  std::vector<int64_t> pos_off(C + 1, 0);
  std::vector<int> opos, oha, ohc;
  for (int c = 0; c < C; c++) {
    const ChainMeta& m = h->meta[c];
    const int P = status[c] == AHSOKA_ST_PHASED ? npos[c] : 0;
    if (status[c] != AHSOKA_ST_PHASED) { nfin[c] = 0; ncl[c] = 0; niter[c] = 0; }
    for (int j = 0; j < P; j++) opos.push_back(pos[m.pos0 + j]);
    for (int i = 0; i < p; i++)
      for (int j = 0; j < P; j++) {
        oha.push_back(ha[(size_t)p * m.pos0 + (size_t)i * P + j]);
        ohc.push_back(hc[(size_t)p * m.pos0 + (size_t)i * P + j]);
      }
    pos_off[c + 1] = (long long)opos.size();
  }
  out->n_chains = C;
  out->ploidy = p;
  out->status = hostdup(status);
  out->pos_off = hostdup(pos_off);
  out->pos = hostdup(opos);
  out->hap_allele = hostdup(oha);
  out->hap_cluster = hostdup(ohc);
  out->n_reads = hostdup(nfin);
  out->n_clusters = hostdup(ncl);
  out->n_iterations = hostdup(niter);
  if (h->prm.want_stages) {
    std::vector<signed char> vf, vp;
    if ((rc = fetch(vf, (const signed char*)h->db.vfull, (size_t)h->sumVm))) return rc;
    if ((rc = fetch(vp, (const signed char*)h->db.vpart, (size_t)h->sumVm))) return rc;
    if ((rc = fetch(frn, (const int*)h->db.fr_name, (size_t)h->sumG))) return rc;
    if ((rc = fetch(frc, (const int*)h->db.fr_cluster, (size_t)h->sumG))) return rc;
    std::vector<int> sd;
    if ((rc = fetch(sd, (const int*)h->db.score_dump, (size_t)h->sumDump))) return rc;
    std::vector<int64_t> vmat_off(C + 1, 0), fread_off(C + 1, 0), score_off(C + 1, 0);
    std::vector<int8_t> ovf((size_t)h->sumVm), ovp((size_t)h->sumVm);
    std::vector<int> ofn, ofc, osc;
    for (int c = 0; c < C; c++) {
      const ChainMeta& m = h->meta[c];
      const bool ran = status[c] != AHSOKA_ST_SKIPPED_LE1 && status[c] != AHSOKA_ST_UNSUPPORTED;
      // device layout is position-major [B][R]; the ABI (and the oracle) is [R][B]
      for (int r = 0; r < m.R; r++)
        for (int k = 0; k < m.B; k++) {
          ovf[(size_t)m.vmat0 + (size_t)r * m.B + k] = ran ? vf[(size_t)m.vmat0 + (size_t)k * m.R + r] : -1;
          ovp[(size_t)m.vmat0 + (size_t)r * m.B + k] = ran ? vp[(size_t)m.vmat0 + (size_t)k * m.R + r] : -1;
        }
      // k_readset writes the matrices before it knows whether S1 is empty; the
      // oracle returns them in that case too.  S2 of a NO_FULL_MATCH chain is never run
      // by either, but vpart is computed up-front here: mask it like the oracle does.
      vmat_off[c + 1] = vmat_off[c] + (long long)m.R * m.B;
      const int n = nfin[c];
      for (int i = 0; i < n; i++) {
        ofn.push_back(frn[m.fr0 + i]);
        ofc.push_back(frc[m.fr0 + i]);
      }
      for (long long e = 0; e < (long long)n * n; e++) osc.push_back(sd[(size_t)m.dump0 + e]);
      fread_off[c + 1] = (long long)ofn.size();
      score_off[c + 1] = (long long)osc.size();
    }
    out->vmat_off = hostdup(vmat_off);
    out->vfull = hostdup(ovf);
    out->vpart = hostdup(ovp);
    out->fread_off = hostdup(fread_off);
    out->fread_name = hostdup(ofn);
    out->fread_cluster = hostdup(ofc);
    out->score_off = hostdup(score_off);
    out->score_q = hostdup(osc);
  }
  // hostdup is a plain malloc: a failed one must not come back as AHSOKA_OK with a NULL array
  bool ok = out->status && out->pos_off && out->pos && out->hap_allele && out->hap_cluster && out->n_reads && out->n_clusters &&
            out->n_iterations;
  if (h->prm.want_stages)
    ok = ok && out->vmat_off && out->vfull && out->vpart && out->fread_off && out->fread_name && out->fread_cluster && out->score_off &&
         out->score_q;
  if (!ok) return AHSOKA_E_NOMEM;  // (the wrapper releases what was allocated)
  out->elapsed_ms = h->last_ms;
  out->kernel_ms = (double)h->k_ms[0] + (double)h->k_ms[1];
  return AHSOKA_OK;
}

extern "C" int ahsoka_batch_download(ahsoka_device_batch* h, ahsoka_phase_result* out) {
  int rc;
  try {
    rc = download_impl(h, out);
  } catch (const std::bad_alloc&) {
    rc = AHSOKA_E_NOMEM;
  } catch (...) {
    rc = AHSOKA_E_HIP;
  }
  if (rc != AHSOKA_OK && rc != AHSOKA_E_ARG && out) ahsoka_result_free(out);  // (zeroes *out as well)
  return rc;
}

extern "C" void ahsoka_batch_free(ahsoka_device_batch* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  free_classes(h);
  for (void* p : h->allocs) (void)hipFree(p);
  for (int i = 0; i < 3; i++)
    if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

extern "C" void ahsoka_result_free(ahsoka_phase_result* r) {
  if (!r) return;
  std::free(r->status); std::free(r->pos_off); std::free(r->pos); std::free(r->hap_allele);
  std::free(r->hap_cluster); std::free(r->n_reads); std::free(r->n_clusters); std::free(r->n_iterations);
  std::free(r->vmat_off); std::free(r->vfull); std::free(r->vpart);
  std::free(r->fread_off); std::free(r->fread_name); std::free(r->fread_cluster);
  std::free(r->score_off); std::free(r->score_q);
  std::memset(r, 0, sizeof(*r));
}

extern "C" int ahsoka_phase_batch(const ahsoka_chain_batch* in, const ahsoka_phase_params* prm, ahsoka_phase_result* out) {
  if (!in || !prm || !out) return AHSOKA_E_ARG;
  auto t0 = std::chrono::steady_clock::now();
  ahsoka_device_batch* h = nullptr;
  int rc = ahsoka_batch_upload(in, prm, &h);
  if (rc) return rc;
  rc = ahsoka_phase_device(h);
  if (!rc) rc = ahsoka_batch_download(h, out);
  ahsoka_batch_free(h);
  if (!rc) out->elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

// Algorithmic bytes of the INPUT side: everything read once,
// 4-byte elements.  The output side, 4*(P + ploidy*P + 1) per chain, depends on
// the result and is added by the caller.
extern "C" int ahsoka_batch_bytes(const ahsoka_chain_batch* in, int32_t ploidy, int64_t* algorithmic_bytes) {
  (void)ploidy;
  if (!in || !algorithmic_bytes) return AHSOKA_E_ARG;
  const int C = in->n_chains;
  long long total = 0;
  for (int c = 0; c < C; c++) {
    const long long R = in->read_off[c + 1] - in->read_off[c];
    const long long L = in->rnode_off[in->read_off[c + 1]] - in->rnode_off[in->read_off[c]];
    const long long B = in->bubble_off[c + 1] - in->bubble_off[c];
    const long long A = in->allele_off[in->bubble_off[c + 1]] - in->allele_off[in->bubble_off[c]];
    const long long Pn = in->apath_off[in->allele_off[in->bubble_off[c + 1]]] - in->apath_off[in->allele_off[in->bubble_off[c]]];
    total += 4 * ((R + 1) + L + R + R) + 4 * ((B + 1) + (A + 1) + Pn);
  }
  *algorithmic_bytes = total;
  return AHSOKA_OK;
}

// ---- rendering, src/alignmentstoreadset.cpp:82-83,415-458 ---------------------
extern "C" int64_t ahsoka_render_chain(const ahsoka_chain_batch* in, const ahsoka_phase_result* res, int32_t c, char* buf,
                                       int64_t buflen) {
  if (!in || !res || c < 0 || c >= in->n_chains || c >= res->n_chains) return AHSOKA_E_ARG;
  std::string s;
  const long long b0 = in->bubble_off[c], B = in->bubble_off[c + 1] - b0;
  s += "chain id: " + std::to_string(in->chain_id ? in->chain_id[c] : c) + "\n";
  s += "size of chain: " + std::to_string(B) + "\n";
  if (res->status[c] == AHSOKA_ST_PHASED) {
    const long long p0 = res->pos_off[c], P = res->pos_off[c + 1] - p0;
    const int32_t* ha = res->hap_allele + (long long)res->ploidy * p0;
    const int32_t* nid = in->node_id + in->node_off[c];
    for (int i = 0; i < res->ploidy; i++) {
      s += "haplotype " + std::to_string(i) + ":\n";
      std::vector<int> used;
      for (long long j = 0; j < P; j++) {
        const int bub = res->pos[p0 + j];
        const int cons = ha[(long long)i * P + j];
        const long long a0 = in->allele_off[b0 + bub], a1 = in->allele_off[b0 + bub + 1];
        if (cons < 0 || cons >= a1 - a0) continue;  // .at(cons) would throw in the reference (:424)
        const long long e0 = in->apath_off[a0 + cons], e1 = in->apath_off[a0 + cons + 1];
        for (long long e = e0; e + 1 < e1; e++) {  // the last node of a path is never printed (:433)
          const int id = nid[in->apath_node[e]];
          if (std::find(used.begin(), used.end(), id) != used.end()) continue;
          const char sign = in->apath_sign ? (char)in->apath_sign[e] : '+';
          s += std::to_string(id) + "(" + sign + "),";
          used.push_back(id);
        }
      }
      s += "\n";
    }
  }
  if (buf && buflen > (int64_t)s.size()) std::memcpy(buf, s.c_str(), s.size() + 1);
  return (int64_t)s.size();
}

// ---- host-only test hooks ------------------------------------------------------
extern "C" int ahsoka_test_introsort_u32(uint32_t* v, int32_t n) {
  if (!v || n < 0) return AHSOKA_E_ARG;
  gnu_sort(v, n, LessHi16());
  return AHSOKA_OK;
}
// the parallelisable form of the same sort (gnusort.h: gnu_sort_par_ref), sequential statement
extern "C" int ahsoka_test_introsort_par_u32(uint32_t* v, int32_t n) {
  if (!v || n < 0 || n > 65535) return AHSOKA_E_ARG;
  std::vector<uint16_t> pl((size_t)n + 1), pr((size_t)n + 1);
  std::vector<uint32_t> cb((size_t)n / 32 + 2), tmp((size_t)n + 1);
  gnu_sort_par_ref(v, n, LessHi16(), pl.data(), pr.data(), cb.data(), tmp.data());
  return AHSOKA_OK;
}
extern "C" double ahsoka_test_detlog(double x) { return det_log(x); }
