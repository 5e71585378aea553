// k_phase.hip -- stages S4..S8 of the per-chain phasing path for gfx950:
// a persistent grid of workgroups, each pulling whole bubble chains from a
// queue and running, for one chain at a time,
//
//   S4  read-pair similarity      (ReadScoring::scoreReadsetLocal,  :308-311)
//   S5  cluster editing           (ClusterEditingSolver::run,       :312-314)
//   S6  coverage / top clusters / consensus      (:346-404, helpers :550-779)
//   S7  haplotype threading DP    (HaploThreader::computePaths,     :320,408)
//   S8  haplotype -> consensus allele                               (:415-424)
//
// (line numbers: src/alignmentstoreadset.cpp of the reference).
//
// Data layout: reads are bit-planes over the chain's covered positions
// (cover word + allele planes, word-major so lanes = reads load coalesced);
// overlap and Hamming distance of a pair are popcounts.  The similarity
// matrix W (int32, Q15.16) and the induced costs IC (2 x int64 per pair) are
// dense n x n, symmetric, in this workgroup's slot of HBM; every access
// pattern below is "one row, consecutive lanes = consecutive columns".  The
// per-row maxima that drive the heuristic's edge choice live in LDS as packed
// 64-bit (value, column) keys so that a row update is one ds_max_u64.
//
// All induced-cost arithmetic is exact integer arithmetic, so the result does
// not depend on the order in which lanes apply updates; that is what lets one
// outer iteration (one merge or one separation of two cliques, with all its
// implied edges) be applied by 256 lanes at once and still equal the oracle's
// edge-by-edge sequential loop bit for bit.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "detmath.h"
#include "gnusort.h"
#include "phase_kernels.h"
#include "../../include/ahsoka_phase.h"

// Diagnostic build only (-DAHK_TIMING): workgroup 0 prints, for its first chain,
// the shader-clock cycles spent in each phase of the cluster-editing loop.
#ifdef AHK_TIMING
#define AHK_T(slot)                                      \
  do {                                                   \
    if (tid == 0) {                                      \
      const long long _t = (long long)clock64();         \
      tacc[slot] += _t - tlast;                          \
      if (heavy_it) tacc2[slot] += _t - tlast;           \
      tlast = _t;                                        \
    }                                                    \
  } while (0)
#else
#define AHK_T(slot) do {} while (0)
#endif

// Diagnostic build only (-DAHK_FTRACE): per wave, cycles between consecutive marks inside the
// argmax and the single-read separation; workgroup 0 prints the sums for its first chain.
#ifdef AHK_FTRACE
#define AHK_F(slot)                                                               \
  do {                                                                            \
    const long long _t = (long long)__builtin_amdgcn_s_memtime();                 \
    if (lane == 0) atomicAdd(&s_ft[wv][slot], (unsigned long long)(_t - ftl));    \
    ftl = _t;                                                                     \
  } while (0)
#else
#define AHK_F(slot) do {} while (0)
#endif

// Diagnostic build only (-DAHK_LINES): workgroup 0 counts, per load/store site of the S5 loop, the wave
// instructions issued and the distinct 128-B lines they touch (what the L1 asks the L2 for if nothing hits);
// printed per chain (tools/lines.sh).  The product build compiles none of this.
#ifdef AHK_LINES
__shared__ unsigned long long g_lines[32], g_reqs[32], g_lanes[32];
__device__ __noinline__ void ahk_count_lines(int slot, const void* p) {
  if (blockIdx.x != 0) return;
  const unsigned long long a = (unsigned long long)p >> 7;
  const unsigned long long ex = __ballot(true);
  unsigned long long rem = ex;
  int cnt = 0;
  while (rem != 0ull) {
    const int l = __builtin_ctzll(rem);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)a, l);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(a >> 32), l);
    const unsigned long long al = ((unsigned long long)hi << 32) | lo;
    rem &= ~__ballot(a == al);
    cnt++;
  }
  const int me = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  if (me == __builtin_ctzll(ex)) {
    atomicAdd(&g_lines[slot], (unsigned long long)cnt);
    atomicAdd(&g_reqs[slot], 1ull);
    atomicAdd(&g_lanes[slot], (unsigned long long)__popcll(ex));
  }
}
#define AHK_L(slot, ptr) ahk_count_lines(slot, (const void*)(ptr))
#else
#define AHK_L(slot, ptr) do {} while (0)
#endif

// tuning switches (tools/variants.sh builds and benches combinations on one box)
#ifndef AHK_P3B_SINGLE
#define AHK_P3B_SINGLE 1
#endif
#ifndef AHK_LANE_EXPAND_MAX
#define AHK_LANE_EXPAND_MAX 8
#endif
#ifndef AHK_DPP_SUM
#define AHK_DPP_SUM 1
#endif
#ifndef AHK_MINMAX_TERMS
#define AHK_MINMAX_TERMS 1
#endif

namespace ahk {

// Element access with a 32-bit BYTE offset: "base + zext(32-bit offset)" is what the
// global_load/store saddr form takes (scalar base, one VGPR offset); indexing with
// base[zext(i)] scales AFTER the extension, i.e. a 64-bit shift and a 64-bit add in the VALU per
// access.  Offsets stay far below 2^32 (a slot's largest array is n^2 * 16 B = 64 MB).
#ifndef AHK_AT32
#define AHK_AT32 1
#endif
#if AHK_AT32
#define AHK_AT(T, base, i) (*(T*)((char*)(base) + (size_t)((unsigned int)(i) * (unsigned int)sizeof(T))))
#else
#define AHK_AT(T, base, i) (((T*)(base))[(unsigned int)(i)])
#endif
#define WM_AT(i) AHK_AT(int, Wm, i)
#define IC_AT(i) AHK_AT(IcT, IC, i)  // (IcT: IcPair or IcPair32, a template parameter wherever this is used)
#define LV_AT(i) AHK_AT(unsigned long long, LV, i)
#define NZ_AT(i) AHK_AT(unsigned long long, NZ, i)

constexpr int kLaneExpandMax = AHK_LANE_EXPAND_MAX;  // P3a: bits per word up to which a lane expands its own word
constexpr int ROLE_NONE = 0, ROLE_A = 1, ROLE_B = 2, ROLE_FA = 3, ROLE_FB = 4;

// Triangle terms in 32-bit: W_PERM = INT_MAX and W_FORB = -INT_MAX act as +-infinity
// under min/negation.  (A term with BOTH edges infinite never reaches a live
// edge: the third edge of such a triangle is already implied.)
__device__ __forceinline__ int wval(int w) { return w; }
// (min/max forms of: "both positive -> the smaller", and "opposite signs -> the smaller
// magnitude"; two and five VALU instructions, no compares.  Inputs are never INT_MIN.)
#if AHK_MINMAX_TERMS
__device__ __forceinline__ int icf_term(int p, int q) {
  const int m = p < q ? p : q;
  return m > 0 ? m : 0;
}
__device__ __forceinline__ int icp_term(int p, int q) {
  const int np = -p, nq = -q;
  const int a = np < q ? np : q, b = p < nq ? p : nq;
  const int m = a > b ? a : b;
  return m > 0 ? m : 0;
}
#else
__device__ __forceinline__ int icf_term(int p, int q) {
  return (p > 0 && q > 0) ? (p < q ? p : q) : 0;
}
__device__ __forceinline__ int icp_term(int p, int q) {
  const int np = -p, nq = -q;
  if (p < 0 && q > 0) return q < np ? q : np;
  if (p > 0 && q < 0) return nq < p ? nq : p;
  return 0;
}
#endif
// Packed key of an edge (a,b), a < b, with induced cost val >= 0:
//   (val+1) << 25 | stale << 24 | (4095-a) << 12 | (4095-b)
// One unsigned 64-bit max picks the largest value and, among equal values, the
// lexicographically smallest pair -- the heuristic's global tie rule.  Within one
// row that is also "smallest other end", so the per-row maxima are kept in this
// very form and the argmax is a plain max over them.  The stale bit marks a row
// maximum that is only an upper bound (its edge went down or died): among equal
// values a stale key wins the max, which is exactly "some stale row ties the
// maximum -> rescan before deciding".  The ends are 12-bit fields (n <= 4096 = kMaxReads), which
// leaves 39 bits for val+1: exact for every narrow (32-bit induced cost) chain; the wide instance
// checks its chain's bound 2 * max row sum of |W| < 2^39 - 1 once, after S4b, and reports a chain
// beyond it AHSOKA_ST_UNSUPPORTED (weights are clamped to 2^30: more than 256 neighbours at the clamp).
// 0 = no live edge.
constexpr int kKeyEndBits = 12;
constexpr unsigned int kKeyEndMax = (1u << kKeyEndBits) - 1u;            // 4095
constexpr int kKeyValShift = 2 * kKeyEndBits + 1;                         // 25
constexpr unsigned long long kStale = 1ull << (2 * kKeyEndBits);          // bit 24
constexpr unsigned long long kPairMask = (1ull << (2 * kKeyEndBits)) - 1ull;
constexpr long long kKeyValMax = (1ll << (64 - kKeyValShift)) - 2;       // largest val a key can carry
static_assert(kMaxReads <= (1 << kKeyEndBits), "packed edge keys hold 12-bit read indices");
__device__ __forceinline__ unsigned long long mkkey(long long val, int r, int c) {
  const unsigned int a = r < c ? r : c, b = r < c ? c : r;
  return ((unsigned long long)(val + 1) << kKeyValShift) | (unsigned long long)(((kKeyEndMax - a) << kKeyEndBits) | (kKeyEndMax - b));
}
__device__ __forceinline__ long long key_val(unsigned long long k) { return (long long)(k >> kKeyValShift) - 1; }
__device__ __forceinline__ int key_a(unsigned long long k) { return (int)kKeyEndMax - (int)((k >> kKeyEndBits) & (unsigned long long)kKeyEndMax); }
__device__ __forceinline__ int key_b(unsigned long long k) { return (int)kKeyEndMax - (int)(k & (unsigned long long)kKeyEndMax); }
// the other end of the edge in key k, seen from its end r
__device__ __forceinline__ int key_other(unsigned long long k, int r) {
  const int a = key_a(k), b = key_b(k);
  return a == r ? b : a;
}
// fold the new value of an edge into the maximum of the row of one of its ends:
// a maximum that sat on this very edge and went down is now only an upper bound
__device__ __forceinline__ void fold_key(unsigned long long* best, unsigned long long c) {
  const unsigned long long k = *best;
  if (((k ^ c) & kPairMask) == 0ull && c < (k & ~kStale)) atomicOr(best, kStale);
  atomicMax(best, c);
}

// value of a 64-bit register in lane l, l wave-uniform (v_readlane, no LDS crossbar)
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, l);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}
// number of set bits of a wave-uniform mask below this lane
__device__ __forceinline__ int mask_rank(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}

// sum over the 64 lanes, result in every lane: DPP row shifts / row broadcasts (VALU only;
// __shfl_xor would be six trips through the LDS crossbar)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_add_step(int v) {
  return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROWMASK, 0xf, false);
}
__device__ __forceinline__ int wave_sum_i(int v) {
#if AHK_DPP_SUM
  v = dpp_add_step<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add_step<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add_step<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add_step<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row holds the row sum
  v = dpp_add_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the sum
  return __builtin_amdgcn_readlane(v, 63);
#else
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
#endif
}
// inclusive prefix sum over the 64 lanes (same DPP sequence, result per lane)
__device__ __forceinline__ int wave_scan_i(int v) {
  v = dpp_add_step<0x111, 0xf>(v);
  v = dpp_add_step<0x112, 0xf>(v);
  v = dpp_add_step<0x114, 0xf>(v);
  v = dpp_add_step<0x118, 0xf>(v);
  v = dpp_add_step<0x142, 0xa>(v);
  v = dpp_add_step<0x143, 0xc>(v);
  return v;
}
// (for sums known to stay below 2^31: the S4a counters are at most 1023 per pair)
__device__ __forceinline__ long long wave_sum_ll(long long v) { return (long long)wave_sum_i((int)v); }
// max over the 64 lanes, result in every lane.  DPP row shifts / row broadcasts
// run in the VALU (a few cycles each); __shfl_xor would go through the LDS
// crossbar (ds_bpermute, ~100 cycles a step).  0 is the identity.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned long long dpp_max_step(unsigned long long v) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)v, CTRL, ROWMASK, 0xf, false);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(v >> 32), CTRL, ROWMASK, 0xf, false);
  const unsigned long long t = ((unsigned long long)hi << 32) | lo;
  return t > v ? t : v;
}
#ifndef AHK_MAX_SPLIT
#define AHK_MAX_SPLIT 1
#endif
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned int dpp_max_step32(unsigned int v) {
  const unsigned int t = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);
  return t > v ? t : v;  // (the compiler folds the DPP move into v_max_u32: one instruction per step)
}
__device__ __forceinline__ unsigned int wave_max_u32(unsigned int v) {
  v = dpp_max_step32<0x111, 0xf>(v);
  v = dpp_max_step32<0x112, 0xf>(v);
  v = dpp_max_step32<0x114, 0xf>(v);
  v = dpp_max_step32<0x118, 0xf>(v);
  v = dpp_max_step32<0x142, 0xa>(v);
  v = dpp_max_step32<0x143, 0xc>(v);
  return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#if AHK_MAX_SPLIT
  // a 64-bit maximum as two 32-bit ones: the high words first, then the low words of the lanes
  // that hold the maximal high word.  A 64-bit step costs two DPP moves, a 64-bit compare and
  // two selects; a 32-bit step is ONE v_max_u32 with a DPP operand -- 16 instead of 36
  // instructions per reduction, and the reductions are a third of a fast iteration.
  const unsigned int hi = (unsigned int)(v >> 32), lo = (unsigned int)v;
  const unsigned int mh = wave_max_u32(hi);
  const unsigned int ml = wave_max_u32(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | ml;
#else
  v = dpp_max_step<0x111, 0xf>(v);  // row_shr:1
  v = dpp_max_step<0x112, 0xf>(v);  // row_shr:2
  v = dpp_max_step<0x114, 0xf>(v);  // row_shr:4
  v = dpp_max_step<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row holds the row maximum
  v = dpp_max_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_max_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the maximum
  return readlane_u64(v, 63);
#endif
}

// fold_key of the keys (c1 into bestF[x], c2 into bestP[x]) of every active lane, for lanes that
// mostly share their row x (consecutive edges of a row): 64 LDS atomics on one address take
// turns, a few hundred cycles each wave instruction, so the lanes of a row are combined first
// (one maximum, one "a stored maximum went down" flag, applied by one lane).  Any order of the
// single updates gives a valid state (an upper bound per row, exact unless marked stale), and so
// does this.  Up to four rows per call are combined; lanes of further rows update on their own.
__device__ __forceinline__ void fold_rows_grouped(bool act, int x, unsigned long long c1, unsigned long long c2, unsigned long long* bestF,
                                                  unsigned long long* bestP, int lane) {
  unsigned long long rem = __ballot(act);
#pragma unroll 1
  for (int g = 0; g < 4 && rem != 0ull; g++) {
    const int l0 = __builtin_ctzll(rem);
    const int xs = __builtin_amdgcn_readlane(x, l0);
    const bool mine = act && x == xs;
    const unsigned long long kF0 = bestF[xs], kP0 = bestP[xs];
    const unsigned long long mF = wave_max_u64(mine ? c1 : 0ull), mP = wave_max_u64(mine ? c2 : 0ull);
    const bool stF = mine && ((kF0 ^ c1) & kPairMask) == 0ull && c1 < (kF0 & ~kStale);
    const bool stP = mine && ((kP0 ^ c2) & kPairMask) == 0ull && c2 < (kP0 & ~kStale);
    const unsigned long long bF = __ballot(stF), bP = __ballot(stP);
    if (lane == l0) {
      if (bF != 0ull) atomicOr(&bestF[xs], kStale);
      atomicMax(&bestF[xs], mF);
      if (bP != 0ull) atomicOr(&bestP[xs], kStale);
      atomicMax(&bestP[xs], mP);
    }
    rem &= ~__ballot(mine);
  }
  if (act && ((rem >> lane) & 1ull)) {
    fold_key(&bestF[x], c1);
    fold_key(&bestP[x], c2);
  }
}

// does the one triangle (y, rs, b) of a staged-row edge change its induced costs?  ov = W[rs][y] (old
// weight of the pair that is rewritten to +-infinity), wb = W[rs][b] (third edge)
__device__ __forceinline__ bool staged_changes(int ov, int wb, bool perm) {
  if (perm) return icf_term(W_PERM, wb) != icf_term(ov, wb) || icp_term(W_PERM, wb) != icp_term(ov, wb);
  return icf_term(W_FORB, wb) != icf_term(ov, wb) || icp_term(W_FORB, wb) != icp_term(ov, wb);
}

// is the pair (role r1, role r2) rewritten by this iteration, and to what:
// bit (r1*5 + r2) of a 25-bit table
constexpr unsigned int pair_bit(int r1, int r2) { return 1u << (r1 * 5 + r2); }
constexpr unsigned int kPairAB = pair_bit(ROLE_A, ROLE_B) | pair_bit(ROLE_B, ROLE_A);
constexpr unsigned int kPairMerge = kPairAB | pair_bit(ROLE_B, ROLE_FA) | pair_bit(ROLE_FA, ROLE_B) |
                                    pair_bit(ROLE_A, ROLE_FB) | pair_bit(ROLE_FB, ROLE_A);
__device__ __forceinline__ bool in_s(int r1, int r2, bool merge) {
  return ((merge ? kPairMerge : kPairAB) >> (r1 * 5 + r2)) & 1u;
}
__device__ __forceinline__ int new_w(int r1, int r2, bool merge) {
  return (merge && ((kPairAB >> (r1 * 5 + r2)) & 1u)) ? W_PERM : W_FORB;
}


// Induced costs are stored for both orientations of an edge: every read of row
// r touches IC[r][.] only (neighbours of a read cluster in a few 128-B lines),
// and only the few edges that actually change (~8 % of the candidates) pay for
// the second, scattered store.  (A single copy at [min][max] made half of all
// reads column accesses: one HBM line per 16 useful bytes.)
__device__ __forceinline__ unsigned ic_index(int r, int c, int n) { return (unsigned)(r * n + c); }
// LIVE-WORD SUMMARY (LDS, 16 bits per read): bit i of lwm[r] is set while the bitmap words [i << glog, (i + 1) << glog) of
// row r may hold a live edge (glog = 0 for rows of at most 16 words -- the bit then says exactly "word i is not empty" --,
// 1 up to 32 words, 2 up to 64; a set bit over an empty word only costs a load that is masked out afterwards, a clear
// bit over a live edge would be a wrong result: bits are cleared only by the lane that has just written a zero word,
// and only for glog == 0).  With it a pass that scans a row's induced costs knows which 64-column words to ask for
// WITHOUT first waiting for the row's bitmap: the bitmap words and the induced costs travel in the same round trip.
#ifndef AHK_LWM
#define AHK_LWM 1  // 0: bitmap first, then the induced costs of the live words (two dependent round trips)
#endif
__device__ __forceinline__ unsigned long long lwm_words(unsigned int m, int glog, int nw) {
  unsigned long long t = m;
  if (glog == 1) {
    t = (t | (t << 8)) & 0x00ff00ffull;
    t = (t | (t << 4)) & 0x0f0f0f0full;
    t = (t | (t << 2)) & 0x33333333ull;
    t = (t | (t << 1)) & 0x55555555ull;
    t |= t << 1;
  } else if (glog == 2) {
    t = (t | (t << 24)) & 0x000000ff000000ffull;
    t = (t | (t << 12)) & 0x000f000f000f000full;
    t = (t | (t << 6)) & 0x0303030303030303ull;
    t = (t | (t << 3)) & 0x1111111111111111ull;
    t *= 15ull;
  }
  return nw >= 64 ? t : (t & ((1ull << nw) - 1ull));
}
// (two rows share a 32-bit word of the summary: the atomic leaves the other row's half alone)
__device__ __forceinline__ void lwm_clear(unsigned short* lwm, int r, int bit) {
  atomicAnd((unsigned int*)lwm + (r >> 1), ~((1u << bit) << ((r & 1) << 4)));
}

// Exact maxima of rows r0 and r1 (r1 < 0: none) over their live edges.  Lane = bit
// of a 64-column bitmap word; the non-empty words of both rows form one work list
// that is consumed four words at a time: the four induced-cost loads of a batch are
// issued before any of them is used (a load-use pair per loop trip would cost one
// HBM round trip per word, and a row's live edges sit in 2-4 words).
#ifndef AHK_RESCAN_BATCH
#define AHK_RESCAN_BATCH 4
#endif
constexpr int kRescanBatch = AHK_RESCAN_BATCH;  // bitmap words of a rescan whose induced costs are requested together
template <class IcT>
__device__ __forceinline__ void rescan_rows2(int r0, int r1, int n, int nw, int lane, const unsigned long long* __restrict__ LV,
                                             const IcT* __restrict__ IC, const unsigned short* lwm, int glog, unsigned long long* keys /* F0 P0 F1 P1 */) {
  const unsigned long long bm0 = lane < nw ? LV_AT((unsigned)(r0 * nw + lane)) : 0ull;
  const unsigned long long bm1 = (r1 >= 0 && lane < nw) ? LV_AT((unsigned)(r1 * nw + lane)) : 0ull;
  if (lane < nw) { AHK_L(5, &LV_AT((unsigned)(r0 * nw + lane))); if (r1 >= 0) AHK_L(5, &LV_AT((unsigned)(r1 * nw + lane))); }
#if AHK_LWM
  // the words to scan come from the LDS summary: their induced costs are requested together with the bitmap rows
  unsigned long long todo0 = lwm_words((unsigned)__builtin_amdgcn_readfirstlane((int)lwm[r0]), glog, nw);
  unsigned long long todo1 = r1 >= 0 ? lwm_words((unsigned)__builtin_amdgcn_readfirstlane((int)lwm[r1]), glog, nw) : 0ull;
#else
  unsigned long long todo0 = __ballot(bm0 != 0ull), todo1 = __ballot(bm1 != 0ull);
#endif
  unsigned long long kF0 = 0, kP0 = 0, kF1 = 0, kP1 = 0;
  while ((todo0 | todo1) != 0ull) {
    IcT v[kRescanBatch];
    int wl[kRescanBatch];
    bool second[kRescanBatch];
#pragma unroll
    for (int c = 0; c < kRescanBatch; c++) {
      wl[c] = -1;
      second[c] = false;
      v[c].f = 0;  // (defined on every path: an undefined value becomes a register that lives across the caller's loop)
      v[c].p = 0;
      if (todo0 != 0ull) {
        wl[c] = __builtin_ctzll(todo0);
        todo0 &= todo0 - 1ull;
      } else if (todo1 != 0ull) {
        wl[c] = __builtin_ctzll(todo1);
        todo1 &= todo1 - 1ull;
        second[c] = true;
      }
#if AHK_LWM
      if (wl[c] >= 0) {  // uniform: every lane of the word asks (the last word's lanes beyond the row stay inside it)
        const int col = (wl[c] << 6) + lane;
        v[c] = IC_AT(ic_index(second[c] ? r1 : r0, col < n ? col : n - 1, n));
        AHK_L(6, &IC_AT(ic_index(second[c] ? r1 : r0, col < n ? col : n - 1, n)));
      }
#else
      if (wl[c] >= 0 && ((readlane_u64(second[c] ? bm1 : bm0, wl[c]) >> lane) & 1ull)) {
        v[c] = IC_AT(ic_index(second[c] ? r1 : r0, (wl[c] << 6) + lane, n));
        AHK_L(6, &IC_AT(ic_index(second[c] ? r1 : r0, (wl[c] << 6) + lane, n)));
      }
#endif
    }
#pragma unroll
    for (int c = 0; c < kRescanBatch; c++) {
      if (wl[c] < 0) continue;
      const unsigned long long w = readlane_u64(second[c] ? bm1 : bm0, wl[c]);
      if (((w >> lane) & 1ull) == 0ull) continue;
      const int rr = second[c] ? r1 : r0, col = (wl[c] << 6) + lane;
      const unsigned long long k1 = mkkey(v[c].f, rr, col), k2 = mkkey(v[c].p, rr, col);
      if (second[c]) {
        kF1 = k1 > kF1 ? k1 : kF1;
        kP1 = k2 > kP1 ? k2 : kP1;
      } else {
        kF0 = k1 > kF0 ? k1 : kF0;
        kP0 = k2 > kP0 ? k2 : kP0;
      }
    }
  }
  keys[0] = wave_max_u64(kF0);
  keys[1] = wave_max_u64(kP0);
  keys[2] = 0ull;  // (always defined: an undefined value here becomes a register that lives across the whole main loop)
  keys[3] = 0ull;
  if (r1 >= 0) {
    keys[2] = wave_max_u64(kF1);
    keys[3] = wave_max_u64(kP1);
  }
}

// rows dlist[0..nd): wave wv takes pairs of rows
template <class IcT, int NWV>
__device__ __forceinline__ void rescan_list(const unsigned short* dlist, int nd, int wv, int lane, int n, int nw,
                                            const unsigned long long* __restrict__ LV, const IcT* __restrict__ IC,
                                            const unsigned short* lwm, int glog, unsigned long long* bestF, unsigned long long* bestP) {
  for (int di = 2 * wv; di < nd; di += 2 * NWV) {
    // (the rows are wave-uniform: as scalars, their address arithmetic runs on the scalar unit)
    const int r0 = __builtin_amdgcn_readfirstlane((int)dlist[di]);
    const int r1 = di + 1 < nd ? __builtin_amdgcn_readfirstlane((int)dlist[di + 1]) : -1;
    unsigned long long keys[4];
    rescan_rows2(r0, r1, n, nw, lane, LV, IC, lwm, glog, keys);
    if (lane == 0) {
      bestF[r0] = keys[0];
      bestP[r0] = keys[1];
      if (r1 >= 0) {
        bestF[r1] = keys[2];
        bestP[r1] = keys[3];
      }
    }
  }
}

// S4a sort key of a scored pair: (Hamming rate d/o as floor(d * 2^20 / o), overlap o) -- the exact order by
// rate, ties by overlap.  o <= 4095 covered positions (kMaxBubbles) take 12 bits; the rate of a pair that
// differs everywhere (d == o: 2^20, the only value above 2^20 - 2^20/o) is stored as 2^20 - 2, which keeps
// its place in the order, still gives d back as ceil(rate * o / 2^20), and leaves 0xFFFFFFFF free as the
// "not scored" mark.  (oracle.cpp: the same function)
__device__ __forceinline__ unsigned int rate_key(unsigned int d, unsigned int o) {
  unsigned int rate = (d << 20) / o;
  if (rate > (1u << 20) - 2u) rate = (1u << 20) - 2u;
  return (rate << 12) | o;
}

struct GreaterCnt {  // cmp() of the reference, :699-701: coverage descending
  __host__ __device__ bool operator()(unsigned int x, unsigned int y) const { return (x >> 16) > (y >> 16); }
};

struct ChainCtx {
  int n, P, Wp, G;
  const unsigned long long* cover;  // word-major, stride G
  const unsigned long long* alle;
  long long plane_stride;
  int planes;
};

__device__ __forceinline__ void pair_overlap(const ChainCtx& cx, int i, int j, int* o_out, int* d_out) {
  int o = 0, d = 0;
  for (int w = 0; w < cx.Wp; w++) {
    const long long oi = (long long)w * cx.G + i, oj = (long long)w * cx.G + j;
    const unsigned long long both = cx.cover[oi] & cx.cover[oj];
    unsigned long long df = cx.alle[oi] ^ cx.alle[oj];
    for (int pl = 1; pl < cx.planes; pl++) df |= cx.alle[pl * cx.plane_stride + oi] ^ cx.alle[pl * cx.plane_stride + oj];
    o += __popcll(both);
    d += __popcll(both & df);
  }
  *o_out = o;
  *d_out = d;
}

// NOTE: the descriptor structs are passed BY VALUE on purpose: pointers that
// arrive in the kernarg segment are known to be global, so the compiler emits
// global_load/global_store.  Read through a pointer-to-struct they degrade to
// flat_* instructions, which count on lgkmcnt as well -- every LDS wait then
// also waits for all outstanding HBM traffic (measured: whole-kernel slowdown).
#ifndef AHK_FAST_BATCH
#define AHK_FAST_BATCH 5
#endif
constexpr int kFastBatch = AHK_FAST_BATCH;  // live words of a row a wave of the fast path loads together
#ifndef AHK_P3B_FULL
#define AHK_P3B_FULL 1
#endif
#ifndef AHK_P3B_SPECIAL
#define AHK_P3B_SPECIAL 1
#endif
#ifndef AHK_P3B_CH
#define AHK_P3B_CH 4
#endif
constexpr int kP3bChunk = AHK_P3B_CH;  // triangles per load batch of an edge in P3b
#ifndef AHK_S7_MARGINAL
#define AHK_S7_MARGINAL 1  // threading DP: predecessor minimum by marginal minima, O(T 2^p), instead of the O(T^2) scan
#endif
#ifndef AHK_HEAVY_WIDE
#define AHK_HEAVY_WIDE 1
#endif
#if AHK_HEAVY_WIDE
#define tri_seg_h tri_seg_wide
#else
#define tri_seg_h tri_seg
#endif
#ifndef AHK_HEAVY
#define AHK_HEAVY 1  // P3b, edges with long member lists: 1 = member lists in chunks over all lanes; 2 = one wave per edge, lanes = members; 0 = off
#endif
#ifndef AHK_HEAVY_MIN
#define AHK_HEAVY_MIN 48
#endif
#ifndef AHK_HEAVY_MAXITEMS
#define AHK_HEAVY_MAXITEMS 768
#endif
constexpr int kHeavyChunk = 16;  // members per chunk (AHK_HEAVY == 1)
constexpr int kHeavyCap = 128;   // heavy edges per round: their accumulators live in rmask/rolenz
#ifndef AHK_IC_LATE
#define AHK_IC_LATE 1  // P3b: load an edge's induced costs only once its delta is known to be non-zero
#endif
constexpr int kHeavyMin = AHK_HEAVY_MIN;            // member-list length from which an edge counts as heavy
constexpr int kHeavyMaxItems = AHK_HEAVY_MAXITEMS;  // beyond that many heavy edges the plain loop keeps every lane busy anyway

// exact 64-bit sum over the 64 lanes (DPP row shifts / broadcasts), result in every lane
template <int CTRL, int ROWMASK>
__device__ __forceinline__ long long dpp_add_step64(long long v) {
  const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(unsigned long long)v, CTRL, ROWMASK, 0xf, false);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)((unsigned long long)v >> 32), CTRL, ROWMASK, 0xf, false);
  return v + (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
  v = dpp_add_step64<0x111, 0xf>(v);
  v = dpp_add_step64<0x112, 0xf>(v);
  v = dpp_add_step64<0x114, 0xf>(v);
  v = dpp_add_step64<0x118, 0xf>(v);
  v = dpp_add_step64<0x142, 0xa>(v);
  v = dpp_add_step64<0x143, 0xc>(v);
  return (long long)readlane_u64((unsigned long long)v, 63);
}

// Triangle terms of edge (x,b), b not itself rewritten, over the members L[0..cnt) of ONE
// list whose pairs (x,w) all get the same new weight Nv (W_PERM for the opposite clique of
// a merge, W_FORB otherwise): with Nv known at compile time the two "new" terms fold to
// max(q,0) / max(-q,0) / 0 -- about half the instructions per triangle.  Full chunks of
// four members run without per-load predicates (each one costs an exec-mask branch) and
// read their ids with one 8-byte LDS load (L must be 8-byte aligned); padding slots load
// nothing and read as 0,0: all four terms vanish.
template <int Nv>
__device__ __forceinline__ void tri_seg(const int* __restrict__ Wm, int n, int x, int b, const unsigned short* L, int cnt,
                                        long long& df, long long& dp, int lslot = 12) {
  int w0 = 0;
#if AHK_P3B_SINGLE
  // a single member (one read opposite a clique): the most common list of all
  if (cnt == 1) {
    const int w = L[0];
    // (W is symmetric: W[x][w] is read as W[w][x] -- row w is the row every edge of this
    // iteration reads its third edge from, row x would be one more HBM line per row x)
    const int ov = WM_AT((unsigned)(w * n + x)), wb = WM_AT((unsigned)(w * n + b));
    AHK_L(lslot, &WM_AT((unsigned)(w * n + x)));
    AHK_L(lslot, &WM_AT((unsigned)(w * n + b)));
    df += icf_term(Nv, wb) - icf_term(ov, wb);
    dp += icp_term(Nv, wb) - icp_term(ov, wb);
    return;
  }
#endif
#if AHK_P3B_FULL
  static_assert(kP3bChunk == 4, "full-chunk loop reads four u16 ids at once");
  for (; w0 + 4 <= cnt; w0 += 4) {
    const uint2 pk = *reinterpret_cast<const uint2*>(L + w0);
    const int wq[4] = {(int)(pk.x & 0xffffu), (int)(pk.x >> 16), (int)(pk.y & 0xffffu), (int)(pk.y >> 16)};
    int ovv[4], wbv[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      ovv[c] = WM_AT((unsigned)(x * n + wq[c]));
      wbv[c] = WM_AT((unsigned)(wq[c] * n + b));
      AHK_L(lslot, &WM_AT((unsigned)(x * n + wq[c])));
      AHK_L(lslot + 1, &WM_AT((unsigned)(wq[c] * n + b)));
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
      df += icf_term(Nv, wbv[c]) - icf_term(ovv[c], wbv[c]);
      dp += icp_term(Nv, wbv[c]) - icp_term(ovv[c], wbv[c]);
    }
  }
#endif
  for (; w0 < cnt; w0 += kP3bChunk) {
    int ovv[kP3bChunk], wbv[kP3bChunk];
#pragma unroll
    for (int c = 0; c < kP3bChunk; c++) {
      const bool ok = w0 + c < cnt;
      const int w = ok ? (int)L[w0 + c] : 0;
      ovv[c] = ok ? WM_AT((unsigned)(x * n + w)) : 0;
      wbv[c] = ok ? WM_AT((unsigned)(w * n + b)) : 0;
      if (ok) { AHK_L(lslot, &WM_AT((unsigned)(x * n + w))); AHK_L(lslot + 1, &WM_AT((unsigned)(w * n + b))); }
    }
#pragma unroll
    for (int c = 0; c < kP3bChunk; c++) {
      df += icf_term(Nv, wbv[c]) - icf_term(ovv[c], wbv[c]);
      dp += icp_term(Nv, wbv[c]) - icp_term(ovv[c], wbv[c]);
    }
  }
}


// the same over a full chunk of the heavy-edge pass, EIGHT members per load batch: that pass is
// a chain of dependent load batches per lane (its arithmetic hides behind other waves), so its
// length is what counts; it has few live values, sixteen loads in flight fit the registers.
template <int Nv>
__device__ __forceinline__ void tri_seg_wide(const int* __restrict__ Wm, int n, int x, int b, const unsigned short* L, int cnt,
                                             long long& df, long long& dp, int lslot = 14) {
  int w0 = 0;
  for (; w0 + 8 <= cnt; w0 += 8) {
    const uint4 pk = *reinterpret_cast<const uint4*>(L + w0);
    const int wq[8] = {(int)(pk.x & 0xffffu), (int)(pk.x >> 16), (int)(pk.y & 0xffffu), (int)(pk.y >> 16),
                       (int)(pk.z & 0xffffu), (int)(pk.z >> 16), (int)(pk.w & 0xffffu), (int)(pk.w >> 16)};
    int ovv[8], wbv[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
      ovv[c] = WM_AT((unsigned)(x * n + wq[c]));
      wbv[c] = WM_AT((unsigned)(wq[c] * n + b));
      AHK_L(lslot, &WM_AT((unsigned)(x * n + wq[c])));
      AHK_L(lslot + 1, &WM_AT((unsigned)(wq[c] * n + b)));
    }
#pragma unroll
    for (int c = 0; c < 8; c++) {
      df += icf_term(Nv, wbv[c]) - icf_term(ovv[c], wbv[c]);
      dp += icp_term(Nv, wbv[c]) - icp_term(ovv[c], wbv[c]);
    }
  }
  if (w0 < cnt) tri_seg<Nv>(Wm, n, x, b, L + w0, cnt - w0, df, dp, lslot);
}

#ifndef AHK_WAVES_PER_SIMD
#define AHK_WAVES_PER_SIMD 5
#endif
#ifndef AHK_INIT_ROWS
#define AHK_INIT_ROWS 4
#endif
constexpr int kInitRows = AHK_INIT_ROWS;  // consecutive rows a wave of S5-init prices together (they share every third-edge gather)
// IC32: induced costs as 2 x uint32 (IcPair32).  Exact for a chain whose induced costs provably stay
// below 2^32: they are sums of non-negative terms, and every term of icf(a,b) / icp(a,b) is at most
// |W[b][x]| where that is finite and |W[a][x]| where it is not (never both infinite on a live edge),
// so icf, icp <= S(a) + S(b) with S(r) the sum of the initial |weights| of row r.  Every row scan
// of the induced costs (single-read separations, rescans) then moves half the HBM sectors; the
// kernel is bound by those (+10 % at config 3).  A chain that fails the test right after S4b is
// put on the `deferred` list and left to the 64-bit instance, launched behind this one with that
// list as its queue (Cdev = its length, on the device).
// NWV: wavefronts per workgroup.  4 is the throughput form (five workgroups per CU, 1280 chains in
// flight); 8 and 16 are the LATENCY forms for a size class whose chains all have a CU (or half of
// one) to themselves -- config 5's largest chains, a batch of few chains, a shard at N = 8: the
// trip-count-bound passes of an iteration (candidate listing, edge pass, rescans, heavy chunks,
// S5 init, S4) shorten with the lane count, the strictly sequential parts (argmax, the two
// dependent round trips of a single-read separation) do not.  Results do not depend on NWV.
// RW: words of the per-word role tables (a compile-time stride: as a run-time value it cost 3 % at config 3):
// 32 for classes of at most 2048 reads, 64 beyond; the wide forms need NWV * 8 for their pending lists.
template <bool IC32, int NWV, int RW>
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? AHK_WAVES_PER_SIMD : 4) void k_phase_t(DeviceBatch db, SlotWorkspace ws, PhaseLds lds, const int* __restrict__ queue,
                                                       int C, const int* __restrict__ Cdev, int* __restrict__ qhead,
                                                       int* __restrict__ deferred, int* __restrict__ ndeferred) {
  using IcT = typename std::conditional<IC32, IcPair32, IcPair>::type;
  constexpr int kThreads = NWV * 64;
  if (Cdev) C = *Cdev;
  extern __shared__ __align__(16) unsigned long long smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  // the wave index as a SCALAR: everything derived from it (the row a wave works on, its
  // share of a list, addresses of wave-uniform loads) then lives in scalar registers and is
  // computed by the scalar unit; as "threadIdx.x >> 6" the compiler must treat it as divergent
#ifndef AHK_SCALAR_WV
#define AHK_SCALAR_WV 1
#endif
#if AHK_SCALAR_WV
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#else
  const int wv = tid >> 6;
#endif
  const int nmax = lds.nmax;
  unsigned long long* bestF = smem;               // [nmax]  (S4: per-wave key buffers; S6: histograms)
  unsigned long long* bestP = bestF + nmax;       // [nmax]
  unsigned short* cl = (unsigned short*)(bestP + nmax);  // [nmax] clique id, later cluster id
  // the member lists of the four roles of an iteration share ONE pool (they are disjoint sets of reads: at most n entries;
  // every list starts on a 16-byte boundary for the 8- and 16-byte reads of the triangle loops).  As four arrays of nmax
  // entries they were 8 B of LDS per read -- the 6 B saved pay for the live-word summary and keep the staged weight row
  // (config 3) or a fourth workgroup per CU (config 4).
  unsigned short* lpool = cl + nmax;                      // [nmax + 32]
  unsigned short* rowbuf = lpool + nmax + 32;             // [4][nmax] as u16 = the first 2*nmax entries of the candidate-edge list (absent if !lds.items_lds)
  unsigned short* dlist = rowbuf + (lds.items_lds ? 4 * (size_t)nmax : 0);  // [nmax] rows whose maximum must be rescanned
  unsigned int* multiw = (unsigned int*)(dlist + nmax);   // [nmax/32] bit per read: its clique has > 1 member
  // RW = words of a bitmap row the class can have, 32 (n <= 2048) or 64: sizes the per-word role tables
  // (and, in the wide forms, gives room for the per-wave pending lists that re-use the tables: 128 u32 per wave)
  static_assert(RW >= NWV * 8 && (RW == 32 || RW == 64 || RW == 128), "role tables");
  unsigned long long* rmask = (unsigned long long*)(multiw + ((nmax + 31) / 32 + 1) / 2 * 2);  // [4][RW] role masks per 64-column word
  unsigned long long* rolenz = rmask + 4 * RW;             // [4][RW] OR of the members' nz-rows
  unsigned short* s_pref = (unsigned short*)rolenz;         // [4][64] word prefixes of the member lists: P2 only, before rolenz is written
                                                            // (as a static array it was the 512 B that cost config 3 an LDS extra)
  unsigned char* role = (unsigned char*)(rolenz + 4 * RW);  // [nmax]
  int* wrow = (int*)(role + nmax);                         // [nmax] weights of a single read's row (see "staged row")
  // per-wave scratch of S4a (key buffer, u32) and S5 init (neighbour list, u16): the four-wave form
  // re-uses bestF/bestP and the role lists, which hold exactly four such buffers; the wide forms
  // have a region of their own (one workgroup or two per CU: the LDS is there)
  unsigned short* lwm = (unsigned short*)(wrow + (lds.stage_row ? nmax : 0));  // [nmax] live-word summary of every row (see lwm_words)
  unsigned int* wscr = NWV > 4 ? (unsigned int*)(((size_t)(lwm + (AHK_LWM ? nmax : 0)) + 7) & ~(size_t)7) : nullptr;  // [NWV][nmax]
  __shared__ int s_chain, s_cnt[4], s_nd, s_items, s_nh, s_nslow;
#ifdef AHK_TIMING
  __shared__ int s_chg;
#endif
  __shared__ unsigned long long s_rk[2 * NWV < 16 ? 16 : 2 * NWV];
#ifdef AHK_FTRACE
  __shared__ unsigned long long s_ft[4][16];
  long long ftl = 0;
  bool ft_first = blockIdx.x == 0;
#endif
  __shared__ int s_asc[4][8];
#if AHK_LWM
  __shared__ int s_clr[2];  // single-read separation: word + 1 of row eu / ev whose last live edge has just died (0: none)
#endif
  __shared__ int s_back;

  int* Wm = ws.W + (size_t)blockIdx.x * ws.nmax * ws.nmax;
  IcT* IC = (IcT*)(ws.IC + (size_t)blockIdx.x * ws.nmax * ws.nmax);  // (a slot holds nmax^2 entries of the wide form)
  unsigned long long* LV = ws.live + (size_t)blockIdx.x * ws.nmax * ((ws.nmax + 63) / 64);
  unsigned long long* NZ = ws.nz + (size_t)blockIdx.x * ws.nmax * ((ws.nmax + 63) / 64);
  unsigned int* items_glb = ws.items + (size_t)blockIdx.x * ws.nmax * ws.nmax;
  double* rates = ws.rates + (size_t)blockIdx.x * 2 * ws.nmax;

  const int ploidy = db.ploidy;

  for (;;) {
    __syncthreads();
    if (tid == 0) s_chain = atomicAdd(qhead, 1);
    __syncthreads();
    const int qi = __builtin_amdgcn_readfirstlane(s_chain);  // (uniform values read from LDS / HBM are made scalar: see wv above)
    if (qi >= C) break;  // every wave of every workgroup reaches this
    const int c = queue[qi];
    if (db.status[c] != AHSOKA_ST_PHASED) continue;
    const ChainMeta m = db.meta[c];
    ChainCtx cx;
    cx.n = __builtin_amdgcn_readfirstlane(db.nfin[c]);
    cx.P = __builtin_amdgcn_readfirstlane(db.npos[c]);
    cx.Wp = (cx.P + 63) >> 6;
    cx.G = m.G;
    cx.cover = db.cover + m.cov0;
    cx.alle = db.alle + m.cov0;
    cx.plane_stride = db.alle_plane_stride;
    cx.planes = db.planes;
    const int n = cx.n;

    // ================= S4a: per-read local Hamming-rate estimates ==========
    {
      unsigned int* keybuf = (NWV > 4 ? wscr : (unsigned int*)bestF) + (size_t)wv * nmax;  // (4 waves x nmax u32 = 16*nmax bytes = bestF + bestP)
      for (int i = wv; i < n; i += NWV) {
        int cnt = 0;
        long long dall = 0, oall = 0;
        for (int j = lane; j < n; j += 64) {
          unsigned int key = 0xFFFFFFFFu;
          if (j != i) {
            int o, d;
            pair_overlap(cx, i, j, &o, &d);
            if (o >= db.min_overlap && o > 0) {
              key = rate_key((unsigned)d, (unsigned)o);
              cnt++;
              dall += d;
              oall += o;
            }
          }
          keybuf[j] = key;
        }
        cnt = wave_sum_i(cnt);
        dall = wave_sum_ll(dall);
        oall = wave_sum_ll(oall);
        double s = 0.05, f = 0.5;
        if (cnt > 0) {
          int msel = cnt / ploidy;
          if (msel == 0) msel = 1;
          // smallest key T with #(key <= T) >= msel
          unsigned int lo = 0, hi = 0xFFFFFFFEu;
          while (lo < hi) {
            const unsigned int mid = lo + ((hi - lo) >> 1);
            int cle = 0;
            for (int j = lane; j < n; j += 64) cle += (keybuf[j] <= mid) ? 1 : 0;
            cle = wave_sum_i(cle);
            if (cle >= msel) hi = mid;
            else lo = mid + 1;
          }
          const unsigned int T = lo;
          int cl_lt = 0;
          long long ds = 0, os = 0;
          for (int j = lane; j < n; j += 64) {
            const unsigned int key = keybuf[j];
            if (key < T) {
              const unsigned int o = key & 4095u, rate = key >> 12;
              const unsigned long long dd = ((unsigned long long)rate * o + ((1u << 20) - 1)) >> 20;
              cl_lt++;
              ds += (long long)dd;
              os += o;
            }
          }
          cl_lt = wave_sum_i(cl_lt);
          ds = wave_sum_ll(ds);
          os = wave_sum_ll(os);
          const unsigned int oT = T & 4095u, rT = T >> 12;
          const long long dT = (long long)(((unsigned long long)rT * oT + ((1u << 20) - 1)) >> 20);
          ds += (long long)(msel - cl_lt) * dT;
          os += (long long)(msel - cl_lt) * (long long)oT;
          const long long dr = dall - ds, orr = oall - os;
          s = os > 0 ? (double)ds / (double)os : 0.05;
          f = orr > 0 ? (double)dr / (double)orr : 0.5;
        }
        if (s < 0.001) s = 0.001;
        if (s > 0.999) s = 0.999;
        if (f < 0.001) f = 0.001;
        if (f > 0.999) f = 0.999;
        if (lane == 0) {
          rates[i] = s;
          rates[n + i] = f;
        }
      }
    }
    __syncthreads();

    // ================= S4b: similarity matrix W ============================
    long long smax = 0;  // (this wave's rows) largest row sum of |W|: decides whether 32-bit induced costs are exact
    for (int i = wv; i < n; i += NWV) {
      const double si = rates[i], fi = rates[n + i];
      long long srow = 0;
      for (int j = lane; j < n; j += 64) {
        int q = 0;
        if (j != i) {
          int o, d;
          pair_overlap(cx, i, j, &o, &d);
          if (o >= db.min_overlap && o > 0) {
            // symmetric in (i,j): (a+b)*0.5 is commutative, so W[i][j] == W[j][i]
            const double S = (si + rates[j]) * 0.5;
            const double D = (fi + rates[n + j]) * 0.5;
            const double t1 = (double)d * (det_log(S) - det_log(D));
            const double t2 = (double)(o - d) * (det_log(1.0 - S) - det_log(1.0 - D));
            q = quantise_score(t1 + t2);
          }
        }
        WM_AT((unsigned)(i * n + j)) = q;
        if (m.dump0 >= 0) db.score_dump[m.dump0 + (size_t)i * n + j] = q;
        srow += q < 0 ? -(long long)q : (long long)q;
      }
      srow = wave_sum_i64(srow);
      smax = srow > smax ? srow : smax;
    }
    if (lane == 0) s_rk[wv] = (unsigned long long)smax;
    for (int u = tid; u < n; u += kThreads) {
      cl[u] = (unsigned short)u;
      role[u] = ROLE_NONE;
    }
    __syncthreads();
    {
      // every induced cost of this chain is at most S(a) + S(b) <= 2 * max row sum: below 2^32, or the chain goes to
      // the wide instance (uniform: every wave reads the same four values); the wide instance in turn needs the
      // value to fit the 39-bit field of the packed edge keys
      unsigned long long sm = s_rk[0];
      for (int k = 1; k < NWV; k++) sm = s_rk[k] > sm ? s_rk[k] : sm;
      if (IC32) {
        if (2ull * sm >= (1ull << 32)) {
          if (tid == 0) deferred[atomicAdd(ndeferred, 1)] = c;
          continue;
        }
      } else if (2ull * sm > (unsigned long long)kKeyValMax) {
        if (tid == 0) {
          db.status[c] = AHSOKA_ST_UNSUPPORTED;
          db.nclusters[c] = 0;
          db.niter[c] = 0;
        }
        continue;
      }
    }

#ifdef AHK_DIAG
    if (lds.stop_after == 1) continue;  // diagnostic build only (results invalid)
#endif

    // ================= S5 init: induced costs of every live edge ===========
    // icf(ab) = max(w_ab,0) + sum_w min(w_aw,w_bw)+ ; icp likewise.  One wave per
    // row a; the row's non-zero columns are compacted into LDS first so the
    // triangle loop only visits actual neighbours of a.  Also writes the row's
    // live-edge bitmap (one u64 per 64 columns): from here on an edge is live
    // iff its bit is set, and IC holds garbage for dead pairs.
    const int nw = (n + 63) >> 6;
    // words per row rounded up to a power of two (>= 8): lane layout of the row passes
    const int nwp_log = nw <= 8 ? 3 : nw <= 16 ? 4 : nw <= 32 ? 5 : 6;
    const int nwp = 1 << nwp_log;
    const int glog = nw <= 16 ? 0 : nw <= 32 ? 1 : 2;  // bitmap words per bit of the live-word summary, as a shift
    const int RP = 64 >> nwp_log;  // rows a wave handles at once
    // Each unordered pair {a,b} is computed once, by the wave of its lower end a (lanes = the
    // neighbours b > a), and stored for both orientations; the row maxima are folded in with LDS
    // atomics.  A wave takes kInitRows CONSECUTIVE rows at a time: reads are sorted by first
    // position, so neighbouring rows have nearly the same neighbours, and over the union of their
    // neighbour lists every third-edge gather W[w][b] (one row segment of W per w: 3-4 HBM lines)
    // is made ONCE and priced against all rows of the group -- a non-neighbour w of a row has
    // weight 0 there and both of its terms vanish, so the union needs no per-row masks.  One row
    // per wave re-read each chain's 1.4 MB of weights ~30 times (a fifth of the kernel's HBM
    // traffic, round 2); four rows per group make a quarter of those gathers and drop the
    // v_readlane of the row weight from three of four terms.
    for (int u = tid; u < n; u += kThreads) { bestF[u] = 0ull; bestP[u] = 0ull; }
    __syncthreads();
    {
      // induced costs of a narrow (IC32) chain are < 2^32 and sums of non-negative terms: 32-bit
      // unsigned accumulators are exact there
      using AccT = typename std::conditional<IC32, unsigned int, long long>::type;
      // nbr: union of the group's neighbours, ascending -- per-wave scratch: the wide forms have their own LDS region, the
      // four-wave form takes the candidate-edge list's LDS part (free here) or, in a class that gave that up for a workgroup
      // per CU, this slot's overflow region in HBM (a few dozen L2-resident lines per wave)
      auto init_rows = [&](unsigned short* nbr) __attribute__((always_inline)) {
      for (int a0 = wv * kInitRows; a0 < n; a0 += NWV * kInitRows) {
        const int R = n - a0 < kInitRows ? n - a0 : kInitRows;
        int deg = 0, dhi = 0;
        unsigned int lw0[kInitRows];  // the rows' live-word summaries
#pragma unroll
        for (int r = 0; r < kInitRows; r++) lw0[r] = 0u;
        for (int w0 = 0; w0 < n; w0 += 64) {
          const int w = w0 + lane;
          bool nzu = false;
#pragma unroll
          for (int r = 0; r < kInitRows; r++) {
            const bool nz = r < R && w < n && WM_AT((unsigned)((a0 + r) * n + w)) != 0;
            const unsigned long long mk = __ballot(nz);
            if (r < R && lane == 0) {
              LV_AT((unsigned)((a0 + r) * nw + (w0 >> 6))) = mk;
              NZ_AT((unsigned)((a0 + r) * nw + (w0 >> 6))) = mk;
            }
            if (mk != 0ull) lw0[r] |= 1u << ((w0 >> 6) >> glog);
            nzu = nzu || nz;
          }
          const unsigned long long mu = __ballot(nzu);
          const unsigned long long mh = __ballot(nzu && w > a0);
          if (nzu) nbr[deg + __popcll(mu & ((1ull << lane) - 1ull))] = (unsigned short)w;
          deg += __popcll(mu);
          dhi += __popcll(mh);
        }
#pragma unroll
        for (int r = 0; r < kInitRows; r++)
          if (AHK_LWM && r < R && lane == 0) lwm[a0 + r] = (unsigned short)lw0[r];
        __threadfence_block();  // (the list is handed from lane to lane of this wave, possibly through global memory)
        // lanes = the union's members b > a0 (the tail of the ascending list): a pair (a_r, b) of the
        // group is this wave's iff b > a_r and W[a_r][b] != 0.  The rows' own weights W[a_r][w] are
        // wave-uniform: 64 of them per row come with ONE load (lane = list index) and are handed out
        // with v_readlane; padding entries read as weight 0 (both terms vanish): nothing is predicated.
        const unsigned short* nhi = nbr + (deg - dhi);
        for (int i0 = 0; i0 < dhi; i0 += 64) {
          const int i = i0 + lane;
          const bool act = i < dhi;
          const int b = act ? (int)nhi[i] : a0;
          int wab[kInitRows];
          AccT f[kInitRows], p[kInitRows];
#pragma unroll
          for (int r = 0; r < kInitRows; r++) {
            wab[r] = r < R ? WM_AT((unsigned)((a0 + r) * n + b)) : 0;  // (a,a) is 0
            f[r] = (AccT)(wab[r] > 0 ? wab[r] : 0);
            p[r] = (AccT)(wab[r] < 0 ? -wab[r] : 0);
          }
          for (int k0 = 0; k0 < deg; k0 += 64) {
            const int kk = k0 + lane;
            const int nb_l = kk < deg ? (int)nbr[kk] : a0;
            int wa_l[kInitRows];
#pragma unroll
            for (int r = 0; r < kInitRows; r++) wa_l[r] = (r < R && kk < deg) ? WM_AT((unsigned)((a0 + r) * n + nb_l)) : 0;
            const int cnt = deg - k0 < 64 ? deg - k0 : 64;
            for (int c0 = 0; c0 < cnt; c0 += 8) {
              int wb[8];
#pragma unroll
              for (int c = 0; c < 8; c++) {
                const int w = __builtin_amdgcn_readlane(nb_l, c0 + c);
                wb[c] = WM_AT((unsigned)(w * n + b));
              }
#pragma unroll
              for (int c = 0; c < 8; c++) {
#pragma unroll
                for (int r = 0; r < kInitRows; r++) {
                  const int wa = __builtin_amdgcn_readlane(wa_l[r], c0 + c);
                  f[r] += (AccT)icf_term(wa, wb[c]);
                  p[r] += (AccT)icp_term(wa, wb[c]);
                }
              }
            }
          }
#pragma unroll
          for (int r = 0; r < kInitRows; r++) {
            const int a = a0 + r;
            if (act && r < R && b > a && wab[r] != 0) {
              IcT v;
              v.f = f[r];
              v.p = p[r];
              IC_AT((unsigned)(a * n + b)) = v;
              IC_AT((unsigned)(b * n + a)) = v;
              const unsigned long long kf = mkkey((long long)f[r], a, b), kp = mkkey((long long)p[r], a, b);
              atomicMax(&bestF[a], kf);
              atomicMax(&bestP[a], kp);
              atomicMax(&bestF[b], kf);
              atomicMax(&bestP[b], kp);
            }
          }
        }
      }
      };
      if (NWV > 4) init_rows((unsigned short*)(wscr + (size_t)wv * nmax));
      else if (lds.items_lds) init_rows(rowbuf + (size_t)wv * nmax);
      else init_rows((unsigned short*)items_glb + (size_t)wv * ws.nmax);
    }
    for (int u = tid; u < (n + 31) / 32; u += kThreads) multiw[u] = 0;
    if (tid < 4) s_cnt[tid] = 0;
    if (tid == 0) { s_nd = 0; s_items = 0; s_nh = 0; s_nslow = 0; }
    __syncthreads();
#ifdef AHK_DIAG
    if (lds.stop_after == 2) continue;
#endif

    // ================= S5 main loop ========================================
    long long iters = 0;
#ifdef AHK_LINES
    if (tid < 32) { g_lines[tid] = 0ull; g_reqs[tid] = 0ull; g_lanes[tid] = 0ull; }
    __syncthreads();
#endif
#ifdef AHK_FTRACE
    if (tid < 64) s_ft[tid >> 4][tid & 15] = 0ull;
    __syncthreads();
    long long ft_nfast = 0, ft_nother = 0;
    ftl = (long long)__builtin_amdgcn_s_memtime();
#endif
#ifdef AHK_TIMING
    int s_items_last = 0;
    if (tid == 0) s_chg = 0;
    long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = (long long)clock64(), nretry = 0, nresc = 0, sumT = 0, sumM = 0, sumW = 0;
    long long hM[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hMc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hT[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hTc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, titer = tlast, nmerge = 0, sumnd = 0, nfast = 0;
    long long tacc2[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, nheavy = 0;
    long long pl_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long pcat_c[4] = {0, 0, 0, 0}, pcat_n[4] = {0, 0, 0, 0}, pcat_m[4] = {0, 0, 0, 0}, pcat_f[4] = {0, 0, 0, 0};
    bool heavy_it = false;
    long long h2[32], h2c[32], h2m[32];
    for (int q = 0; q < 32; q++) { h2[q] = 0; h2c[q] = 0; h2m[q] = 0; }
#endif
    for (;;) {
      // ---- P1: edge with the largest icf and edge with the largest icp ----
      // Row maxima may be stale upper bounds (their maximum decreased in an
      // earlier iteration and the row was not revisited): a stale row is
      // rescanned only when its bound ties the global maximum ("lazy deletion").
      long long vf, vp;
      unsigned int abf, abp;
      for (;;) {
        // the row maxima are packed edge keys (mkkey): the argmax with the
        // heuristic's tie rule is one 64-bit max over them, and its stale bit says
        // whether a stale bound ties the maximum
        unsigned long long gF = 0ull, gP = 0ull;
        for (int u = tid; u < n; u += kThreads) {
          const unsigned long long k1 = bestF[u], k2 = bestP[u];
          gF = k1 > gF ? k1 : gF;
          gP = k2 > gP ? k2 : gP;
        }
        AHK_F(0);  // (since the end of the previous iteration: the LDS scan)
        gF = wave_max_u64(gF);
        gP = wave_max_u64(gP);
        if (lane == 0) {
          s_rk[wv] = gF; s_rk[NWV + wv] = gP;
        }
        AHK_F(1);
        __syncthreads();
        AHK_F(2);
#pragma unroll
        for (int k = 0; k < NWV; k++) {
          gF = s_rk[k] > gF ? s_rk[k] : gF;
          gP = s_rk[NWV + k] > gP ? s_rk[NWV + k] : gP;
        }
        gF = readlane_u64(gF, 0);  // (every lane holds the same two keys: make them scalars)
        gP = readlane_u64(gP, 0);
        vf = key_val(gF);
        vp = key_val(gP);
        abf = ((unsigned)key_a(gF) << 16) | (unsigned)key_b(gF);
        abp = ((unsigned)key_a(gP) << 16) | (unsigned)key_b(gP);
        const bool need = ((gF | gP) & kStale) != 0ull;
        AHK_T(0);
        if (!need) break;  // uniform
        // collect the stale rows whose bound ties a maximum, rescan them exactly
        for (int u = tid; u < n; u += kThreads) {
          const unsigned long long k1 = bestF[u], k2 = bestP[u];
          if (((k1 & kStale) && (k1 >> kKeyValShift) == (gF >> kKeyValShift)) || ((k2 & kStale) && (k2 >> kKeyValShift) == (gP >> kKeyValShift)))
            dlist[atomicAdd(&s_nd, 1)] = (unsigned short)u;
        }
        __syncthreads();
        {
          rescan_list<IcT, NWV>(dlist, __builtin_amdgcn_readfirstlane(s_nd), wv, lane, n, nw, LV, IC, lwm, glog, bestF, bestP);
        }
        __syncthreads();
#ifdef AHK_TIMING
        nretry++; nresc += s_nd;
#endif
        AHK_T(1);
        if (tid == 0) s_nd = 0;  // next use is behind the reduction barrier above
      }
      if (vf < 0 || vp < 0) break;  // no live edge left (uniform)
      iters++;
      const bool merge = vf >= vp;
      const int eu = __builtin_amdgcn_readfirstlane(merge ? (int)(abf >> 16) : (int)(abp >> 16));
      const int ev = __builtin_amdgcn_readfirstlane(merge ? (int)(abf & 0xffff) : (int)(abp & 0xffff));
      const int cu = __builtin_amdgcn_readfirstlane((int)cl[eu]), cv = __builtin_amdgcn_readfirstlane((int)cl[ev]);

      // ---- fast path: separation of two single reads (9 of 10 iterations) ------
      // The only rewritten pair is (eu,ev); the only induced costs that can change
      // are those of the live edges (x,b), x in {eu,ev}, through the one triangle
      // (x,w,b) with w the other end -- and only where W[w][b] != 0.  Two waves per
      // row x (lane = column bit of a live word): every live edge of the row is
      // loaded, the candidates among them updated, and the row's exact new maximum
      // falls out of the same pass (the general path would queue both rows for a
      // rescan: their maximum is the edge that dies).  Two dependent HBM round
      // trips per iteration instead of about ten.
      AHK_F(3);
      if (!merge && !(((multiw[eu >> 5] >> (eu & 31)) | (multiw[ev >> 5] >> (ev & 31))) & 1u)) {
        constexpr int NSUB = NWV / 2;  // waves per row
        const int half = wv / NSUB, sub = wv % NSUB;
        const int x = half ? ev : eu, w = half ? eu : ev;
        unsigned long long lvn = 0ull, nzo = 0ull;
        if (lane < nw) {
          lvn = LV_AT((unsigned)(x * nw + lane));
          nzo = NZ_AT((unsigned)(w * nw + lane));
          AHK_L(0, &LV_AT((unsigned)(x * nw + lane)));
          AHK_L(1, &NZ_AT((unsigned)(w * nw + lane)));
        }
        const int ov = WM_AT((unsigned)(eu * n + ev));  // wave-uniform
#if AHK_LWM
        // ONE round trip: the words of row x that may hold a live edge are known from the LDS summary, so the
        // induced costs of those words and the third-edge weights W[w][.] of the same columns are requested
        // together with the two bitmap rows (every lane of a word asks: a row's live edges are spread over all
        // four lines of a word's induced costs anyway) and masked with the bitmaps once everything is here.
        unsigned long long todo = lwm_words((unsigned)__builtin_amdgcn_readfirstlane((int)lwm[x]), glog, nw);
        bool first = true;
#else
        if (lane == (w >> 6)) {
          lvn &= ~(1ull << (w & 63));
          if (sub == 0) LV_AT((unsigned)(x * nw + lane)) = lvn;
        }
        unsigned long long todo = __ballot(lvn != 0ull);
#endif
        unsigned long long kF = 0ull, kP = 0ull;
        AHK_F(4);
        // this wave takes the live words whose rank is sub modulo NSUB (four waves: even / odd rank), kFastBatch per trip
        while (todo != 0ull) {
          int lw[kFastBatch];
#pragma unroll
          for (int c = 0; c < kFastBatch; c++) {
#pragma unroll
            for (int q = 0; q < NSUB - 1; q++)
              if (q < sub && todo != 0ull) todo &= todo - 1ull;
            lw[c] = todo != 0ull ? __builtin_ctzll(todo) : -1;
            if (todo != 0ull) todo &= todo - 1ull;
#pragma unroll
            for (int q = 0; q < NSUB - 1; q++)
              if (q >= sub && todo != 0ull) todo &= todo - 1ull;
          }
          IcT icv[kFastBatch];
          int wb[kFastBatch];
          bool on[kFastBatch];
#if AHK_LWM
#pragma unroll
          for (int c = 0; c < kFastBatch; c++) {
            icv[c].f = 0;  // (defined on every path, see rescan_rows2)
            icv[c].p = 0;
            wb[c] = 0;
            if (lw[c] >= 0) {  // uniform
              const int b = (lw[c] << 6) + lane;
              const int bc = b < n ? b : n - 1;  // (lanes beyond the row stay inside it; their bits are clear)
              icv[c] = IC_AT(ic_index(x, bc, n));
              wb[c] = WM_AT((unsigned)(w * n + bc));
              AHK_L(2, &IC_AT(ic_index(x, bc, n)));
              AHK_L(3, &WM_AT((unsigned)(w * n + bc)));
            }
          }
          if (first) {  // uniform: the edge (x, w) itself dies
            first = false;
            if (lane == (w >> 6)) {
              lvn &= ~(1ull << (w & 63));
              if (sub == 0) {
                LV_AT((unsigned)(x * nw + lane)) = lvn;
                // (the summary bit of a word that has just become empty is cleared BEHIND the barrier below: the waves
                // of a row split its words by rank in the summary, and a wave that read the summary after the bit
                // had gone would take a word its partner also takes -- seen once in 10 000 chains as a different
                // iteration count between two runs)
                s_clr[half] = (glog == 0 && lvn == 0ull) ? lane + 1 : 0;
              }
            }
          }
#pragma unroll
          for (int c = 0; c < kFastBatch; c++) {
            on[c] = false;
            if (lw[c] >= 0) {
              const unsigned long long wd = readlane_u64(lvn, lw[c]);
              const unsigned long long cd = wd & readlane_u64(nzo, lw[c]);
              on[c] = ((wd >> lane) & 1ull) != 0ull;
              if (((cd >> lane) & 1ull) == 0ull) wb[c] = 0;
            }
          }
#else
#pragma unroll
          for (int c = 0; c < kFastBatch; c++) {
            on[c] = false;
            wb[c] = 0;
            icv[c].f = 0;  // (defined on every path, see rescan_rows2)
            icv[c].p = 0;
            if (lw[c] >= 0) {
              const unsigned long long wd = readlane_u64(lvn, lw[c]);
              const unsigned long long cd = wd & readlane_u64(nzo, lw[c]);
              if ((wd >> lane) & 1ull) {
                const int b = (lw[c] << 6) + lane;
                on[c] = true;
                icv[c] = IC_AT(ic_index(x, b, n));
                AHK_L(2, &IC_AT(ic_index(x, b, n)));
                if ((cd >> lane) & 1ull) { wb[c] = WM_AT((unsigned)(w * n + b)); AHK_L(3, &WM_AT((unsigned)(w * n + b))); }
              }
            }
          }
#endif
#pragma unroll
          for (int c = 0; c < kFastBatch; c++) {
            if (!on[c]) continue;
            const int b = (lw[c] << 6) + lane;
            IcT ic = icv[c];
            const long long df = (long long)icf_term(W_FORB, wb[c]) - (long long)icf_term(ov, wb[c]);
            const long long dp = (long long)icp_term(W_FORB, wb[c]) - (long long)icp_term(ov, wb[c]);
            if (df != 0 || dp != 0) {
              ic.f += df;
              ic.p += dp;
              IC_AT(ic_index(x, b, n)) = ic;
              IC_AT(ic_index(b, x, n)) = ic;
              AHK_L(4, &IC_AT(ic_index(b, x, n)));
            }
            const unsigned long long r1 = mkkey(ic.f, x, b), r2 = mkkey(ic.p, x, b);
            if (df != 0 || dp != 0) {
              fold_key(&bestF[b], r1);
              fold_key(&bestP[b], r2);
            }
            kF = r1 > kF ? r1 : kF;
            kP = r2 > kP ? r2 : kP;
          }
        }
#if AHK_LWM
        if (first && lane == (w >> 6)) {  // (a row whose summary was empty: only the dying edge's word is rewritten)
          lvn &= ~(1ull << (w & 63));
          if (sub == 0) {
            LV_AT((unsigned)(x * nw + lane)) = lvn;
            s_clr[half] = 0;
          }
        }
#endif
        AHK_F(5);
#if defined(AHK_PROBE_VALU) || defined(AHK_PROBE_SALU)
        // sensitivity probe (diagnostic builds only): N extra dependent vector / scalar instructions per wave
        // and single-read separation -- what does an instruction cost in this loop?  (tools/probe.sh)
        {
#ifdef AHK_PROBE_VALU
          unsigned int pv = (unsigned int)lane;
#pragma unroll
          for (int q = 0; q < AHK_PROBE_VALU; q++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(pv));
          if (pv == 0xffffffffu) kF = 0ull;  // (never: keeps the chain alive)
#endif
#ifdef AHK_PROBE_SALU
          unsigned int ps = (unsigned int)wv;
#pragma unroll
          for (int q = 0; q < AHK_PROBE_SALU; q++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(ps) : : "scc");
          if (ps == 0xffffffffu) kF = 0ull;
#endif
        }
#endif
        kF = wave_max_u64(kF);
        kP = wave_max_u64(kP);
        if (lane == 0) { s_rk[wv] = kF; s_rk[NWV + wv] = kP; }
        AHK_F(6);
        __syncthreads();
        AHK_F(7);
        if (tid < 2) {
          const int xr = tid ? ev : eu;
          unsigned long long a1 = 0ull, b1 = 0ull;  // (the row's waves: NSUB partial maxima each)
#pragma unroll
          for (int q = 0; q < NSUB; q++) {
            const unsigned long long a2 = s_rk[NSUB * tid + q], b2 = s_rk[NWV + NSUB * tid + q];
            a1 = a2 > a1 ? a2 : a1;
            b1 = b2 > b1 ? b2 : b1;
          }
          bestF[xr] = a1;
          bestP[xr] = b1;
#if AHK_LWM
          if (s_clr[tid] != 0) lwm_clear(lwm, xr, s_clr[tid] - 1);  // (half 0 = row eu, half 1 = row ev)
#endif
          WM_AT((unsigned)(xr * n + (tid ? eu : ev))) = W_FORB;  // every wave has read the old weight by now
        }
        __syncthreads();
        AHK_F(8);
#ifdef AHK_FTRACE
        ft_nfast++;
#endif
#ifdef AHK_TIMING
        AHK_T(11);
        nfast++;
        titer = (long long)clock64();
#endif
        continue;
      }
#ifdef AHK_FTRACE
      ft_nother++;
#endif

      // ---- staged row: when one side of the decision is a single read rs (96 % of the
      // iterations that get here: a read joining or leaving a clique), every edge (y,b) of the
      // other side's rows y has the one triangle (y,rs,b), whose two other edges W[rs][y] and
      // W[rs][b] both sit in row rs.  That row goes to LDS now; P3b then prices those edges
      // (thousands per iteration, 4 % of them change) without touching HBM.
      const int rs = !lds.stage_row ? -1
                                    : __builtin_amdgcn_readfirstlane(!((multiw[eu >> 5] >> (eu & 31)) & 1u) ? eu : (!((multiw[ev >> 5] >> (ev & 31)) & 1u) ? ev : -1));
      if (rs >= 0)
        for (int z = tid; z < n; z += kThreads) { wrow[z] = WM_AT((unsigned)(rs * n + z)); AHK_L(8, &WM_AT((unsigned)(rs * n + z))); }

      // ---- P2: roles, member lists and per-word role masks --------------------
      // (s_cnt was zeroed before the last barrier).  rmask[R][k] has bit c set iff
      // column 64k+c plays role R: the row passes below work on whole 64-column
      // words with these masks instead of looking at single columns.
      for (int z0 = 0; z0 < nw * 64; z0 += kThreads) {
        const int z = z0 + tid;
        int r = ROLE_NONE;
        if (z < n) {
          const int cz = cl[z];
          if (cz == cu) r = ROLE_A;
          else if (cz == cv) r = ROLE_B;
          else if (merge) {
            const bool fa = WM_AT((unsigned)(eu * n + z)) == W_FORB, fb = WM_AT((unsigned)(ev * n + z)) == W_FORB;
            AHK_L(7, &WM_AT((unsigned)(eu * n + z)));
            AHK_L(7, &WM_AT((unsigned)(ev * n + z)));
            if (fa && !fb) r = ROLE_FA;
            if (fb && !fa) r = ROLE_FB;
          }
          role[z] = (unsigned char)r;
        }
        const unsigned long long m1 = __ballot(r == ROLE_A), m2 = __ballot(r == ROLE_B);
        const unsigned long long m3 = __ballot(r == ROLE_FA), m4 = __ballot(r == ROLE_FB);
        const int k = (z0 >> 6) + wv;
        if (lane == 0 && k < nw) {
          rmask[0 * RW + k] = m1;
          rmask[1 * RW + k] = m2;
          rmask[2 * RW + k] = m3;
          rmask[3 * RW + k] = m4;
        }
      }
      __syncthreads();
      // member lists in ASCENDING read order, straight from the masks: wave R scans the word
      // popcounts of role R, then every member finds its slot by a popcount below its bit
      // (consecutive list entries are neighbouring columns: the row accesses W[x][w] over a
      // list touch few lines, and the heavy-edge pass relies on the order)
      if (wv < 4) {  // (four roles: waves 0..3)
        const unsigned long long mw = lane < nw ? rmask[wv * RW + lane] : 0ull;
        const int pc = __popcll(mw);
        const int incl = wave_scan_i(pc);
        s_pref[wv * 64 + lane] = (unsigned short)(incl - pc);
        if (lane == 63) s_cnt[wv] = incl;
      }
      __syncthreads();
      const int nA = __builtin_amdgcn_readfirstlane(s_cnt[0]), nB = __builtin_amdgcn_readfirstlane(s_cnt[1]);
      const int nFa = __builtin_amdgcn_readfirstlane(s_cnt[2]), nFb = __builtin_amdgcn_readfirstlane(s_cnt[3]);
      const int nT = nA + nB + nFa + nFb;
      // the four lists, one behind the other in the pool
      unsigned short* const listA = lpool;
      unsigned short* const listB = listA + ((nA + 7) & ~7);
      unsigned short* const listFa = listB + ((nB + 7) & ~7);
      unsigned short* const listFb = listFa + ((nFa + 7) & ~7);
      for (int z = tid; z < n; z += kThreads) {
        const int r = role[z];
        if (r == ROLE_NONE) continue;
        const int k = z >> 6;
        const int slot = (int)s_pref[(r - 1) * 64 + k] + __popcll(rmask[(r - 1) * RW + k] & ((1ull << (z & 63)) - 1ull));
        unsigned short* Lz = r == ROLE_A ? listA : r == ROLE_B ? listB : r == ROLE_FA ? listFa : listFb;
        Lz[slot] = (unsigned short)z;
      }
      __syncthreads();
      // Edges priced from the staged row (the far end b is not rewritten itself, the row's only member
      // list is {rs}): roles whose rows qualify
      unsigned int ldsrole = 0u;
      if (rs >= 0) {
        if (rs == eu) ldsrole = (nFa == 0 ? 1u << ROLE_B : 0u) | (1u << ROLE_FB);
        else ldsrole = (nFb == 0 ? 1u << ROLE_A : 0u) | (1u << ROLE_FA);
      }
#ifdef AHK_TIMING
      heavy_it = (nA + nFa > 64 && nB + nFb <= 1) || (nB + nFb > 64 && nA + nFa <= 1);
      nheavy += heavy_it ? 1 : 0;
#endif
      // ---- P2b: per role, the OR of the nz-rows of its members (wave R <-> role R):
      // "some member w of the role has W[w][b] != 0".  One pass per iteration,
      // shared by every row of P3a; lists longer than 32 are not filtered.
      if (wv < 4) {
        const unsigned short* Lr = wv == 0 ? listA : wv == 1 ? listB : wv == 2 ? listFa : listFb;
        const int nr = wv == 0 ? nA : wv == 1 ? nB : wv == 2 ? nFa : nFb;
        unsigned long long acc = 0ull;
        if (nr > 32) acc = ~0ull;
        else if (lane < nw) {
          for (int w0 = 0; w0 < nr; w0 += 8) {
            unsigned long long t[8];
#pragma unroll
            for (int c = 0; c < 8; c++)
            {
              t[c] = (w0 + c < nr) ? NZ_AT((unsigned)(__builtin_amdgcn_readfirstlane((int)Lr[w0 + c < nr ? w0 + c : 0]) * nw + lane)) : 0ull;
              if (w0 + c < nr) AHK_L(9, &NZ_AT((unsigned)(__builtin_amdgcn_readfirstlane((int)Lr[w0 + c < nr ? w0 + c : 0]) * nw + lane)));
            }
#pragma unroll
            for (int c = 0; c < 8; c++) acc |= t[c];
          }
        }
        if (lane < RW) rolenz[wv * RW + lane] = acc;
      }
      __syncthreads();
      AHK_T(2);
#ifdef AHK_TIMING
      sumT += nT;
#endif

      // ---- P3a: candidate edges (x,b), x in T ----------------------------------
      // One wave per row x, ONE LANE PER 64-COLUMN WORD.  With D = columns whose
      // edge to x is rewritten now:   live' = live & ~D   (stored),
      // nz' = nz | D (stored; nz = "W[x][.] != 0", the third-edge filter),
      // candidates = live' & (nz-rows of the rewritten columns, OR-ed) -- an edge
      // (x,b) can only change if some rewritten neighbour w of x has W[w][b] != 0.
      // Edges inside T are kept unfiltered and listed once, from their lower end.
      // A row whose stored maximum sits on a dying edge is queued for an exact
      // rescan in P4; every other row maximum is maintained lazily.
      {
        unsigned int* items_lds = (unsigned int*)rowbuf;
        const int cap = lds.items_lds ? 2 * nmax : 0;
        // Lane layout: a wave works on RP = 64/nwp rows at once, lane = (row slot,
        // 64-column word), nwp = nw rounded up to a power of two (>= 8).  With
        // 10 words per row that is 4 rows per wave pass, 16 per workgroup pass.
        const int slot = lane >> nwp_log, kw = lane & (nwp - 1);
        const bool wordok = kw < nw;
        // this lane's word of every role mask / neighbour mask: read from LDS once
        unsigned long long mA = 0ull, mB = 0ull, mFa = 0ull, mFb = 0ull, oA = 0ull, oB = 0ull, oFa = 0ull, oFb = 0ull;
        if (wordok) {
          mA = rmask[kw]; mB = rmask[RW + kw]; mFa = rmask[2 * RW + kw]; mFb = rmask[3 * RW + kw];
          oA = rolenz[kw]; oB = rolenz[RW + kw]; oFa = rolenz[2 * RW + kw]; oFb = rolenz[3 * RW + kw];
        }
        const unsigned long long Tm = mA | mB | mFa | mFb;
        const unsigned long long segmask = (nwp == 64 ? ~0ull : ((1ull << nwp) - 1ull)) << (slot << nwp_log);
        auto row_of = [&](int xi) -> int {
          if (xi >= nT) return -1;
          if (xi < nA) return listA[xi];
          if (xi < nA + nB) return listB[xi - nA];
          if (xi < nA + nB + nFa) return listFa[xi - nA - nB];
          return listFb[xi - nA - nB - nFa];
        };
        // the bitmap words of pass k+1 are requested before pass k is processed: with
        // hundreds of rewritten rows (a large clique) every pass would otherwise wait
        // for its own HBM round trip
        int x_nx = row_of(wv * RP + slot);  // (a workgroup pass handles NWV * RP rows)
        unsigned long long lv_nx = 0ull, nz_nx = 0ull;
        if (x_nx >= 0 && wordok) {
          lv_nx = LV_AT((unsigned)(x_nx * nw + kw));
          nz_nx = NZ_AT((unsigned)(x_nx * nw + kw));
        }
        for (int xi0 = wv * RP; xi0 < nT; xi0 += NWV * RP) {
          const int xi = xi0 + slot;
          const int x = x_nx;
          const unsigned long long lvw = lv_nx, nzw = nz_nx;
          x_nx = row_of(xi + NWV * RP);
          lv_nx = 0ull;
          nz_nx = 0ull;
          if (x_nx >= 0 && wordok) {
            lv_nx = LV_AT((unsigned)(x_nx * nw + kw));
            nz_nx = NZ_AT((unsigned)(x_nx * nw + kw));
            AHK_L(10, &LV_AT((unsigned)(x_nx * nw + kw)));
            AHK_L(11, &NZ_AT((unsigned)(x_nx * nw + kw)));
          }
          // the role of x follows from which list it came from
          const int rx = xi < nA ? ROLE_A : xi < nA + nB ? ROLE_B : xi < nA + nB + nFa ? ROLE_FA : ROLE_FB;
          const unsigned long long D = rx == ROLE_A ? (mB | mFb) : rx == ROLE_B ? (mA | mFa) : rx == ROLE_FA ? mB : mA;
          // does a stored row maximum sit on a dying edge?
          bool hit = false;
          if (x >= 0) {
            const unsigned long long kF = bestF[x], kP = bestP[x];
            const int cF = key_other(kF, x), cP = key_other(kP, x);
            hit = (kF && kw == (cF >> 6) && ((D >> (cF & 63)) & 1ull)) || (kP && kw == (cP >> 6) && ((D >> (cP & 63)) & 1ull));
          }
          const unsigned long long hm = __ballot(hit) & segmask;
          if (hm != 0ull && kw == 0 && x >= 0) dlist[atomicAdd(&s_nd, 1)] = (unsigned short)x;
#if defined(AHK_TIMING) && AHK_TIMING > 1
          AHK_T(6);
#endif
          unsigned long long cand = 0ull;
          if (x >= 0 && wordok) {
            const unsigned long long lvn = lvw & ~D;
            const unsigned long long nzn = nzw | D;
            if (lvn != lvw) {
              LV_AT((unsigned)(x * nw + kw)) = lvn;
              if (AHK_LWM && glog == 0 && lvn == 0ull) lwm_clear(lwm, x, kw);
            }
            if (nzn != nzw) NZ_AT((unsigned)(x * nw + kw)) = nzn;
            const int xw = x >> 6;
            const unsigned long long low = kw < xw ? ~0ull : (kw == xw ? ((1ull << (x & 63)) - 1ull) : 0ull);
            // columns with a non-zero third edge to some rewritten neighbour of x
            const unsigned long long nzc = rx == ROLE_A ? (oB | oFb) : rx == ROLE_B ? (oA | oFa) : rx == ROLE_FA ? oB : oA;
            cand = lvn & (nzc | Tm) & ~(Tm & low);
          }
          // Staged-row rows: their candidates are priced right here, from the row in LDS, and only
          // the 8 % that change are listed at all (the list, its overflow to HBM and the classifying
          // pass shrink accordingly).  Sparse words by their own lane, dense words by the whole wave.
          if (ldsrole != 0u) {  // uniform
            const bool elig = x >= 0 && wordok && ((ldsrole >> rx) & 1u);
            const unsigned long long cs = elig ? (cand & ~Tm) : 0ull;
            const int pcs = __popcll(cs);
            const bool perm = merge && (rx == ROLE_A || rx == ROLE_B);
            unsigned long long surv = 0ull;
            // (a pair that becomes permanent from a weight <= 0, or forbidden from a weight > 0, changes the
            // induced costs of EVERY candidate -- all have a non-zero third edge: no test per column needed)
            const int ovx = pcs > 0 ? wrow[x] : 0;
            const bool all = pcs > 0 && (perm ? ovx <= 0 : ovx > 0);
            if (all) surv = cs;
            if (pcs > 0 && !all && pcs <= kLaneExpandMax) {
              unsigned long long mbits = cs;
              while (mbits != 0ull) {
                const int bit = __builtin_ctzll(mbits);
                mbits &= mbits - 1ull;
                if (staged_changes(ovx, wrow[(kw << 6) + bit], perm)) surv |= 1ull << bit;
              }
            }
            unsigned long long todoS = __ballot(pcs > kLaneExpandMax && !all);
            while (todoS != 0ull) {
              const int l = __builtin_ctzll(todoS);
              todoS &= todoS - 1ull;
              const unsigned long long w = readlane_u64(cs, l);
              const int xl = __builtin_amdgcn_readlane(x, l);
              const bool perml = __builtin_amdgcn_readlane((int)perm, l) != 0;
              const int kwl = l & (nwp - 1);
              bool ch = false;
              if ((w >> lane) & 1ull) ch = staged_changes(wrow[xl], wrow[(kwl << 6) + lane], perml);
              const unsigned long long bal = __ballot(ch);
              if (lane == l) surv = bal;
            }
            if (elig) cand = surv | (cand & Tm);
          }
          // Item slots by an exclusive scan of the popcounts (order: lane = (row, word),
          // then bit -- the same as a word-by-word walk).  Sparse words (the rule late
          // in a chain: 2-3 live edges per word) are expanded by their own lane, all
          // lanes in parallel; dense words one at a time by the whole wave, lane = bit.
          if (__ballot(cand != 0ull) != 0ull) {
            const int pc = __popcll(cand);
            const int incl = wave_scan_i(pc);
            int base = 0;
            if (lane == 63) base = atomicAdd(&s_items, incl);
            int idx = __builtin_amdgcn_readlane(base, 63) + incl - pc;
            unsigned long long todo = __ballot(pc > kLaneExpandMax);
            if (pc <= kLaneExpandMax) {
              unsigned long long mbits = cand;
              while (mbits != 0ull) {
                const int b = (kw << 6) + __builtin_ctzll(mbits);
                mbits &= mbits - 1ull;
                const unsigned int it = ((unsigned)rx << 28) | ((unsigned)role[b] << 24) | ((unsigned)x << 12) | (unsigned)b;
                if (idx < cap) items_lds[idx] = it;
                else { items_glb[idx - cap] = it; AHK_L(24, &items_glb[idx - cap]); }
                idx++;
              }
            }
            while (todo != 0ull) {
              const int l = __builtin_ctzll(todo);
              todo &= todo - 1ull;
              const unsigned long long w = readlane_u64(cand, l);
              const int xl = __builtin_amdgcn_readlane(x, l);
              const int rxl = __builtin_amdgcn_readlane(rx, l);
              const int basel = __builtin_amdgcn_readlane(idx, l);
              const int kwl = l & (nwp - 1);
              if ((w >> lane) & 1ull) {
                const int b = (kwl << 6) + lane;
                const int id2 = basel + mask_rank(w);
                const unsigned int it = ((unsigned)rxl << 28) | ((unsigned)role[b] << 24) | ((unsigned)xl << 12) | (unsigned)b;
                if (id2 < cap) items_lds[id2] = it;
                else { items_glb[id2 - cap] = it; AHK_L(24, &items_glb[id2 - cap]); }
              }
            }
          }
#if defined(AHK_TIMING) && AHK_TIMING > 1
          AHK_T(7);
#endif
        }
      }
      __syncthreads();
      AHK_T(5);

      // ---- P3b: one lane per candidate edge ---------------------------------------
      {
        const unsigned int* items_lds = (const unsigned int*)rowbuf;
        const int cap = lds.items_lds ? 2 * nmax : 0;
        const int M = __builtin_amdgcn_readfirstlane(s_items);
#ifdef AHK_TIMING
        sumM += M;
        s_items_last = M;
#endif
#if AHK_HEAVY
        // Heavy edges.  The member lists an edge (x,b) must walk depend only on the role of x:
        // A: B u Fb, B: A u Fa, FA: B, FB: A.  When one clique is large and the other a single
        // read -- the shape of the late iterations: 7 % of them, two thirds of the loop's time
        // -- the ~100 edges of the single read's row each walk 100-300 members, four per
        // dependent load batch, on one or two waves, while every other lane has one triangle
        // per edge and then waits at the barrier.  Such edges go to a list of their own (hl)
        // and are handled after the plain loop.
        unsigned int heavy = 0u;  // bit per role of x
        {
          const int lenA = nB + nFb, lenB = nA + nFa;
          if (lenA >= kHeavyMin) heavy |= 1u << ROLE_A;
          if (lenB >= kHeavyMin) heavy |= 1u << ROLE_B;
          if (nB >= kHeavyMin) heavy |= 1u << ROLE_FA;
          if (nA >= kHeavyMin) heavy |= 1u << ROLE_FB;
        }
        unsigned int* hl = items_glb + (size_t)(ws.nmax * ws.nmax / 2);  // (overflow items use the first half at most)
        int nh = 0;
#endif
#if defined(AHK_TIMING) && AHK_TIMING == 1
        AHK_T(8);  // (level 1: "decode" = heavy-list building, "tri" = the plain loop, P3b = the heavy pass)
#endif
        // any other edge: its triangles from HBM, applied at once
        auto plain_item = [&](const unsigned int it, const int rx, const int rb) {
          const int x = (int)((it >> 12) & 0xfffu), b = (int)(it & 0xfffu);
          const unsigned ici = ic_index(x, b, n);
          // Only ~4 % of the candidates change.  The kernel moves HBM lines at the rate the
          // memory system can deliver them for this access pattern, so the entry (one 128-B
          // line for 16 useful bytes) is loaded only once the delta is known to be non-zero.
#if !AHK_IC_LATE
          IcT ic = IC_AT(ici);
#endif
          long long df = 0, dp = 0;
#if defined(AHK_TIMING) && AHK_TIMING > 1
          AHK_T(8);
#endif
          {
            const unsigned short* L1 = (rx == ROLE_A || rx == ROLE_FA) ? listB : listA;
            const int n1 = (rx == ROLE_A || rx == ROLE_FA) ? nB : nA;
            const unsigned short* L2 = rx == ROLE_A ? listFb : listFa;
            const int n2 = rx == ROLE_A ? nFb : (rx == ROLE_B ? nFa : 0);
#if AHK_P3B_SPECIAL
            if (rb == ROLE_NONE) {
              // The common case: b is not itself rewritten, so the third edge (w,b) keeps its
              // weight, and within one member list the new weight of (x,w) is one constant
              // (tri_seg).
              if (merge && (rx == ROLE_A || rx == ROLE_B)) tri_seg<W_PERM>(Wm, n, x, b, L1, n1, df, dp);
              else tri_seg<W_FORB>(Wm, n, x, b, L1, n1, df, dp);
              if (n2 > 0) tri_seg<W_FORB>(Wm, n, x, b, L2, n2, df, dp);
            } else
#endif
            {
            const int nc = n1 + n2;
            for (int w0 = 0; w0 < nc; w0 += kP3bChunk) {
              // the sixteen loads of a chunk of eight triangles are independent
              int ws4[kP3bChunk], ov4[kP3bChunk], wb4[kP3bChunk];
#pragma unroll
              for (int c = 0; c < kP3bChunk; c++) {
                const int wi = w0 + c;
                const int w = wi < n1 ? L1[wi] : (wi < nc ? L2[wi - n1] : -1);
                ws4[c] = w;
                ov4[c] = w >= 0 ? WM_AT((unsigned)(x * n + w)) : 0;
                wb4[c] = w >= 0 ? WM_AT((unsigned)(w * n + b)) : 0;
                if (w >= 0) { AHK_L(18, &WM_AT((unsigned)(x * n + w))); AHK_L(18, &WM_AT((unsigned)(w * n + b))); }
              }
#pragma unroll
              for (int c = 0; c < kP3bChunk; c++) {
                const int w = ws4[c];
                if (w < 0) continue;
                // the role of w follows from the list it sits in
                const int rw = (w0 + c < n1) ? ((rx == ROLE_A || rx == ROLE_FA) ? ROLE_B : ROLE_A) : (rx == ROLE_A ? ROLE_FB : ROLE_FA);
                const int Nv = new_w(rx, rw, merge);
                const int wbo = wb4[c];
                const int wbn = in_s(rw, rb, merge) ? new_w(rw, rb, merge) : wbo;
                df += icf_term(Nv, wbn) - icf_term(ov4[c], wbo);
                dp += icp_term(Nv, wbn) - icp_term(ov4[c], wbo);
              }
            }
            }
          }
          if (rb != ROLE_NONE) {
            // triangles whose changed edge is (b,w) only: w in C(b) \ C(x)
            const unsigned short* M1 = (rb == ROLE_A || rb == ROLE_FA) ? listB : listA;
            const int m1 = (rb == ROLE_A || rb == ROLE_FA) ? nB : nA;
            const unsigned short* M2 = rb == ROLE_A ? listFb : listFa;
            const int m2 = rb == ROLE_A ? nFb : (rb == ROLE_B ? nFa : 0);
            for (int pass = 0; pass < 2; pass++) {
              const unsigned short* L = pass == 0 ? M1 : M2;
              const int nl = pass == 0 ? m1 : m2;
              for (int wi = 0; wi < nl; wi++) {
                const int w = L[wi];
                const int rw = role[w];
                if (w == x || in_s(rx, rw, merge)) continue;  // (x,w) changes too: counted above
                const int Nv = new_w(rb, rw, merge);
                const int ov = WM_AT((unsigned)(w * n + b));
                const int xw = WM_AT((unsigned)(x * n + w));
                AHK_L(18, &WM_AT((unsigned)(w * n + b)));
                AHK_L(18, &WM_AT((unsigned)(x * n + w)));
                df += icf_term(Nv, xw) - icf_term(ov, xw);
                dp += icp_term(Nv, xw) - icp_term(ov, xw);
              }
            }
          }
#if defined(AHK_TIMING) && AHK_TIMING > 1
          AHK_T(9);
#endif
          if (df == 0 && dp == 0) return;
#if AHK_IC_LATE
          IcT ic = IC_AT(ici);
#endif
          AHK_L(16, &IC_AT(ici));
          AHK_L(17, &IC_AT(ic_index(b, x, n)));
          ic.f += df;
          ic.p += dp;
          IC_AT(ici) = ic;
          IC_AT(ic_index(b, x, n)) = ic;
          // fold the new value into both row maxima; a maximum that sat on this
          // edge and went down is now only an upper bound -> mark the row stale
          {
            const unsigned long long c1 = mkkey(ic.f, x, b), c2 = mkkey(ic.p, x, b);
            fold_key(&bestF[x], c1);
            fold_key(&bestP[x], c2);
            fold_key(&bestF[b], c1);
            fold_key(&bestP[b], c2);
          }
#if defined(AHK_TIMING) && AHK_TIMING > 1
          AHK_T(10);
#endif
        };
        // Staged-row edges are priced from LDS; the few that change are parked in a per-wave list
        // (rmask/rolenz are dead after P3a: 128 entries per wave) and applied 64 at a time, so the
        // induced-cost round trip is paid once per 64 changes instead of once per loop trip.
        unsigned int* pend = (unsigned int*)rmask + wv * 128;
        int npend = 0;
        auto staged_delta = [&](const unsigned int it, long long& df, long long& dp) {
          const int rx = (int)(it >> 28);
          const int ov = wrow[(it >> 12) & 0xfffu], wb = wrow[it & 0xfffu];
          if (merge && (rx == ROLE_A || rx == ROLE_B)) {
            df = (long long)icf_term(W_PERM, wb) - (long long)icf_term(ov, wb);
            dp = (long long)icp_term(W_PERM, wb) - (long long)icp_term(ov, wb);
          } else {
            df = (long long)icf_term(W_FORB, wb) - (long long)icf_term(ov, wb);
            dp = (long long)icp_term(W_FORB, wb) - (long long)icp_term(ov, wb);
          }
        };
        auto apply_pending = [&](const int cnt) {
          const bool act = lane < cnt;
          int x = 0;
          unsigned long long c1 = 0ull, c2 = 0ull;
          if (act) {
            const unsigned int it = pend[lane];
            x = (int)((it >> 12) & 0xfffu);
            const int b = (int)(it & 0xfffu);
            long long df, dp;
            staged_delta(it, df, dp);
            const unsigned ici = ic_index(x, b, n);
            IcT ic = IC_AT(ici);
            AHK_L(19, &IC_AT(ici));
            AHK_L(20, &IC_AT(ic_index(b, x, n)));
            ic.f += df;
            ic.p += dp;
            IC_AT(ici) = ic;
            IC_AT(ic_index(b, x, n)) = ic;
            c1 = mkkey(ic.f, x, b);
            c2 = mkkey(ic.p, x, b);
            fold_key(&bestF[b], c1);
            fold_key(&bestP[b], c2);
          }
          fold_rows_grouped(act, x, c1, c2, bestF, bestP, lane);  // (consecutive entries are edges of one row)
        };
        // Pass A (only when some role is staged or heavy): every item is classified -- heavy edges
        // to hl, staged-row edges priced on the spot, anything else to a list of its own (slow,
        // growing down from the end of hl's region: each item lands in one list at most).  A trip
        // is a few dozen instructions, so the items that overflowed to HBM are requested four
        // trips ahead.  Pass B prices what is left, one lane per edge with its triangles from HBM:
        // the whole item list when nothing is staged or heavy, else the slow list, plus hl when it
        // turns out long enough to keep every lane busy that way.
        const bool classified = ldsrole != 0u || heavy != 0u;  // uniform
        unsigned int* slow_end = items_glb + (size_t)ws.nmax * ws.nmax;  // slow item k at slow_end[-1 - k]
        int nslow = 0;
        if (classified) {
          auto classify = [&](const unsigned int it, const bool valid) {
            bool chg = false, hv = false, sl = false;
            if (valid) {
              const int rx = (int)(it >> 28), rb = (int)((it >> 24) & 7u);
              hv = ((heavy >> rx) & 1u) && rb == ROLE_NONE;
              if (!hv) {
                if (((ldsrole >> rx) & 1u) && rb == ROLE_NONE) {
                  chg = true;  // (P3a lists a staged-row edge only if its induced costs change)
                } else {
                  sl = true;
                }
              }
            }
            const unsigned long long mk = __ballot(hv), ms = __ballot(sl), cm = __ballot(chg);
            if (mk != 0ull) {
              int base = 0;
              if (lane == 0) base = atomicAdd(&s_nh, __popcll(mk));
              base = __builtin_amdgcn_readlane(base, 0);
              if (hv) hl[base + mask_rank(mk)] = it;
            }
            if (ms != 0ull) {
              int base = 0;
              if (lane == 0) base = atomicAdd(&s_nslow, __popcll(ms));
              base = __builtin_amdgcn_readlane(base, 0);
              if (sl) slow_end[-1 - (base + mask_rank(ms))] = it;
            }
            if (cm != 0ull) {  // wave-uniform
#ifdef AHK_TIMING
              if (lane == 0) atomicAdd(&s_chg, __popcll(cm));
#endif
              if (chg) pend[npend + mask_rank(cm)] = it;
              npend += __popcll(cm);
              if (npend >= 64) {
#if defined(AHK_TIMING) && AHK_TIMING == 1
                const long long _f0 = (long long)clock64();
#endif
                apply_pending(64);
#if defined(AHK_TIMING) && AHK_TIMING == 1
                if (tid == 0) { pl_t[1] += (long long)clock64() - _f0; pl_t[5]++; }
#endif
                unsigned int mv = 0u;
                if (lane < npend - 64) mv = pend[64 + lane];
                if (lane < npend - 64) pend[lane] = mv;
                npend -= 64;
              }
            }
          };
          auto fetch = [&](const int i) -> unsigned int { return i < M ? (i < cap ? items_lds[i] : items_glb[i - cap]) : 0u; };
          unsigned int q[4];
#pragma unroll
          for (int u = 0; u < 4; u++) q[u] = fetch(tid + u * kThreads);
          for (int i0 = 0; i0 < M; i0 += 4 * kThreads) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int ib = i0 + u * kThreads;
              if (ib < M) {  // uniform
                const unsigned int it = q[u];
                q[u] = fetch(ib + tid + 4 * kThreads);
                classify(it, ib + tid < M);
              }
            }
          }
#if defined(AHK_TIMING) && AHK_TIMING == 1
          const long long _f1 = (long long)clock64();
          if (tid == 0) pl_t[0] += _f1 - tlast;
#endif
          if (npend > 0) apply_pending(npend);
          npend = 0;
#if defined(AHK_TIMING) && AHK_TIMING == 1
          const long long _f2 = (long long)clock64();
          if (tid == 0) pl_t[2] += _f2 - _f1;
#endif
          __syncthreads();  // hl and slow complete; (the heavy pass re-uses the words of the per-wave lists)
#if defined(AHK_TIMING) && AHK_TIMING == 1
          if (tid == 0) { pl_t[3] += (long long)clock64() - _f2; pl_t[6]++; tlast = (long long)clock64(); }
#endif
          nh = __builtin_amdgcn_readfirstlane(s_nh);
          nslow = __builtin_amdgcn_readfirstlane(s_nslow);
        }
        {
          // pass B: segment 0 = the item list or the slow list, segment 1 = hl when one lane per edge is the better deal
          const int cnt0 = classified ? nslow : M;
          const int cnt1 = (classified && nh > kHeavyMaxItems) ? nh : 0;
          if (cnt1 > 0) nh = 0;
          auto fetchB = [&](const int i) -> unsigned int {
            if (i >= cnt0) return hl[i - cnt0];
            if (classified) return slow_end[-1 - i];
            return i < cap ? items_lds[i] : items_glb[i - cap];
          };
          unsigned int it_nx = 0u;
          if (tid < cnt0 + cnt1) it_nx = fetchB(tid);
          for (int i = tid; i < cnt0 + cnt1; i += kThreads) {
            const unsigned int it = it_nx;
            if (i + kThreads < cnt0 + cnt1) it_nx = fetchB(i + kThreads);
            plain_item(it, (int)(it >> 28), (int)((it >> 24) & 7u));
          }
        }
#if defined(AHK_TIMING) && AHK_TIMING == 1
        __syncthreads();
        {
          // plain-loop cycles by shape: one side a single read and the other side's rows walk no second list / a second list / neither side single
          const int bigF = nA == 1 ? nFa : nFb;
          const int cat = (nA == 1 || nB == 1) ? (merge ? (bigF == 0 ? 1 : 2) : 0) : 3;
          if (tid == 0) {
            const long long _t = (long long)clock64();
            pcat_c[cat] += _t - tlast; pcat_n[cat]++; pcat_m[cat] += M; pcat_f[cat] += bigF;
          }
        }
        AHK_T(9);
#endif
#if AHK_HEAVY == 1
        // Heavy edges in chunks: the member lists are cut into chunks of kHeavyChunk that are
        // dealt out over all 256 lanes (consecutive lanes = consecutive edges of a row, same
        // chunk: the coalescing of the plain loop); partial sums are exact integers, added
        // into per-edge LDS accumulators (kHeavyCap edges per round), then applied one lane
        // per edge.  75 dependent load batches of one lane become ~26 of every lane.
        if (nh > 0) {  // uniform
          unsigned long long* hacc = rmask;  // [2 * kHeavyCap]: rmask and rolenz are dead after P3a
          for (int r0 = 0; r0 < nh; r0 += kHeavyCap) {
            const int nr = nh - r0 < kHeavyCap ? nh - r0 : kHeavyCap;
            for (int k = tid; k < 2 * nr; k += kThreads) hacc[k] = 0ull;
            __syncthreads();
            // chunks per edge: the largest count over the heavy roles spans the grid
            int chmax = 0;
            {
              const int cA = (nB + kHeavyChunk - 1) / kHeavyChunk + (nFb + kHeavyChunk - 1) / kHeavyChunk;
              const int cB = (nA + kHeavyChunk - 1) / kHeavyChunk + (nFa + kHeavyChunk - 1) / kHeavyChunk;
              const int cFa = (nB + kHeavyChunk - 1) / kHeavyChunk, cFb = (nA + kHeavyChunk - 1) / kHeavyChunk;
              if ((heavy >> ROLE_A) & 1u) chmax = cA;
              if (((heavy >> ROLE_B) & 1u) && cB > chmax) chmax = cB;
              if (((heavy >> ROLE_FA) & 1u) && cFa > chmax) chmax = cFa;
              if (((heavy >> ROLE_FB) & 1u) && cFb > chmax) chmax = cFb;
            }
            const int V = nr * chmax;
            for (int v = tid; v < V; v += kThreads) {
              const int c = v / nr, hh = v - c * nr;
              const unsigned int it = hl[r0 + hh];
              const int rx = (int)(it >> 28);
              const int x = (int)((it >> 12) & 0xfffu), b = (int)(it & 0xfffu);
              const unsigned short* L1 = (rx == ROLE_A || rx == ROLE_FA) ? listB : listA;
              const int n1 = (rx == ROLE_A || rx == ROLE_FA) ? nB : nA;
              const unsigned short* L2 = rx == ROLE_A ? listFb : listFa;
              const int n2 = rx == ROLE_A ? nFb : (rx == ROLE_B ? nFa : 0);
              const int ch1 = (n1 + kHeavyChunk - 1) / kHeavyChunk, ch2 = (n2 + kHeavyChunk - 1) / kHeavyChunk;
              long long df = 0, dp = 0;
              if (c < ch1) {
                const int o = c * kHeavyChunk;
                const int cnt = n1 - o < kHeavyChunk ? n1 - o : kHeavyChunk;
                if (merge && (rx == ROLE_A || rx == ROLE_B)) tri_seg_h<W_PERM>(Wm, n, x, b, L1 + o, cnt, df, dp);
                else tri_seg_h<W_FORB>(Wm, n, x, b, L1 + o, cnt, df, dp);
              } else if (c < ch1 + ch2) {
                const int o = (c - ch1) * kHeavyChunk;
                const int cnt = n2 - o < kHeavyChunk ? n2 - o : kHeavyChunk;
                tri_seg_h<W_FORB>(Wm, n, x, b, L2 + o, cnt, df, dp);
              }
              if (df != 0) atomicAdd(&hacc[2 * hh], (unsigned long long)df);
              if (dp != 0) atomicAdd(&hacc[2 * hh + 1], (unsigned long long)dp);
            }
            __syncthreads();
            for (int hh0 = 0; hh0 < nr; hh0 += kThreads) {
              const int hh = hh0 + tid;
              long long df = 0, dp = 0;
              if (hh < nr) {
                df = (long long)hacc[2 * hh];
                dp = (long long)hacc[2 * hh + 1];
              }
              const bool act = df != 0 || dp != 0;
              int x = 0;
              unsigned long long c1 = 0ull, c2 = 0ull;
              if (act) {
                const unsigned int it = hl[r0 + hh];
                x = (int)((it >> 12) & 0xfffu);
                const int b = (int)(it & 0xfffu);
                const unsigned ici = ic_index(x, b, n);
                IcT ic = IC_AT(ici);
                AHK_L(21, &IC_AT(ici));
                AHK_L(22, &IC_AT(ic_index(b, x, n)));
                ic.f += df;
                ic.p += dp;
                IC_AT(ici) = ic;
                IC_AT(ic_index(b, x, n)) = ic;
#ifdef AHK_TIMING
                atomicAdd(&s_chg, 1);
#endif
                c1 = mkkey(ic.f, x, b);
                c2 = mkkey(ic.p, x, b);
                fold_key(&bestF[b], c1);
                fold_key(&bestP[b], c2);
              }
              fold_rows_grouped(act, x, c1, c2, bestF, bestP, lane);  // (the heavy edges of a row sit side by side)
            }
            __syncthreads();
          }
        }
#elif AHK_HEAVY == 2
        // One wave per heavy edge (x,b); lanes = members w of x's opposite lists, which P2
        // keeps in ascending order.  Only members with a non-zero third edge W[w][b] can
        // contribute (a zero makes both terms vanish): row b of the NZ bitmap says which --
        // typically a third of a clique, one contiguous stretch of the sorted list.  W[b][w]
        // (= W[w][b], W is symmetric) and W[x][w] are row accesses with consecutive lanes on
        // neighbouring columns: a few lines per edge instead of one row segment per member.
        // Sums are exact integers: a 64-bit DPP reduction, applied by lane 0.  (The bitmap row
        // of the wave's next edge is requested before the current edge is worked on.)
        unsigned long long nz_nx = 0ull;
        if (wv < nh && lane < nw) nz_nx = NZ_AT((unsigned)((int)(hl[wv] & 0xfffu) * nw + lane));
        for (int h = wv; h < nh; h += 4) {
          const unsigned int it = hl[h];
          const int rx = (int)(it >> 28);
          const int x = (int)((it >> 12) & 0xfffu), b = (int)(it & 0xfffu);
          const unsigned long long nzw = nz_nx;
          nz_nx = 0ull;
          if (h + 4 < nh && lane < nw) nz_nx = NZ_AT((unsigned)((int)(hl[h + 4] & 0xfffu) * nw + lane));
          long long df = 0, dp = 0;
#pragma unroll
          for (int sg = 0; sg < 2; sg++) {
            const unsigned short* L = sg == 0 ? ((rx == ROLE_A || rx == ROLE_FA) ? listB : listA) : (rx == ROLE_A ? listFb : listFa);
            const int cnt = sg == 0 ? ((rx == ROLE_A || rx == ROLE_FA) ? nB : nA) : (rx == ROLE_A ? nFb : (rx == ROLE_B ? nFa : 0));
            const bool perm = sg == 0 && merge && (rx == ROLE_A || rx == ROLE_B);
            for (int p0 = 0; p0 < cnt; p0 += 128) {
              // two passes of 64 members in flight
              int w[2], ov[2], wb[2];
              bool in[2];
#pragma unroll
              for (int c = 0; c < 2; c++) {
                const int i = p0 + 64 * c + lane;
                w[c] = i < cnt ? (int)L[i] : -1;
                // bit w of the bitmap row: the word sits in lane w >> 6
                const int wl = w[c] < 0 ? 0 : w[c] >> 6;
                const unsigned int lo = (unsigned int)__shfl((int)(unsigned int)nzw, wl, 64);
                const unsigned int hi = (unsigned int)__shfl((int)(unsigned int)(nzw >> 32), wl, 64);
                const unsigned long long word = ((unsigned long long)hi << 32) | lo;
                in[c] = w[c] >= 0 && ((word >> (w[c] & 63)) & 1ull);
                ov[c] = 0;
                wb[c] = 0;
              }
              if (__ballot(in[0] || in[1]) == 0ull) continue;
#pragma unroll
              for (int c = 0; c < 2; c++)
                if (in[c]) {
                  wb[c] = WM_AT((unsigned)(b * n + w[c]));
                  ov[c] = WM_AT((unsigned)(x * n + w[c]));
                }
#pragma unroll
              for (int c = 0; c < 2; c++) {
                // (lanes without a third edge hold 0,0: every term vanishes)
                if (perm) {
                  df += icf_term(W_PERM, wb[c]) - icf_term(ov[c], wb[c]);
                  dp += icp_term(W_PERM, wb[c]) - icp_term(ov[c], wb[c]);
                } else {
                  df += icf_term(W_FORB, wb[c]) - icf_term(ov[c], wb[c]);
                  dp += icp_term(W_FORB, wb[c]) - icp_term(ov[c], wb[c]);
                }
              }
            }
          }
          df = wave_sum_i64(df);
          dp = wave_sum_i64(dp);
          if (lane == 0 && (df != 0 || dp != 0)) {
            const unsigned ici = ic_index(x, b, n);
            IcT ic = IC_AT(ici);
            ic.f += df;
            ic.p += dp;
            IC_AT(ici) = ic;
            IC_AT(ic_index(b, x, n)) = ic;
#ifdef AHK_TIMING
            atomicAdd(&s_chg, 1);
#endif
            const unsigned long long c1 = mkkey(ic.f, x, b), c2 = mkkey(ic.p, x, b);
            fold_key(&bestF[x], c1);
            fold_key(&bestP[x], c2);
            fold_key(&bestF[b], c1);
            fold_key(&bestP[b], c2);
          }
        }
#endif
      }
      __syncthreads();
      AHK_T(3);

      // ---- P4: rewrite the weights of S, merge the cliques, rescan queued rows ----
      {
        const int N1 = merge ? W_PERM : W_FORB;
        // all rewritten pairs, flattened over the workgroup: A x (B u Fb), then B x Fa
        // (a big clique against a single read is 300 pairs, not 300 loop trips)
        {
          const int ny = nB + nFb;
          const int np1 = nA * ny, np2 = nB * nFa;
          for (int i = tid; i < np1 + np2; i += kThreads) {
            int x, y, N;
            if (i < np1) {
              const int xi = i / ny, yi = i - xi * ny;
              x = listA[xi];
              y = yi < nB ? listB[yi] : listFb[yi - nB];
              N = yi < nB ? N1 : W_FORB;
            } else {
              const int j = i - np1;
              const int xi = j / nFa, yi = j - xi * nFa;
              x = listB[xi];
              y = listFa[yi];
              N = W_FORB;
            }
            WM_AT((unsigned)(x * n + y)) = N;
            WM_AT((unsigned)(y * n + x)) = N;
            AHK_L(23, &WM_AT((unsigned)(x * n + y)));
            AHK_L(23, &WM_AT((unsigned)(y * n + x)));
          }
        }
        if (merge) {
          for (int yi = tid; yi < nB; yi += kThreads) {
            const int y = listB[yi];
            cl[y] = (unsigned short)cu;
            atomicOr(&multiw[y >> 5], 1u << (y & 31));
          }
          for (int xi = tid; xi < nA; xi += kThreads) {
            const int x = listA[xi];
            atomicOr(&multiw[x >> 5], 1u << (x & 31));
          }
        }
        if (tid < 4) s_cnt[tid] = 0;
        if (tid == 0) { s_items = 0; s_nh = 0; s_nslow = 0; }
        AHK_T(6);  // (diagnostic build: weight rewrites vs rescans)
        // rows whose maximum died: exact rescan now (they would tie the next argmax anyway)
        rescan_list<IcT, NWV>(dlist, __builtin_amdgcn_readfirstlane(s_nd), wv, lane, n, nw, LV, IC, lwm, glog, bestF, bestP);
        AHK_T(7);
      }
      __syncthreads();
#ifdef AHK_TIMING
      {
        const int Mh = s_items_last;
        const int na_ = nA + nFa, nb_ = nB + nFb;
        const int Th = na_ > nb_ ? na_ : nb_, Tl = na_ > nb_ ? nb_ : na_;
        int bm = 0, bt = 0;
        while (bm < 7 && Mh > (32 << bm)) bm++;
        while (bt < 7 && Th > (2 << bt)) bt++;
        const int bl = Tl <= 1 ? 0 : Tl <= 8 ? 1 : Tl <= 64 ? 2 : 3;
        const long long tn = (long long)clock64();
        hM[bm]++; hMc[bm] += tn - titer; hT[bt]++; hTc[bt] += tn - titer;
        h2[bt * 4 + bl]++; h2c[bt * 4 + bl] += tn - titer; h2m[bt * 4 + bl] += Mh; titer = tn;
        nmerge += merge ? 1 : 0; sumnd += s_nd;
      }
#endif
      if (tid == 0) s_nd = 0;  // next append is behind the P1 reduction barrier
      AHK_T(4);
      AHK_F(9);
#ifdef AHK_TIMING
      heavy_it = false;
#endif
    }
    __syncthreads();
#ifdef AHK_FTRACE
    if (tid == 0 && ft_first) {
      printf("ftrace chain %d n=%d fast=%lld other=%lld | per wave: scan reduce barrier decide | rt1 rt2+update reduce barrier commit | general\n", c, n, ft_nfast, ft_nother);
      for (int q = 0; q < 4; q++)
        printf("ftrace wave %d: %llu %llu %llu %llu | %llu %llu %llu %llu %llu | %llu\n", q, s_ft[q][0], s_ft[q][1], s_ft[q][2], s_ft[q][3], s_ft[q][4], s_ft[q][5], s_ft[q][6],
               s_ft[q][7], s_ft[q][8], s_ft[q][9]);
    }
    ft_first = false;
    __syncthreads();
#endif
#ifdef AHK_TIMING
    if (tid == 0 && blockIdx.x == 0)
      printf("chain %d n=%d iters=%lld retries=%lld rescans=%lld sumT=%lld sumM=%lld sumW=%lld | cycles P1=%lld resc=%lld P2=%lld P3a=%lld (loadwait=%lld proc=%lld) P3b=%lld P4=%lld | P3b: decode=%lld tri=%lld fold=%lld\n", c, n, iters,
             nretry, nresc, sumT, sumM, (long long)s_chg, tacc[0], tacc[1], tacc[2], tacc[5], tacc[6], tacc[7], tacc[3], tacc[4], tacc[8], tacc[9], tacc[10]);
    if (tid == 0 && blockIdx.x == 0) {
      printf("merges=%lld sum_nd=%lld fast=%lld fast_cycles=%lld\n", nmerge, sumnd, nfast, tacc[11]);
      printf("plain loop wave 0: passA trips+flushes=%lld (flushes %lld, n=%lld) tail flush=%lld barrier=%lld passA count=%lld\n", pl_t[0], pl_t[1], pl_t[5], pl_t[2], pl_t[3], pl_t[6]);
      for (int q = 0; q < 4; q++)
        printf("plain loop cat %d (0 sep single-sided, 1 merge single-sided no F walk, 2 merge single-sided F walk, 3 other): iters=%lld cycles=%lld items=%lld sumF=%lld\n", q, pcat_n[q],
               pcat_c[q], pcat_m[q], pcat_f[q]);
      printf("heavy iters=%lld | Mcycles P1=%lld P2=%lld P3a=%lld (P4: rewrite=%lld rescan=%lld) P3b=%lld P4wait=%lld | P3b: decode=%lld tri=%lld fold=%lld\n", nheavy, tacc2[0] / 1000000,
             tacc2[2] / 1000000, tacc2[5] / 1000000, tacc2[6] / 1000000, tacc2[7] / 1000000, tacc2[3] / 1000000, tacc2[4] / 1000000, tacc2[8] / 1000000, tacc2[9] / 1000000, tacc2[10] / 1000000);
      for (int b = 0; b < 8; b++)
        for (int q = 0; q < 4; q++)
          if (h2[b * 4 + q]) printf("ncmax<=%d ncmin_bucket=%d: iters=%lld Mcyc=%lld avgM=%lld\n", 2 << b, q, h2[b * 4 + q], h2c[b * 4 + q] / 1000000, h2m[b * 4 + q] / h2[b * 4 + q]);
      for (int b = 0; b < 8; b++) printf("M<=%d: iters=%lld cyc/iter=%lld | T<=%d: iters=%lld cyc/iter=%lld\n", 32 << b, hM[b], hM[b] ? hMc[b] / hM[b] : 0, 2 << b, hT[b], hT[b] ? hTc[b] / hT[b] : 0);
    }
#endif
#ifdef AHK_LINES
    if (tid == 0 && blockIdx.x == 0) {
      unsigned long long tot = 0ull;
      for (int q = 0; q < 32; q++) tot += g_lines[q];
      printf("lines chain %d n=%d iters=%lld total_lines=%llu\n", c, n, iters, tot);
      for (int q = 0; q < 32; q++) {  // (one value per call: the device printf garbles longer argument lists)
        printf("lines site %d insts %llu\n", q, g_reqs[q]);
        printf("lines site %d lanes %llu\n", q, g_lanes[q]);
        printf("lines site %d lines %llu\n", q, g_lines[q]);
      }
    }
    __syncthreads();
#endif
#ifdef AHK_DIAG
    if (lds.stop_after == 3) continue;
#endif

    // (the per-slot tables of S6/S7 are addressed only from here on: computed here, their
    // pointers do not occupy scalar registers during S5, which spills them as it is)
    unsigned int* sortbuf = ws.sortbuf + (size_t)blockIdx.x * 4 * ws.nmax;
    int* tp_k = ws.tp_k + (size_t)blockIdx.x * ws.Pmax;
    int* tp_gcid = ws.tp_gcid + (size_t)blockIdx.x * ws.Pmax * 8;
    double* tp_cov = ws.tp_cov + (size_t)blockIdx.x * ws.Pmax * 8;
    int* tp_cons = ws.tp_cons + (size_t)blockIdx.x * ws.Pmax * 8;
    int* tp_consmap = ws.tp_consmap + (size_t)blockIdx.x * ws.Pmax * 8;
    int* tp_ntup = ws.tp_ntup + (size_t)blockIdx.x * ws.Pmax;
    int* tp_tuples = ws.tp_tuples + (size_t)blockIdx.x * ws.Pmax * ws.Tmax;
    double* tp_score = ws.tp_score + (size_t)blockIdx.x * ws.Pmax * ws.Tmax;
    unsigned short* tp_pred = ws.tp_pred + (size_t)blockIdx.x * ws.Pmax * ws.Tmax;
    double* tp_best = ws.tp_best + (size_t)blockIdx.x * ws.Tmax;
    // ---- clusters: ordered by smallest member ------------------------------
    // (bestF is free now: reuse as scratch)
    unsigned int* scratch = (unsigned int*)bestF;  // [4*nmax] u32 available
    for (int u = tid; u < n; u += kThreads) scratch[u] = 0xFFFFFFFFu;
    __syncthreads();
    for (int u = tid; u < n; u += kThreads) atomicMin(&scratch[cl[u]], (unsigned)u);
    __syncthreads();
    if (wv == 0) {
      int base = 0;
      for (int u0 = 0; u0 < n; u0 += 64) {
        const int u = u0 + lane;
        const bool lead = u < n && scratch[cl[u]] == (unsigned)u;
        const unsigned long long mk = __ballot(lead);
        if (lead) scratch[nmax + u] = base + __popcll(mk & ((1ull << lane) - 1ull));
        base += __popcll(mk);
      }
      if (lane == 0) s_back = base;
    }
    __syncthreads();
    const int K = __builtin_amdgcn_readfirstlane(s_back);
    for (int u = tid; u < n; u += kThreads) {
      const int kc = (int)scratch[nmax + scratch[cl[u]]];
      db.fr_cluster[m.fr0 + u] = kc;
      scratch[2 * (size_t)nmax + u] = (unsigned)kc;
    }
    __syncthreads();
    for (int u = tid; u < n; u += kThreads) cl[u] = (unsigned short)scratch[2 * (size_t)nmax + u];
    if (tid == 0) {
      db.nclusters[c] = K;
      db.niter[c] = iters;
    }
    __syncthreads();

    // ================= S6: coverage, covMap, consensus =====================
    const int P = cx.P;
    const int nA_all = 1 << cx.planes;
    {
      unsigned int* hist = (unsigned int*)bestF + (size_t)(wv & 3) * nmax;  // per wave, [K]
      unsigned int* sb = sortbuf + (size_t)(wv & 3) * ws.nmax;
      for (int j = wv; j < P && wv < 4; j += 4) {  // (per-wave histogram and sort buffer: four waves)
        for (int k = lane; k < K; k += 64) hist[k] = 0;
        const int wj = j >> 6, bj = j & 63;
        int tot = 0;
        for (int i = lane; i < n; i += 64) {
          if ((cx.cover[(long long)wj * cx.G + i] >> bj) & 1ull) {
            atomicAdd(&hist[cl[i]], 1u);
            tot++;
          }
        }
        tot = wave_sum_i(tot);
        // clusters present at this position, ascending id (std::map order)
        int M = 0;
        for (int k0 = 0; k0 < K; k0 += 64) {
          const int k = k0 + lane;
          const unsigned int h = k < K ? hist[k] : 0;
          const unsigned long long mk = __ballot(h > 0);
          if (h > 0) sb[M + __popcll(mk & ((1ull << lane) - 1ull))] = (h << 16) | (unsigned)k;
          M += __popcll(mk);
        }
        // sort(): by coverage descending, std::sort from ascending-id order (:707-727)
        __threadfence_block();  // sb is global memory shared between the lanes of this wave
        if (lane < 8) s_asc[wv & 3][lane] = lane < M ? (int)sb[lane] : 0;
        if (lane == 0) gnu_sort(sb, M, GreaterCnt());
        __threadfence_block();
        // get_pos_to_clusters_map cut-off (:763-772)
        int cut = M < 2 * ploidy ? M : 2 * ploidy;
        {
          const int lim = cut;
          for (int i = ploidy; i < lim; i++) {
            const double cv = (double)(sb[i] >> 16) / (double)tot;
            if (cv < (1.0 / (8.0 * (double)ploidy))) { cut = i; break; }
          }
        }
        // consensus of each selected cluster at this position (:604-655)
        int mycons = 0;  // lane i < cut holds the consensus of covMap[j][i]
        for (int ci = 0; ci < cut; ci++) {
          const int cid = (int)(sb[ci] & 0xffffu);
          int best_cnt = 0, best_a = 0;
          for (int a = 0; a < nA_all; a++) {
            int cnt = 0;
            for (int i0 = 0; i0 < n; i0 += 64) {
              const int i = i0 + lane;
              bool hit = false;
              if (i < n && cl[i] == cid) {
                const long long o = (long long)wj * cx.G + i;
                if ((cx.cover[o] >> bj) & 1ull) {
                  int al = (int)((cx.alle[o] >> bj) & 1ull);
                  for (int pl = 1; pl < cx.planes; pl++) al |= (int)((cx.alle[pl * cx.plane_stride + o] >> bj) & 1ull) << pl;
                  hit = al == a;
                }
              }
              cnt += __popcll(__ballot(hit));
            }
            if (cnt > best_cnt) { best_cnt = cnt; best_a = a; }  // lowest allele among maxima (:629-646)
          }
          if (lane == ci) mycons = best_a;
        }
        // tables for the threader
        if (lane == 0) tp_k[j] = cut;
        if (lane < cut) {
          const unsigned int e = sb[lane];
          tp_gcid[(size_t)j * 8 + lane] = (int)(e & 0xffffu);
          tp_consmap[(size_t)j * 8 + lane] = mycons;
          if (db.compaction_mode != 0) {
            tp_cov[(size_t)j * 8 + lane] = (double)(e >> 16) / (double)tot;
            tp_cons[(size_t)j * 8 + lane] = mycons;
          } else {
            // reference compaction (:379-402): coverage = i-th cluster present at the
            // position in ascending id; consensus = i-th covMap cluster in ascending id
            tp_cov[(size_t)j * 8 + lane] = (double)((unsigned)s_asc[wv & 3][lane] >> 16) / (double)tot;
          }
        }
        if (db.compaction_mode == 0) {
          // rank of my cluster id among the selected ones
          const int myid = lane < cut ? (int)(sb[lane] & 0xffffu) : 0x7fffffff;
          int rank = 0;
          for (int t = 0; t < cut; t++) {
            const int oid = __shfl(myid, t, 64);
            rank += (oid < myid) ? 1 : 0;
          }
          if (lane < cut) tp_cons[(size_t)j * 8 + rank] = mycons;
        }
      }
    }
    __syncthreads();

    // ================= S7: threading DP ====================================
    // tuples of position j: ordered ploidy-tuples of local cluster indices,
    // code = sum_i t_i * k^(ploidy-1-i), ascending; genotype-conform ones, or
    // all of them when none conforms.
    {
      const long long Tmax = ws.Tmax;
      {
        for (int j = wv; j < P; j += NWV) {  // one wave per position
          const int k = tp_k[j];
          int T = 1;
          for (int i = 0; i < ploidy; i++) T *= k;
          // wanted allele counts
          int want[kMaxAlleles];
          for (int a = 0; a < kMaxAlleles; a++) want[a] = 0;
          if (db.genotype) {
            const int bub = db.pos[m.pos0 + j];
            const long long a0 = db.allele_off[m.b0 + bub], a1 = db.allele_off[m.b0 + bub + 1];
            for (long long a = a0; a < a1 && a - a0 < kMaxAlleles; a++) want[a - a0] = db.genotype[a];
          } else {
            want[0] = 1;
            want[1] = 1;
          }
          int nconf = 0;
          for (int c0 = 0; c0 < T; c0 += 64) {
            const int code = c0 + lane;
            bool ok = false;
            if (code < T) {
              int have[kMaxAlleles];
              for (int a = 0; a < kMaxAlleles; a++) have[a] = 0;
              int rem = code;
              for (int i = 0; i < ploidy; i++) {
                const int t = rem % k;
                rem /= k;
                const int al = tp_cons[(size_t)j * 8 + t];
                for (int a = 0; a < kMaxAlleles; a++) have[a] += (a == al) ? 1 : 0;
              }
              ok = true;
              for (int a = 0; a < kMaxAlleles; a++) ok = ok && (have[a] == want[a]);
            }
            const unsigned long long mk = __ballot(ok);
            if (ok) tp_tuples[(size_t)j * Tmax + nconf + __popcll(mk & ((1ull << lane) - 1ull))] = code;
            nconf += __popcll(mk);
          }
          if (nconf == 0) {
            for (int code = lane; code < T; code += 64) tp_tuples[(size_t)j * Tmax + code] = code;
            nconf = T;
          }
          if (lane == 0) tp_ntup[j] = nconf;
        }
      }
      __syncthreads();
      for (int j = 0; j < P; j++) {
        const int k = tp_k[j], nt = tp_ntup[j];
        const int kp = j > 0 ? tp_k[j - 1] : 1, ntp = j > 0 ? tp_ntup[j - 1] : 0;
        for (int r = tid; r < nt; r += kThreads) {
          int t[kMaxPloidy], g[kMaxPloidy];
          int rem = tp_tuples[(size_t)j * Tmax + r];
          for (int i = ploidy - 1; i >= 0; i--) { t[i] = rem % k; rem /= k; }
          for (int i = 0; i < ploidy; i++) g[i] = tp_gcid[(size_t)j * 8 + t[i]];
          double cc = 0.0;
          for (int i = 0; i < ploidy; i++) {
            int cn = 0;
            for (int i2 = 0; i2 < ploidy; i2++) cn += (t[i2] == t[i]) ? 1 : 0;
            const double cov = tp_cov[(size_t)j * 8 + t[i]];
            const double lo = ((double)cn - 0.5) / (double)ploidy;
            const double hi = ((double)cn + 0.5) / (double)ploidy;
            if (cov < lo || cov > hi) cc += 1.0;
          }
          if (j == 0) {
            tp_score[r] = cc;
            tp_pred[r] = 0xffff;
            continue;
          }
#if AHK_S7_MARGINAL
          // (the predecessor minimum is taken below, by marginal minima; here: the column cost)
          tp_score[(size_t)j * Tmax + r] = cc;
        }
        __syncthreads();
        if (j == 0) continue;  // (uniform)
        // ---- predecessor minimum by MARGINAL MINIMA: O(T * 2^p) instead of O(T^2) ----------
        // The transition cost depends only on how many slots change cluster: with A(q,r) the set
        // of slots on which predecessor q and tuple r agree, cost = c(p - |A|), c(m) =
        // switch*m + (m > 0 ? affine : 0), non-decreasing in m (both parameters >= 0).  Hence
        //   min_q [score(q) + c(p - |A(q,r)|)] = min over subsets S of the slots of
        //       c(p - |S|) + min { score(q) : q agrees with r on (at least) S }
        // -- a predecessor that agrees on more than S is merely over-charged under S and
        // appears with its true cost under its own set.  For every S one table over the
        // clusters of the previous position holds the minimum score of the predecessors with
        // given clusters on S (LDS, ds_min_u64 on the bit patterns of the non-negative doubles)
        // and, among the predecessors that reach it, the lowest list index; S = all slots is a
        // plain lookup of the tuple code.  Taking candidates in (value, index) order reproduces
        // the first minimum of the sequential scan, strict '<' and all: a tie partner with a
        // lower index wins its own class or loses it to an even lower index of equal value.
        {
          unsigned short* idxmap = (unsigned short*)smem;                          // [kp^p] tuple code -> list index
          int pw[kMaxPloidy + 1];
          pw[0] = 1;
#pragma unroll
          for (int i = 1; i <= kMaxPloidy; i++) pw[i] = pw[i - 1] * kp;
          const int ncode = pw[ploidy];
          unsigned long long* tabS = (unsigned long long*)(idxmap + ((ncode + 3) & ~3));  // [kp^(p-1)] min score bits
          unsigned int* tabQ = (unsigned int*)(tabS + pw[ploidy - 1]);                      // [kp^(p-1)] lowest index reaching it
          const int* tup_prev = tp_tuples + (size_t)(j - 1) * Tmax;
          const double* sc_prev = tp_score + (size_t)(j - 1) * Tmax;
          for (int cidx2 = tid; cidx2 < ncode; cidx2 += kThreads) idxmap[cidx2] = 0xffffu;
          __syncthreads();
          for (int q = tid; q < ntp; q += kThreads) idxmap[tup_prev[q]] = (unsigned short)q;
          __syncthreads();
          // slot i of tuple r lives in cluster gcid[j][t_i]: its index among the previous
          // position's clusters (or -1) -- per thread, eight clusters at most
          int locp[8];
#pragma unroll
          for (int t = 0; t < 8; t++) {
            locp[t] = -1;
            if (t < k) {
              const int gid = tp_gcid[(size_t)j * 8 + t];
              for (int l = 0; l < kp; l++)
                if (tp_gcid[(size_t)(j - 1) * 8 + l] == gid) locp[t] = l;
            }
          }
          auto loc_of = [&](int t) {  // (constant indices only: locp stays in registers)
            int v = -1;
#pragma unroll
            for (int u = 0; u < 8; u++) v = (t == u) ? locp[u] : v;
            return v;
          };
          // S = all slots: exact predecessor with the same clusters
          for (int r = tid; r < nt; r += kThreads) {
            int rem = tp_tuples[(size_t)j * Tmax + r];
            int code = 0;
            bool ok = true;
            for (int i = ploidy - 1; i >= 0; i--) {
              const int l = loc_of(rem % k);
              rem /= k;
              ok = ok && l >= 0;
              code += (l < 0 ? 0 : l) * pw[ploidy - 1 - i];
            }
            double best = __builtin_inf();
            int bq = 0;
            if (ok) {
              const unsigned int q0 = idxmap[code];
              if (q0 != 0xffffu) {
                best = sc_prev[q0] + (db.switch_cost * (double)0 + 0.0);
                bq = (int)q0;
              }
            }
            tp_best[r] = best;
            tp_pred[(size_t)j * Tmax + r] = (unsigned short)bq;
          }
          const int nsub = (1 << ploidy) - 1;  // proper subsets
          for (int S = 0; S < nsub; S++) {
            const int nS = __popc((unsigned)S), sw = ploidy - nS;
            const int nent = pw[nS];
            for (int e = tid; e < nent; e += kThreads) { tabS[e] = ~0ull; tabQ[e] = 0xffffffffu; }
            __syncthreads();
            for (int pass = 0; pass < 2; pass++) {
              for (int q = tid; q < ntp; q += kThreads) {
                int rem = tup_prev[q];
                int idx = 0, mul = 1;
                for (int i = ploidy - 1; i >= 0; i--) {
                  const int tq = rem % kp;
                  rem /= kp;
                  if ((S >> i) & 1) { idx += tq * mul; mul *= kp; }
                }
                const unsigned long long sb = (unsigned long long)__double_as_longlong(sc_prev[q]);
                if (pass == 0) atomicMin(&tabS[idx], sb);
                else if (sb == tabS[idx]) atomicMin(&tabQ[idx], (unsigned int)q);
              }
              __syncthreads();
            }
            const double add = db.switch_cost * (double)sw + (sw > 0 ? db.affine_cost : 0.0);
            for (int r = tid; r < nt; r += kThreads) {
              int rem = tp_tuples[(size_t)j * Tmax + r];
              int idx = 0, mul = 1;
              bool ok = true;
              for (int i = ploidy - 1; i >= 0; i--) {
                const int l = loc_of(rem % k);
                rem /= k;
                if ((S >> i) & 1) {
                  ok = ok && l >= 0;
                  idx += (l < 0 ? 0 : l) * mul;
                  mul *= kp;
                }
              }
              if (!ok) continue;
              const unsigned long long sb = tabS[idx];
              if (sb == ~0ull) continue;
              const double v = __longlong_as_double((long long)sb) + add;
              const int qq = (int)tabQ[idx];
              const double best = tp_best[r];
              const int bq = tp_pred[(size_t)j * Tmax + r];
              if (v < best || (v == best && qq < bq)) {
                tp_best[r] = v;
                tp_pred[(size_t)j * Tmax + r] = (unsigned short)qq;
              }
            }
            __syncthreads();
          }
          for (int r = tid; r < nt; r += kThreads) tp_score[(size_t)j * Tmax + r] = tp_score[(size_t)j * Tmax + r] + tp_best[r];
        }
        __syncthreads();
      }
#else
          double best = __builtin_inf();
          int bq = 0;
          for (int q = 0; q < ntp; q++) {
            int remq = tp_tuples[(size_t)(j - 1) * Tmax + q];
            int sw = 0;
            for (int i = ploidy - 1; i >= 0; i--) {
              const int tq = remq % kp;
              remq /= kp;
              sw += (tp_gcid[(size_t)(j - 1) * 8 + tq] != g[i]) ? 1 : 0;
            }
            const double s = tp_score[(size_t)(j - 1) * Tmax + q] + (db.switch_cost * (double)sw + (sw > 0 ? db.affine_cost : 0.0));
            if (s < best) { best = s; bq = q; }
          }
          tp_score[(size_t)j * Tmax + r] = cc + best;
          tp_pred[(size_t)j * Tmax + r] = (unsigned short)bq;
        }
        __syncthreads();
      }
#endif
      // ---- backtrace + S8 -------------------------------------------------
      // final state: the FIRST minimum of the last column.  Scores are non-negative doubles
      // (their bit patterns order like the values), so lane-strided running minima of
      // (score bits, index) pairs and one 64-bit DPP min-reduce of the bits, then of the
      // indices among the lanes that hold the minimum, pick exactly the tuple a sequential
      // scan with strict '<' picks.
      if (wv == 0) {
        const int ntl = tp_ntup[P - 1];
        unsigned long long bs = ~0ull;
        int br = 0x7fffffff;
        for (int r = lane; r < ntl; r += 64) {
          const unsigned long long v = (unsigned long long)__double_as_longlong(tp_score[(size_t)(P - 1) * Tmax + r]);
          if (v < bs) { bs = v; br = r; }  // (ascending r per lane: ties keep the lower index)
        }
        const unsigned long long gmin = ~wave_max_u64(~bs);
        const unsigned long long idxkey = bs == gmin ? (unsigned long long)(0x7fffffff - br) : 0ull;
        const int cur0 = 0x7fffffff - (int)wave_max_u64(idxkey);
        if (lane == 0) s_back = cur0;
      }
      __syncthreads();
      if (tid == 0) {
        int cur = s_back;
        int* ha = db.hap_allele + (long long)ploidy * m.pos0;
        int* hc = db.hap_cluster + (long long)ploidy * m.pos0;
        for (int j = P - 1; j >= 0; j--) {
          const int k = tp_k[j];
          int rem = tp_tuples[(size_t)j * Tmax + cur];
          for (int i = ploidy - 1; i >= 0; i--) {
            const int t = rem % k;
            rem /= k;
            hc[(size_t)i * P + j] = tp_gcid[(size_t)j * 8 + t];
            ha[(size_t)i * P + j] = tp_consmap[(size_t)j * 8 + t];  // new_consensus[j][c_id], :420-422
          }
          cur = tp_pred[(size_t)j * Tmax + cur];
        }
      }
    }
  }
}

size_t phase_lds_bytes(const PhaseLds& l) {
  const int nwv = l.nwv > 4 ? l.nwv : 4;
  const int rw0 = l.nmax > 2048 ? 64 : 32;
  const int rw = rw0 > nwv * 8 ? rw0 : nwv * 8;
  size_t b = (size_t)l.nmax * 8 * 2;       // bestF, bestP
  b += (size_t)l.nmax * 2 * 2 + 64;        // cl + the pool of the four member lists
  b += (size_t)l.nmax * 2 * (l.items_lds ? 5 : 1);  // rowbuf[4] (the LDS part of the candidate-edge list), dlist
  b += (size_t)(((l.nmax + 31) / 32 + 1) / 2 * 2) * 4;  // multiw (padded to 8 bytes)
  b += (size_t)2 * 4 * rw * 8;             // rmask, rolenz (and the per-wave pending lists that re-use them)
  b += (size_t)l.nmax;                     // role
  if (l.stage_row) b += (size_t)l.nmax * 4;  // wrow
  if (AHK_LWM) b += (size_t)l.nmax * 2;    // lwm
  if (nwv > 4) b += 8 + (size_t)nwv * l.nmax * 4;  // per-wave scratch of the wide forms (S4a keys, S5-init neighbour lists)
  {  // the threading DP re-uses the region: tuple-code map (u16) + one marginal table (u64 + u32)
    size_t codes = 1, marg = 1;
    for (int i = 0; i < l.ploidy; i++) { marg = codes; codes *= (size_t)(2 * l.ploidy); }
    const size_t s7 = ((codes + 3) & ~(size_t)3) * 2 + marg * 12 + 16;
    if (s7 > b) b = s7;
  }
  return (b + 15) & ~(size_t)15;
}

namespace {
template <int NWV, int RW>
int max_blocks(const PhaseLds& l) {
  int nb = 0, nb32 = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_phase_t<false, NWV, RW>, NWV * 64, phase_lds_bytes(l)) != hipSuccess) return 1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb32, k_phase_t<true, NWV, RW>, NWV * 64, phase_lds_bytes(l)) != hipSuccess) return 1;
  if (nb32 < nb) nb = nb32;
  return nb < 1 ? 1 : nb;
}
template <int NWV, int RW>
void launch_nwv(const DeviceBatch& db, const SlotWorkspace& ws, const PhaseLds& l, const int* queue, int C, int* ctl, int* deferred, int slots,
                hipStream_t stream) {
  // narrow induced costs first; the chains it had to defer (none in the BASELINE configurations 1-3 and 5) go
  // through the wide-cost instance, whose workgroups find an empty list otherwise
  hipLaunchKernelGGL((k_phase_t<true, NWV, RW>), dim3(slots), dim3(NWV * 64), phase_lds_bytes(l), stream, db, ws, l, queue, C, (const int*)nullptr, ctl,
                     deferred, ctl + 2);
  hipLaunchKernelGGL((k_phase_t<false, NWV, RW>), dim3(slots), dim3(NWV * 64), phase_lds_bytes(l), stream, db, ws, l, (const int*)deferred, 0,
                     (const int*)(ctl + 2), ctl + 1, (int*)nullptr, (int*)nullptr);
}
}  // namespace

// (the instances: four waves with 32- or 64-word role tables, eight waves, sixteen waves)
int phase_max_blocks_per_cu(const PhaseLds& l) {
  if (l.nwv == 16) return max_blocks<16, 128>(l);
  if (l.nwv == 8) return max_blocks<8, 64>(l);
  return l.nmax > 2048 ? max_blocks<4, 64>(l) : max_blocks<4, 32>(l);
}

void launch_phase(const DeviceBatch& db, const SlotWorkspace& ws, const PhaseLds& l, const int* queue, int C, int* ctl, int* deferred,
                  int slots, hipStream_t stream) {
  if (l.nwv == 16) launch_nwv<16, 128>(db, ws, l, queue, C, ctl, deferred, slots, stream);
  else if (l.nwv == 8) launch_nwv<8, 64>(db, ws, l, queue, C, ctl, deferred, slots, stream);
  else if (l.nmax > 2048) launch_nwv<4, 64>(db, ws, l, queue, C, ctl, deferred, slots, stream);
  else launch_nwv<4, 32>(db, ws, l, queue, C, ctl, deferred, slots, stream);
}

}  // namespace ahk
