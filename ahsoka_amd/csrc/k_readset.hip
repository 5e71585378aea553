// k_readset.hip -- stages S1..S3 of the per-chain phasing path as one HIP
// kernel for gfx950: one workgroup (4 wavefronts) per bubble chain.
//
//   S1   full-match readset        src/alignmentstoreadset.cpp:90-135
//   S1b  filter                    :146-163
//   S1c  positions to revisit      :174-208
//   S2   partial-match readset     :210-254
//   S2b  filter                    :262-282
//   S3   ReadSet::sort()           :296-297
//
// is_subset (:495-548) on allele paths without repeated nodes is a bit test
// (paths that repeat a node take the exact multiset-counting path): every
// alignment's node list becomes a bitset over the chain's nodes, held in LDS
// (one u64 column per lane, so a wave-uniform node index hits 64 different
// banks), and an allele matches when all of its path nodes (S1) or all inner
// nodes (S2) are set.  The R x B "first matching allele" matrices are written
// position-major so that consecutive lanes (alignments) store consecutive
// bytes.  Everything after that is per-read bookkeeping over those matrices;
// the read order the reference gets from std::sort is reproduced by one lane
// running the same introsort (gnusort.h) on an LDS array.
#include <hip/hip_runtime.h>

#include "gnusort.h"
#include "phase_kernels.h"
#include "../../include/ahsoka_phase.h"

namespace ahk {




// ReadSet::sort() -- the reference's std::sort by first position -- by ONE WAVEFRONT, in the
// rank-based form of gnusort.h (gnu_sort_par_ref is the sequential statement pinned against
// std::sort): every partition is two ballot sweeps (positions of the elements not less / not
// greater than the pivot), a parallel count of the swaps, the swaps themselves and a closed
// form of the cut; the final insertion sort is a stable rank inside every leaf.  The array and
// the scratch arrays live in LDS; call with the 64 lanes of one wave, n <= 65535.
//   a[n], posL[n], posR[n] (u16), cutbit[(n+31)/32] (u32), tmp[n]
__device__ __forceinline__ int wave_sum_small(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// lanes of ONE wave hand data to each other through LDS between these points: a fence at
// wavefront scope plus a scheduling barrier (the hardware runs a wave's LDS operations in order)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __noinline__ void wave_sort_hi16(unsigned int* a, int n, int lane, unsigned short* posL, unsigned short* posR,
                                            unsigned int* cutbit, unsigned int* tmp, int* stk /* [3][64] */) {
  n = __builtin_amdgcn_readfirstlane(n);
  if (n <= 1) return;
  for (int w = lane; w < (n + 31) / 32; w += 64) cutbit[w] = w == 0 ? 1u : 0u;
  int depth = 0;
  for (int t = n; t > 1; t >>= 1) depth++;
  depth *= 2;
  int sp = 1;
  if (lane == 0) { stk[0] = 0; stk[64] = n; stk[128] = depth; }
  const LessHi16 less;
  while (sp > 0) {
    --sp;
    wave_lds_sync();
    int first = stk[sp], last = stk[64 + sp], d = stk[128 + sp];
    while (last - first > 16) {
      if (d == 0) {  // depth limit: heapsort, one lane (adversarial inputs only)
        if (lane == 0) {
          PtrArray<unsigned int> arr{a};
          gs_heapsort(arr, first, last, less);
        }
        break;
      }
      --d;
      // median of a[first+1], a[mid], a[last-1] to a[first] (uniform reads, lane 0 swaps)
      {
        const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
        const unsigned int va = a[ia], vb = a[ib], vc = a[ic];
        int pick;
        if (less(va, vb)) pick = less(vb, vc) ? ib : (less(va, vc) ? ic : ia);
        else pick = less(va, vc) ? ia : (less(vb, vc) ? ic : ib);
        const unsigned int vf = a[first], vpk = a[pick];
        wave_lds_sync();
        if (lane == 0) {
          a[first] = vpk;
          a[pick] = vf;
        }
        wave_lds_sync();
      }
      const unsigned int pivot = a[first];
      int nL = 0, nR = 0;
      for (int i0 = first + 1; i0 < last; i0 += 64) {  // positions not less than the pivot, ascending
        const int i = i0 + lane;
        const bool fl = i < last && !less(a[i < last ? i : first], pivot);
        const unsigned long long mk = __ballot(fl);
        if (fl) posL[nL + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned short)i;
        nL += __popcll(mk);
      }
      for (int j0 = last - 1; j0 > first; j0 -= 64) {  // positions not greater than the pivot, descending
        const int j = j0 - lane;
        const bool fr = j > first && !less(pivot, a[j > first ? j : first]);
        const unsigned long long mk = __ballot(fr);
        if (fr) posR[nR + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned short)j;
        nR += __popcll(mk);
      }
      wave_lds_sync();
      const int nmin = nL < nR ? nL : nR;
      int K = 0;
      for (int k0 = 0; k0 < nmin; k0 += 64) {
        const int k = k0 + lane;
        const bool sw = k < nmin && posL[k < nmin ? k : 0] < posR[k < nmin ? k : 0];
        K += __popcll(__ballot(sw));
      }
      for (int k = lane; k < K; k += 64) {  // disjoint positions
        const int pl = posL[k], pr = posR[k];
        const unsigned int t = a[pl];
        a[pl] = a[pr];
        a[pr] = t;
      }
      wave_lds_sync();
      const int c1 = K < nL ? (int)posL[K] : 0x7fffffff, c2 = K > 0 ? (int)posR[K - 1] : 0x7fffffff;
      const int cut = c1 < c2 ? c1 : c2;
      if (lane == 0) {
        cutbit[cut >> 5] |= 1u << (cut & 31);
        stk[sp] = cut;
        stk[64 + sp] = last;
        stk[128 + sp] = d;
      }
      ++sp;
      last = cut;
    }
  }
  // final pass: stable rank inside every leaf
  wave_lds_sync();
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    if (i < n) {
      int f = i;
      while (!((cutbit[f >> 5] >> (f & 31)) & 1u)) f--;
      int l = i + 1;
      while (l < n && !((cutbit[l >> 5] >> (l & 31)) & 1u)) l++;
      const unsigned int vi = a[i];
      int rank = f;
      for (int j = f; j < l; j++) {
        const unsigned int vj = a[j];
        rank += (less(vj, vi) || (j < i && !less(vi, vj))) ? 1 : 0;
      }
      tmp[rank] = vi;
    }
  }
  wave_lds_sync();
  for (int i = lane; i < n; i += 64) a[i] = tmp[i];
  wave_lds_sync();
}

__device__ __forceinline__ bool tba_test(const unsigned int* tba, int k) {
  return (tba[k >> 5] >> (k & 31)) & 1u;
}

// merged "first matching allele" of read name g at position k: the allele loop
// is outside the alignment loop in the reference (:98-106, :214-222), so the
// lowest allele any alignment of that name matches wins, and among those the
// earliest alignment created the Read (its identity becomes the Read's mapq).
// need_gt90: only alignments whose `alignment.id*100 > 90` holds count -- a FLOAT
// compare in the reference (:245), carried as bit 0 of read_flags.  S2 adds a
// variant to an EXISTING read only from such alignments; the variant that
// creates the read is exempt.
__device__ __forceinline__ int merged_allele(const signed char* __restrict__ v, long long R, int k,
                                             const long long* __restrict__ name_off,
                                             const int* __restrict__ name_aln, int g, int* creator,
                                             const unsigned char* __restrict__ flags = nullptr, bool need_gt90 = false) {
  int best_a = 127, best_r = -1;
  for (long long ai = name_off[g]; ai < name_off[g + 1]; ai++) {
    int r = name_aln[ai];
    int a = v[(long long)k * R + r];
    if (need_gt90 && !(flags[r] & 1)) continue;
    if (a >= 0 && a < best_a) {
      best_a = a;
      best_r = r;
    }
  }
  *creator = best_r;
  return best_r >= 0 ? best_a : -1;
}

// std::includes on sorted ranges is a MULTISET test (quirk Q5): for an allele
// path that repeats a node the bit test is not enough, the alignment must visit
// that node at least as often.  Rare (cyclic bubbles): exact counting loop.
__device__ bool multiset_includes(const int* __restrict__ path, long long pb, long long pe, const int* __restrict__ rn,
                                  long long rb, long long re) {
  for (long long e = pb; e < pe; e++) {
    const int nd = path[e];
    bool seen_before = false;
    for (long long e2 = pb; e2 < e; e2++) seen_before = seen_before || path[e2] == nd;
    if (seen_before) continue;
    int cp = 0, cr = 0;
    for (long long e2 = e; e2 < pe; e2++) cp += path[e2] == nd;
    for (long long q = rb; q < re; q++) cr += rn[q] == nd;
    if (cr < cp) return false;
  }
  return true;
}

// bytes of the LDS region that k_readset uses first for the per-lane node windows and then for the
// per-read arrays and the sort's scratch
__host__ __device__ inline size_t readset_region_bytes(const ReadsetLds& l) {
  const size_t capN = (size_t)(l.maxG < kMaxReads ? l.maxG : kMaxReads);
  const size_t win = (size_t)l.wordsN * kBlock * 8;
  const size_t per_read = capN * 4 * 3 + ((capN / 32 + 2 + 1) & ~(size_t)1) * 4 + capN * 2 * 6;
  const size_t r = win > per_read ? win : per_read;
  return (r + 15) & ~(size_t)15;
}

__global__ __launch_bounds__(kBlock) void k_readset(DeviceBatch db, ReadsetLds lds, const int* __restrict__ list) {
  extern __shared__ __align__(16) unsigned long long smem[];
  const int c = list[blockIdx.x];
  const ChainMeta m = db.meta[c];
  const int tid = threadIdx.x;
  const int B = m.B, R = m.R, G = m.G;

  // LDS: the per-lane node windows of phase A and the per-read arrays of the later phases share one
  // region (the windows are dead once the match matrices are in HBM); the position masks follow
  const int capN = lds.maxG < kMaxReads ? lds.maxG : kMaxReads;  // reads that can reach the final readset
  const int PW = (lds.maxB + 32) >> 5;                            // u32 words of a position mask
  unsigned long long* bits = smem;                                   // [wordsN][kBlock]   (phase A)
  unsigned int* keys = (unsigned int*)smem;                          // [capN]             (phase C on)
  unsigned int* srt = keys + capN;                                   // [capN]
  unsigned int* stmp = srt + capN;                                   // [capN]   sort scratch
  unsigned int* cutbit = stmp + capN;                                // [capN/32 + 2]
  unsigned short* kname = (unsigned short*)(cutbit + ((capN / 32 + 2 + 1) & ~1));  // [capN]
  unsigned short* insname = kname + capN;
  unsigned short* kfirst = insname + capN;
  unsigned short* insfirst = kfirst + capN;
  unsigned short* posL = insfirst + capN;                            // [capN]   sort scratch
  unsigned short* posR = posL + capN;
  unsigned int* tba = (unsigned int*)((char*)smem + readset_region_bytes(lds));  // [PW]
  unsigned int* firsts = tba + PW;
  unsigned int* lasts = firsts + PW;
  unsigned int* posm = lasts + PW;
  unsigned short* cidx = (unsigned short*)(posm + PW);               // [maxB+1]
  __shared__ int s_maxpos, s_any, s_n, s_P;
  __shared__ int s_stk[3 * 64];

  for (int w = tid; w < PW; w += kBlock) { tba[w] = 0; firsts[w] = 0; lasts[w] = 0; posm[w] = 0; }
  if (tid == 0) { s_maxpos = -1; s_any = 0; s_n = 0; s_P = 0; }
  __syncthreads();

  if (B <= 1 || (m.flags & 1)) {  // :86 (<=1 bubble) or a compiled-in limit
    if (tid == 0) {
      db.status[c] = (m.flags & 1) ? AHSOKA_ST_UNSUPPORTED : AHSOKA_ST_SKIPPED_LE1;
      db.nfin[c] = 0;
      db.npos[c] = 0;
    }
    return;
  }

  signed char* vfull = db.vfull + m.vmat0;
  signed char* vpart = db.vpart + m.vmat0;
  const long long* aoff = db.allele_off + m.b0;
  const int wordsN = (m.N + 63) >> 6;

  // ---- phase A: per alignment, first fully / partially matching allele -----
  // The lane's node bitset is a WINDOW of WW words starting at the word of the alignment's smallest
  // chain node: an alignment touches a short stretch of the chain, whatever the chain's length
  // (kMaxNodes = 8192 nodes would be 128 words per lane otherwise).  Exact for any alignment: a
  // node below the window is not on the read; a node above it is looked up in the read's node
  // list (only alignments that span more than WW*64 node indices ever take that branch).
  const int WW = wordsN < lds.wordsN ? wordsN : lds.wordsN;
  for (int r = tid; r < R; r += kBlock) {
    const long long e0 = db.rnode_off[m.r0 + r], e1 = db.rnode_off[m.r0 + r + 1];
    int mn = 0x7fffffff, mx = -1;
    for (long long e = e0; e < e1; e++) {
      const int nd = db.rnode[e];
      if (nd >= 0) { mn = nd < mn ? nd : mn; mx = nd > mx ? nd : mx; }
    }
    const int w0 = mx < 0 ? 0 : (mn >> 6);
    const bool wide = mx >= 0 && (mx >> 6) - w0 >= WW;
    for (int w = 0; w < WW; w++) bits[(size_t)w * kBlock + tid] = 0ull;
    for (long long e = e0; e < e1; e++) {
      const int nd = db.rnode[e];
      if (nd >= 0 && (nd >> 6) - w0 < WW) bits[(size_t)((nd >> 6) - w0) * kBlock + tid] |= 1ull << (nd & 63);
    }
    auto has = [&](int nd) -> bool {
      const int wi = (nd >> 6) - w0;
      if (wi >= 0 && wi < WW) return (bits[(size_t)wi * kBlock + tid] >> (nd & 63)) & 1ull;
      if (!wide || wi < 0) return false;
      for (long long e = e0; e < e1; e++)
        if (db.rnode[e] == nd) return true;
      return false;
    };
    for (int k = 0; k < B; k++) {
      int vf = -1, vp = -1;
      const long long a0 = aoff[k], a1 = aoff[k + 1];
      for (long long a = a0; a < a1 && (vf < 0 || vp < 0); a++) {
        const long long p0 = db.apath_off[a], p1 = db.apath_off[a + 1];
        if (db.apath_dup[a]) {  // path repeats a node: exact multiset containment
          const bool inner_d = p1 - p0 >= 2 ? multiset_includes(db.apath_node, p0 + 1, p1 - 1, db.rnode, e0, e1) : true;
          const bool full_d = multiset_includes(db.apath_node, p0, p1, db.rnode, e0, e1);
          if (vp < 0 && inner_d) vp = (int)(a - a0);
          if (vf < 0 && full_d) vf = (int)(a - a0);
          continue;
        }
        bool inner = true;
        for (long long e = p0 + 1; e < p1 - 1; e++) inner = inner && has(db.apath_node[e]);
        bool full = inner;
        if (p1 > p0) {
          full = full && has(db.apath_node[p0]);
          if (p1 - p0 > 1) full = full && has(db.apath_node[p1 - 1]);
        }
        if (vp < 0 && inner) vp = (int)(a - a0);
        if (vf < 0 && full) vf = (int)(a - a0);
      }
      vfull[(long long)k * R + r] = (signed char)vf;
      vpart[(long long)k * R + r] = (signed char)vp;
    }
  }
  __syncthreads();  // (from here on the window region holds the per-read arrays)

  const long long* name_off = db.name_off + m.name0;
  const int* name_aln = db.name_aln + m.r0;

  // ---- phase B: S1b filter and S1c boundary sets ---------------------------
  // S1 walks the bubbles in the iteration order of the reference's unordered_map
  // (:76,:90): the first hit IN THAT ORDER creates the Read and fixes its mapq.
  const int* s1 = db.s1_order + m.b0;
  for (int g = tid; g < G; g += kBlock) {
    int nvar = 0, first = B, last = -1, mapq = -1;
    for (int kk = 0; kk < B; kk++) {
      const int k = s1[kk];
      int cr;
      int a = merged_allele(vfull, R, k, name_off, name_aln, g, &cr);
      if (a < 0) continue;
      if (nvar == 0) mapq = db.read_mapq[m.r0 + cr];
      nvar++;
      first = k < first ? k : first;
      last = k > last ? k : last;
    }
    if (nvar > 0) {
      atomicMax(&s_maxpos, last);
      s_any = 1;
      if (nvar >= db.min_variants && mapq >= db.min_mapq) {
        atomicOr(&firsts[first >> 5], 1u << (first & 31));
        atomicOr(&lasts[last >> 5], 1u << (last & 31));
      }
    }
  }
  __syncthreads();
  if (!s_any) {  // reference: back() of an empty vector, UB (quirk Q4)
    if (tid == 0) { db.status[c] = AHSOKA_ST_NO_FULL_MATCH; db.nfin[c] = 0; db.npos[c] = 0; }
    return;
  }
  if (tid == 0) {
    // boundaries = {el, el+1 : el in lasts \ firsts}; to_be_added = [0,maxpos) U boundaries
    unsigned int carry = 0;
    for (int w = 0; w < PW; w++) {
      unsigned int bd = lasts[w] & ~firsts[w];
      unsigned int t = bd | (bd << 1) | carry;
      carry = bd >> 31;
      int lo = w * 32;
      unsigned int range = 0;
      if (s_maxpos > lo) range = (s_maxpos - lo >= 32) ? 0xffffffffu : ((1u << (s_maxpos - lo)) - 1u);
      t |= range;
      unsigned int valid = 0;  // positions >= B have no alleles (:216)
      if (B > lo) valid = (B - lo >= 32) ? 0xffffffffu : ((1u << (B - lo)) - 1u);
      tba[w] = t & valid;
    }
  }
  __syncthreads();

  // ---- phase C: S2 variants restricted to to_be_added, S2b filter ----------
  const unsigned char* fl = db.read_flags + m.r0;
  for (int g = tid; g < G; g += kBlock) {
    int nvar = 0, first = -1, afirst = 0, rcre = 0, mapq = -1;
    for (int k = 0; k < B; k++) {
      if (!tba_test(tba, k)) continue;
      int cr;
      // the first hit creates the Read (any identity); later positions need > 90
      int a = merged_allele(vpart, R, k, name_off, name_aln, g, &cr, fl, first >= 0);
      if (a < 0) continue;
      nvar++;
      if (first < 0) { first = k; afirst = a; rcre = cr; mapq = db.read_mapq[m.r0 + cr]; }
    }
    if (nvar >= db.min_variants && mapq >= db.min_mapq) {
      int slot = atomicAdd(&s_n, 1);
      // creation order of the Read objects in partial_readset: (position, allele, alignment)
      if (slot < capN) keys[slot] = ((unsigned)first << 20) | ((unsigned)afirst << 16) | (unsigned)rcre;  // (12-bit position, 4-bit allele, 16-bit alignment)
      if (slot < capN) {
        kname[slot] = (unsigned short)g;
        kfirst[slot] = (unsigned short)first;
      }
      for (int k = first; k < B; k++) {
        if (!tba_test(tba, k)) continue;
        int cr;
        if (merged_allele(vpart, R, k, name_off, name_aln, g, &cr, fl, k != first) >= 0)
          atomicOr(&posm[k >> 5], 1u << (k & 31));
      }
    }
  }
  __syncthreads();
  const int n = s_n;
  if (n == 0 || n > kMaxReads) {  // :276-279 warning + continue
    if (tid == 0) {
      db.status[c] = n == 0 ? AHSOKA_ST_EMPTY_READSET : AHSOKA_ST_UNSUPPORTED;
      db.nfin[c] = 0;
      db.npos[c] = 0;
    }
    return;
  }

  // ---- phase D: insertion order (rank of the creation key), then S3 --------
  for (int i = tid; i < n; i += kBlock) {
    unsigned int ki = keys[i];
    int rank = 0;
    for (int j = 0; j < n; j++) rank += (keys[j] < ki) ? 1 : 0;
    srt[rank] = ((ki >> 20) << 16) | (unsigned)rank;
    insname[rank] = kname[i];
    insfirst[rank] = kfirst[i];
  }
  if (tid == 0) {
    int P = 0;
    for (int k = 0; k < B; k++) {
      cidx[k] = (unsigned short)P;
      if ((posm[k >> 5] >> (k & 31)) & 1u) {
        db.pos[m.pos0 + P] = k;
        P++;
      }
    }
    s_P = P;
  }
  __syncthreads();
  // ReadSet::sort(): std::sort on firstPosition, in the rank-based form one wavefront can run
  // (gnusort.h; the same routine k_sortperm uses)
  if (tid < 64) wave_sort_hi16(srt, n, tid, posL, posR, cutbit, stmp, s_stk);
  __syncthreads();

  // ---- phase E: compact variant bit-planes of the final reads --------------
  const int P = s_P;
  const int Wp = (P + 63) >> 6;
  for (int i = tid; i < n; i += kBlock) {
    const int g = insname[srt[i] & 0xffffu];
    const int gfirst = insfirst[srt[i] & 0xffffu];
    db.fr_name[m.fr0 + i] = g;
    unsigned long long cw = 0, pl0 = 0, pl1 = 0, pl2 = 0, pl3 = 0;
    int curw = 0;
    for (int k = 0; k <= B; k++) {
      int j = -1, a = 0;
      if (k < B) {
        if (!((posm[k >> 5] >> (k & 31)) & 1u)) continue;
        j = cidx[k];
        int cr;
        a = k < gfirst ? -1 : merged_allele(vpart, R, k, name_off, name_aln, g, &cr, fl, k != gfirst);
      }
      const int wj = (k < B) ? (j >> 6) : Wp;  // k == B flushes the tail
      while (curw < wj) {
        const long long o = m.cov0 + (long long)curw * G + i;
        db.cover[o] = cw;
        db.alle[o] = pl0;
        if (db.planes > 1) db.alle[db.alle_plane_stride + o] = pl1;
        if (db.planes > 2) db.alle[2 * db.alle_plane_stride + o] = pl2;
        if (db.planes > 3) db.alle[3 * db.alle_plane_stride + o] = pl3;
        cw = pl0 = pl1 = pl2 = pl3 = 0;
        curw++;
      }
      if (k < B && a >= 0) {
        const unsigned long long bit = 1ull << (j & 63);
        cw |= bit;
        if (a & 1) pl0 |= bit;
        if (a & 2) pl1 |= bit;
        if (a & 4) pl2 |= bit;
        if (a & 8) pl3 |= bit;
      }
    }
  }
  if (tid == 0) {
    db.status[c] = AHSOKA_ST_PHASED;
    db.nfin[c] = n;
    db.npos[c] = P;
  }
}

size_t readset_lds_bytes(const ReadsetLds& l) {
  size_t b = readset_region_bytes(l);
  b += (size_t)4 * ((l.maxB + 32) >> 5) * 4;  // tba, firsts, lasts, posm
  b += (size_t)(l.maxB + 2) * 2;              // cidx
  return (b + 15) & ~(size_t)15;
}

void launch_readset(const DeviceBatch& db, const int* list, int count, const ReadsetLds& l, hipStream_t stream) {
  if (count > 0) hipLaunchKernelGGL(k_readset, dim3(count), dim3(kBlock), readset_lds_bytes(l), stream, db, l, list);
}

// ===========================================================================
// k_readset_lds: the same stages for chains whose alignment bitsets fit the LDS
// as a whole (all of BASELINE configs 1-4).  Differences from k_readset:
//  * the chain's read-path CSR is streamed from HBM ONCE, coalesced (consecutive
//    lanes = consecutive entries; the alignment an entry belongs to comes from a
//    binary search in the chain's offsets, staged in LDS), into one node bitset
//    per alignment in LDS (ds_or_b64);
//  * no R x B match matrices in HBM: "first matching allele" of (alignment,
//    bubble) is recomputed from the LDS bitsets by every pass that needs it (S1,
//    S2 statistics, covered positions, final bit-planes) -- k_readset writes two
//    R x B byte matrices and gathers them back four times, ~6x this chain's
//    input bytes; here HBM sees the inputs once and the final bit-planes once;
//  * the matrices are still written, position-major and coalesced, when the
//    caller asked for the stage dumps (want_stages).
// Everything after the matching (filters, boundary sets, creation order, the
// reference's std::sort, bit-planes) is the code of k_readset.
// ===========================================================================
struct MatchCtx {
  const unsigned long long* bits;   // [wordsN][R] node bitset of every alignment
  const unsigned long long* mfull;  // [SA][wordsN] node mask of every allele path
  const unsigned long long* minner; // [SA][wordsN] ... without its first and last node
  const unsigned short* ahead;      // [B+1] first allele of every bubble (chain-local allele index)
  const unsigned char* adup;        // [SA] path repeats a node: exact multiset test on the node lists
  const unsigned char* bwin;        // [B] first | last << 4 of the bitset words the bubble's allele paths touch (wordsN <= 4)
  int R, wordsN;
  const DeviceBatch* db;
  long long a_base;       // global index of the chain's first allele
  long long e_base;       // first rnode entry of the chain
  const unsigned int* roff;  // [R+1] entry offsets relative to e_base
};

// the node bitset of one alignment: in registers when the chain has at most 256 nodes
struct AlnBits {
  unsigned long long w[4];
  int r;
};
__device__ __forceinline__ AlnBits load_bits(const MatchCtx& mc, int r) {
  AlnBits ab;
  ab.r = r;
#pragma unroll
  for (int w = 0; w < 4; w++) ab.w[w] = w < mc.wordsN ? mc.bits[(size_t)w * mc.R + r] : ~0ull;
  return ab;
}

// first allele of bubble k that the alignment matches fully (PARTIAL = false: all path nodes)
// or partially (inner nodes only); -1 if none.  k is wave-uniform: mask reads are LDS broadcasts.
template <bool PARTIAL>
__device__ __forceinline__ int match_allele(const MatchCtx& mc, const AlnBits& ab, int k) {
  const unsigned long long* mk = PARTIAL ? mc.minner : mc.mfull;
  const int a0 = mc.ahead[k], a1 = mc.ahead[k + 1];
  for (int a = a0; a < a1; a++) {
    bool ok;
    if (mc.adup[a]) {  // path repeats a node: exact multiset containment on the node list
      const DeviceBatch& db = *mc.db;
      const long long p0 = db.apath_off[mc.a_base + a], p1 = db.apath_off[mc.a_base + a + 1];
      const long long e0 = mc.e_base + mc.roff[ab.r], e1 = mc.e_base + mc.roff[ab.r + 1];
      if (PARTIAL) ok = p1 - p0 >= 2 ? multiset_includes(db.apath_node, p0 + 1, p1 - 1, db.rnode, e0, e1) : true;
      else ok = multiset_includes(db.apath_node, p0, p1, db.rnode, e0, e1);
    } else if (mc.wordsN <= 4) {
      unsigned long long miss = 0ull;
#pragma unroll
      for (int w = 0; w < 4; w++)
        if (w < mc.wordsN) miss |= mk[(size_t)a * mc.wordsN + w] & ~ab.w[w];
      ok = miss == 0ull;
    } else {
      unsigned long long miss = 0ull;
      for (int w = 0; w < mc.wordsN; w++) miss |= mk[(size_t)a * mc.wordsN + w] & ~mc.bits[(size_t)w * mc.R + ab.r];
      ok = miss == 0ull;
    }
    if (ok) return a - a0;
  }
  return -1;
}

// One read name = one Read: its alignments (usually exactly one).  The name's entry of the
// per-chain CSR and, for a single alignment, that alignment's bitset are loaded ONCE per
// pass, not per bubble (a global load per bubble made every pass latency-bound).
struct NameAln {
  long long n0, n1;  // range in name_aln
  int r;             // the alignment if n1 - n0 == 1
  AlnBits ab;
};
__device__ __forceinline__ NameAln load_name(const MatchCtx& mc, const long long* __restrict__ name_off,
                                             const int* __restrict__ name_aln, int g) {
  NameAln na;
  na.n0 = name_off[g];
  na.n1 = name_off[g + 1];
  na.r = na.n1 - na.n0 == 1 ? name_aln[na.n0] : -1;
  na.ab = load_bits(mc, na.r < 0 ? 0 : na.r);
  return na;
}

// merged "first matching allele" of a read name at position k (see merged_allele above)
template <bool PARTIAL>
__device__ __forceinline__ int merged_match(const MatchCtx& mc, int k, const NameAln& na, const int* __restrict__ name_aln,
                                            int* creator, const unsigned char* __restrict__ flags = nullptr,
                                            bool need_gt90 = false) {
  if (na.r >= 0) {  // the common case
    *creator = na.r;
    if (need_gt90 && !(flags[na.r] & 1)) { *creator = -1; return -1; }
    const int a = match_allele<PARTIAL>(mc, na.ab, k);
    if (a < 0) *creator = -1;
    return a;
  }
  int best_a = 127, best_r = -1;
  for (long long ai = na.n0; ai < na.n1; ai++) {
    const int r = name_aln[ai];
    if (need_gt90 && !(flags[r] & 1)) continue;
    const AlnBits ab = load_bits(mc, r);
    const int a = match_allele<PARTIAL>(mc, ab, k);
    if (a >= 0 && a < best_a) {
      best_a = a;
      best_r = r;
    }
  }
  *creator = best_r;
  return best_r >= 0 ? best_a : -1;
}

// Four read names of one lane at a time.  The passes over (name, bubble) are chains of
// dependent LDS reads (allele range -> masks -> compare); interleaving four names gives the
// lane four independent chains and lets one load of an allele mask serve four tests.  Valid for
// the common shape only: every name of the group has exactly one alignment, no allele path of
// the chain repeats a node, at most 256 nodes (bitsets in registers); otherwise the per-name
// code runs.  res[j] = first matching allele of name j at bubble k or -1.
// nab[j] = the COMPLEMENT of name j's node bitset (an allele matches when mask & ~bits == 0).  The nodes of
// one bubble are neighbours in the chain-local numbering, so all of its allele paths touch one bitset word
// (two where the bubble straddles a 64-node boundary): bwin[k] names the words, the test of an allele is
// then one 64-bit AND per name instead of wordsN of them (the match passes are bound by these instructions).
template <bool PARTIAL>
__device__ __forceinline__ void match4(const MatchCtx& mc, const unsigned long long (&nab)[4][4], int k, int (&res)[4]) {
  const unsigned long long* mk = PARTIAL ? mc.minner : mc.mfull;
  const int a0 = mc.ahead[k], a1 = mc.ahead[k + 1];
  const int win = __builtin_amdgcn_readfirstlane((int)mc.bwin[k]);
  const int wlo = win & 15, whi = win >> 4;
#pragma unroll
  for (int j = 0; j < 4; j++) res[j] = -1;
  if (wlo >= whi) {  // one word (or no node at all): uniform
    unsigned long long s[4];
#pragma unroll
    for (int j = 0; j < 4; j++) s[j] = wlo == 0 ? nab[j][0] : wlo == 1 ? nab[j][1] : wlo == 2 ? nab[j][2] : nab[j][3];
    for (int a = a0; a < a1; a++) {
      const unsigned long long mw = mk[(size_t)a * mc.wordsN + wlo];
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (res[j] < 0 && (mw & s[j]) == 0ull) res[j] = a - a0;
    }
    return;
  }
  for (int a = a0; a < a1; a++) {
    unsigned long long mw[4];
#pragma unroll
    for (int w = 0; w < 4; w++) mw[w] = w < mc.wordsN ? mk[(size_t)a * mc.wordsN + w] : 0ull;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const unsigned long long miss = (mw[0] & nab[j][0]) | (mw[1] & nab[j][1]) | (mw[2] & nab[j][2]) | (mw[3] & nab[j][3]);
      if (res[j] < 0 && miss == 0ull) res[j] = a - a0;
    }
  }
}

__global__ __launch_bounds__(kBlock, 3) void k_readset_lds(DeviceBatch db, const int* __restrict__ list) {
  extern __shared__ __align__(16) unsigned long long smem[];
  const int c = list[blockIdx.x];
  const ChainMeta m = db.meta[c];
  const int tid = threadIdx.x;
  const int B = m.B, R = m.R, G = m.G;

  // the LDS arrays are laid out from THIS chain's sizes (readset_lds2_bytes of the same sizes is what
  // the host checked against the budget; a launch asks for the largest need among its chains)
  const int wordsN = (m.N + 63) >> 6;
  const long long a_base = db.allele_off[m.b0];
  const int SA = (int)(db.allele_off[m.b0 + B] - a_base);
  const int Gs = G > 0 ? G : 1, Bs = B > 0 ? B : 1;
  const int PW = (Bs + 32) >> 5;                                        // u32 words of a position mask
  const int capN = Gs;                                                  // (slots of the per-read arrays)
  unsigned long long* bits = smem;                                     // [wordsN][R]
  unsigned long long* mfull = bits + (size_t)wordsN * R;               // [SA][wordsN]
  unsigned long long* minner = mfull + (size_t)wordsN * SA;            // [SA][wordsN]
  unsigned int* roff = (unsigned int*)(minner + (size_t)wordsN * SA);  // [R + 1]
  unsigned int* tba = roff + ((R + 2) & ~1);                            // [PW]
  unsigned int* firsts = tba + PW;
  unsigned int* lasts = firsts + PW;
  unsigned int* posm = lasts + PW;
  // [G], 16-byte aligned (by element arithmetic: a round trip through an integer would turn every array behind it into a
  // generic pointer, i.e. flat_* instead of ds_* instructions -- 110 of them in this kernel when it was written that way)
  unsigned int* keys = posm + PW + ((4 - ((int)((posm + PW) - (unsigned int*)smem) & 3)) & 3);
  unsigned int* srt = keys + Gs;           // [G]
  unsigned short* kname = (unsigned short*)(srt + Gs);  // [G]
  unsigned short* insname = kname + Gs;                 // [G]
  unsigned short* kfirst = insname + Gs;                // [G]
  unsigned short* insfirst = kfirst + Gs;               // [G]
  unsigned short* cidx = insfirst + Gs;                 // [B+1]
  unsigned short* ahead = cidx + ((Bs + 2) & ~1);       // [B+1]
  unsigned char* adup = (unsigned char*)(ahead + ((Bs + 2) & ~1));  // [SA]
  unsigned char* bwin = adup + ((SA + 15) & ~15);                    // [B] word window of the bubble's allele paths
  __shared__ int s_maxpos, s_any, s_n, s_P, s_anydup;

  for (int w = tid; w < PW; w += kBlock) { tba[w] = 0; firsts[w] = 0; lasts[w] = 0; posm[w] = 0; }
  if (tid == 0) { s_maxpos = -1; s_any = 0; s_n = 0; s_P = 0; s_anydup = 0; }

  if (B <= 1 || (m.flags & 1)) {  // :86 (<=1 bubble) or a compiled-in limit  (uniform)
    if (tid == 0) {
      db.status[c] = (m.flags & 1) ? AHSOKA_ST_UNSUPPORTED : AHSOKA_ST_SKIPPED_LE1;
      db.nfin[c] = 0;
      db.npos[c] = 0;
    }
    return;
  }
  // the scalars above are written by other waves from here on (s_anydup in the mask loop below,
  // s_maxpos / s_any / s_n later): their initialisation must have landed first
  __syncthreads();

#ifdef AHK_RS_TIMING
  long long rs_t[8], rs_last = (long long)clock64();
  int rs_i = 0;
#define RS_T() do { const long long _t = (long long)clock64(); rs_t[rs_i++] = _t - rs_last; rs_last = _t; } while (0)
#else
#define RS_T() do {} while (0)
#endif
  // ---- phase A: the read-path CSR, once, coalesced -> one node bitset per alignment ----
  const long long e_base = db.rnode_off[m.r0];
  for (int i = tid; i < wordsN * R; i += kBlock) bits[i] = 0ull;
  for (int r = tid; r <= R; r += kBlock) roff[r] = (unsigned int)(db.rnode_off[m.r0 + r] - e_base);
  // the allele paths as node masks (full path / inner nodes), once per chain
  for (int k = tid; k <= B; k += kBlock) ahead[k] = (unsigned short)(db.allele_off[m.b0 + k] - a_base);
  unsigned short* s1l = cidx;  // S1 visiting order (phase B only; cidx is written in phase D)
  for (int k = tid; k < B; k += kBlock) s1l[k] = (unsigned short)db.s1_order[m.b0 + k];
  for (int a = tid; a < SA; a += kBlock) {
    for (int w = 0; w < wordsN; w++) {
      mfull[(size_t)a * wordsN + w] = 0ull;
      minner[(size_t)a * wordsN + w] = 0ull;
    }
    const long long p0 = db.apath_off[a_base + a], p1 = db.apath_off[a_base + a + 1];
    adup[a] = db.apath_dup[a_base + a];
    if (adup[a]) s_anydup = 1;
    for (long long e = p0; e < p1; e++) {
      const int nd = db.apath_node[e];
      mfull[(size_t)a * wordsN + (nd >> 6)] |= 1ull << (nd & 63);
      if (e > p0 && e < p1 - 1) minner[(size_t)a * wordsN + (nd >> 6)] |= 1ull << (nd & 63);
    }
  }
  __syncthreads();
#ifdef AHK_RS_TIMING
  rs_t[6] = (long long)clock64() - rs_last;  // phase A: arrays zeroed, offsets and allele masks in LDS
#endif
  // the bitset words every bubble's allele paths touch (see match4)
  for (int k = tid; k < B; k += kBlock) {
    int lo = 15, hi = 0;
    for (int a = ahead[k]; a < ahead[k + 1]; a++)
      for (int w = 0; w < wordsN && w < 15; w++)
        if (mfull[(size_t)a * wordsN + w] != 0ull) {
          lo = w < lo ? w : lo;
          hi = w > hi ? w : hi;
        }
    bwin[k] = (unsigned char)(lo > hi ? 0 : (lo | (hi << 4)));
  }
  {
    // each wave streams one contiguous quarter of the chain's entries, 4 x 64 per trip (the four
    // loads are in flight together); the alignment of an entry is found by walking forward
    // from the previous entry's (entries of a lane are 256 apart: a few alignments further)
    const unsigned int E = roff[R];
    const int wv = tid >> 6, lane = tid & 63;
    const unsigned int q0 = (unsigned int)(((unsigned long long)E * wv) >> 2), q1 = (unsigned int)(((unsigned long long)E * (wv + 1)) >> 2);
    int rcur = 0;
    {  // the alignment of this wave's first entry
      int lo = 0, hi = R;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (roff[mid] <= q0) lo = mid;
        else hi = mid;
      }
      rcur = lo;
    }
    for (unsigned int e0 = q0; e0 < q1; e0 += 256) {
      int nd[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const unsigned int e = e0 + 64 * c + lane;
        nd[c] = e < q1 ? db.rnode[e_base + e] : -1;
      }
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const unsigned int e = e0 + 64 * c + lane;
        if (e >= q1) continue;
        while (rcur + 1 < R && roff[rcur + 1] <= e) rcur++;
        if (nd[c] >= 0) atomicOr(&bits[(size_t)(nd[c] >> 6) * R + rcur], 1ull << (nd[c] & 63));  // (-1: a node of another chain)
      }
    }
  }
  __syncthreads();

  RS_T();  // 0: phase A
  MatchCtx mc;
  mc.bits = bits;
  mc.mfull = mfull;
  mc.minner = minner;
  mc.ahead = ahead;
  mc.adup = adup;
  mc.bwin = bwin;
  mc.R = R;
  mc.wordsN = wordsN;
  mc.db = &db;
  mc.a_base = a_base;
  mc.e_base = e_base;
  mc.roff = roff;

  if (m.vmat0 >= 0) {  // stage dump: the raw outcome for every (alignment, bubble), position-major
    signed char* vfull = db.vfull + m.vmat0;
    signed char* vpart = db.vpart + m.vmat0;
    for (int r = tid; r < R; r += kBlock) {
      const AlnBits ab = load_bits(mc, r);
      for (int k = 0; k < B; k++) {
        vfull[(long long)k * R + r] = (signed char)match_allele<false>(mc, ab, k);
        vpart[(long long)k * R + r] = (signed char)match_allele<true>(mc, ab, k);
      }
    }
  }

  const long long* name_off = db.name_off + m.name0;
  const int* name_aln = db.name_aln + m.r0;

  // ---- phase B: S1b filter and S1c boundary sets ---------------------------
  // S1 walks the bubbles in the iteration order of the reference's unordered_map
  // (:76,:90): the first hit IN THAT ORDER creates the Read and fixes its mapq.
  const bool quad_ok = !s_anydup && wordsN <= 4;  // (uniform) the four-names-at-a-time form applies
  for (int g0 = tid; g0 < G; g0 += 4 * kBlock) {
    NameAln na[4];
    bool val[4];
    bool quad = quad_ok;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int g = g0 + j * kBlock;
      val[j] = g < G;
      na[j] = load_name(mc, name_off, name_aln, val[j] ? g : g0);
      quad = quad && na[j].r >= 0;
    }
    int nvar[4], first[4], last[4], mapq[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { nvar[j] = 0; first[j] = B; last[j] = -1; mapq[j] = -1; }
    if (quad) {
      unsigned long long ab[4][4];  // complemented bitsets
      int mq[4];  // (a single alignment per name: the creator is known up front)
#pragma unroll
      for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int w = 0; w < 4; w++) ab[j][w] = ~na[j].ab.w[w];
        mq[j] = db.read_mapq[m.r0 + na[j].r];
      }
      for (int kk = 0; kk < B; kk++) {
        const int k = s1l[kk];
        int res[4];
        match4<false>(mc, ab, k, res);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (res[j] < 0) continue;
          mapq[j] = mq[j];
          nvar[j]++;
          first[j] = k < first[j] ? k : first[j];
          last[j] = k > last[j] ? k : last[j];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (!val[j]) continue;
        for (int kk = 0; kk < B; kk++) {
          const int k = s1l[kk];
          int cr;
          int a = merged_match<false>(mc, k, na[j], name_aln, &cr);
          if (a < 0) continue;
          if (nvar[j] == 0) mapq[j] = db.read_mapq[m.r0 + cr];
          nvar[j]++;
          first[j] = k < first[j] ? k : first[j];
          last[j] = k > last[j] ? k : last[j];
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (!val[j] || nvar[j] == 0) continue;
      atomicMax(&s_maxpos, last[j]);
      s_any = 1;
      if (nvar[j] >= db.min_variants && mapq[j] >= db.min_mapq) {
        atomicOr(&firsts[first[j] >> 5], 1u << (first[j] & 31));
        atomicOr(&lasts[last[j] >> 5], 1u << (last[j] & 31));
      }
    }
  }
  __syncthreads();
  if (!s_any) {  // reference: back() of an empty vector, UB (quirk Q4)
    if (tid == 0) { db.status[c] = AHSOKA_ST_NO_FULL_MATCH; db.nfin[c] = 0; db.npos[c] = 0; }
    return;
  }
  if (tid == 0) {
    // boundaries = {el, el+1 : el in lasts \ firsts}; to_be_added = [0,maxpos) U boundaries
    unsigned int carry = 0;
    for (int w = 0; w < PW; w++) {
      unsigned int bd = lasts[w] & ~firsts[w];
      unsigned int t = bd | (bd << 1) | carry;
      carry = bd >> 31;
      int lo = w * 32;
      unsigned int range = 0;
      if (s_maxpos > lo) range = (s_maxpos - lo >= 32) ? 0xffffffffu : ((1u << (s_maxpos - lo)) - 1u);
      t |= range;
      unsigned int valid = 0;  // positions >= B have no alleles (:216)
      if (B > lo) valid = (B - lo >= 32) ? 0xffffffffu : ((1u << (B - lo)) - 1u);
      tba[w] = t & valid;
    }
  }
  __syncthreads();

  RS_T();  // 1: phase B
  // ---- phase C: S2 variants restricted to to_be_added, S2b filter ----------
  const unsigned char* fl = db.read_flags + m.r0;
  for (int g0 = tid; g0 < G; g0 += 4 * kBlock) {
    NameAln na[4];
    bool val[4];
    bool quad = quad_ok;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int g = g0 + j * kBlock;
      val[j] = g < G;
      na[j] = load_name(mc, name_off, name_aln, val[j] ? g : g0);
      quad = quad && na[j].r >= 0;
    }
    int nvar[4], first[4], afirst[4], rcre[4], mapq[4];
    unsigned long long cv0[4], cv1[4];  // covered positions below 128 (the rest: second loop)
#pragma unroll
    for (int j = 0; j < 4; j++) { nvar[j] = 0; first[j] = -1; afirst[j] = 0; rcre[j] = 0; mapq[j] = -1; cv0[j] = 0ull; cv1[j] = 0ull; }
    if (quad) {
      unsigned long long ab[4][4];  // complemented bitsets
      bool gt90[4];
      int mq[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int w = 0; w < 4; w++) ab[j][w] = ~na[j].ab.w[w];
        gt90[j] = (fl[na[j].r] & 1) != 0;
        mq[j] = db.read_mapq[m.r0 + na[j].r];
      }
      for (int k = 0; k < B; k++) {
        if (!tba_test(tba, k)) continue;
        int res[4];
        match4<true>(mc, ab, k, res);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          // the first hit creates the Read (any identity); later positions need id*100 > 90 (:245)
          if (res[j] < 0 || (first[j] >= 0 && !gt90[j])) continue;
          nvar[j]++;
          if (first[j] < 0) { first[j] = k; afirst[j] = res[j]; rcre[j] = na[j].r; mapq[j] = mq[j]; }
          if (k < 64) cv0[j] |= 1ull << k;
          else if (k < 128) cv1[j] |= 1ull << (k - 64);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (!val[j]) continue;
        for (int k = 0; k < B; k++) {
          if (!tba_test(tba, k)) continue;
          int cr;
          int a = merged_match<true>(mc, k, na[j], name_aln, &cr, fl, first[j] >= 0);
          if (a < 0) continue;
          nvar[j]++;
          if (first[j] < 0) { first[j] = k; afirst[j] = a; rcre[j] = cr; mapq[j] = db.read_mapq[m.r0 + cr]; }
          if (k < 64) cv0[j] |= 1ull << k;
          else if (k < 128) cv1[j] |= 1ull << (k - 64);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (!val[j] || !(nvar[j] >= db.min_variants && mapq[j] >= db.min_mapq)) continue;
      const int g = g0 + j * kBlock;
      int slot = atomicAdd(&s_n, 1);
      // creation order of the Read objects in partial_readset: (position, allele, alignment)
      if (slot < capN) {  // (a readset beyond kMaxReads is reported below: its reads need no slot)
        keys[slot] = ((unsigned)first[j] << 20) | ((unsigned)afirst[j] << 16) | (unsigned)rcre[j];  // (12-bit position, 4-bit allele, 16-bit alignment)
        kname[slot] = (unsigned short)g;
        kfirst[slot] = (unsigned short)first[j];
      }
      if ((unsigned int)cv0[j]) atomicOr(&posm[0], (unsigned int)cv0[j]);
      if ((unsigned int)(cv0[j] >> 32)) atomicOr(&posm[1], (unsigned int)(cv0[j] >> 32));
      if ((unsigned int)cv1[j]) atomicOr(&posm[2], (unsigned int)cv1[j]);
      if ((unsigned int)(cv1[j] >> 32)) atomicOr(&posm[3], (unsigned int)(cv1[j] >> 32));
      for (int k = first[j] > 128 ? first[j] : 128; k < B; k++) {
        if (!tba_test(tba, k)) continue;
        int cr;
        if (merged_match<true>(mc, k, na[j], name_aln, &cr, fl, k != first[j]) >= 0)
          atomicOr(&posm[k >> 5], 1u << (k & 31));
      }
    }
  }
  __syncthreads();
  const int n = s_n;
  if (n == 0 || n > kMaxReads) {  // :276-279 warning + continue
    if (tid == 0) {
      db.status[c] = n == 0 ? AHSOKA_ST_EMPTY_READSET : AHSOKA_ST_UNSUPPORTED;
      db.nfin[c] = 0;
      db.npos[c] = 0;
    }
    return;
  }

  RS_T();  // 2: phase C
  // ---- phase D: insertion order (rank of the creation key), then S3 --------
  for (int i = tid; i < n; i += kBlock) {
    unsigned int ki = keys[i];
    int rank = 0;
    int j = 0;
    for (; j + 4 <= n; j += 4) {  // (keys is 16-byte aligned: four keys per LDS read, all lanes the same address)
      const uint4 k4 = *reinterpret_cast<const uint4*>(keys + j);
      rank += (k4.x < ki ? 1 : 0) + (k4.y < ki ? 1 : 0) + (k4.z < ki ? 1 : 0) + (k4.w < ki ? 1 : 0);
    }
    for (; j < n; j++) rank += (keys[j] < ki) ? 1 : 0;
    srt[rank] = ((ki >> 20) << 16) | (unsigned)rank;
    insname[rank] = kname[i];
    insfirst[rank] = kfirst[i];
  }
  if (tid == 0) {
    int P = 0;
    for (int k = 0; k < B; k++) {
      cidx[k] = (unsigned short)P;
      if ((posm[k >> 5] >> (k & 31)) & 1u) {
        db.pos[m.pos0 + P] = k;
        P++;
      }
    }
    s_P = P;
  }
  __syncthreads();
  RS_T();  // 3: ranks + position scan
  // ReadSet::sort() -- the reference's std::sort, whose tie order is the very sequence of its
  // comparisons: inherently one lane's work (~0.3 ms for 600 reads) -- is NOT done here, where
  // it would hold 50 KB of LDS and 255 idle lanes hostage: the sort keys and the bit-planes go
  // out in CREATION order (rank i) and k_sortperm, one wavefront per chain at 8x the
  // occupancy, sorts and permutes them into the final order.
  for (int i = tid; i < n; i += kBlock) db.sortkey[m.fr0 + i] = srt[i];
  RS_T();  // 4: (sort: k_sortperm)

  // ---- phase E: compact variant bit-planes of the reads, in creation order --------------
  const int P = s_P;
  const int Wp = (P + 63) >> 6;
  for (int i0 = tid; i0 < n; i0 += 4 * kBlock) {
    NameAln na[4];
    bool val[4];
    int gfirst[4];
    bool quad = quad_ok && db.planes <= 2;  // (two bit-planes: up to four alleles per bubble; more take the per-name code)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int i = i0 + j * kBlock;
      val[j] = i < n;
      const int g = insname[val[j] ? i : i0];
      gfirst[j] = insfirst[val[j] ? i : i0];
      if (val[j]) db.name_tmp[m.fr0 + i] = g;
      na[j] = load_name(mc, name_off, name_aln, g);
      quad = quad && na[j].r >= 0;
    }
    if (quad) {
      unsigned long long ab[4][4];  // complemented bitsets
      bool gt90[4];
      unsigned long long cw[4], p0[4], p1[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int w = 0; w < 4; w++) ab[j][w] = ~na[j].ab.w[w];
        gt90[j] = (fl[na[j].r] & 1) != 0;
        cw[j] = p0[j] = p1[j] = 0ull;
      }
      int curw = 0;
      for (int k = 0; k <= B; k++) {
        int jp = -1;
        int res[4] = {-1, -1, -1, -1};
        if (k < B) {
          if (!((posm[k >> 5] >> (k & 31)) & 1u)) continue;
          jp = cidx[k];
          match4<true>(mc, ab, k, res);
        }
        const int wj = (k < B) ? (jp >> 6) : Wp;  // k == B flushes the tail
        while (curw < wj) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (val[j]) {
              const long long o = m.cov0 + (long long)curw * G + (i0 + j * kBlock);
              db.cover_tmp[o] = cw[j];
              db.alle_tmp[o] = p0[j];
              if (db.planes > 1) db.alle_tmp[db.alle_plane_stride + o] = p1[j];
            }
            cw[j] = p0[j] = p1[j] = 0ull;
          }
          curw++;
        }
        if (k < B) {
          const unsigned long long bit = 1ull << (jp & 63);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            int a = k < gfirst[j] ? -1 : res[j];
            if (k != gfirst[j] && !gt90[j]) a = -1;  // a later variant needs id*100 > 90 (:245)
            if (a < 0) continue;
            cw[j] |= bit;
            if (a & 1) p0[j] |= bit;
            if (a & 2) p1[j] |= bit;
          }
        }
      }
      continue;
    }
#pragma unroll 1
    for (int j = 0; j < 4; j++) {  // (no indexing of the per-group arrays here: a dynamic index would put them in scratch)
      const int i = i0 + j * kBlock;
      if (i >= n) break;
      const int gf = insfirst[i];
      const NameAln nj = load_name(mc, name_off, name_aln, insname[i]);
      unsigned long long cw = 0, pl0 = 0, pl1 = 0, pl2 = 0, pl3 = 0;
      int curw = 0;
      for (int k = 0; k <= B; k++) {
        int jp = -1, a = 0;
        if (k < B) {
          if (!((posm[k >> 5] >> (k & 31)) & 1u)) continue;
          jp = cidx[k];
          int cr;
          a = k < gf ? -1 : merged_match<true>(mc, k, nj, name_aln, &cr, fl, k != gf);
        }
        const int wj = (k < B) ? (jp >> 6) : Wp;  // k == B flushes the tail
        while (curw < wj) {
          const long long o = m.cov0 + (long long)curw * G + i;
          db.cover_tmp[o] = cw;
          db.alle_tmp[o] = pl0;
          if (db.planes > 1) db.alle_tmp[db.alle_plane_stride + o] = pl1;
          if (db.planes > 2) db.alle_tmp[2 * db.alle_plane_stride + o] = pl2;
          if (db.planes > 3) db.alle_tmp[3 * db.alle_plane_stride + o] = pl3;
          cw = pl0 = pl1 = pl2 = pl3 = 0;
          curw++;
        }
        if (k < B && a >= 0) {
          const unsigned long long bit = 1ull << (jp & 63);
          cw |= bit;
          if (a & 1) pl0 |= bit;
          if (a & 2) pl1 |= bit;
          if (a & 4) pl2 |= bit;
          if (a & 8) pl3 |= bit;
        }
      }
    }
  }
  RS_T();  // 5: phase E
#ifdef AHK_RS_TIMING
  if (tid == 0 && blockIdx.x == 0)
    printf("k_readset_lds chain %d R=%d n=%d cycles: A=%lld (masks and offsets staged after %lld) B=%lld C=%lld rank=%lld sort=%lld E=%lld\n", c, R, n, rs_t[0],
           rs_t[6], rs_t[1], rs_t[2], rs_t[3], rs_t[4], rs_t[5]);
#endif
  if (tid == 0) {
    db.status[c] = AHSOKA_ST_PHASED;
    db.nfin[c] = n;
    db.npos[c] = P;
  }
}

size_t readset_lds2_bytes(const ReadsetLds2& l) {
  size_t b = (size_t)l.maxBitWords * 8;       // bitsets
  b += (size_t)l.maxMaskWords * 8 * 2;        // allele masks, full and inner
  b += (size_t)((l.maxB + 2) & ~1) * 2 * 2;   // cidx padding, ahead
  b += (size_t)((l.maxSA + 15) & ~15);        // adup
  b += (size_t)((l.maxB + 15) & ~15);         // bwin
  b += (size_t)((l.maxR + 2) & ~1) * 4;       // roff
  b += (size_t)4 * ((l.maxB + 32) >> 5) * 4 + 16;  // tba, firsts, lasts, posm (+ the alignment of keys)
  b += (size_t)l.maxG * 4 * 2;                // keys, srt
  b += (size_t)l.maxG * 2 * 4;                // kname, insname, kfirst, insfirst
  b += (size_t)(l.maxB + 2) * 2;              // cidx
  return (b + 15) & ~(size_t)15;
}

// ReadSet::sort() and the permutation of the creation-ordered outputs of k_readset_lds into
// the sorted order: one wavefront per chain.  The wave runs the reference's std::sort in its
// rank-based form (wave_sort_hi16) on the keys in LDS; then it moves names and bit-plane words,
// lanes = reads (coalesced stores, gathered loads of data that was just written).
__global__ __launch_bounds__(64) void k_sortperm(DeviceBatch db, const int* __restrict__ list, int maxG) {
  extern __shared__ __align__(16) unsigned long long smem[];
  __shared__ int s_stk[3 * 64];
  const int c = list[blockIdx.x];
  if (db.status[c] != AHSOKA_ST_PHASED) return;
  const ChainMeta m = db.meta[c];
  const int lane = threadIdx.x;
  const int n = db.nfin[c], G = m.G;
  const int cap = (maxG + 1) & ~1;
  unsigned int* srt = (unsigned int*)smem;            // [cap]
  unsigned int* tmp = srt + cap;                      // [cap]
  unsigned int* cutbit = tmp + cap;                   // [cap/32 + 2]
  unsigned short* posL = (unsigned short*)(cutbit + ((cap / 32 + 2 + 1) & ~1));  // [cap]
  unsigned short* posR = posL + cap;                  // [cap]
  for (int i = lane; i < n; i += 64) srt[i] = db.sortkey[m.fr0 + i];
  wave_lds_sync();
  wave_sort_hi16(srt, n, lane, posL, posR, cutbit, tmp, s_stk);
  const int Wp = (db.npos[c] + 63) >> 6;
  for (int i = lane; i < n; i += 64) {
    const int src = (int)(srt[i] & 0xffffu);  // creation rank of the read that sorts to place i
    db.fr_name[m.fr0 + i] = db.name_tmp[m.fr0 + src];
    for (int w = 0; w < Wp; w++) {
      const long long o = m.cov0 + (long long)w * G + i, q = m.cov0 + (long long)w * G + src;
      db.cover[o] = db.cover_tmp[q];
      for (int pl = 0; pl < db.planes; pl++) db.alle[pl * db.alle_plane_stride + o] = db.alle_tmp[pl * db.alle_plane_stride + q];
    }
  }
}

void launch_readset_lds(const DeviceBatch& db, const int* list, int count, size_t lds_bytes, int maxG, hipStream_t stream) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_readset_lds, dim3(count), dim3(kBlock), lds_bytes, stream, db, list);
  const size_t cap = (size_t)((maxG + 1) & ~1);  // srt + tmp (u32), cutbit, posL + posR (u16)
  hipLaunchKernelGGL(k_sortperm, dim3(count), dim3(64), cap * 12 + (cap / 32 + 4) * 4 + 32, stream, db, list, maxG);
}

}  // namespace ahk
