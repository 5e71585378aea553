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

__global__ __launch_bounds__(kBlock) void k_readset(DeviceBatch db, ReadsetLds lds) {
  extern __shared__ __align__(16) unsigned long long smem[];
  const int c = blockIdx.x;
  const ChainMeta m = db.meta[c];
  const int tid = threadIdx.x;
  const int B = m.B, R = m.R, G = m.G;

  unsigned long long* bits = smem;                                   // [wordsN][kBlock]
  unsigned int* tba = (unsigned int*)(bits + (size_t)lds.wordsN * kBlock);  // [32]
  unsigned int* firsts = tba + 32;
  unsigned int* lasts = firsts + 32;
  unsigned int* posm = lasts + 32;
  unsigned int* keys = posm + 32;          // [maxG]
  unsigned int* srt = keys + lds.maxG;     // [maxG]
  unsigned short* kname = (unsigned short*)(srt + lds.maxG);  // [maxG]
  unsigned short* insname = kname + lds.maxG;                 // [maxG]
  unsigned short* kfirst = insname + lds.maxG;                // [maxG]
  unsigned short* insfirst = kfirst + lds.maxG;               // [maxG]
  unsigned short* cidx = insfirst + lds.maxG;                 // [maxB+1]
  __shared__ int s_maxpos, s_any, s_n, s_P;

  if (tid < 32) { tba[tid] = 0; firsts[tid] = 0; lasts[tid] = 0; posm[tid] = 0; }
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
  for (int r = tid; r < R; r += kBlock) {
    for (int w = 0; w < wordsN; w++) bits[(size_t)w * kBlock + tid] = 0ull;
    const long long e0 = db.rnode_off[m.r0 + r], e1 = db.rnode_off[m.r0 + r + 1];
    for (long long e = e0; e < e1; e++) {
      int nd = db.rnode[e];
      if (nd >= 0) bits[(size_t)(nd >> 6) * kBlock + tid] |= 1ull << (nd & 63);
    }
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
        for (long long e = p0 + 1; e < p1 - 1; e++) {
          int nd = db.apath_node[e];
          inner = inner && ((bits[(size_t)(nd >> 6) * kBlock + tid] >> (nd & 63)) & 1ull);
        }
        bool full = inner;
        if (p1 > p0) {
          int nd = db.apath_node[p0];
          full = full && ((bits[(size_t)(nd >> 6) * kBlock + tid] >> (nd & 63)) & 1ull);
          if (p1 - p0 > 1) {
            nd = db.apath_node[p1 - 1];
            full = full && ((bits[(size_t)(nd >> 6) * kBlock + tid] >> (nd & 63)) & 1ull);
          }
        }
        if (vp < 0 && inner) vp = (int)(a - a0);
        if (vf < 0 && full) vf = (int)(a - a0);
      }
      vfull[(long long)k * R + r] = (signed char)vf;
      vpart[(long long)k * R + r] = (signed char)vp;
    }
  }
  __syncthreads();

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
    for (int w = 0; w < 32; w++) {
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
      keys[slot] = ((unsigned)first << 21) | ((unsigned)afirst << 17) | (unsigned)rcre;
      kname[slot] = (unsigned short)g;
      kfirst[slot] = (unsigned short)first;
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
    srt[rank] = ((ki >> 21) << 16) | (unsigned)rank;
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
  if (tid == 0) gnu_sort(srt, n, LessHi16());  // ReadSet::sort(): std::sort on firstPosition
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
  size_t b = (size_t)l.wordsN * kBlock * 8;   // bitsets
  b += 4 * 32 * 4;                            // tba, firsts, lasts, posm
  b += (size_t)l.maxG * 4 * 2;                // keys, srt
  b += (size_t)l.maxG * 2 * 4;                // kname, insname, kfirst, insfirst
  b += (size_t)(l.maxB + 2) * 2;              // cidx
  return (b + 15) & ~(size_t)15;
}

void launch_readset(const DeviceBatch& db, int C, const ReadsetLds& l, hipStream_t stream) {
  hipLaunchKernelGGL(k_readset, dim3(C), dim3(kBlock), readset_lds_bytes(l), stream, db, l);
}

}  // namespace ahk
