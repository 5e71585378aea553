// phase_kernels.h -- device-side data layout shared by the two kernels and the
// host code that launches them.  gfx950 only.
#pragma once
#include <stdint.h>

// Revision tag of the kernels' memory behaviour.  Bumped with every change of k_phase / k_readset
// that can move their HBM traffic or time: a committed counter summary (profiles/*_pmc_hbm.json)
// carries the tag it was taken with, and bench.py reports `roofline.traffic` only from a summary
// of the SAME revision (never a counter of one kernel version next to a time of another).
#define AHK_KERNEL_REV "r03.8"

namespace ahk {

// compiled-in device limits (a chain beyond them comes back AHSOKA_ST_UNSUPPORTED)
constexpr int kMaxBubbles = 4095;   // positions are 12-bit fields of sort keys (k_readset creation keys, S4a rate keys)
constexpr int kMaxNodes = 8192;     // chain-local node indices: k_readset's node window covers 2048 of them around an alignment
constexpr int kMaxNames = 8192;     // distinct read names per chain (k_readset LDS: 16 B per name)
constexpr int kMaxAlns = 65535;     // alignments per chain (16-bit field of the creation keys)
constexpr int kMaxReads = 4096;     // reads in the final readset (k_phase: 12-bit ends of the packed edge keys, 64-word bitmap rows)
constexpr int kPosWords = (kMaxBubbles + 32) / 32;  // u32 words of a position bit mask in LDS
constexpr int kWinWords = 32;       // u64 words of k_readset's per-alignment node window
constexpr int kMaxAlleles = 16;     // alleles per bubble (4 bit planes)
constexpr int kMaxPloidy = 4;       // tuples = (2p)^p <= 4096
constexpr int kBlock = 256;         // threads per workgroup (4 wavefronts)
constexpr size_t kReadsetFitBytes = 144 * 1024;  // LDS budget of k_readset_lds per chain (at least one workgroup per CU of 160 KB)

constexpr int32_t W_PERM = 0x7fffffff;
constexpr int32_t W_FORB = -0x7fffffff;  // -INT_MAX: its negation is W_PERM

struct alignas(16) IcPair {  // induced cost of forbidding / of making permanent
  long long f, p;
};
// the same in 32 bits (induced costs are sums of non-negative terms: unsigned), for chains whose induced
// costs provably stay below 2^32 (k_phase decides per chain from the row sums of |W|): half the sectors
// of every row scan of the induced costs
struct alignas(8) IcPair32 {
  unsigned int f, p;
};

// per-chain description, computed on the host at upload
struct ChainMeta {
  long long b0;        // first bubble (global index)
  long long r0;        // first alignment
  long long n0;        // first node
  long long name0;     // first entry of name_off (per-chain CSR name -> alignments)
  long long vmat0;     // offset of this chain's R*B match matrices, -1: none (k_readset_lds without stage dumps)
  long long fr0;       // offset of per-final-read arrays (capacity = G)
  long long cov0;      // offset of cover words (capacity G * Wb)
  long long pos0;      // offset of pos[] output (capacity B)
  long long dump0;     // offset into score dump (n*n) or -1
  int B, R, N, G;      // bubbles, alignments, nodes, distinct names
  int Wb;              // ceil(B/64): words of a position bit mask
  int flags;           // bit0: chain exceeds a device limit; bit1: alignment bitsets fit the LDS (k_readset_lds)
  int pad0, pad1;
};

struct DeviceBatch {
  // inputs (ChainBatch as uploaded)
  const long long* allele_off;
  const long long* apath_off;
  const int* apath_node;
  const unsigned char* apath_dup;  // [SA] 1 if the allele path repeats a node (multiset containment needed)
  const int* genotype;  // may be null
  const long long* rnode_off;
  const int* rnode;
  const int* read_mapq;
  const unsigned char* read_flags;  // [SR] bit0: float(id)*100 > 90 (reference :245)
  const int* s1_order;              // [SB] per chain: bubble ids in S1 visiting order (reference :76,90)
  // derived on host
  const ChainMeta* meta;
  const long long* name_off;  // [sum(G)+C] per-chain CSR
  const int* name_aln;        // [sum(R)] alignment index within chain
  // k_readset workspace / outputs
  signed char* vfull;  // [B][R] per chain, position-major
  signed char* vpart;
  int* status;         // [C]
  int* nfin;           // [C] reads in the final readset
  int* npos;           // [C] covered positions
  int* pos;            // [sum B]
  int* fr_name;        // [sum G]
  unsigned long long* cover;  // [sum G*Wb] word-major: cover[w*G + i]
  unsigned long long* alle;   // [planes][sum G*Wb]
  // k_readset_lds -> k_sortperm: the same in creation order, the sort keys, the names
  unsigned long long* cover_tmp;
  unsigned long long* alle_tmp;
  unsigned int* sortkey;      // [sum G]
  int* name_tmp;              // [sum G]
  long long alle_plane_stride;
  int planes;
  // k_phase outputs
  int* hap_allele;     // [ploidy][P] per chain at ploidy*pos0
  int* hap_cluster;
  int* fr_cluster;     // [sum G]
  int* nclusters;      // [C]
  long long* niter;    // [C]
  int* score_dump;     // optional
  // params
  int ploidy, min_mapq, min_variants, min_overlap, compaction_mode;
  double switch_cost, affine_cost;
};

struct SlotWorkspace {
  int* W;               // [slots][nmax*nmax]
  IcPair* IC;           // [slots][nmax*nmax]
  unsigned long long* live;  // [slots][nmax*ceil(nmax/64)] live-edge bitmap rows
  unsigned long long* nz;    // same shape: W[x][b] != 0 (finite or resolved)
  unsigned int* items;       // [slots][nmax*nmax] overflow of the per-iteration edge list
  double* rates;        // [slots][2*nmax]   same_i, diff_i
  // threading tables
  int* tp_k;            // [slots][Pmax]         clusters per position
  int* tp_gcid;         // [slots][Pmax*8]       covMap
  double* tp_cov;       // [slots][Pmax*8]       coverage as HaploThreader indexes it
  int* tp_cons;         // [slots][Pmax*8]       consensus as HaploThreader indexes it
  int* tp_consmap;      // [slots][Pmax*8]       consensus of covMap[j][i]
  int* tp_ntup;         // [slots][Pmax]
  int* tp_tuples;       // [slots][Pmax*Tmax]
  double* tp_score;     // [slots][Pmax*Tmax]
  unsigned short* tp_pred;  // [slots][Pmax*Tmax]
  double* tp_best;          // [slots][Tmax] running predecessor minimum of the current position
  unsigned int* sortbuf;    // [slots][nmax] cluster sort scratch (per wave slices)
  long long nmax, Pmax, Tmax;
};

}  // namespace ahk

// ---- launchers (defined next to their kernels) ------------------------------
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
namespace ahk {
struct ReadsetLds {
  int wordsN;  // u64 words of a node bitset (max over the batch)
  int maxG;    // max distinct names
  int maxB;
};
struct PhaseLds {
  int nmax;        // capacity of the per-read LDS arrays
  int stop_after;  // diagnostic builds (-DAHK_DIAG) only: 1/2/3 = stop after S4 / S5 init / S5 loop
  int ploidy;      // the threading DP's tables ((2p)^p tuple codes, (2p)^(p-1) marginal minima) share the LDS
  int stage_row;   // 1: room for the staged weight row (4 B per read) -- set when it does not cost a workgroup per CU
  int items_lds;   // 1: the first 2*nmax candidate edges of an iteration are listed in LDS (8 B per read), 0: all in HBM
                   //    -- dropped when that buys a workgroup per CU
  int nwv;         // wavefronts per workgroup: 4 (throughput form; 0 means 4) or 8 / 16 (latency forms, k_phase.hip)
};
struct ReadsetLds2 {  // k_readset_lds: the sizes ONE chain's LDS layout follows from
  int maxBitWords;   // ceil(N/64) * R: u64 words of all alignment bitsets
  int maxMaskWords;  // ceil(N/64) * alleles: u64 words of one set of allele-path masks
  int maxR, maxG, maxB, maxSA;
};
size_t readset_lds_bytes(const ReadsetLds& l);
size_t readset_lds2_bytes(const ReadsetLds2& l);
void launch_readset(const DeviceBatch& db, const int* list, int count, const ReadsetLds& l, hipStream_t stream);
// lds_bytes: the largest readset_lds2_bytes need among the listed chains; maxG: their largest name count
void launch_readset_lds(const DeviceBatch& db, const int* list, int count, size_t lds_bytes, int maxG, hipStream_t stream);
size_t phase_lds_bytes(const PhaseLds& l);
// ctl: three device ints of the class -- cursor of the queue, cursor of the deferred list, length of the
// deferred list -- zeroed by the caller before every call; deferred: [C] device ints
void launch_phase(const DeviceBatch& db, const SlotWorkspace& ws, const PhaseLds& l, const int* queue, int C, int* ctl, int* deferred,
                  int slots, hipStream_t stream);
int phase_max_blocks_per_cu(const PhaseLds& l);
}  // namespace ahk
#endif
