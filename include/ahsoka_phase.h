/*
 * ahsoka_phase.h -- C-ABI of the MI355X-native bubble-chain phasing path.
 *
 * This is the drop-in boundary for the one hot path of Ahsoka that this
 * repository implements: the per-chain loop body of
 *
 *     void alignmentsToReadset(AlignmentReader&, Graph&, pathToAlleles&,
 *                              string readsetfile, bool, size_sorting&, mutex&)
 *         reference: src/alignmentstoreadset.cpp:55-493
 *         callers:   src/polyassembly.cpp:171 (single thread) and :187
 *
 * The reference has no ABI boundary there (it is an in-process C++ call that
 * borrows std::unordered_map / std::vector objects).  The replacement takes
 * the same information flattened into CSR arrays of plain int32/int64
 * ("ChainBatch"), phases every chain of the batch on the GPU, and returns the
 * per-chain haplotype allele assignment; the host then renders the node lists
 * exactly as src/alignmentstoreadset.cpp:415-477 does (ahsoka_render_chain).
 *
 * Conventions: plain pointers and sizes only, no C++ or torch types; every
 * entry point returns 0 on success or a negative AHSOKA_E_* code and never
 * throws across the boundary; inputs are caller-owned and read-only; results
 * are callee-allocated and released with ahsoka_result_free; no global mutable
 * state -- distinct handles may be used from distinct threads.
 */
#ifndef AHSOKA_PHASE_H
#define AHSOKA_PHASE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AHSOKA_ABI_VERSION 3

/* ---- error codes ------------------------------------------------------- */
#define AHSOKA_OK 0
#define AHSOKA_E_ARG (-1)        /* null / inconsistent argument              */
#define AHSOKA_E_NODEVICE (-2)   /* no HIP device / HIP runtime error at init */
#define AHSOKA_E_HIP (-3)        /* HIP runtime error during the call         */
#define AHSOKA_E_NOMEM (-4)      /* host or device allocation failed          */
#define AHSOKA_E_UNSUPPORTED (-5)/* input exceeds a compiled-in limit         */
#define AHSOKA_E_IO (-6)         /* file could not be read / written          */
#define AHSOKA_E_PARSE (-7)      /* malformed GFA / GAF                       */

/* ---- per-chain status (replaces the reference's silent `continue`s) ----- */
#define AHSOKA_ST_PHASED 0          /* haplotype lines were produced              */
#define AHSOKA_ST_SKIPPED_LE1 1     /* <=1 bubble: alignmentstoreadset.cpp:86     */
#define AHSOKA_ST_EMPTY_READSET 2   /* filtered partial readset empty: :276-279   */
#define AHSOKA_ST_NO_FULL_MATCH 3   /* S1 readset empty: reference UB at :193 (Q4)*/
#define AHSOKA_ST_UNSUPPORTED 4     /* chain exceeds a device limit               */

/*
 * ChainBatch: everything alignmentsToReadset reads, flattened.
 *
 * Chains are stored in the order they are to be reported (the reference's
 * size-descending `size_sorting`, src/polyassembly.cpp:136-140, is produced by
 * the ingest; the phasing itself is order-independent).
 *
 *  - A chain has B = bubble_off[c+1]-bubble_off[c] bubbles; bubble k of the
 *    chain IS the reference's bubble id k (ids are dense 0..B-1,
 *    src/graph.cpp:351-365) and therefore the "position" of every variant.
 *  - Bubble g (global index bubble_off[c]+k) has alleles
 *    allele_off[g]..allele_off[g+1]; allele a's node path is
 *    apath_node[apath_off[a]..apath_off[a+1]) in the order
 *    ChainsToReadsetDetailed produced it (src/chainstoreadset.cpp:161-203).
 *    Nodes are CHAIN-LOCAL indices into node_id[node_off[c]..node_off[c+1]).
 *  - Alignments: one entry per GAF line that touches the chain, in GAF line
 *    order (the reference stores one copy per path node,
 *    src/alignmentreader.cpp:176-183; copies are idempotent).
 *    rnode[] holds the alignment's path as chain-local node indices, -1 for a
 *    node that is not part of this chain.  read_mapq is
 *    (int)(std::stof(id) * 100) (src/alignmentstoreadset.cpp:117).
 *    read_name is a chain-local dense id of the read NAME in order of first
 *    appearance: alignments with equal name are the same Read for
 *    ReadSet::getByName (src/alignmentstoreadset.cpp:116,233).
 *  - apath_sign (optional, may be NULL => all '+'): orientation character
 *    printed for apath_node[i] when it is followed by apath_node[i+1]
 *    (src/alignmentstoreadset.cpp:441-447); value is '+' or '-'.
 *  - genotype (optional, may be NULL): allele dosage per allele, same indexing
 *    as allele_off.  NULL reproduces the reference literal {0:1,1:1}
 *    (src/alignmentstoreadset.cpp:340-344).
 *  - read_flags (optional, may be NULL): per alignment, bit 0 =
 *    `alignment.id*100 > 90` evaluated as the reference does, in single
 *    precision on the float identity (src/alignmentstoreadset.cpp:245 --
 *    NOT the truncated read_mapq: 0.905f*100 = 90.5 > 90 but (int) = 90).
 *    S2 adds a variant to an EXISTING Read only from alignments with this bit.
 *    NULL derives the bit as read_mapq > 90 (exact unless an identity lies
 *    strictly between 0.90 and 0.91).  ahsoka_ingest / ahsoka_synth_batch
 *    always fill it.
 *  - s1_order (optional, may be NULL): per chain the B bubble ids in the order
 *    S1 visits them, i.e. the iteration order of the chain's copied
 *    unordered_map<int, vector<vector<int>>> (src/alignmentstoreadset.cpp:76,90;
 *    quirk Q2).  The order decides which of several same-name alignments
 *    creates the S1 Read and with it the Read's mapq for the S1b filter.
 *    Indexed like bubble_off.  NULL = the order libstdc++ gives a map filled
 *    with the keys 0..B-1 ascending (what ChainsToReadsetDetailed does,
 *    src/chainstoreadset.cpp:171-198), computed by the library.
 */
typedef struct ahsoka_chain_batch {
  int32_t n_chains;
  const int32_t* chain_id;    /* [C]       reference chain id (for output)    */
  const int64_t* bubble_off;  /* [C+1]                                        */
  const int64_t* allele_off;  /* [SB+1]    SB = bubble_off[C]                 */
  const int64_t* apath_off;   /* [SA+1]    SA = allele_off[SB]                */
  const int32_t* apath_node;  /* [SP]      SP = apath_off[SA], chain-local    */
  const uint8_t* apath_sign;  /* [SP] or NULL                                 */
  const int32_t* genotype;    /* [SA] or NULL                                 */
  const int64_t* node_off;    /* [C+1]                                        */
  const int32_t* node_id;     /* [SN]      original (digit) node ids          */
  const int64_t* read_off;    /* [C+1]     alignments per chain               */
  const int64_t* rnode_off;   /* [SR+1]    SR = read_off[C]                   */
  const int32_t* rnode;       /* [SL]      chain-local node index or -1       */
  const int32_t* read_mapq;   /* [SR]                                         */
  const int32_t* read_name;   /* [SR]      chain-local dense name id          */
  const uint8_t* read_flags;  /* [SR] or NULL   bit0: float(id)*100 > 90 (:245) */
  const int32_t* s1_order;    /* [SB] or NULL   S1 bubble visiting order (:90)  */
} ahsoka_chain_batch;

/* Literals of the reference, made parameters. */
typedef struct ahsoka_phase_params {
  int32_t ploidy;             /* reference literal 2 (alignmentstoreadset.cpp:306) */
  int32_t min_mapq;           /* 93  (:158,270)                                    */
  int32_t min_variants;       /* 2   ("getVariantCount() > 1", :158,270)           */
  int32_t min_overlap;        /* 1   (:311)                                        */
  double switch_cost;         /* 32.0 (:320)                                       */
  double affine_switch_cost;  /* 8.0  (:320)                                       */
  int32_t compaction_mode;    /* 0 = vectors compacted as the reference does at
                                 :379-402 (quirk Q12); 1 = WhatsHap caller
                                 convention (i-th entry <-> i-th covMap cluster)   */
  int32_t device;             /* HIP device ordinal                                */
  int32_t want_stages;        /* !=0: also return the per-stage dumps below        */
  int32_t max_slots;          /* 0 = automatic.  >0 caps the number of chains in flight
                                 per size class (workgroups of the persistent grid, each
                                 with its own slot of HBM working state): a memory knob;
                                 results do not depend on it                          */
  int32_t symmetry_opt;       /* HaploThreader's symmetryOptimization: reference literal
                                 false (alignmentstoreadset.cpp:320).  Only 0 is implemented;
                                 anything else is rejected with AHSOKA_E_UNSUPPORTED      */
  int32_t row_limit;          /* HaploThreader's rowLimit: reference literal 0 = unlimited
                                 (:320).  Only 0 is implemented (AHSOKA_E_UNSUPPORTED)    */
  int32_t wave_mode;          /* 0 = automatic.  4 / 8 / 16 force that many wavefronts per workgroup
                                 in the cluster-editing kernel (4: throughput form, five chains
                                 per CU; 8 / 16: latency forms for chains that have a CU to
                                 themselves).  A performance knob; results do not depend on it */
} ahsoka_phase_params;

/*
 * Result.  hap_allele is the `cons` of src/alignmentstoreadset.cpp:420-422:
 * for chain c, covered position j (0 <= j < P_c) and haplotype i,
 *     hap_allele[hap_off[c] + i*P_c + j],   pos[pos_off[c] + j] = bubble id.
 * hap_off[c] = ploidy * pos_off[c].
 */
typedef struct ahsoka_phase_result {
  int32_t n_chains;
  int32_t ploidy;
  int32_t* status;       /* [C] AHSOKA_ST_*                                  */
  int64_t* pos_off;      /* [C+1]                                            */
  int32_t* pos;          /* [SPos] covered bubble ids, ascending             */
  int32_t* hap_allele;   /* [ploidy*SPos]                                    */
  int32_t* hap_cluster;  /* [ploidy*SPos] threaded global cluster id         */
  int32_t* n_reads;      /* [C] reads in the final (sorted) readset          */
  int32_t* n_clusters;   /* [C]                                              */
  int64_t* n_iterations; /* [C] cluster-editing outer iterations             */
  /* stage dumps, only when params.want_stages (else NULL) */
  int64_t* vmat_off;     /* [C+1] offsets into vfull/vpart (R_c*B_c each)    */
  int8_t* vfull;         /* S1: first fully matching allele or -1            */
  int8_t* vpart;         /* S2: first partially matching allele or -1        */
  int64_t* fread_off;    /* [C+1] offsets into final-read arrays             */
  int32_t* fread_name;   /* final readset (after ReadSet::sort): name id     */
  int32_t* fread_cluster;/* cluster index of each final read                 */
  int64_t* score_off;    /* [C+1] offsets into score_q (n_c*n_c each)        */
  int32_t* score_q;      /* S4: quantised similarity, dense n x n            */
  double elapsed_ms;     /* wall time of the phase call                      */
  double kernel_ms;      /* device time between first and last kernel        */
  void* _owner;          /* internal                                         */
} ahsoka_phase_result;

/* Fill params with the reference's literals. */
void ahsoka_default_params(ahsoka_phase_params* prm);

/* "hip:<gfx name>:<CUs>" or an error string; never fails hard. */
int ahsoka_backend_info(char* buf, size_t buflen);
int ahsoka_abi_version(void);
/* revision tag of the device kernels (static string): measurement bookkeeping -- a counter
 * summary under profiles/ is only ever paired with timings of the same revision */
const char* ahsoka_kernel_revision(void);

/*
 * THE hot path.  Replaces one call of alignmentsToReadset
 * (src/alignmentstoreadset.cpp:55).  Host buffers in, host buffers out.
 */
int ahsoka_phase_batch(const ahsoka_chain_batch* in, const ahsoka_phase_params* prm,
                       ahsoka_phase_result* out);
void ahsoka_result_free(ahsoka_phase_result* out);

/*
 * The same call over several HIP devices of one node (SURVEY.md 8.b2 "device list";
 * the reference's own attempt at parallelism is the two-thread split of
 * src/polyassembly.cpp:190-215).  Chains are independent, so they are split by a static
 * cost-weighted partition (longest processing time first), phased concurrently -- one
 * host thread, device handle and HIP stream per entry of `devices`, all owned by the
 * library, no data-path collective -- and merged back into the caller's chain order:
 * the result is bit-identical to ahsoka_phase_batch whatever the device list.
 * An ordinal may repeat ({0,0}: two shards on one GPU).  prm->device is ignored.
 */
int ahsoka_phase_batch_multi(const ahsoka_chain_batch* in, const ahsoka_phase_params* prm,
                             const int32_t* devices, int32_t n_devices, ahsoka_phase_result* out);

/*
 * Device-resident variant used by the benchmark: upload once, phase many
 * times (the timed region starts with inputs already in HBM), download once.
 */
typedef struct ahsoka_device_batch ahsoka_device_batch;
int ahsoka_batch_upload(const ahsoka_chain_batch* in, const ahsoka_phase_params* prm,
                        ahsoka_device_batch** handle);
int ahsoka_phase_device(ahsoka_device_batch* handle); /* kernels only, async + sync */
int ahsoka_batch_download(ahsoka_device_batch* handle, ahsoka_phase_result* out);
void ahsoka_batch_free(ahsoka_device_batch* handle);
/* per-kernel device time of the last ahsoka_phase_device call (hipEvents on
 * the launch stream); names/ms arrays are owned by the handle. */
int ahsoka_batch_kernel_times(ahsoka_device_batch* handle, int32_t* n, const char*** names,
                              const float** ms);
/*
 * How the last ahsoka_phase_device call laid out its size classes (chains are binned by the size of
 * their final readset; every class has its own pool of slots, queue and launch of the
 * cluster-editing kernel).  out: 8 int32 per class -- largest readset the class holds, chains,
 * slots, wavefronts per workgroup, workgroups per CU, bytes of LDS per workgroup, and whether the
 * workgroup keeps the first part of an iteration's candidate-edge list (8 B per read) and the staged
 * weight row (4 B per read) in LDS.  The chains in flight are what the 160 KB of LDS per CU allow:
 * this is the view of that budget (tests pin it for the BASELINE shapes).  cap = int32 capacity of out.
 */
int ahsoka_batch_class_info(ahsoka_device_batch* handle, int32_t* n_classes, int32_t* out, int32_t cap);
/*
 * The result of the last ahsoka_phase_device call as ONE device-resident int32
 * record block with a layout that depends only on the batch shape, so that ranks
 * can exchange results with a single fixed-size collective (RCCL gather) without
 * touching the host:   [ status[C] | npos[C] | pos[SB] | hap_allele[ploidy*SB] ]
 * chain c: pos at bubble_off[c] (npos[c] valid entries), haplotype i of chain c at
 * C+C+SB + ploidy*bubble_off[c] + i*npos[c].  *dev_ptr stays owned by the handle.
 */
int ahsoka_batch_result_device(ahsoka_device_batch* handle, void** dev_ptr, int64_t* n_int32);
/* bytes of the batch as laid out in HBM, and algorithmic bytes */
int ahsoka_batch_bytes(const ahsoka_chain_batch* in, int32_t ploidy, int64_t* algorithmic_bytes);

/* ---- host side of the path (own C++; reference stays C++ too) ----------- */

/*
 * Render the haplotype node lists of chain c exactly as
 * src/alignmentstoreadset.cpp:82-83,415-458 writes them into
 * "<o>-result.txt".  Returns the number of bytes written to buf (excluding
 * the terminating NUL) or the required size if buflen is too small.
 */
int64_t ahsoka_render_chain(const ahsoka_chain_batch* in, const ahsoka_phase_result* res,
                            int32_t c, char* buf, int64_t buflen);

/*
 * Ingest: GFA + GAF text -> ChainBatch, reproducing the numbering of
 * Graph::ReadGraph / findBubbles (src/graph.cpp:188-249,333-501),
 * ChainsToReadsetDetailed (src/chainstoreadset.cpp:161-203) and
 * AlignmentReader::readAlignmentfile (src/alignmentreader.cpp:69-186).
 * The returned batch owns its arrays; release with ahsoka_batch_host_free.
 */
int ahsoka_ingest(const char* gfa_path, const char* gaf_path, ahsoka_chain_batch** out);
void ahsoka_batch_host_free(ahsoka_chain_batch* b);
/* Text dump of what the ingest derived (CHAIN/BUBBLE/ALLELE/ALN/ORDER lines, the
 * format of oracle/ref_harness.cpp): pins the ingest against the reference's own
 * parsers and feeds <o>-bubbleinfo.txt (src/polyassembly.cpp:95-110). */
int ahsoka_ingest_dump(const char* gfa_path, const char* gaf_path, const char* out_path);

/*
 * Deterministic synthetic workload: C chains of
 * B simple biallelic bubbles, R reads per chain, splitmix64 seeded.
 * b_mode 0: every chain has exactly B bubbles; 1: B ~ Pareto(1.8) in [1,500]
 * and R = ceil(40*B/11) (config 5).  Optionally also writes GFA/GAF text.
 */
typedef struct ahsoka_synth_spec {
  int32_t n_chains, n_bubbles, n_reads, ploidy;
  int32_t b_mode;
  double mean_span;   /* 7 (c1-c4) or 10 (c5) */
  double allele_err;  /* 0.02 */
  uint64_t seed;
  int32_t first_chain; /* generate chains [first_chain, first_chain+n_chains) of the
                          global workload: rank sharding without communication */
} ahsoka_synth_spec;
int ahsoka_synth_batch(const ahsoka_synth_spec* spec, ahsoka_chain_batch** out);
/* chains picked by global index instead of a range (a shard of a cost-weighted partition) */
int ahsoka_synth_select(const ahsoka_synth_spec* spec, const int32_t* chains, int32_t n,
                        ahsoka_chain_batch** out);
/* bubbles / reads of chains [first_chain, first_chain+n_chains) without generating them */
int ahsoka_synth_shapes(const ahsoka_synth_spec* spec, int32_t* n_bubbles, int32_t* n_reads);
int ahsoka_synth_write_text(const ahsoka_synth_spec* spec, const char* gfa_path,
                            const char* gaf_path);
/* truth haplotypes the reads were drawn from: per chain ploidy rows of B alleles,
 * chains concatenated; returns the bytes needed (fills out if cap suffices) */
int64_t ahsoka_synth_truth(const ahsoka_synth_spec* spec, uint8_t* out, int64_t cap);

/* test hooks (host-only, no GPU): the device code's re-statement of the
 * libstdc++ introsort, run on the host, so it can be pinned against std::sort */
int ahsoka_test_introsort_u32(uint32_t* keys_hi16_payload_lo16, int32_t n);
/* the rank-based (parallelisable) statement of the same sort: must give the same permutation */
int ahsoka_test_introsort_par_u32(uint32_t* keys_hi16_payload_lo16, int32_t n);
double ahsoka_test_detlog(double x);
/* the static partition of ahsoka_phase_batch_multi and the per-shard host work up to the first
 * upload, without touching a device: owner_out [C] or NULL; *blocks_out 1 = contiguous blocks
 * (views of the caller's arrays), 0 = LPT (packed copies); *serial_ms = cost model + partition,
 * *shard_ms_max = longest per-shard view/pack time (one per device thread, concurrent);
 * check != 0 validates every shard like a caller's batch */
int ahsoka_test_multi_plan(const ahsoka_chain_batch* in, const ahsoka_phase_params* prm,
                           int32_t n_devices, int32_t* owner_out, int32_t* blocks_out,
                           double* serial_ms, double* shard_ms_max, int32_t check);

#ifdef __cplusplus
}
#endif
#endif /* AHSOKA_PHASE_H */
