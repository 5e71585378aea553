"""GPU parity tests proper: the HIP path, called through the C-ABI, against the
CPU oracle on the same seeded inputs -- bit-exact (everything on this path is
integer / index work; the one floating-point stage, S4, is compared after its
quantisation to Q15.16, which is what the later stages consume).

Oracle provenance: stages S1/S2/S6/S8 follow the reference text, is_subset is
pinned on the reference's compiled function (test_oracle_pin.py); S3/S4/S5/S7
are re-statements of WhatsHap code that is not available => parity unpinned
for those.
"""
import numpy as np
import pytest

from ahsoka_amd import capi
import oracle_binding as ob

pytestmark = pytest.mark.gpu


def _compare(batch, prm, stages=True, n_threads=8):
    g = capi.phase_batch(batch, prm)
    o = ob.phase_batch(batch, prm, n_threads=n_threads)
    np.testing.assert_array_equal(g.status, o.status)
    np.testing.assert_array_equal(g.n_reads, o.n_reads)
    if stages and prm.want_stages:
        np.testing.assert_array_equal(g.stages["vfull"], o.stages["vfull"])
        np.testing.assert_array_equal(g.stages["vpart"], o.stages["vpart"])
        np.testing.assert_array_equal(g.stages["fread_name"], o.stages["fread_name"])
        np.testing.assert_array_equal(g.stages["score_q"], o.stages["score_q"])
        np.testing.assert_array_equal(g.stages["fread_cluster"], o.stages["fread_cluster"])
    np.testing.assert_array_equal(g.n_clusters, o.n_clusters)
    np.testing.assert_array_equal(g.n_iterations, o.n_iterations)
    np.testing.assert_array_equal(g.pos_off, o.pos_off)
    np.testing.assert_array_equal(g.pos, o.pos)
    np.testing.assert_array_equal(g.hap_cluster, o.hap_cluster)
    np.testing.assert_array_equal(g.hap_allele, o.hap_allele)
    return g, o


def test_c1_all_stages():
    """BASELINE.json configs[0]: 10 chains x 5 bubbles, 50 reads."""
    b = capi.synth_batch(**capi.CONFIGS["c1"])
    g, _ = _compare(b, capi.default_params(want_stages=1))
    assert (g.status == capi.ST_PHASED).all()


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("shape", [(6, 3, 12), (4, 20, 200), (3, 70, 300), (2, 130, 260)])
def test_small_shapes_all_stages(seed, shape):
    """(every form of the cluster-editing kernel: 4, 8 and 16 wavefronts per workgroup -- the automatic choice
    for so few chains would be 16 only)"""
    nc, nb, nr = shape
    b = capi.synth_batch(n_chains=nc, n_bubbles=nb, n_reads=nr, seed=seed)
    for wm in (4, 8, 16):
        _compare(b, capi.default_params(want_stages=1, wave_mode=wm))


def test_compaction_mode_1():
    b = capi.synth_batch(n_chains=6, n_bubbles=12, n_reads=150, seed=11)
    _compare(b, capi.default_params(want_stages=1, compaction_mode=1))


def test_high_error_many_clusters():
    b = capi.synth_batch(n_chains=4, n_bubbles=25, n_reads=300, seed=5, allele_err=0.15)
    _compare(b, capi.default_params(want_stages=1))


def test_edge_chains():
    """<=1 bubble chains are skipped (alignmentstoreadset.cpp:86); chains with no
    reads have an empty S1 readset (reference UB, status NO_FULL_MATCH)."""
    b1 = capi.synth_batch(n_chains=3, n_bubbles=1, n_reads=20, seed=3)
    g, _ = _compare(b1, capi.default_params(want_stages=1))
    assert (g.status == capi.ST_SKIPPED_LE1).all()
    b2 = capi.synth_batch(n_chains=3, n_bubbles=6, n_reads=0, seed=3)
    g, _ = _compare(b2, capi.default_params(want_stages=1))
    assert (g.status == capi.ST_NO_FULL_MATCH).all()
    # all reads below the identity filter -> EMPTY_READSET
    b3 = capi.synth_batch(n_chains=2, n_bubbles=6, n_reads=40, seed=3)
    g, _ = _compare(b3, capi.default_params(want_stages=1, min_mapq=100))
    assert (g.status == capi.ST_EMPTY_READSET).all()


def test_ploidy4_small():
    """configs[3] shape at reduced size: tetraploid, genotype = truth dosage."""
    b = capi.synth_batch(n_chains=3, n_bubbles=8, n_reads=160, ploidy=4, seed=0xA450CA04)
    _compare(b, capi.default_params(want_stages=1, ploidy=4))


def test_c2_subset_final_results():
    """configs[1] shape (20 bubbles, 500 reads/chain), 48 chains, end results."""
    cfg = dict(capi.CONFIGS["c2"])
    cfg["n_chains"] = 48
    b = capi.synth_batch(**cfg)
    g, _ = _compare(b, capi.default_params(), stages=False)
    assert (g.status == capi.ST_PHASED).all()


def test_c3_few_chains_final_results():
    """configs[2] shape (50 bubbles, 1000 reads/chain), 6 chains."""
    cfg = dict(capi.CONFIGS["c3"])
    cfg["n_chains"] = 6
    b = capi.synth_batch(**cfg)
    _compare(b, capi.default_params(), stages=False)


def test_render_matches_haplotypes():
    b = capi.synth_batch(**capi.CONFIGS["c1"])
    g = capi.phase_batch(b)
    txt = capi.render_chain(b, g, 0)
    lines = txt.split("\n")
    assert lines[0] == "chain id: 0" and lines[1] == "size of chain: 5"
    assert lines[2] == "haplotype 0:" and lines[4] == "haplotype 1:"
    a = b.arrays()
    hap = g.chain_haps(0)
    # chain 0: nodes 1..16, bubble k -> source 3k+1, inner 3k+2+allele
    exp = "".join(f"{3 * k + 1}(+),{3 * k + 2 + int(hap[0][j])}(+)," for j, k in enumerate(g.chain_pos(0)))
    assert lines[3] == exp


def _dense_names(names):
    remap, out = {}, names.copy()
    for i, v in enumerate(names):
        out[i] = remap.setdefault(int(v), len(remap))
    return out


def test_duplicate_read_names():
    """several alignments with one name are one Read (getByName,
    src/alignmentstoreadset.cpp:116,233): the kernel's per-name merge path"""
    b = capi.synth_batch(n_chains=3, n_bubbles=10, n_reads=90, seed=21)
    a = b.arrays()
    names = a["read_name"].copy()
    rng = np.random.default_rng(5)
    for c in range(3):
        r0, r1 = a["read_off"][c], a["read_off"][c + 1]
        for _ in range(25):
            i, j = rng.integers(r0, r1, size=2)
            names[max(i, j)] = names[min(i, j)]
        names[r0:r1] = _dense_names(names[r0:r1])
    a["read_name"] = names
    nb = capi.ChainBatch(arrays={k: v for k, v in a.items() if v is not None})
    g, o = _compare(nb, capi.default_params(want_stages=1))
    assert (g.n_reads < 90).all()


def test_reads_with_foreign_nodes_and_empty_middle():
    """toy fixture through the ingest: indel bubble (2-node allele path: empty
    middle matches every alignment, quirk Q6), 3-inner-node bubble, reads that
    leave the chain (-1 nodes), a repeated read name"""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    b = capi.ingest(os.path.join(gold, "toy.gfa"), os.path.join(gold, "toy.gaf"))
    _compare(b, capi.default_params(want_stages=1, min_mapq=90))
    _compare(b, capi.default_params(want_stages=1))


def test_superbubble_fixture():
    """a seven-inner-node superbubble with six allele paths listed sink -> source (quirk Q7),
    between two simple bubbles: complex allele paths through the match kernel"""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    b = capi.ingest(os.path.join(gold, "sbub.gfa"), os.path.join(gold, "sbub.gaf"))
    g, _ = _compare(b, capi.default_params(want_stages=1))
    assert g.status[0] == capi.ST_PHASED and g.stages["vfull"].max() >= 3


def test_c1_text_roundtrip_through_ingest():
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    b = capi.ingest(os.path.join(gold, "c1.gfa"), os.path.join(gold, "c1.gaf"))
    g, _ = _compare(b, capi.default_params(want_stages=1))
    assert (g.status == capi.ST_PHASED).all()
    txt = capi.render_chain(b, g, 0)
    assert txt.startswith("chain id: 9\nsize of chain: 5\nhaplotype 0:\n")


def test_c2_full_vs_oracle():
    """BASELINE configs[1] at full size: 1000 chains x 20 bubbles x 500 reads."""
    b = capi.synth_batch(**capi.CONFIGS["c2"])
    g, _ = _compare(b, capi.default_params(), stages=False, n_threads=16)
    assert (g.status == capi.ST_PHASED).all()


def test_c4_shape_two_chains():
    """BASELINE configs[3] shape: ploidy 4, 30 bubbles, 2000 reads/chain."""
    cfg = dict(capi.CONFIGS["c4"])
    cfg["n_chains"] = 2
    b = capi.synth_batch(**cfg)
    _compare(b, capi.default_params(ploidy=4), stages=False, n_threads=2)


def test_c5_skewed_sample():
    """BASELINE configs[4] shape: power-law bubble counts (1..500), includes
    1-bubble chains (SKIPPED) and multi-word position masks."""
    cfg = dict(capi.CONFIGS["c5"])
    cfg["n_chains"] = 400
    b = capi.synth_batch(**cfg)
    g, _ = _compare(b, capi.default_params(), stages=False, n_threads=16)
    assert (g.status == capi.ST_SKIPPED_LE1).sum() > 50
    assert np.diff(b.arrays()["bubble_off"]).max() > 64


def test_multi_alignment_fixture():
    """reads with several alignments under one name: the S1 visiting order (:76,90) and the
    float guard `id*100 > 90` (:245) decide which positions are phased (tests/golden/multi.*)"""
    from test_multi_alignment import check_expected, multi_batch
    g, _ = _compare(multi_batch(), capi.default_params(want_stages=1))
    check_expected(g)
    # NULL s1_order = the library derives the reference's order itself: same result
    g2, _ = _compare(multi_batch(s1_order=None), capi.default_params(want_stages=1))
    np.testing.assert_array_equal(g.hap_allele, g2.hap_allele)
    np.testing.assert_array_equal(g.pos, g2.pos)
    # NULL read_flags = documented fallback (guard on the truncated identity): position 10 is lost
    g3, _ = _compare(multi_batch(read_flags=None), capi.default_params(want_stages=1))
    assert 10 not in g3.chain_pos(0).tolist()


def test_slot_reuse_c5_all_stages():
    """the persistent grid re-uses a slot's HBM state (W, IC, LV, NZ, threading tables) for
    chain after chain: 1500 chains of the power-law config through at most 16 slots per size
    class, every stage against the oracle"""
    cfg = dict(capi.CONFIGS["c5"])
    cfg["n_chains"] = 1500
    b = capi.synth_batch(**cfg)
    _compare(b, capi.default_params(want_stages=1, max_slots=16, wave_mode=4), n_threads=16)
    _compare(b, capi.default_params(max_slots=16), stages=False, n_threads=16)  # (automatic: wide forms for the small classes)


def test_slot_reuse_large_then_small_chain():
    """ONE slot, chains of very different size in the same size class, largest first (the queue
    order): stale live/non-zero bitmaps, clique bits or row maxima of a large chain must not
    leak into the smaller chain that follows it in the slot"""
    big = capi.synth_batch(n_chains=2, n_bubbles=50, n_reads=1000, seed=0xA450CA03).arrays()
    small = capi.synth_batch(n_chains=3, n_bubbles=30, n_reads=560, seed=77).arrays()
    mid = capi.synth_batch(n_chains=2, n_bubbles=40, n_reads=760, seed=78).arrays()
    b = _concat([small, big, mid])
    g, o = _compare(b, capi.default_params(want_stages=1, max_slots=1, wave_mode=4), n_threads=8)
    assert g.n_reads.max() > 1.6 * g.n_reads.min() and g.n_reads.min() > g.n_reads.max() // 2  # one size class
    for wm in (8, 16):
        _compare(b, capi.default_params(want_stages=1, max_slots=1, wave_mode=wm), n_threads=8)


def test_slot_reuse_ploidy4():
    """64 tetraploid chains through 8 slots: the threading tables (up to 4096 tuples per
    position) are slot state too"""
    b = capi.synth_batch(n_chains=64, n_bubbles=12, n_reads=240, ploidy=4, seed=0xA450CA04)
    _compare(b, capi.default_params(want_stages=1, ploidy=4, max_slots=8, wave_mode=4), n_threads=16)
    _compare(b, capi.default_params(want_stages=1, ploidy=4, max_slots=8, wave_mode=16), n_threads=16)


def _concat(parts):
    """concatenate ChainBatch array dicts (chains keep their own local indices)"""
    out = {}
    for k in ("chain_id", "apath_node", "node_id", "rnode", "read_mapq", "read_name", "read_flags", "s1_order"):
        out[k] = np.concatenate([p[k] for p in parts])
    out["chain_id"] = np.arange(len(out["chain_id"]), dtype=np.int32)
    for k, dep in (("bubble_off", None), ("allele_off", "bubble_off"), ("apath_off", "allele_off"), ("node_off", None),
                   ("read_off", None), ("rnode_off", "read_off")):
        acc, base = [np.zeros(1, dtype=np.int64)], 0
        for p in parts:
            acc.append(p[k][1:] + base)
            base += p[k][-1]
        out[k] = np.concatenate(acc)
    return capi.ChainBatch(arrays=out)


def test_over_limit_chain_is_reported_per_chain():
    """a chain beyond a device limit comes back AHSOKA_ST_UNSUPPORTED; the other chains of the
    batch are phased as usual (the limits: phase_kernels.h -- 8192 names, 8192 nodes, 4096 final
    reads, 4095 bubbles)"""
    ok = capi.synth_batch(n_chains=3, n_bubbles=9, n_reads=80, seed=5).arrays()
    many_names = capi.synth_batch(n_chains=1, n_bubbles=12, n_reads=8300, seed=6).arrays()   # > 8192 names
    many_nodes = capi.synth_batch(n_chains=1, n_bubbles=2800, n_reads=50, seed=7).arrays()   # 8401 nodes > 8192
    many_final = capi.synth_batch(n_chains=1, n_bubbles=40, n_reads=8000, seed=8).arrays()   # > 4096 final reads
    b = _concat([ok, many_names, many_nodes, many_final])
    g = capi.phase_batch(b, capi.default_params())
    assert g.status.tolist() == [0, 0, 0, capi.ST_UNSUPPORTED, capi.ST_UNSUPPORTED, capi.ST_UNSUPPORTED]
    o = ob.phase_batch(capi.ChainBatch(arrays=ok), capi.default_params())
    np.testing.assert_array_equal(g.hap_allele, o.hap_allele)
    np.testing.assert_array_equal(g.pos, o.pos)
    assert g.n_reads[3:].sum() == 0 and g.pos_off[-1] == o.pos_off[-1]


def test_chains_beyond_the_round2_limits_are_phased():
    """the reference's chain loop has no size limit (src/alignmentstoreadset.cpp:75-490).  Chains the
    round-2 kernels refused -- a 1500-bubble chain at 40x depth (4501 nodes, 5455 read names, ~3500
    final reads: a real-shape hifiasm chain), a 40-bubble chain whose 3600 reads leave ~2100 final
    reads that nearly all overlap (150 000 iterations), and a 700-bubble chain (2101 nodes) -- are
    phased and bit-equal to the oracle, stage dumps of the long chain included; the three go
    through ONE batch, i.e. through the windowed fallback readset kernel, the 64-word bitmap rows
    and the 12-bit edge keys of k_phase, next to an ordinary chain."""
    import concurrent.futures as cf

    parts = [capi.synth_batch(n_chains=1, n_bubbles=1500, n_reads=5455, seed=9, mean_span=10.0).arrays(),
             capi.synth_batch(n_chains=1, n_bubbles=40, n_reads=3600, seed=8).arrays(),
             capi.synth_batch(n_chains=1, n_bubbles=700, n_reads=900, seed=7).arrays(),
             capi.synth_batch(n_chains=2, n_bubbles=9, n_reads=80, seed=5).arrays()]
    b = _concat(parts)
    g = capi.phase_batch(b, capi.default_params())
    assert (g.status == capi.ST_PHASED).all()
    assert g.n_reads[0] > 3000 and g.n_reads[1] > 2048 and np.diff(g.pos_off)[0] > 1400
    # the oracle is single-threaded per chain (~100 s for each of the two large ones): one thread per chain
    with cf.ThreadPoolExecutor(4) as ex:
        outs = list(ex.map(lambda a: ob.phase_batch(capi.ChainBatch(arrays=a), capi.default_params(), n_threads=2), parts))
    c = 0
    for o in outs:
        for lc in range(o.n_chains):
            assert g.status[c] == o.status[lc]
            assert g.n_reads[c] == o.n_reads[lc] and g.n_clusters[c] == o.n_clusters[lc]
            assert g.n_iterations[c] == o.n_iterations[lc], (c, g.n_iterations[c], o.n_iterations[lc])
            np.testing.assert_array_equal(g.chain_pos(c), o.chain_pos(lc))
            np.testing.assert_array_equal(g.chain_haps(c), o.chain_haps(lc))
            c += 1
    # all stages of a mid-size chain beyond the old node limit, through one slot
    _compare(capi.ChainBatch(arrays=parts[2]), capi.default_params(want_stages=1, max_slots=1))


def test_c5_full_size_vs_oracle():
    """BASELINE configs[4] at FULL size on one GPU (50 000 chains, power-law bubble counts,
    35 000 one-bubble chains that are skipped): every chain bit-exact against the oracle,
    with 64 slots per size class (heavy slot re-use) and with the automatic slot count"""
    b = capi.synth_batch(**capi.CONFIGS["c5"])
    o = ob.phase_batch(b, capi.default_params(), n_threads=16)
    for slots in (64, 0):
        g = capi.phase_batch(b, capi.default_params(max_slots=slots))
        np.testing.assert_array_equal(g.status, o.status)
        np.testing.assert_array_equal(g.n_reads, o.n_reads)
        np.testing.assert_array_equal(g.n_iterations, o.n_iterations)
        np.testing.assert_array_equal(g.pos, o.pos)
        np.testing.assert_array_equal(g.hap_cluster, o.hap_cluster)
        np.testing.assert_array_equal(g.hap_allele, o.hap_allele)
    assert (g.status == capi.ST_PHASED).sum() > 10000 and g.n_reads.max() > 600


def _perm_accuracy(h, t):
    """fraction of positions where the haplotype rows equal the truth rows under the best
    global assignment of predicted to true haplotypes"""
    import itertools
    p = h.shape[0]
    return max(int((h[list(perm)] == t).all(axis=0).sum()) for perm in itertools.permutations(range(p))) / max(1, h.shape[1])


def test_c4_full_size_properties():
    """BASELINE configs[3] at FULL size (2000 chains x 30 bubbles x 2000 reads, ploidy 4,
    genotype = truth dosage): idempotence, genotype conformity, agreement with the truth
    haplotypes, slot re-use at ploidy 4, and bit-exact parity with the oracle on a sample"""
    cfg = dict(capi.CONFIGS["c4"])
    b = capi.synth_batch(**cfg)
    dev = capi.DeviceBatch(b, capi.default_params(ploidy=4))
    dev.phase()
    g = dev.download()
    dev.phase()
    g2 = dev.download()
    classes = dev.class_info()
    dev.free()
    # the LDS budget of its size class: four workgroups per CU WITH the staged weight row --
    # two bytes of LDS per read too many cost it the fourth workgroup (-9 %) for half of round 3 without any test noticing
    assert len(classes) == 1 and classes[0]["waves"] == 4, classes
    assert classes[0]["workgroups_per_cu"] == 4 and classes[0]["stage_row"] == 1, classes
    np.testing.assert_array_equal(g.hap_allele, g2.hap_allele)
    np.testing.assert_array_equal(g.hap_cluster, g2.hap_cluster)
    np.testing.assert_array_equal(g.n_iterations, g2.n_iterations)
    assert (g.status == capi.ST_PHASED).all()
    assert np.diff(g.pos_off).min() >= 25
    a = b.arrays()
    truth = capi.synth_truth(**cfg)
    conform = tot = 0
    acc = []
    for c in range(0, 2000, 7):
        h = g.chain_haps(c)
        pos = g.chain_pos(c)
        t = truth[c * 120:(c + 1) * 120].reshape(4, 30)[:, pos]
        # dosage of allele 1 per position equals the supplied genotype wherever a conform tuple existed
        conform += int((h.sum(axis=0) == t.sum(axis=0)).sum())
        tot += h.shape[1]
        acc.append(_perm_accuracy(h, t))
    assert conform / tot > 0.9
    assert np.mean(acc) > 0.6  # (4 haplotypes, 2 % allele errors, reads of ~8 bubbles)
    # the same chains through 8 slots: identical bits
    sub = dict(cfg)
    sub["n_chains"] = 40
    bs = capi.synth_batch(**sub)
    gs = capi.phase_batch(bs, capi.default_params(ploidy=4, max_slots=8))
    for c in range(40):
        np.testing.assert_array_equal(gs.chain_haps(c), g.chain_haps(c))
    np.testing.assert_array_equal(gs.n_iterations, g.n_iterations[:40])
    # oracle on 4 chains spread over the batch (17 s each on one core)
    idx = [0, 666, 1333, 1999]
    import concurrent.futures as cf

    def one(c):
        s1 = dict(cfg)
        s1["n_chains"] = 1
        s1["first_chain"] = c
        return ob.phase_batch(capi.synth_batch(**s1), capi.default_params(ploidy=4))
    with cf.ThreadPoolExecutor(4) as ex:
        for c, o in zip(idx, ex.map(one, idx)):
            np.testing.assert_array_equal(o.chain_haps(0), g.chain_haps(c))
            assert o.n_iterations[0] == g.n_iterations[c]


def test_c3_full_size_properties():
    """BASELINE configs[2] at FULL size (10k chains x 50 bubbles x 1000 reads):
    size-independent properties of a diploid phasing, plus bit-exact parity with
    the oracle on a sample of the same chains."""
    cfg = dict(capi.CONFIGS["c3"])
    b = capi.synth_batch(**cfg)
    dev = capi.DeviceBatch(b, capi.default_params())
    dev.phase()
    g = dev.download()
    dev.phase()                      # idempotence: the same call gives the same bits
    g2 = dev.download()
    classes = dev.class_info()
    dev.free()
    # the LDS budget of the bench workload's size class: five workgroups per CU with BOTH extras (the LDS part of the
    # candidate-edge list and the staged weight row); 512 B of LDS too many had silently cost it the staged row (-2.4 %)
    assert len(classes) == 1 and classes[0]["waves"] == 4 and classes[0]["slots"] == 1280, classes
    assert classes[0]["workgroups_per_cu"] == 5 and classes[0]["items_lds"] == 1 and classes[0]["stage_row"] == 1, classes
    np.testing.assert_array_equal(g.hap_allele, g2.hap_allele)
    np.testing.assert_array_equal(g.n_iterations, g2.n_iterations)
    assert (g.status == capi.ST_PHASED).all()
    P = np.diff(g.pos_off)
    assert P.min() >= 45 and P.max() == 50
    # genotype {0:1,1:1}: wherever a genotype-conform tuple exists the two haplotypes
    # are complementary; positions covered by one haplotype's reads only are the exception
    comp = tot = 0
    for c in range(0, 10000, 97):
        h = g.chain_haps(c)
        comp += int((h[0] ^ h[1]).sum())
        tot += h.shape[1]
    assert comp / tot > 0.97
    # the result is a correct phasing: against the haplotypes the reads were drawn
    # from (2 % allele errors), best of the two global assignments per chain
    truth = capi.synth_truth(**cfg)
    ok = tot = 0
    for c in range(0, 10000, 13):
        t = truth[c * 100:(c + 1) * 100].reshape(2, 50)[:, g.chain_pos(c)]
        h = g.chain_haps(c)
        ok += max(int((h == t).all(axis=0).sum()), int((h == t[::-1]).all(axis=0).sum()))
        tot += h.shape[1]
    assert ok / tot > 0.99
    # oracle on every 400th chain (25 chains), bit-exact
    idx = list(range(0, 10000, 400))
    for c in idx:
        sub = dict(cfg)
        sub["n_chains"] = 1
        sub["first_chain"] = c
        o = ob.phase_batch(capi.synth_batch(**sub), capi.default_params())
        np.testing.assert_array_equal(o.chain_haps(0), g.chain_haps(c))
        assert o.n_iterations[0] == g.n_iterations[c]


@pytest.mark.parametrize("seed", range(24))
def test_random_irregular_batches(seed):
    """differential fuzz: irregular bubbles (1..5 alleles, 2..5-node paths in either
    direction), reads with foreign / repeated nodes, repeated names"""
    from random_batches import random_batch

    b = random_batch(1000 + seed)
    _compare(b, capi.default_params(want_stages=1, min_mapq=93 if seed % 2 else 90, wave_mode=(0, 4, 8)[seed % 3]))


@pytest.mark.parametrize("over", [
    dict(min_overlap=2), dict(min_variants=3), dict(switch_cost=5.0, affine_switch_cost=1.0),
    dict(min_mapq=96), dict(switch_cost=0.0, affine_switch_cost=0.0),
])
def test_parameters_are_honoured(over):
    """the reference's literals are parameters of the ABI: non-default values"""
    b = capi.synth_batch(n_chains=5, n_bubbles=16, n_reads=220, seed=31, allele_err=0.06)
    _compare(b, capi.default_params(want_stages=1, **over))


def test_cli_end_to_end(tmp_path):
    """`Ahsoka phase -g .. -a .. -o ..` (the reference's command line) writes the same
    haplotype text as rendering the library result"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "ahsoka_amd", "Ahsoka")
    gold = os.path.join(root, "tests", "golden")
    prefix = str(tmp_path / "out")
    subprocess.run([exe, "phase", "-g", os.path.join(gold, "c1.gfa"), "-a", os.path.join(gold, "c1.gaf"), "-o", prefix],
                   check=True, capture_output=True)
    b = capi.ingest(os.path.join(gold, "c1.gfa"), os.path.join(gold, "c1.gaf"))
    g = capi.phase_batch(b)
    expect = "".join(capi.render_chain(b, g, c) for c in range(b.n_chains))
    assert open(prefix + "-result.txt").read() == expect
    # -d 0,0: the chains sharded over a device list (here: twice the same GPU), same text
    prefix2 = str(tmp_path / "out2")
    subprocess.run([exe, "phase", "-g", os.path.join(gold, "c1.gfa"), "-a", os.path.join(gold, "c1.gaf"), "-o", prefix2,
                    "-d", "0,0"], check=True, capture_output=True)
    assert open(prefix2 + "-result.txt").read() == expect
    assert open(prefix + "-bubbleinfo.txt").read().startswith("chain id: 0size: 5\nbubble id: 0\nnode id: 160,158,159,157,\n")
    # per-chain files hold the node lines only (src/alignmentstoreadset.cpp:411-412,449,456)
    lines = open(prefix + "-chain9-result.txt").read().split("\n")
    assert len(lines) == 3 and lines[0].endswith("),") and lines[2] == ""


def test_multi_device_entry_equals_single_device():
    """ahsoka_phase_batch_multi: chains split by cost over a device list, one library-owned
    handle per entry, results merged into the batch order.  With one GPU the list {0,0} /
    {0,0,0} runs the shards side by side on it: every field, stage dumps included, must
    equal the single-device call bit for bit (and with it the oracle)."""
    cfg = dict(capi.CONFIGS["c5"])
    cfg["n_chains"] = 700
    cfg["first_chain"] = 18000          # a stretch with one 142-bubble chain: skewed costs
    b = capi.synth_batch(**cfg)
    prm = capi.default_params(want_stages=1)
    one = capi.phase_batch(b, prm)
    for devs in ([0, 0], [0, 0, 0]):
        m = capi.phase_batch_multi(b, devs, prm)
        for k in ("status", "pos_off", "pos", "hap_allele", "hap_cluster", "n_reads", "n_clusters", "n_iterations"):
            np.testing.assert_array_equal(getattr(m, k), getattr(one, k), err_msg=k)
        for k in ("vmat_off", "vfull", "vpart", "fread_off", "fread_name", "fread_cluster", "score_off", "score_q"):
            np.testing.assert_array_equal(m.stages[k], one.stages[k], err_msg=k)
    o = ob.phase_batch(b, prm, n_threads=16)
    np.testing.assert_array_equal(one.hap_allele, o.hap_allele)
    # both forms of the partition must have run: contiguous blocks (zero-copy views of the caller's
    # arrays, offsets re-based) and LPT (packed copies made by the device threads)
    paths = {capi.multi_plan(b, nd, prm)[1] for nd in (2, 3)}
    eq = capi.synth_batch(n_chains=36, n_bubbles=14, n_reads=160, seed=4242)       # equal chains: blocks
    one_eq = capi.phase_batch(eq, prm)
    for devs in ([0, 0], [0, 0, 0, 0, 0]):
        paths.add(capi.multi_plan(eq, len(devs), prm)[1])
        m = capi.phase_batch_multi(eq, devs, prm)
        for k in ("status", "pos_off", "pos", "hap_allele", "hap_cluster", "n_reads", "n_clusters", "n_iterations"):
            np.testing.assert_array_equal(getattr(m, k), getattr(one_eq, k), err_msg=k)
        for k in ("vmat_off", "vfull", "vpart", "fread_off", "fread_name", "fread_cluster", "score_off", "score_q"):
            np.testing.assert_array_equal(m.stages[k], one_eq.stages[k], err_msg=k)
    # a few chains of very different size, largest first (the reference's size_sorting): no cut balances them
    sk = _concat([capi.synth_batch(n_chains=1, n_bubbles=nb, n_reads=nr, seed=900 + nb).arrays()
                  for nb, nr in ((60, 700), (30, 400), (28, 380), (12, 150), (10, 120), (9, 100), (8, 90))])
    paths.add(capi.multi_plan(sk, 3, prm)[1])
    one_sk = capi.phase_batch(sk, prm)
    m = capi.phase_batch_multi(sk, [0, 0, 0], prm)
    for k in ("status", "pos_off", "pos", "hap_allele", "hap_cluster", "n_iterations"):
        np.testing.assert_array_equal(getattr(m, k), getattr(one_sk, k), err_msg=k)
    np.testing.assert_array_equal(m.stages["score_q"], one_sk.stages["score_q"])
    assert paths == {True, False}, paths
    # ploidy 4 (genotype array travels with the shards), no stages
    b4 = capi.synth_batch(n_chains=9, n_bubbles=12, n_reads=240, ploidy=4, seed=0xA450CA04)
    p4 = capi.default_params(ploidy=4)
    np.testing.assert_array_equal(capi.phase_batch_multi(b4, [0, 0], p4).hap_allele, capi.phase_batch(b4, p4).hap_allele)
    # a single entry is the plain call on that device; an unknown ordinal is an argument error
    np.testing.assert_array_equal(capi.phase_batch_multi(b, [0], prm).hap_allele, one.hap_allele)
    with pytest.raises(capi.AhsokaError):
        capi.phase_batch_multi(b, [0, 4096], prm)


def test_degenerate_batches():
    """empty batch, a chain without bubbles, a chain without reads, a single chain"""
    from random_batches import random_batch
    empty = capi.ChainBatch(arrays={k: np.zeros(0 if not k.endswith("_off") else 1, dtype=np.int64 if k.endswith("_off") else np.int32)
                                    for k in ("chain_id", "bubble_off", "allele_off", "apath_off", "apath_node", "node_off", "node_id",
                                              "read_off", "rnode_off", "rnode", "read_mapq", "read_name")})
    g = capi.phase_batch(empty)
    assert g.n_chains == 0
    for seed in (5000, 5001):
        b = random_batch(seed, n_chains=1, max_bubbles=0, max_reads=3)
        _compare(b, capi.default_params(want_stages=1))
    b = capi.synth_batch(n_chains=1, n_bubbles=9, n_reads=70, seed=77)
    _compare(b, capi.default_params(want_stages=1))


@pytest.mark.parametrize("ploidy", [3, 4])
def test_other_ploidies(ploidy):
    """ploidy is a parameter of the ABI (reference literal 2, quirk Q17); tuples = (2p)^p"""
    b = capi.synth_batch(n_chains=4, n_bubbles=10, n_reads=150, ploidy=ploidy, seed=900 + ploidy)
    _compare(b, capi.default_params(want_stages=1, ploidy=ploidy))
    _compare(b, capi.default_params(want_stages=1, ploidy=ploidy, wave_mode=4))
    # reference literal genotype {0:1,1:1} with ploidy > 2: nothing conforms, all tuples are used
    a = b.arrays()
    a.pop("genotype")
    nb = capi.ChainBatch(arrays={k: v for k, v in a.items() if v is not None})
    _compare(nb, capi.default_params(want_stages=1, ploidy=ploidy))


@pytest.mark.parametrize("seed", range(8))
def test_allele_paths_with_repeated_nodes(seed):
    """std::includes is a multiset test (quirk Q5): paths that visit a node twice need
    the alignment to visit it twice"""
    from random_batches import random_batch

    b = random_batch(7000 + seed, dup_paths=True)
    g, o = _compare(b, capi.default_params(want_stages=1, min_mapq=90))
    assert (g.status != capi.ST_UNSUPPORTED).all()


def test_device_result_block_matches_download():
    """the packed device-resident result block (what ranks gather over RCCL) decodes
    to exactly what ahsoka_batch_download returns"""
    from ahsoka_amd import shard
    b = capi.synth_batch(n_chains=40, n_bubbles=17, n_reads=120, seed=314)
    dev = capi.DeviceBatch(b, capi.default_params())
    dev.phase()
    res = dev.download()
    t = dev.result_tensor()
    blk = t.cpu().numpy()
    # the bench gathers a torch-owned device copy of the block: same content
    np.testing.assert_array_equal(t.clone().cpu().numpy(), blk)
    status, pos_off, pos, hap = shard.decode_block(blk, b.arrays()["bubble_off"], 2)
    dev.free()
    np.testing.assert_array_equal(status, res.status)
    np.testing.assert_array_equal(pos_off, res.pos_off)
    np.testing.assert_array_equal(pos, res.pos)
    np.testing.assert_array_equal(hap, res.hap_allele)


def test_wide_induced_costs_instance():
    """k_phase keeps induced costs as 2 x uint32 where a chain's row sums of |W| prove them below 2^32
    and defers the other chains to its 64-bit instance.  1800 error-free reads over 16 bubbles with
    long spans give ~1170 final reads whose largest row sum is 2^31.7: this chain takes the wide
    instance (every other test and configuration takes the narrow one), two of them through one slot."""
    b = capi.synth_batch(n_chains=2, n_bubbles=16, n_reads=1800, ploidy=2, seed=12, mean_span=15.0, allele_err=0.0)
    g, o = _compare(b, capi.default_params(want_stages=1, max_slots=1, wave_mode=4))
    _compare(b, capi.default_params(want_stages=1, max_slots=1, wave_mode=16))
    assert (g.status == capi.ST_PHASED).all()
    n = int(o.n_reads[0])
    w = np.asarray(o.stages["score_q"][: n * n], dtype=np.int64).reshape(n, n)
    assert 2 * int(np.abs(w).sum(axis=1).max()) >= 1 << 32  # (the premise: this chain must be deferred)
