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
    nc, nb, nr = shape
    b = capi.synth_batch(n_chains=nc, n_bubbles=nb, n_reads=nr, seed=seed)
    _compare(b, capi.default_params(want_stages=1))


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
