"""Reads with several alignments under one name (GraphAligner split alignments):
the two places where the reference's result depends on more than the merged
variant matrix.

* S1 visits bubbles in the iteration order of a copied unordered_map
  (src/alignmentstoreadset.cpp:76,90); the first hit creates the Read and fixes
  its mapq (:117), which the S1b filter (:158) then tests.
* S2 adds a variant to an existing Read only if `alignment.id*100 > 90`,
  evaluated in float (:245): identity 0.905 passes, its truncated mapq 90 would not.

tests/golden/multi.{gfa,gaf} (see make_golden.py: multi()) is built so that both
decide which positions get phased.  The expected positions follow from the
reference text; the visiting order and the flag themselves are pinned on the
compiled reference (test_ingest_golden.py).  CPU part: the oracle; the GPU part is
in test_gpu_parity.py::test_multi_alignment_fixture.
"""
import os

import numpy as np

from ahsoka_amd import capi
import oracle_binding as ob

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def multi_batch(**drop):
    b = capi.ingest(os.path.join(GOLD, "multi.gfa"), os.path.join(GOLD, "multi.gaf"))
    if not drop:
        return b
    a = b.arrays()
    for k, v in drop.items():
        a[k] = v(a) if callable(v) else v
    return capi.ChainBatch(arrays={k: v for k, v in a.items() if v is not None})


def check_expected(res):
    """chain 0 (20 bubbles): 10 only through the 0.905 alignment of mB; 12 and 13 only if mD
    (created by its 0.96 alignment, because bubble 0 precedes bubble 12 in S1 order) survives
    S1b.  chain 1 (6 bubbles): 5 only if mA is created by its 0.95 alignment (bubble 5 first)."""
    assert (res.status == capi.ST_PHASED).all()
    assert res.chain_pos(0).tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13]
    assert res.chain_pos(1).tolist() == [0, 1, 2, 3, 4, 5]


def test_oracle_expected_positions():
    check_expected(ob.phase_batch(multi_batch(), capi.default_params()))


def test_fixture_discriminates_the_float_guard():
    """with the guard evaluated on the truncated mapq (read_flags absent) position 10 is lost"""
    r = ob.phase_batch(multi_batch(read_flags=None), capi.default_params())
    assert 10 not in r.chain_pos(0).tolist() and 12 in r.chain_pos(0).tolist()


def test_fixture_discriminates_the_s1_order():
    asc = ob.phase_batch(multi_batch(s1_order=lambda a: np.concatenate([np.arange(20), np.arange(6)]).astype(np.int32)),
                         capi.default_params())
    assert 5 not in asc.chain_pos(1).tolist()          # ascending: mA created by its 0.92 alignment
    assert 12 in asc.chain_pos(0).tolist()
    rev = ob.phase_batch(multi_batch(s1_order=lambda a: np.concatenate([np.arange(19, -1, -1), np.arange(5, -1, -1)]).astype(np.int32)),
                         capi.default_params())
    assert 12 not in rev.chain_pos(0).tolist()         # plain reverse: mD created by its 0.92 alignment
    assert 5 in rev.chain_pos(1).tolist()


def test_null_s1_order_is_the_reference_order():
    """s1_order == NULL means: the order the reference's own map construction gives"""
    a = ob.phase_batch(multi_batch(), capi.default_params(want_stages=1))
    b = ob.phase_batch(multi_batch(s1_order=None), capi.default_params(want_stages=1))
    np.testing.assert_array_equal(a.pos, b.pos)
    np.testing.assert_array_equal(a.hap_allele, b.hap_allele)
    np.testing.assert_array_equal(a.stages["score_q"], b.stages["score_q"])
