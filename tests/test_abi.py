"""The C-ABI library loads without a GPU and exports every symbol that
include/ahsoka_phase.h declares; compute entry points refuse to run without a
device instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

from ahsoka_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ahsoka_phase.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ahsoka_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(capi.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    for name in declared_symbols():
        assert hasattr(L, name), name
    assert L.ahsoka_abi_version() == 3


def test_default_params_are_the_reference_literals():
    p = capi.default_params()
    # src/alignmentstoreadset.cpp:306,158,311,320
    assert (p.ploidy, p.min_mapq, p.min_variants, p.min_overlap) == (2, 93, 2, 1)
    assert (p.switch_cost, p.affine_switch_cost, p.compaction_mode) == (32.0, 8.0, 0)
    # HaploThreader(ploidy, 32.0, 8.0, false, 0), :320
    assert (p.symmetry_opt, p.row_limit) == (0, 0)


def test_threader_options_other_than_the_reference_literals_are_refused():
    """symmetryOptimization / rowLimit of :320 are at the boundary; only (false, 0) is implemented"""
    L = capi.lib()
    b = capi.synth_batch(**capi.CONFIGS["c1"])
    h = C.c_void_p()
    for bad in (dict(symmetry_opt=1), dict(row_limit=5)):
        prm = capi.default_params(**bad)
        assert L.ahsoka_batch_upload(b.ptr(), C.byref(prm), C.byref(h)) == capi.E_UNSUPPORTED, bad


def test_argument_errors():
    L = capi.lib()
    assert L.ahsoka_phase_batch(None, None, None) == capi.E_ARG
    assert L.ahsoka_synth_batch(None, None) == capi.E_ARG
    bad = capi.synth_spec(n_chains=1, n_bubbles=0, n_reads=1)
    out = C.POINTER(capi.CChainBatch)()
    assert L.ahsoka_synth_batch(C.byref(bad), C.byref(out)) == capi.E_ARG
    with pytest.raises(capi.AhsokaError):
        capi.ingest("/nonexistent.gfa", "/nonexistent.gaf")


def test_parameter_validation_precedes_the_device():
    """bad parameters are argument errors whether or not a device exists"""
    L = capi.lib()
    b = capi.synth_batch(**capi.CONFIGS["c1"])
    h = C.c_void_p()
    for bad in (dict(ploidy=1), dict(ploidy=9), dict(min_overlap=-1), dict(max_slots=-1), dict(switch_cost=-1.0),
                dict(affine_switch_cost=float("nan")), dict(wave_mode=5), dict(wave_mode=32)):
        prm = capi.default_params(**bad)
        assert L.ahsoka_batch_upload(b.ptr(), C.byref(prm), C.byref(h)) == capi.E_ARG, bad
    # s1_order must be a permutation of the chain's bubble ids
    a = b.arrays()
    a["s1_order"] = a["s1_order"].copy()
    a["s1_order"][1] = a["s1_order"][0]
    bad_b = capi.ChainBatch(arrays={k: v for k, v in a.items() if v is not None})
    prm = capi.default_params()
    assert L.ahsoka_batch_upload(bad_b.ptr(), C.byref(prm), C.byref(h)) == capi.E_ARG


def test_multi_device_entry_argument_errors():
    import numpy as np

    L = capi.lib()
    b = capi.synth_batch(**capi.CONFIGS["c1"])
    prm = capi.default_params()
    r = capi.CPhaseResult()
    dv = np.zeros(2, dtype=np.int32)
    assert L.ahsoka_phase_batch_multi(b.ptr(), C.byref(prm), dv.ctypes.data_as(capi.i32p), 0, C.byref(r)) == capi.E_ARG
    assert L.ahsoka_phase_batch_multi(b.ptr(), C.byref(prm), None, 2, C.byref(r)) == capi.E_ARG
    neg = np.array([0, -1], dtype=np.int32)
    assert L.ahsoka_phase_batch_multi(b.ptr(), C.byref(prm), neg.ctypes.data_as(capi.i32p), 2, C.byref(r)) == capi.E_ARG


def test_no_cpu_fallback_without_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    b = capi.synth_batch(**capi.CONFIGS["c1"])
    with pytest.raises(capi.AhsokaError) as e:
        capi.phase_batch(b)
    assert e.value.code == capi.E_NODEVICE
    with pytest.raises(capi.AhsokaError) as e:
        capi.phase_batch_multi(b, [0, 1])
    assert e.value.code == capi.E_NODEVICE
    assert capi.backend_info() == "hip:no-device"


def test_batch_validation_rejects_out_of_range_nodes():
    import numpy as np

    b = capi.synth_batch(n_chains=1, n_bubbles=3, n_reads=4, seed=1)
    a = b.arrays()
    a["rnode"] = a["rnode"].copy()
    a["rnode"][0] = 10_000
    bad = capi.ChainBatch(arrays={k: v for k, v in a.items() if v is not None})
    r = capi.CPhaseResult()
    prm = capi.default_params()
    rc = capi.lib().ahsoka_phase_batch(bad.ptr(), C.byref(prm), C.byref(r))
    assert rc in (capi.E_ARG, capi.E_NODEVICE)
    h = C.c_void_p()
    assert capi.lib().ahsoka_batch_upload(bad.ptr(), C.byref(prm), C.byref(h)) == capi.E_ARG


@pytest.mark.parametrize("defect", ["rnode_last", "apath_node_last", "read_name_mid", "s1_order_dup", "rnode_off_dip"])
def test_threaded_validation_finds_a_defect_anywhere(defect):
    """The batch is validated on several host threads, each over a range of chains: a defect in
    the last range (or in the middle) must come back E_ARG exactly as one in the first."""
    import numpy as np

    b = capi.synth_batch(n_chains=400, n_bubbles=6, n_reads=40, seed=5)
    a = {k: (v.copy() if v is not None else None) for k, v in b.arrays().items()}
    if defect == "rnode_last":
        a["rnode"][-1] = 1_000_000
    elif defect == "apath_node_last":
        a["apath_node"][-1] = -2
    elif defect == "read_name_mid":
        a["read_name"][len(a["read_name"]) // 2] = 10_000
    elif defect == "s1_order_dup":
        if a.get("s1_order") is None:
            pytest.skip("generator does not emit s1_order")
        a["s1_order"][-1] = a["s1_order"][-2]
    elif defect == "rnode_off_dip":
        a["rnode_off"][len(a["rnode_off"]) // 2] -= 1_000
    bad = capi.ChainBatch(arrays={k: v for k, v in a.items() if v is not None})
    h = C.c_void_p()
    prm = capi.default_params()
    assert capi.lib().ahsoka_batch_upload(bad.ptr(), C.byref(prm), C.byref(h)) == capi.E_ARG
    # and the untouched batch passes validation (what comes next needs a device)
    assert capi.lib().ahsoka_batch_upload(b.ptr(), C.byref(prm), C.byref(h)) in (0, capi.E_NODEVICE)
    if h.value:
        capi.lib().ahsoka_batch_free(h)


def test_synth_is_deterministic_and_shardable():
    import numpy as np

    full = capi.synth_batch(n_chains=6, n_bubbles=7, n_reads=30, seed=42).arrays()
    lo = capi.synth_batch(n_chains=4, n_bubbles=7, n_reads=30, seed=42, first_chain=0).arrays()
    hi = capi.synth_batch(n_chains=2, n_bubbles=7, n_reads=30, seed=42, first_chain=4).arrays()
    np.testing.assert_array_equal(full["rnode"], np.concatenate([lo["rnode"], hi["rnode"]]))
    np.testing.assert_array_equal(full["read_mapq"], np.concatenate([lo["read_mapq"], hi["read_mapq"]]))
    np.testing.assert_array_equal(full["node_id"], np.concatenate([lo["node_id"], hi["node_id"]]))
    np.testing.assert_array_equal(full["chain_id"], np.arange(6))
    assert full["read_mapq"].min() >= 90 and full["read_mapq"].max() <= 99


def test_algorithmic_bytes_formula():
    b = capi.synth_batch(n_chains=3, n_bubbles=5, n_reads=10, seed=7)
    a = b.arrays()
    R, L, B = 30, len(a["rnode"]), 15
    A, Pn = 2 * B, 3 * 2 * B
    exp = 4 * ((R + 3) + L + R + R) + 4 * ((B + 3) + (A + 3) + Pn)
    assert b.algorithmic_input_bytes() == exp


def test_bench_picks_newest_pmc_summary(tmp_path, monkeypatch):
    """profiles/r<round>_v<version>_pmc_hbm.json are ordered numerically (v12 after v9), and a summary is
    only ever used for the kernel revision it was taken with (no counter of one kernel version next to
    a time of another)"""
    import json
    import bench

    prof = tmp_path / "profiles"
    prof.mkdir()
    for tag, val, rev in (("r01_v9", 9.0, "A"), ("r01_v12", 12.0, "A"), ("r01_v3", 3.0, "A"), ("r01_v13", 13.0, "B")):
        (prof / f"{tag}_pmc_hbm.json").write_text(json.dumps(
            {"config": "c3", "chains": 10000, "kernel_rev": rev, "kernels": {"k_phase": {"hbm_bytes_fetch_x2": val}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_traffic("c3", 10000, "A") == (12.0, os.path.join("profiles", "r01_v12_pmc_hbm.json"))
    assert bench.pmc_traffic("c3", 10000, "B")[0] == 13.0
    assert bench.pmc_traffic("c3", 10000, "C") == (None, None)   # another kernel revision: no figure
    assert bench.pmc_traffic("c3", 123, "A") == (None, None)
    assert capi.kernel_revision().startswith("r")


def test_multi_plan_blocks_for_equal_chains_are_views_and_lpt_for_skewed():
    """ahsoka_phase_batch_multi's static partition (host only): equal chains -> contiguous blocks,
    i.e. zero-copy views of the caller's arrays; power-law chains -> LPT.  Every shard validates
    as a batch of its own, every chain has exactly one owner, blocks are ranges in input order."""
    import numpy as np

    b = capi.synth_batch(n_chains=240, n_bubbles=12, n_reads=120, seed=77)
    for nd in (2, 3, 8):
        owner, blocks, serial_ms, shard_ms = capi.multi_plan(b, nd)
        assert blocks
        assert (np.diff(owner) >= 0).all() and set(owner.tolist()) == set(range(nd))
        cnt = np.bincount(owner, minlength=nd)
        assert cnt.max() - cnt.min() <= max(2, 240 // nd // 8)
    sk = capi.synth_batch(n_chains=600, n_bubbles=0, n_reads=0, b_mode=1, mean_span=10.0, seed=0xA450CA05)
    owner, blocks, _, _ = capi.multi_plan(sk, 4)
    assert len(owner) == 600 and set(owner.tolist()) <= set(range(4))
    if not blocks:  # LPT: the shards interleave
        assert (np.diff(owner) < 0).any()
    # more devices than chains: empty shards are legal
    tiny = capi.synth_batch(n_chains=3, n_bubbles=4, n_reads=30, seed=5)
    owner, _, _, _ = capi.multi_plan(tiny, 8)
    assert len(owner) == 3
