"""The reference-side binding (integration/ref_bridge.cpp, the stub of INTEGRATION.md section 2,
compiled with the reference's own parsers into oracle/_ref/ref_bridge) builds the same ChainBatch
as the library's ingest, and on a GPU the same rendered result as the library's own path."""
import os
import subprocess

import numpy as np
import pytest

from ahsoka_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BRIDGE = os.path.join(ROOT, "oracle", "_ref", "ref_bridge")
GOLD = os.path.join(ROOT, "tests", "golden")
STEMS = ["toy", "c1", "multi", "sbub"]


def _bridge():
    if not os.path.exists(BRIDGE) and os.path.isdir("/root/reference/src"):
        subprocess.call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/ref_bridge"])
    if not os.path.exists(BRIDGE):
        pytest.skip("oracle/_ref/ref_bridge not built (needs the reference sources and the HIP library)")
    return BRIDGE


def canonical_dump(b: capi.ChainBatch) -> str:
    """The text ref_bridge --dump prints, from a ChainBatch: nodes by their original ids; a read's
    node that is on none of the chain's allele paths prints as -1 (the bridge only indexes those)."""
    a = b.arrays()
    out = []
    for c in range(b.n_chains):
        b0, b1 = int(a["bubble_off"][c]), int(a["bubble_off"][c + 1])
        r0, r1 = int(a["read_off"][c]), int(a["read_off"][c + 1])
        n0 = int(a["node_off"][c])
        out.append(f"CHAIN {int(a['chain_id'][c])} B={b1 - b0} R={r1 - r0}")
        on_path = set()
        for g in range(b0, b1):
            for al in range(int(a["allele_off"][g]), int(a["allele_off"][g + 1])):
                e0, e1 = int(a["apath_off"][al]), int(a["apath_off"][al + 1])
                loc = [int(x) for x in a["apath_node"][e0:e1]]
                on_path.update(loc)
                ids = " ".join(str(int(a["node_id"][n0 + x])) for x in loc)
                signs = " ".join(chr(int(s)) for s in a["apath_sign"][e0:e1 - 1])
                out.append(f"A {g - b0} {al - int(a['allele_off'][g])} : {ids} |" + (" " + signs if signs else ""))
        out.append("S1 " + " ".join(str(int(x)) for x in a["s1_order"][b0:b1]) if b1 > b0 else "S1")
        for r in range(r0, r1):
            e0, e1 = int(a["rnode_off"][r]), int(a["rnode_off"][r + 1])
            ids = " ".join(str(int(a["node_id"][n0 + x])) if (x >= 0 and int(x) in on_path) else "-1" for x in a["rnode"][e0:e1])
            out.append(f"R {int(a['read_name'][r])} {int(a['read_mapq'][r])} {int(a['read_flags'][r])} :" + (" " + ids if ids else ""))
    return "\n".join(out) + "\n"


@pytest.mark.parametrize("stem", STEMS)
def test_bridge_builds_the_batch_the_ingest_builds(stem):
    exe = _bridge()
    gfa, gaf = os.path.join(GOLD, stem + ".gfa"), os.path.join(GOLD, stem + ".gaf")
    got = subprocess.run([exe, "--dump", gfa, gaf], capture_output=True, text=True, check=True).stdout
    want = canonical_dump(capi.ingest(gfa, gaf))
    assert got == want


def test_bridge_batch_on_a_power_law_graph(tmp_path):
    exe = _bridge()
    gfa, gaf = str(tmp_path / "x.gfa"), str(tmp_path / "x.gaf")
    capi.synth_write_text(gfa, gaf, n_chains=40, n_bubbles=0, n_reads=0, ploidy=2, b_mode=1, mean_span=10.0, seed=77)
    got = subprocess.run([exe, "--dump", gfa, gaf], capture_output=True, text=True, check=True).stdout
    assert got == canonical_dump(capi.ingest(gfa, gaf))


@pytest.mark.gpu
@pytest.mark.parametrize("stem", STEMS)
def test_bridge_phases_to_the_same_text(stem):
    """reference parsers + flattening + ahsoka_phase_batch + ahsoka_render_chain == ingest + the same calls"""
    if not os.path.exists(BRIDGE):
        pytest.skip("oracle/_ref/ref_bridge not built")
    gfa, gaf = os.path.join(GOLD, stem + ".gfa"), os.path.join(GOLD, stem + ".gaf")
    got = subprocess.run([BRIDGE, "--phase", gfa, gaf], capture_output=True, text=True, check=True).stdout
    b = capi.ingest(gfa, gaf)
    r = capi.phase_batch(b)
    want = "".join(capi.render_chain(b, r, c) for c in range(b.n_chains))
    assert got == want and len(want) > 0
