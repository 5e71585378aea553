"""Host ingest (ahsoka_amd/csrc/ingest.cpp) against the reference's own parsers
and bubble finder: chain / bubble numbering, Bubble{source,sink,inner}, allele
paths and orientation signs, the per-chain alignment lists with the
(int)(id*100) identity and the number of copies the reference stores, and the
size_sorting order.  Expected files were produced by oracle/_ref/ref_dump =
/root/reference/src/{graph,alignmentreader,chainstoreadset}.cpp compiled
unchanged (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from ahsoka_amd import capi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("stem", ["c1", "toy", "multi", "sbub"])
def test_dump_equals_reference(stem, tmp_path):
    out = str(tmp_path / "dump.txt")
    capi.ingest_dump(os.path.join(GOLD, stem + ".gfa"), os.path.join(GOLD, stem + ".gaf"), out)
    assert open(out).read() == open(os.path.join(GOLD, stem + ".ref.txt")).read()


def _parse(path):
    chains, alleles, alns, order, s1 = {}, {}, {}, [], {}
    for ln in open(path):
        t = ln.split()
        if t[0] == "CHAIN":
            chains[int(t[1])] = int(t[2])
        elif t[0] == "ALLELE":
            n = int(t[4])
            alleles.setdefault((int(t[1]), int(t[2])), []).append(([int(x) for x in t[5:5 + n]], t[6 + n:]))
        elif t[0] == "ALN":
            n = int(t[6])
            alns.setdefault(int(t[1]), []).append((t[2], int(t[3]), int(t[4]), [int(x) for x in t[7:7 + n]]))
        elif t[0] == "S1ORDER":
            s1[int(t[1])] = [int(x) for x in t[2:]]
        elif t[0] == "ORDER":
            order = [int(x) for x in t[1:]]
    return chains, alleles, alns, order, s1


@pytest.mark.parametrize("stem", ["c1", "toy", "multi", "sbub"])
def test_batch_equals_reference(stem):
    chains, alleles, alns, order, s1 = _parse(os.path.join(GOLD, stem + ".ref.txt"))
    b = capi.ingest(os.path.join(GOLD, stem + ".gfa"), os.path.join(GOLD, stem + ".gaf"))
    a = b.arrays()
    assert list(a["chain_id"]) == order
    for ci, cid in enumerate(order):
        b0, b1 = a["bubble_off"][ci], a["bubble_off"][ci + 1]
        assert b1 - b0 == chains[cid]
        assert list(a["s1_order"][b0:b1]) == s1[cid]  # S1 bubble visiting order (:76,90)
        nid = a["node_id"][a["node_off"][ci]:a["node_off"][ci + 1]]
        for k in range(b1 - b0):
            exp = alleles[(cid, k)]
            a0, a1 = a["allele_off"][b0 + k], a["allele_off"][b0 + k + 1]
            assert a1 - a0 == len(exp)
            for ai in range(a1 - a0):
                e0, e1 = a["apath_off"][a0 + ai], a["apath_off"][a0 + ai + 1]
                assert list(nid[a["apath_node"][e0:e1]]) == exp[ai][0]
                assert [chr(c) for c in a["apath_sign"][e0:e1 - 1]] == exp[ai][1]
        r0, r1 = a["read_off"][ci], a["read_off"][ci + 1]
        exp_al = alns.get(cid, [])
        assert r1 - r0 == len(exp_al)
        names = {}
        for r in range(r0, r1):
            name, mapq, gt90, nodes = exp_al[r - r0]
            assert a["read_mapq"][r] == mapq
            assert a["read_flags"][r] == gt90  # float(id)*100 > 90 (:245)
            loc = a["rnode"][a["rnode_off"][r]:a["rnode_off"][r + 1]]
            back = [int(nid[x]) if x >= 0 else None for x in loc]
            assert all(bk is None or bk == nd for bk, nd in zip(back, nodes)) and len(back) == len(nodes)
            # nodes mapped to -1 are exactly those that are not nodes of this chain
            assert all((nd in set(nid.tolist())) == (bk is not None) for bk, nd in zip(back, nodes))
            names.setdefault(name, len(names))
            assert a["read_name"][r] == names[name]


def test_s1_visiting_order_golden(tmp_path):
    """S1 walks a chain's bubbles in the iteration order of the reference's copied
    unordered_map (src/alignmentstoreadset.cpp:76,90): the ingest's order for chains of
    2..1000 bubbles against what the reference build printed (tests/golden/s1order.ref.txt)"""
    gold = {}
    for ln in open(os.path.join(GOLD, "s1order.ref.txt")):
        t = ln.split()
        gold[int(t[1])] = [int(x) for x in t[2:]]
    assert len(gold) >= 40 and max(gold) == 1000
    not_reverse = 0
    for B, exp in gold.items():
        assert sorted(exp) == list(range(B))
        g, a = str(tmp_path / "x.gfa"), str(tmp_path / "x.gaf")
        capi.synth_write_text(g, a, n_chains=1, n_bubbles=B, n_reads=0, seed=1)
        out = str(tmp_path / "dump.txt")
        capi.ingest_dump(g, a, out)
        line = [ln for ln in open(out) if ln.startswith("S1ORDER ")]
        assert len(line) == 1 and [int(x) for x in line[0].split()[2:]] == exp, B
        # the direct generator and the library's NULL default carry the same order
        sb = capi.synth_batch(n_chains=1, n_bubbles=B, n_reads=0, seed=1).arrays()
        assert list(sb["s1_order"]) == exp
        not_reverse += exp != list(range(B - 1, -1, -1))
    assert not_reverse > 10  # beyond 13 bubbles the table rehashes: not simply reversed


def test_dump_live_against_reference_build(tmp_path):
    """a mid-size synthetic graph through the reference's own parsers, right now"""
    import subprocess
    ref_dump = os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "ref_dump")
    if not os.path.exists(ref_dump):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    g, a = str(tmp_path / "x.gfa"), str(tmp_path / "x.gaf")
    capi.synth_write_text(g, a, n_chains=7, n_bubbles=0, n_reads=0, b_mode=1, mean_span=10.0, seed=0xA450CA05, first_chain=13)
    exp = subprocess.run([ref_dump, g, a], capture_output=True, text=True, check=True).stdout
    out = str(tmp_path / "dump.txt")
    capi.ingest_dump(g, a, out)
    assert open(out).read() == exp


def test_superbubble_path_listing():
    """seven inner nodes, six paths, listed sink -> source depth-first in link order (quirk Q7);
    a ladder of 12 stacked sub-bubbles (4096 paths) stops at the 1024-path bound instead of
    walking all of it -- such a chain is beyond the device's 16 alleles anyway"""
    ref = [ln.split() for ln in open(os.path.join(GOLD, "sbub.ref.txt")) if ln.startswith("ALLELE 0 1 ")]
    assert len(ref) == 6 and all(t[4] == "5" for t in ref)
    listed = [tuple(int(x) for x in t[5:10]) for t in ref]
    # the reference sees this bubble as source 12, sink 4 (BUBBLE line): paths run sink -> source, in link order
    assert listed != sorted(listed) and all(p[0] == 4 and p[-1] == 12 for p in listed)
    b = capi.ingest(os.path.join(GOLD, "sbub.gfa"), os.path.join(GOLD, "sbub.gaf"))
    a = b.arrays()
    assert np.diff(a["allele_off"]).tolist() == [2, 6, 2]


def test_allele_path_bound(tmp_path):
    import time
    S, L = [], []

    def utg(i):
        return f"utg{i:06d}l"

    def link(x, y):
        return [f"L\t{utg(x)}\t+\t{utg(y)}\t+\t0M", f"L\t{utg(y)}\t-\t{utg(x)}\t-\t0M"]
    # simple bubble, then ONE superbubble made of 22 stacked diamonds closed by a long bypass
    # node (so that no inner junction is itself a bubble end), then a simple bubble
    K = 22
    nid = 1
    edges = [(1, 2), (1, 3), (2, 4), (3, 4)]
    junction = 4
    first = junction
    nid = 5
    for _ in range(K):
        a, bq, j = nid, nid + 1, nid + 2
        edges += [(junction, a), (junction, bq), (a, j), (bq, j)]
        junction = j
        nid += 3
    bypass = nid
    edges += [(first, bypass), (bypass, junction)]
    nid += 1
    edges += [(junction, nid), (junction, nid + 1), (nid, nid + 2), (nid + 1, nid + 2)]
    for i in range(1, nid + 3):
        S.append(f"S\t{utg(i)}\tA")
    for x, y in edges:
        L += link(x, y)
    gfa = tmp_path / "ladder.gfa"
    gfa.write_text("\n".join(S + L) + "\n")
    gaf = tmp_path / "ladder.gaf"
    gaf.write_text("")
    t0 = time.time()
    b = capi.ingest(str(gfa), str(gaf))
    dt = time.time() - t0
    na = np.diff(b.arrays()["allele_off"])
    assert dt < 5.0                       # 2^22 paths are not walked
    assert na.max() == 1024               # the bound; > 16 alleles => the device reports the chain UNSUPPORTED


def test_identity_truncation():
    """(int)(stof(id)*100): 0.935 -> 93, 0.931 -> 93, 1 -> 100, 0.92 -> 92"""
    b = capi.ingest(os.path.join(GOLD, "toy.gfa"), os.path.join(GOLD, "toy.gaf"))
    a = b.arrays()
    assert sorted(set(a["read_mapq"].tolist())) == [92, 93, 94, 95, 96, 97, 98, 99, 100]


def test_float_guard_flag():
    """`alignment.id*100 > 90` (src/alignmentstoreadset.cpp:245) is a float compare: 0.905
    truncates to mapq 90 but passes the guard; the reference build's ALN line says so"""
    b = capi.ingest(os.path.join(GOLD, "multi.gfa"), os.path.join(GOLD, "multi.gaf"))
    a = b.arrays()
    sel = (a["read_mapq"] == 90)
    assert sel.sum() == 1 and a["read_flags"][sel][0] == 1
    assert (a["read_flags"][a["read_mapq"] > 90] == 1).all()


def test_parse_errors(tmp_path):
    gfa = tmp_path / "x.gfa"
    gaf = tmp_path / "x.gaf"
    gfa.write_text("S\tutg000001l\tA\nS\tutg000002l\tA\nL\tutg000001l\t+\tutg000002l\t+\t0M\n")
    gaf.write_text("r\t1\t0\t1\t+\t>utg000001l\t1\t0\t1\t1\t1\t60\tNM:i:0\tAS:f:0\tdv:f:0\n")  # 15 tokens
    with pytest.raises(capi.AhsokaError) as e:
        capi.ingest(str(gfa), str(gaf))
    assert e.value.code == capi.E_PARSE
    gaf.write_text("r\t1\t0\t1\t+\t>utg000001l\t1\t0\t1\t1\t1\t60\tNM:i:0\tAS:f:0\tdv:f:0\tid:f:0.95\n")
    b = capi.ingest(str(gfa), str(gaf))
    assert b.n_chains == 0  # a single edge forms no bubble


def test_allele_path_listing_above_the_bound_is_marked_truncated(tmp_path):
    """findPathsComplex (src/chainstoreadset.cpp:84-116) lists every path of a superbubble; the
    ingest stops at 1024 and must SAY so: a layered superbubble of 11 x 2 inner nodes has 2^11
    paths -> the dump carries a TRUNCATED line after the bubble's 1024 ALLELE lines, a 10-layer one
    (exactly 1024 paths... listed in full) carries none below the bound (9 layers: 512)."""
    def layered(layers):
        names = lambda i: "utg%06dl" % i
        # one simple bubble first (so the chain has 2 bubbles), then the layered superbubble, whose sink
        # is the highest node: the reference's DFS starts at the sink and follows its '-' links
        head, src = (2, 3), 4
        nid = 5
        lay = []
        for _ in range(layers):
            lay.append((nid, nid + 1)); nid += 2
        snk = nid
        n_nodes = snk
        S = "".join("S\t%s\tA\n" % names(i) for i in range(1, n_nodes + 1))
        L = []
        def link(a, b):
            L.append("L\t%s\t+\t%s\t+\t0M\n" % (names(a), names(b)))
            L.append("L\t%s\t-\t%s\t-\t0M\n" % (names(b), names(a)))
        for x in head:
            link(1, x)
            link(x, src)
        for x in lay[0]:
            link(src, x)
        for k in range(layers - 1):
            for x in lay[k]:
                for y in lay[k + 1]:
                    link(x, y)
        for x in lay[-1]:
            link(x, snk)
        return S + "".join(L)
    for layers, expect_cut in ((11, True), (9, False)):
        gfa = tmp_path / ("l%d.gfa" % layers)
        gfa.write_text(layered(layers))
        out = tmp_path / ("l%d.txt" % layers)
        capi.ingest_dump(str(gfa), "", str(out))
        lines = open(out).read().split("\n")
        big = [ln for ln in lines if ln.startswith("BUBBLE ") and int(ln.split()[5]) == 2 * layers]
        assert len(big) == 1, "the layered part must be ONE bubble"
        cid, bid = big[0].split()[1:3]
        n_alleles = sum(1 for ln in lines if ln.startswith("ALLELE %s %s " % (cid, bid)))
        cut = [ln for ln in lines if ln.startswith("TRUNCATED")]
        if expect_cut:
            assert n_alleles == 1024 and cut == ["TRUNCATED %s %s 1024" % (cid, bid)]
        else:
            assert n_alleles == 2 ** layers and cut == []
