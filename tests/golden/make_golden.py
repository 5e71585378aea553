#!/usr/bin/env python3
"""Regenerates the golden fixtures of tests/golden/ -- run in the build
container, where /root/reference exists:

    make -C oracle _ref && python tests/golden/make_golden.py

Inputs are this repository's own synthetic / hand-made GFA+GAF files (data);
the expected outputs (*.ref.txt) are what the REFERENCE's own parsers and
bubble finder compute for them: oracle/_ref/ref_dump links
/root/reference/src/{graph,alignmentreader,chainstoreadset}.cpp unchanged
(see oracle/Makefile).  No reference source text is stored here, only the
vectors.  `is_subset.ref.json` holds input/output vectors of the reference's
compiled is_subset (src/alignmentstoreadset.cpp:495-548).
"""
import ctypes as C
import json
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref.so")


def utg(i):
    return f"utg{i:06d}l"


def link(a, ao, b, bo):
    """both directions of one link, as hifiasm writes them"""
    flip = {"+": "-", "-": "+"}
    return [f"L\t{utg(a)}\t{ao}\t{utg(b)}\t{bo}\t0M", f"L\t{utg(b)}\t{flip[bo]}\t{utg(a)}\t{flip[ao]}\t0M"]


def toy():
    """Two chains.  Chain X: simple bubble, indel bubble (direct source->sink
    edge: a 2-node allele path, quirk Q6), 3-inner-node bubble (DFS paths,
    quirk Q7).  Chain Y: two simple bubbles.  Plus a node in no chain and a tip."""
    S, L = [], []
    for i in list(range(1, 13)) + list(range(20, 27)) + [40, 41]:
        S.append(f"S\t{utg(i)}\tACGT")
    # chain X: 1 -{2,3}- 4 -{5 | direct}- 6 -{7,8,9}- 10 ; 11,12 unused tips
    for a, b in [(1, 2), (1, 3), (2, 4), (3, 4), (4, 5), (5, 6), (4, 6), (6, 7), (6, 8), (6, 9), (7, 10), (8, 10), (9, 10)]:
        L += link(a, "+", b, "+")
    # chain Y: 20 -{21,22}- 23 -{24,25}- 26
    for a, b in [(20, 21), (20, 22), (21, 23), (22, 23), (23, 24), (23, 25), (24, 26), (25, 26)]:
        L += link(a, "+", b, "+")
    L += link(40, "+", 41, "+")  # a lone edge: nodes of no chain
    gfa = "\n".join(S + L) + "\n"
    reads = [
        ("r1", [1, 2, 4, 5, 6, 7, 10], "0.97"),
        ("r2", [1, 3, 4, 6, 8, 10], "0.95"),
        ("r3", [4, 5, 6, 9, 10], "0.935"),
        ("r4", [1, 2, 4], "0.92"),
        ("r5", [2, 4, 6, 7], "0.99"),
        ("r5", [6, 8, 10], "0.96"),          # same name again: same Read for getByName
        ("r6", [20, 21, 23, 24, 26], "0.98"),
        ("r7", [20, 22, 23, 25, 26], "0.94"),
        ("r8", [21, 23, 25], "0.931"),
        ("r9", [10, 40, 41, 20, 21, 23], "0.97"),  # spans both chains and chain-less nodes
        ("r10", [40, 41], "0.99"),                 # chain-less only: filed under chain 0
        ("r11", [3, 4, 5, 6, 7, 10], "1"),
    ]
    lines = []
    for name, nodes, ident in reads:
        path = "".join(">" + utg(n) for n in nodes)
        lines.append(f"{name}\t1000\t0\t1000\t+\t{path}\t1000\t0\t1000\t1000\t1000\t60\tNM:i:0\tAS:f:0\tdv:f:0\tid:f:{ident}")
    return gfa, "\n".join(lines) + "\n"


def simple_chain(base, B):
    """S and L lines of B simple biallelic bubbles in series on nodes base..base+3B"""
    S = [f"S\t{utg(base + x)}\tA" for x in range(3 * B + 1)]
    L = []
    for k in range(B):
        s = base + 3 * k
        for a, b in [(s, s + 1), (s, s + 2), (s + 1, s + 3), (s + 2, s + 3)]:
            L += link(a, "+", b, "+")
    return S, L


def multi():
    """Reads with SEVERAL alignments under one name (GraphAligner split alignments),
    built so that the two order/precision details of the reference decide the result:

    * S1 visits bubbles in unordered_map order (src/alignmentstoreadset.cpp:76,90) and
      the first hit creates the Read, fixing its mapq for the S1b filter (:117,158);
    * S2 adds a variant to an existing Read only if `alignment.id*100 > 90` in FLOAT
      (:245): 0.905 passes although (int)(0.905f*100) == 90.

    The reference numbers the 20-bubble chain (nodes 201..261) as chain 0 with bubble k
    = {src 261-3k, inner 259-3k | 260-3k, sink 258-3k}, the 6-bubble chain (nodes
    101..119) as chain 1 with bubble k = {119-3k, 117-3k | 118-3k, 116-3k}
    (BUBBLE lines of multi.ref.txt); S1ORDER is 19..13,0..12 resp. 5..0."""
    S1, L1 = simple_chain(101, 6)
    S2, L2 = simple_chain(201, 20)
    gfa = "\n".join(S1 + S2 + L1 + L2) + "\n"

    def full(top, k0, k1, alleles):  # bubbles k0..k1 of the chain whose bubble 0 source is `top`
        nodes = [top - 3 * k0]
        for k, a in zip(range(k0, k1 + 1), alleles):
            nodes += [top - 2 - 3 * k + a, top - 3 - 3 * k]
        return nodes

    def inner(top, k, a):
        return top - 2 - 3 * k + a

    reads = []
    # ---- chain 0 (top 261) ----
    h0 = [0, 1, 1, 0, 0, 1, 0, 1, 1]
    h1 = [1 - x for x in h0]
    for i, (k0, k1, h, ident) in enumerate([(0, 4, h0, "0.97"), (0, 5, h1, "0.96"), (2, 7, h0, "0.99"), (3, 8, h1, "0.95"),
                                            (1, 6, h0, "0.94"), (4, 8, h1, "0.98"), (0, 3, h1, "0.93"), (5, 8, h0, "0.97")]):
        reads.append((f"a{i}", full(261, k0, k1, h[k0:k1 + 1]), ident))
    # mB: a 0.95 alignment over bubbles 2,3 and a 0.905 alignment touching only the inner node of
    # bubble 10 (nobody else covers 10): position 10 is phased iff the float guard of :245 is used
    reads.append(("mB", full(261, 2, 3, [1, 0]), "0.95"))
    reads.append(("mB", [inner(261, 10, 0)], "0.905"))
    # mD: D1 (0.96) over bubbles 0,1 and D2 (0.92) over bubbles 11,12.  In S1 order (.., 0, 1, .., 11, 12)
    # D1 creates the Read => mapq 96 => kept => last position 12 = maxpos is a boundary => S2 visits 12 and
    # 13.  (Reverse order would make D2 the creator: mapq 92, filtered, 12 and 13 never visited.)
    reads.append(("mD", full(261, 0, 1, [0, 1]), "0.96"))
    reads.append(("mD", full(261, 11, 12, [0, 0]), "0.92"))
    # partial-only coverage of bubbles 12 and 13 (inner nodes, never source AND sink)
    reads.append(("p1", [inner(261, 12, 0), inner(261, 13, 1)], "0.97"))
    reads.append(("p2", [inner(261, 12, 1), inner(261, 13, 0)], "0.98"))
    reads.append(("p3", [inner(261, 12, 0), inner(261, 13, 1)], "0.95"))
    # ---- chain 1 (top 119) ----
    g0 = [1, 0, 0, 1, 1]
    g1 = [1 - x for x in g0]
    for i, (k0, k1, h, ident) in enumerate([(0, 3, g0, "0.97"), (0, 3, g1, "0.96"), (1, 3, g0, "0.99"), (2, 3, g1, "0.95")]):
        # full matches end at bubble 3 (last position 3: boundaries 3,4); each read also touches an inner
        # node of bubbles 4 and 5 -- partial matches only, their sinks 104 / 101 are missing
        reads.append((f"b{i}", full(119, k0, k1, h[k0:k1 + 1]) + [inner(119, 4, h[4]), inner(119, 5, i & 1)], ident))
    # mA: A1 (0.92) over bubbles 0,1 comes first in the file, A2 (0.95) over bubbles 4,5.  S1 order is
    # 5,4,..,0: A2 creates the Read => mapq 95 => kept => last position 5 = maxpos is a boundary => S2
    # visits bubble 5.  (Ascending order would make A1 the creator: mapq 92, filtered, 5 never visited.)
    reads.append(("mA", full(119, 0, 1, [0, 1]), "0.92"))
    reads.append(("mA", full(119, 4, 5, [1, 0]), "0.95"))
    lines = []
    for name, nodes, ident in reads:
        path = "".join(">" + utg(n) for n in nodes)
        lines.append(f"{name}\t1000\t0\t1000\t+\t{path}\t1000\t0\t1000\t1000\t1000\t60\tNM:i:0\tAS:f:0\tdv:f:0\tid:f:{ident}")
    return gfa, "\n".join(lines) + "\n"


def sbub():
    """A chain simple bubble - SUPERBUBBLE - simple bubble.  The superbubble 4 -> 12 has seven
    inner nodes (5..11, two stacked levels of branching) and six source-sink paths: the
    reference lists them depth-first from the sink in link order (findPathsComplex,
    src/chainstoreadset.cpp:84-116, quirk Q7); the L lines are written in an order that makes
    the listing differ from the sorted one."""
    S = [f"S\t{utg(i)}\tA" for i in range(1, 16)]
    L = []
    E = [(1, 2), (1, 3), (2, 4), (3, 4),
         (4, 6), (4, 5), (5, 8), (5, 7), (6, 8), (6, 9), (7, 10), (8, 11), (8, 10), (9, 11), (10, 12), (11, 12),
         (12, 13), (12, 14), (13, 15), (14, 15)]
    for a, b in E:
        L += link(a, "+", b, "+")
    gfa = "\n".join(S + L) + "\n"
    reads = [("s1", [1, 2, 4, 5, 7, 10, 12, 13, 15], "0.97"), ("s2", [1, 3, 4, 6, 9, 11, 12, 14, 15], "0.96"),
             ("s3", [2, 4, 5, 8, 11, 12, 13], "0.95"), ("s4", [3, 4, 6, 8, 10, 12, 14, 15], "0.98"),
             ("s5", [1, 2, 4, 5, 7, 10, 12], "0.99"), ("s6", [4, 6, 9, 11, 12, 14, 15], "0.94")]
    lines = []
    for name, nodes, ident in reads:
        path = "".join(">" + utg(n) for n in nodes)
        lines.append(f"{name}\t1000\t0\t1000\t+\t{path}\t1000\t0\t1000\t1000\t1000\t60\tNM:i:0\tAS:f:0\tdv:f:0\tid:f:{ident}")
    return gfa, "\n".join(lines) + "\n"


S1_SIZES = list(range(2, 31)) + [47, 59, 60, 97, 127, 128, 199, 257, 500, 541, 1000]


def main():
    from ahsoka_amd import capi

    if not os.path.exists(REF_DUMP):
        raise SystemExit("build oracle/_ref first: make -C oracle _ref")
    # fixture 1: BASELINE configs[0] (10 chains x 5 bubbles, 50 reads) as text
    capi.synth_write_text(os.path.join(HERE, "c1.gfa"), os.path.join(HERE, "c1.gaf"), **capi.CONFIGS["c1"])
    # fixture 2: hand-made toy
    gfa, gaf = toy()
    open(os.path.join(HERE, "toy.gfa"), "w").write(gfa)
    open(os.path.join(HERE, "toy.gaf"), "w").write(gaf)
    # fixture 2b: several alignments per read name (S1 order, float guard of :245)
    gfa, gaf = multi()
    open(os.path.join(HERE, "multi.gfa"), "w").write(gfa)
    open(os.path.join(HERE, "multi.gaf"), "w").write(gaf)
    # fixture 2d: a seven-inner-node superbubble (allele-path listing order of findPathsComplex)
    gfa, gaf = sbub()
    open(os.path.join(HERE, "sbub.gfa"), "w").write(gfa)
    open(os.path.join(HERE, "sbub.gaf"), "w").write(gaf)
    for stem in ("c1", "toy", "multi", "sbub"):
        out = subprocess.run([REF_DUMP, os.path.join(HERE, stem + ".gfa"), os.path.join(HERE, stem + ".gaf")],
                             capture_output=True, text=True, check=True).stdout
        open(os.path.join(HERE, stem + ".ref.txt"), "w").write(out)
        # the reference writes a side file next to the GAF; not a fixture
        side = os.path.join(HERE, stem + "-alignment_identities.txt")
        if os.path.exists(side):
            os.remove(side)
    # fixture 2c: S1 bubble visiting order (iteration order of the reference's copied
    # unordered_map, src/alignmentstoreadset.cpp:76,90) for chains of many sizes.  The GFA
    # text is the product's deterministic generator (regenerated by the test, not stored);
    # stored: the S1ORDER line the reference build printed for it.
    import tempfile
    with tempfile.TemporaryDirectory() as td, open(os.path.join(HERE, "s1order.ref.txt"), "w") as f:
        for B in S1_SIZES:
            g, a = os.path.join(td, "x.gfa"), os.path.join(td, "x.gaf")
            capi.synth_write_text(g, a, n_chains=1, n_bubbles=B, n_reads=0, seed=1)
            out = subprocess.run([REF_DUMP, g, a], capture_output=True, text=True, check=True).stdout
            line = [ln for ln in out.split("\n") if ln.startswith("S1ORDER ")]
            assert len(line) == 1
            f.write(f"B {B} " + " ".join(line[0].split()[2:]) + "\n")
    # fixture 3: is_subset vectors from the reference's compiled function
    L = C.CDLL(REF_LIB)
    i32p = C.POINTER(C.c_int32)
    L.ref_is_subset.argtypes = [i32p, C.c_int, i32p, C.c_int, C.c_int]
    rng = random.Random(0xA450CA)
    vecs = []
    for _ in range(400):
        naln = rng.randint(0, 9)
        aln = [rng.randint(1, 12) for _ in range(naln)]
        kind = rng.random()
        if kind < 0.5 and naln >= 2:
            k = rng.randint(2, min(4, naln))
            path = rng.sample(aln, k)
        else:
            path = [rng.randint(1, 12) for _ in range(rng.randint(2, 4))]
        if rng.random() < 0.1:
            path.append(path[0])  # repeated node: std::includes is a multiset test
        for partial in (0, 1):
            pa = (C.c_int32 * len(path))(*path)
            aa = (C.c_int32 * max(1, len(aln)))(*aln) if aln else (C.c_int32 * 1)()
            got = L.ref_is_subset(pa, len(path), aa, len(aln), partial)
            vecs.append({"path": path, "aln": aln, "partial": partial, "expect": int(got)})
    json.dump(vecs, open(os.path.join(HERE, "is_subset.ref.json"), "w"))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
