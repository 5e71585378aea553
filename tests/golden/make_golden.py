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
    for stem in ("c1", "toy"):
        out = subprocess.run([REF_DUMP, os.path.join(HERE, stem + ".gfa"), os.path.join(HERE, stem + ".gaf")],
                             capture_output=True, text=True, check=True).stdout
        open(os.path.join(HERE, stem + ".ref.txt"), "w").write(out)
        # the reference writes a side file next to the GAF; not a fixture
        side = os.path.join(HERE, stem + "-alignment_identities.txt")
        if os.path.exists(side):
            os.remove(side)
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
