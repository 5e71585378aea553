#!/usr/bin/env python3
"""Experiment support: how often could two consecutive single-read separations of the cluster
editing loop be issued together?  Runs ONE synthetic chain through an oracle built with
-DORACLE_CE_TRACE (see the block in oracle/oracle.cpp) and pairs the logged decisions greedily.

    g++ ... -DORACLE_CE_TRACE -shared -o /tmp/cetrace/liboracle.so oracle/oracle.cpp
    python tools/ce_pair_stats.py /tmp/cetrace/liboracle.so c3|c4
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ahsoka_amd import capi  # noqa: E402


def main():
    libp, cfg = sys.argv[1], sys.argv[2]
    trace = "/tmp/cetrace/trace.txt"
    os.environ["ORACLE_CE_TRACE"] = trace
    L = C.CDLL(libp)
    L.oracle_phase_batch.argtypes = [C.POINTER(capi.CChainBatch), C.POINTER(capi.CPhaseParams), C.POINTER(capi.CPhaseResult), C.c_int, C.c_int]
    if cfg == "c3":
        b = capi.synth_batch(n_chains=1, n_bubbles=50, n_reads=1000, ploidy=2, seed=3)
        prm = capi.default_params(ploidy=2)
    else:
        b = capi.synth_batch(n_chains=1, n_bubbles=50, n_reads=2000, ploidy=4, seed=3)
        prm = capi.default_params(ploidy=4)
    r = capi.CPhaseResult()
    rc = L.oracle_phase_batch(b.ptr(), C.byref(prm), C.byref(r), 1, 0)
    assert rc == 0
    rows = [tuple(int(x) for x in ln.split()) for ln in open(trace)]
    n = len(rows)
    fast = [(r[0] == 0 and r[3] == 1 and r[4] == 1) for r in rows]
    print("iterations", n, "fast", sum(fast))
    # greedy pairing: at a fast iteration i, the speculated partner is the runner-up row maximum
    i = steps = pairs = 0
    fail_touch = fail_next = fail_f = 0
    while i < n:
        steps += 1
        if fast[i] and i + 1 < n:
            _, u, v, _, _, vf, vp, a2, b2, v2 = rows[i]
            if a2 >= 0 and a2 not in (u, v) and b2 not in (u, v):
                nx = rows[i + 1]
                if fast[i + 1] and (nx[1], nx[2]) == (a2, b2):
                    if vf < v2:
                        pairs += 1
                        i += 2
                        continue
                    fail_f += 1
                else:
                    fail_next += 1
            else:
                fail_touch += 1
        i += 1
    print("steps", steps, "pairs", pairs, "fail: runner-up touches", fail_touch, "next differs", fail_next, "icf bound", fail_f)


if __name__ == "__main__":
    main()
