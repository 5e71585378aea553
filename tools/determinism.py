#!/usr/bin/env python3
"""Run the device path N times on ONE resident batch and compare every run with the first: iteration counts, cluster
counts and haplotypes must be identical bit for bit (the kernels contain no order-dependent arithmetic; a difference is
a race).  tools/determinism.py [config] [runs]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ahsoka_amd import capi

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
c = dict(capi.CONFIGS[cfg])
b = capi.synth_batch(**c)
dev = capi.DeviceBatch(b, capi.default_params(ploidy=c.get("ploidy", 2)))
dev.phase()
g0 = dev.download()
bad = 0
for r in range(1, runs):
    dev.phase()
    g = dev.download()
    d = int((g.n_iterations != g0.n_iterations).sum()) + int((g.hap_allele != g0.hap_allele).sum()) + int((g.hap_cluster != g0.hap_cluster).sum())
    if d:
        bad += 1
        print(f"run {r}: {int((g.n_iterations != g0.n_iterations).sum())} chains with another iteration count, {int((g.hap_allele != g0.hap_allele).sum())} haplotype entries differ", flush=True)
dev.free()
print(f"{cfg}: {runs} runs x {len(g0.n_iterations)} chains, {bad} runs differ from the first")
sys.exit(1 if bad else 0)
