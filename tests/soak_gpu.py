#!/usr/bin/env python3
"""Extended differential soak (not collected by pytest): GPU vs CPU oracle, all stages, on many
more random irregular batches and synthetic shapes than the regular suite runs.

    python tests/soak_gpu.py [n_random] [n_shapes] [seed_offset]

Prints one line per 10 cases and a final verdict; exits non-zero on the first mismatch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ahsoka_amd import capi  # noqa: E402
import oracle_binding as ob  # noqa: E402
from random_batches import random_batch  # noqa: E402
from test_gpu_parity import _compare  # noqa: E402


def random_case(k, off, rng):
    """one random irregular batch (repeated-node paths, random S1 orders and guard flags, 1-3 slots),
    all stages against the oracle; every seventh also through the device-list entry"""
    b = random_batch(50000 + 1000 * off + k, dup_paths=(k % 5 == 0), n_chains=int(rng.choice([1, 4, 9])),
                     max_bubbles=int(rng.choice([6, 14, 40])), max_reads=int(rng.choice([30, 120, 400])))
    prm = capi.default_params(want_stages=1, min_mapq=int(rng.choice([90, 93, 95])), compaction_mode=int(k % 2),
                              max_slots=int(rng.choice([0, 0, 1, 3])), wave_mode=int(rng.choice([0, 4, 8, 16])))
    g, _ = _compare(b, prm)
    if k % 7 == 0:  # the device-list entry: shards on the same GPU, merged result identical
        m = capi.phase_batch_multi(b, [0, 0, 0], prm)
        np.testing.assert_array_equal(m.hap_allele, g.hap_allele)
        np.testing.assert_array_equal(m.stages["score_q"], g.stages["score_q"])


def shape_case(k, off, rng, max_reads=1600):
    """one random synthetic shape at ploidy 2/3/4 with integer and non-integer threading costs"""
    nb = int(rng.integers(2, 120))
    nr = int(rng.integers(10, max_reads))
    ploidy = int(rng.choice([2, 2, 2, 3, 4]))
    if ploidy > 2:
        nr = min(nr, 500)
        nb = min(nb, 24)
    nc = 3 if nr > 800 else 6
    b = capi.synth_batch(n_chains=nc, n_bubbles=nb, n_reads=nr, ploidy=ploidy, seed=7000 + 1000 * off + k,
                         mean_span=float(rng.choice([3.0, 7.0, 15.0])), allele_err=float(rng.choice([0.0, 0.02, 0.15])))
    _compare(b, capi.default_params(ploidy=ploidy, want_stages=1, max_slots=int(rng.choice([0, 2])), wave_mode=int(rng.choice([0, 4, 8, 16])),
                                    switch_cost=float(rng.choice([32.0, 5.0, 0.5])), affine_switch_cost=float(rng.choice([8.0, 0.0, 1.25]))))
    return nc, nb, nr, ploidy


def main():
    n_random = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    n_shapes = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    off = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # a different region of the random space per run
    rng = np.random.default_rng(20261004 + off)
    done = 0
    for k in range(n_random):
        random_case(k, off, rng)
        done += 1
        if done % 10 == 0:
            print(f"{done} cases ok", flush=True)
    for k in range(n_shapes):
        nc, nb, nr, ploidy = shape_case(k, off, rng)
        done += 1
        if done % 10 == 0:
            print(f"{done} cases ok (shape {nc}x{nb}x{nr} p{ploidy})", flush=True)
    print(f"SOAK OK: {done} cases bit-identical to the oracle")


if __name__ == "__main__":
    main()
