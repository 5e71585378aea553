"""A 60-case slice of the differential soak (tests/soak_gpu.py) inside the regular `-m gpu`
suite, so that the driver's round-end run covers it: 45 random irregular batches (repeated-node
allele paths, random S1 visiting orders and guard flags, 1-3 slots, the device-list entry on every
seventh) and 15 random synthetic shapes at ploidy 2/3/4, all stages bit-exact against the oracle.
The slice uses its own region of the random space (offset 3), not the one the long soak walks."""
import numpy as np
import pytest

import soak_gpu

pytestmark = pytest.mark.gpu


def test_soak_slice_random_batches():
    rng = np.random.default_rng(20261004 + 3)
    for k in range(45):
        soak_gpu.random_case(k, 3, rng)


def test_soak_slice_shapes():
    rng = np.random.default_rng(20261004 + 1003)
    for k in range(15):
        soak_gpu.shape_case(k, 3, rng, max_reads=1100)
