"""ctypes access to oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU checker (oracle/oracle.cpp).  Product code never imports
this module; it lives under tests/ on purpose.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from ahsoka_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "liboracle.so")
REF_LIB = os.path.join(ORACLE_DIR, "_ref", "libref.so")
REF_DUMP = os.path.join(ORACLE_DIR, "_ref", "ref_dump")

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            build()
        L = C.CDLL(ORACLE_LIB)
        L.oracle_phase_batch.argtypes = [C.POINTER(capi.CChainBatch), C.POINTER(capi.CPhaseParams),
                                         C.POINTER(capi.CPhaseResult), C.c_int, C.c_int]
        L.oracle_result_free.argtypes = [C.POINTER(capi.CPhaseResult)]
        L.oracle_result_free.restype = None
        L.oracle_is_subset.argtypes = [capi.i32p, C.c_int32, capi.i32p, C.c_int32, C.c_int32]
        L.oracle_det_log.argtypes = [C.c_double]
        L.oracle_det_log.restype = C.c_double
        L.oracle_std_sort_hi16.argtypes = [C.POINTER(C.c_uint32), C.c_int32]
        L.oracle_std_sort_hi16.restype = None
        _lib = L
    return _lib


def phase_batch(batch, params=None, n_threads=1, brute_ce=False):
    prm = params if params is not None else capi.default_params()
    r = capi.CPhaseResult()
    rc = lib().oracle_phase_batch(batch.ptr(), C.byref(prm), C.byref(r), n_threads, 1 if brute_ce else 0)
    assert rc == 0, rc
    try:
        return capi.result_from_c(r)
    finally:
        lib().oracle_result_free(C.byref(r))


def is_subset(path, aln, partial):
    p = np.ascontiguousarray(path, dtype=np.int32)
    a = np.ascontiguousarray(aln, dtype=np.int32)
    pp = p.ctypes.data_as(capi.i32p) if p.size else C.cast(None, capi.i32p)
    ap = a.ctypes.data_as(capi.i32p) if a.size else C.cast(None, capi.i32p)
    return bool(lib().oracle_is_subset(pp, len(p), ap, len(a), 1 if partial else 0))


def ref_lib():
    """The reference's own is_subset (oracle/_ref, built by `make -C oracle _ref`)."""
    if not os.path.exists(REF_LIB):
        return None
    L = C.CDLL(REF_LIB)
    L.ref_is_subset.argtypes = [capi.i32p, C.c_int, capi.i32p, C.c_int, C.c_int]
    return L
