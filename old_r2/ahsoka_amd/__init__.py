"""ahsoka_amd -- MI355X-native bubble-chain phasing path of Ahsoka.

One hot path, behind a C-ABI (include/ahsoka_phase.h): the per-chain body of
``alignmentsToReadset`` (reference: src/alignmentstoreadset.cpp:55-493) as HIP
kernels for gfx950.  This package is the thin Python plumbing over that
library (ctypes), used by the tests, ``bench.py`` and the multi-GPU driver.
"""
from .capi import (  # noqa: F401
    CONFIGS, ChainBatch, DeviceBatch, PhaseResult, AhsokaError, backend_info, default_params, ingest, lib,
    phase_batch, render_chain, synth_batch, synth_write_text,
    ST_PHASED, ST_SKIPPED_LE1, ST_EMPTY_READSET, ST_NO_FULL_MATCH, ST_UNSUPPORTED,
)
