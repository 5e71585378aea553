#!/usr/bin/env python3
"""Where does the time of one `ahsoka_phase_batch` call go (host-resident ChainBatch in, host result
out)?  Times the pieces of the same path through the staged entry points.  GPU box only.

    python tools/e2e_breakdown.py [c3|c2|c5] [n_chains]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ahsoka_amd import capi  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
    kw = dict(capi.CONFIGS[cfg])
    if len(sys.argv) > 2:
        kw["n_chains"] = int(sys.argv[2])
    t = time.perf_counter()
    b = capi.synth_batch(**kw)
    out = {"config": cfg, "chains": int(b.n_chains), "synth_s": round(time.perf_counter() - t, 3)}
    prm = capi.default_params(ploidy=kw.get("ploidy", 2))
    capi.phase_batch(capi.synth_batch(**dict(capi.CONFIGS["c1"])), capi.default_params())  # context + code objects
    t0 = time.perf_counter()
    d = capi.DeviceBatch(b, prm)
    t1 = time.perf_counter()
    d.phase()  # first call: slot workspaces are allocated
    t2 = time.perf_counter()
    d.phase()
    t3 = time.perf_counter()
    r = d.download()
    t4 = time.perf_counter()
    d.free()
    t5 = time.perf_counter()
    out.update({"upload_s": round(t1 - t0, 3), "phase_first_s": round(t2 - t1, 3), "phase_again_s": round(t3 - t2, 3),
                "download_s": round(t4 - t3, 3), "free_s": round(t5 - t4, 3), "kernels_ms": d.kernel_times_ms() if False else None})
    t6 = time.perf_counter()
    r2 = capi.phase_batch(b, prm)
    t7 = time.perf_counter()
    bub = r2.phased_bubbles(b.arrays()["bubble_off"])
    out.update({"phase_batch_s": round(t7 - t6, 3), "phased_bubbles": int(bub), "end_to_end_bubbles_per_s": round(bub / (t7 - t6))})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
