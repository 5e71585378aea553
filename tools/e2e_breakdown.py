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
    # the in-process multi-device entry: host time before the first upload with eight device threads (the
    # serial part -- cost model and partition -- plus the longest per-shard view / pack, which the device
    # threads do concurrently), and one call over {0,0,...}: eight shards side by side on this one GPU
    _, blocks, serial_ms, shard_ms = capi.multi_plan(b, 8, prm, check=False)
    out.update({"multi8_partition": "blocks (zero-copy views)" if blocks else "LPT (packed copies)",
                "multi8_host_serial_ms": round(serial_ms, 2), "multi8_host_shard_ms_max": round(shard_ms, 2),
                "multi8_host_before_first_upload_ms": round(serial_ms + shard_ms, 2)})
    t8 = time.perf_counter()
    r3 = capi.phase_batch_multi(b, [0] * 8, prm)
    t9 = time.perf_counter()
    assert (r3.hap_allele == r2.hap_allele).all() and (r3.status == r2.status).all()
    out.update({"multi8_same_gpu_s": round(t9 - t8, 3), "multi8_equals_single": True})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
