#!/usr/bin/env python3
"""Host-side ingest rows: time `ahsoka_ingest` on synthetic GFA/GAF text beside
the reference's own parsers (oracle/_ref/ref_dump with REF_TIME=1: graph.cpp, alignmentreader.cpp
and chainstoreadset.cpp compiled where they lie).  CPU only; writes one JSON line.

    python tools/ingest_bench.py [n_chains n_bubbles n_reads] [--ref-timeout S]
"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ahsoka_amd import capi  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ref_dump")


def main():
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    nc, nb, nr = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (1000, 20, 500)
    ref_timeout = 600
    if "--ref-timeout" in sys.argv:
        ref_timeout = int(sys.argv[sys.argv.index("--ref-timeout") + 1])
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        gfa, gaf = os.path.join(td, "x.gfa"), os.path.join(td, "x.gaf")
        capi.synth_write_text(gfa, gaf, n_chains=nc, n_bubbles=nb, n_reads=nr, ploidy=2, seed=2)
        sz = os.path.getsize(gfa) + os.path.getsize(gaf)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            b = capi.ingest(gfa, gaf)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out = {"workload": f"{nc} chains x {nb} bubbles x {nr} reads as GFA+GAF text", "text_MB": round(sz / 1e6, 1),
               "chains": int(b.n_chains), "ours_s": round(best, 3), "ours_MBps": round(sz / 1e6 / best, 1),
               "host_threads": os.cpu_count()}
        del b
        if os.path.exists(REF):
            try:
                env = dict(os.environ, REF_TIME="1")
                r = subprocess.run([REF, gfa, gaf], env=env, capture_output=True, text=True, timeout=ref_timeout)
                ref = json.loads(r.stdout.strip().splitlines()[-1])
                out["reference"] = ref
                out["speedup_vs_reference"] = round(ref["total_s"] / best, 1)
            except subprocess.TimeoutExpired:
                out["reference"] = f"did not finish in {ref_timeout} s"
        else:
            out["reference"] = "oracle/_ref not built"
        print(json.dumps(out))


if __name__ == "__main__":
    main()
