#!/usr/bin/env python3
"""gpurun_out/<tag>_pmc.txt (tools/pmc.sh output) -> profiles/<tag>_pmc_hbm.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ahsoka_amd import capi
tag, config, chains = sys.argv[1], sys.argv[2], int(sys.argv[3])
# the kernel revision the counters were taken with: argument 4, else the library built here (run this
# right after the profile, before the kernels change)
rev = sys.argv[4] if len(sys.argv) > 4 else capi.kernel_revision()
k = {}
for line in open(f"gpurun_out/{tag}_pmc.txt"):
    p = line.split()
    if len(p) >= 4 and p[0] in ("FETCH_SIZE", "WRITE_SIZE"):
        p = [p[0], " ".join(p[1:-2]), p[-2], p[-1]]  # (a templated kernel's name starts with "void ": it has a blank)
        name = next((n for n in ("k_phase", "k_readset_lds", "k_readset", "k_sortperm") if n + "(" in p[1] or p[1].endswith(n)), None)
        if name is None:
            name = "k_phase" if "k_phase" in p[1] else ("k_sortperm" if "k_sortperm" in p[1] else ("k_readset_lds" if "k_readset_lds" in p[1] else "k_readset"))
        d = k.setdefault(name, {})
        d[p[0] + "_KB"] = d.get(p[0] + "_KB", 0.0) + float(p[3])  # (k_phase runs as two instances: narrow and wide induced costs)
for v in k.values():
    v["hbm_bytes_raw"] = (v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024
    v["hbm_bytes_fetch_x2"] = (2 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024
out = {
    "what": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc.sh), python3 bench.py --chains {chains} --steps 1 --warmup 0 (BASELINE configs[2], one launch of each kernel)",
    "config": config, "chains": chains, "kernel_rev": rev,
    "note": "hbm_bytes_fetch_x2 = (2*FETCH_SIZE + WRITE_SIZE)*1024 is the calibrated reading: FETCH_SIZE tallies 64 B per 128-B request on gfx950, for scattered 4/16-B reads as for wide streams (profiles/r01_fetch_calibration.json)",
    "kernels": k,
}
json.dump(out, open(f"profiles/{tag}_pmc_hbm.json", "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
