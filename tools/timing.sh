#!/bin/bash
# per-phase shader-clock breakdown of the S5 loop (diagnostic build -DAHK_TIMING, on the GPU box only:
# the snapshot there is scratch, the tracked library is not touched)
CFG=${1:-c3}; CH=${2:-1280}
make -s -C ahsoka_amd/csrc clean >/dev/null
make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="-DAHK_DIAG -DAHK_TIMING=${AHK_TIMING_LEVEL:-1}" 2>&1 | grep -E "error" 
timeout -k 10 600 python bench.py --config $CFG --chains $CH --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "^chain|^merges|^heavy|^plain|^M<=|^ncmax|kernel_ms" | cut -c1-900
