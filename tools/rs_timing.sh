#!/bin/bash
# per-phase shader-clock breakdown of k_readset_lds (diagnostic build -DAHK_RS_TIMING), workgroup 0's chain
CFG=${1:-c3}; CH=${2:-2560}
cp ahsoka_amd/Ahsoka /tmp/A.keep
make -s -C ahsoka_amd/csrc clean >/dev/null
make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="-DAHK_DIAG -DAHK_RS_TIMING" 2>&1 | grep -E "error"
AHSOKA_STOP_AFTER=1 timeout -k 10 300 python bench.py --config $CFG --chains $CH --steps 2 --warmup 1 --cpu-sample 0 2>&1 | grep -E "^k_readset_lds|k_readset_ms" | cut -c1-400 | head -6
cp /tmp/A.keep ahsoka_amd/Ahsoka
