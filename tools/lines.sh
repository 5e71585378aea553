#!/bin/bash
# where do k_phase's memory requests go?  Diagnostic build -DAHK_LINES: workgroup 0 counts, per load/store site of
# the S5 loop, wave instructions and the distinct 128-B lines they touch, for every chain it runs (GPU box only).
CFG=${1:-c3}; CH=${2:-256}
make -s -C ahsoka_amd/csrc clean >/dev/null
make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="-DAHK_DIAG -DAHK_LINES" 2>&1 | grep -E "error"
timeout -k 10 600 python bench.py --config $CFG --chains $CH --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "^lines|kernel_ms" | cut -c1-400 | tee gpurun_out/lines_raw_$CFG.txt | python3 tools/lines_fmt.py ${LINES_CHAINS:-2}
