#!/bin/bash
# per-wave cycle trace of the argmax and the single-read separation (diagnostic build -DAHK_FTRACE,
# on the GPU box only: the snapshot there is scratch, the tracked library is not touched)
CFG=${1:-c3}; CH=${2:-1280}
make -s -C ahsoka_amd/csrc clean >/dev/null
make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="-DAHK_DIAG -DAHK_FTRACE" 2>&1 | grep -E "error"
timeout -k 10 600 python bench.py --config $CFG --chains $CH --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "^ftrace|kernel_ms" | cut -c1-600
