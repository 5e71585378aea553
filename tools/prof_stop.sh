#!/bin/bash
# time to the end of S4 / S5-init / S5-loop / everything, on the GPU box only (diagnostic
# build -DAHK_DIAG: the product library has no such switch; the snapshot there is scratch)
CFG=${1:-c3}; CH=${2:-1280}
make -s -C ahsoka_amd/csrc clean >/dev/null
make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="-DAHK_DIAG" 2>&1 | grep -E "error"
for sa in 1 2 3 0; do AHSOKA_STOP_AFTER=$sa timeout -k 10 600 python bench.py --config $CFG --chains $CH --steps 1 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$CFG stop_after=$sa k_phase_ms', round(d['roofline']['kernel_ms'],1), 'k_readset_ms', round(d['roofline']['k_readset_ms'],2))"; done
