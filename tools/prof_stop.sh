#!/bin/bash
# time to the end of S5-init / S5-loop / everything (profiling aid: AHSOKA_STOP_AFTER)
CFG=${1:-c3}; CH=${2:-1280}
for sa in 1 2 3 0; do AHSOKA_STOP_AFTER=$sa timeout -k 10 600 python bench.py --config $CFG --chains $CH --steps 1 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$CFG stop_after=$sa k_phase_ms', round(d['roofline']['kernel_ms'],1))"; done
