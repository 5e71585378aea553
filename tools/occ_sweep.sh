#!/bin/bash
# occupancy sweep: rebuild k_phase for 4/5/6 waves per SIMD on the GPU box and time 1536 chains
cd $GRAFT_REPO_ROOT
for w in 4 5 6; do
  make -s -C ahsoka_amd/csrc clean >/dev/null; make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="-DAHK_DIAG -DAHK_WAVES_PER_SIMD=$w" 2>&1 | grep -E "error"
  AHSOKA_WG_PER_CU=$w timeout -k 10 300 python bench.py --chains $((256*w)) --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('waves/SIMD=$w chains', d['config']['chains_total'], 'bubbles/s', round(d['value']), 'k_phase_ms', round(d['roofline']['kernel_ms'],1))"
done
