#!/bin/bash
# config 4: four workgroups per CU with the live-word summary and the staged weight row, or five without the summary?
for v in "-DAHK_DIAG" "-DAHK_DIAG -DAHK_LWM=0"; do
  make -s -C ahsoka_amd/csrc clean >/dev/null
  make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="$v" 2>&1 | grep -E "error"
  for c in c4 c3; do python bench.py --config $c --cpu-sample 0 --steps 2 --warmup 1 2> /tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] $c', round(d['value']), 'k_phase_ms', round(d['roofline']['kernel_ms'],1))"; grep -m1 "^ahsoka: class" /tmp/err.txt; done
done
make -s -C ahsoka_amd/csrc clean >/dev/null; make -s -C ahsoka_amd/csrc ../libahsoka_phase.so 2>&1 | grep -E "error"
