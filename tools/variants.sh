#!/bin/bash
# build k_phase variants on the GPU box and bench each (tuning aid): tools/variants.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
for v in "$@"; do
  make -s -C ahsoka_amd/csrc clean >/dev/null
  make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="$v" 2>&1 | grep -E "error"
  timeout -k 10 300 python bench.py --cpu-sample 0 --steps 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] value', round(d['value']), 'k_phase_ms', round(d['roofline']['kernel_ms'],1))"
done
