#!/bin/bash
# like variants.sh, but benches the tail-bound configurations 2 and 5 per variant (GPU box only)
set -e
cp ahsoka_amd/Ahsoka /tmp/A.keep
for v in "$@"; do
  make -s -C ahsoka_amd/csrc clean >/dev/null
  make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="$v" 2>&1 | grep -E "error" && exit 1
  for cfg in c2 c5; do
    timeout -k 10 200 python bench.py --config $cfg --cpu-sample 0 --steps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] $cfg value', round(d['value']), 'k_phase_ms', round(d['roofline']['kernel_ms'],2))"
  done
done
cp /tmp/A.keep ahsoka_amd/Ahsoka
