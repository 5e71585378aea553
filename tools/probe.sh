#!/bin/bash
# what does one more instruction cost inside the S5 loop?  Diagnostic builds with N extra dependent vector (or scalar)
# instructions per wave and single-read separation, benched on one box next to the unchanged kernel (config 3 and 4).
set -e
for v in "" "-DAHK_PROBE_VALU=128" "-DAHK_PROBE_SALU=128"; do
  make -s -C ahsoka_amd/csrc clean >/dev/null
  make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="$v" 2>&1 | grep -E "error" && exit 1
  timeout -k 10 200 python bench.py --cpu-sample 0 --steps 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] c3 value', round(d['value']), 'k_phase_ms', round(d['roofline']['kernel_ms'],1))"
done
