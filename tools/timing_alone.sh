#!/bin/bash
# per-phase shader-clock breakdown of ONE chain that has the GPU to itself, in the 4- and 8-wave forms
make -s -C ahsoka_amd/csrc clean >/dev/null
make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="-DAHK_DIAG -DAHK_TIMING=1" 2>&1 | grep -E "error"
for wm in 4 8; do
  echo "== wave mode $wm, one chain alone"
  timeout -k 10 300 python bench.py --config c3 --chains 1 --wave-mode $wm --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "^chain|^merges|^heavy|^plain loop wave|kernel_ms" | cut -c1-700
done
for sa in 1 2 3 0; do AHSOKA_STOP_AFTER=$sa timeout -k 10 300 python bench.py --config c3 --chains 1 --wave-mode 4 --steps 1 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().split('\n') if l.startswith('{')][0]); print('alone stop_after=$sa k_phase_ms', round(d['roofline']['kernel_ms'],2))"; done
