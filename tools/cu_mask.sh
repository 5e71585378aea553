#!/bin/bash
# Per-CU bound or chip-wide bound?  The same five workgroups per CU on the whole chip and on every 2nd / 4th /
# 8th CU (a CU-masked stream, diagnostic build): if a chain takes as long on a quarter of the chip as on all
# of it, what the co-resident workgroups share sits inside the CU; if it gets faster, it is chip-wide (L2/HBM).
make -s -C ahsoka_amd/csrc clean >/dev/null
make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="-DAHK_DIAG" 2>&1 | grep -E "error"
for st in 1 2 4 8; do
  CH=$((1280 / st))
  AHSOKA_CU_STRIDE=$st timeout -k 10 300 python bench.py --chains $CH --steps 2 --warmup 1 --cpu-sample 0 2>gpurun_out/cu_mask.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('CU stride $st chains $CH: k_phase_ms', round(d['roofline']['kernel_ms'],1), 'bubbles/s', round(d['value']))"
done
