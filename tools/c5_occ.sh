#!/bin/bash
# config 5 (tail-bound by its largest chains) under a cap on workgroups per CU (diagnostic build, one box)
cp ahsoka_amd/Ahsoka /tmp/A.keep
make -s -C ahsoka_amd/csrc clean >/dev/null; make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="-DAHK_DIAG" 2>&1 | grep error
for w in 5 4 3 2 5; do
  AHSOKA_WG_PER_CU=$w timeout -k 10 300 python bench.py --config c5 --cpu-sample 0 --steps 3 2> gpurun_out/c5occ.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('WG/CU cap $w value', round(d['value']), 'ms_per_step', round(d['ms_per_step'],1), 'k_phase_ms', round(d['roofline']['kernel_ms'],1))"
  grep "^ahsoka: class" gpurun_out/c5occ.err | sort -u | head -5
done
cp /tmp/A.keep ahsoka_amd/Ahsoka
