#!/bin/bash
# the forms of k_phase (4 / 8 / 16 wavefronts per workgroup, 0 = automatic by class size) on the
# latency-bound workloads: config 5 (bound by its largest chain) and config 3 with one chain per CU
run() { python bench.py "$@" --cpu-sample 0 --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value']), 'bubbles/s k_phase_ms', round(d['roofline']['kernel_ms'],2))"; }
for wm in 4 8 16 0; do run --config c5 --wave-mode $wm; done
for wm in 4 8 16; do run --config c3 --chains 256 --wave-mode $wm; done
for wm in 4 8; do run --config c3 --chains 512 --wave-mode $wm; done
for wm in 4 16; do run --config c3 --chains 1 --wave-mode $wm; done
run --config c3
