#!/bin/bash
# same-box A/B of compile-time variants on configs 3 and 4 (GPU box only): tools/ab_flags.sh "<flags A>" "<flags B>" ...
# each variant: parity subset, then c3 (3 steps) and c4 (2 steps); twice, interleaved
set -e
for round in 1 2; do
for v in "$@"; do
  make -s -C ahsoka_amd/csrc clean >/dev/null
  make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="$v" 2>&1 | grep -E "error" && exit 1
  if [ $round = 1 ]; then timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c1 or c2 or fuzz or ploidy" > gpurun_out/variant_test.log 2>&1 || { echo "variant [$v] FAILED parity/timeout"; tail -5 gpurun_out/variant_test.log; exit 1; }; fi
  timeout -k 10 200 python bench.py --cpu-sample 0 --steps 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] c3', round(d['value']), 'k_phase_ms', round(d['roofline']['kernel_ms'],1))"
  timeout -k 10 200 python bench.py --config c4 --cpu-sample 0 --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] c4', round(d['value']), 'k_phase_ms', round(d['roofline']['kernel_ms'],1))"
done
done
