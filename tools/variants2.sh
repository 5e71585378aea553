#!/bin/bash
# like variants.sh, but benches BASELINE configs[2] and configs[3] per variant (GPU box only)
set -e
cp ahsoka_amd/Ahsoka /tmp/A.keep
for v in "$@"; do
  make -s -C ahsoka_amd/csrc clean >/dev/null
  make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="$v" 2>&1 | grep -E "error" && exit 1
  timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c1 or c2 or fuzz" > gpurun_out/variant_test.log 2>&1 || { echo "variant [$v] FAILED parity/timeout"; tail -3 gpurun_out/variant_test.log; exit 1; }
  for cfg in c3 c4; do
    timeout -k 10 200 python bench.py --config $cfg --cpu-sample 0 --steps 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] $cfg value', round(d['value']), 'k_phase_ms', round(d['roofline']['kernel_ms'],1))"
  done
done
cp /tmp/A.keep ahsoka_amd/Ahsoka
