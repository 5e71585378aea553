#!/bin/bash
# build k_phase variants on the GPU box and bench each on the SAME box (run-to-run noise
# across boxes is ~4 %, within a box ~0.5 %):  tools/variants.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
# Every variant must first pass the small parity tests; the script stops at the first failure
# or timeout (no further GPU step after a hang).
set -e
for v in "$@"; do
  make -s -C ahsoka_amd/csrc clean >/dev/null
  make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="$v" 2>&1 | grep -E "error" && exit 1
  timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c1 or c2 or fuzz" > gpurun_out/variant_test.log 2>&1 || { echo "variant [$v] FAILED parity/timeout"; tail -3 gpurun_out/variant_test.log; exit 1; }
  timeout -k 10 200 python bench.py --cpu-sample 0 --steps ${VSTEPS:-3} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant [$v] value', round(d['value']), 'k_phase_ms', round(d['roofline']['kernel_ms'],1))"
done
