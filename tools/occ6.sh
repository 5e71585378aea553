#!/bin/bash
# is a sixth workgroup per CU (80 VGPRs) worth it?  c2 chains are small enough for the LDS to hold six.
set -e
for v in "5:" "6:-DAHK_WAVES_PER_SIMD=6" "5:" "6:-DAHK_WAVES_PER_SIMD=6"; do
  w=${v%%:*}; f=${v#*:}
  make -s -C ahsoka_amd/csrc clean >/dev/null
  make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="$f" 2>&1 | grep -E "error" && exit 1
  timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c1 or c2" > gpurun_out/variant_test.log 2>&1 || { echo "variant [$v] FAILED"; exit 1; }
  timeout -k 10 200 python bench.py --config c2 --chains 15360 --cpu-sample 0 --steps 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wg/cu $w [$f] c2 value', round(d['value']), 'k_phase_ms', round(d['roofline']['kernel_ms'],1))"
done
