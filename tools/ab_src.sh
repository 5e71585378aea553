#!/bin/bash
# same-box A/B of two k_phase sources (GPU box only): tools/ab_src.sh <other k_phase.hip> against the working tree's;
# e.g. git show HEAD:ahsoka_amd/csrc/k_phase.hip > tmp_ab/k_phase_head.hip before the gpurun call
OTHER=${1:?path of the other k_phase.hip}
set -e
cp ahsoka_amd/Ahsoka /tmp/A.keep
cp ahsoka_amd/csrc/k_phase.hip /tmp/k_phase_new.hip
bench() { timeout -k 10 200 python bench.py --cpu-sample 0 --steps 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 value', round(d['value']), 'k_phase_ms', round(d['roofline']['kernel_ms'],1))"; }
for round in 1 2; do
  cp $OTHER ahsoka_amd/csrc/k_phase.hip
  make -s -C ahsoka_amd/csrc clean >/dev/null; make -s -C ahsoka_amd/csrc ../libahsoka_phase.so 2>&1 | grep -E "error" && exit 1
  bench head
  cp /tmp/k_phase_new.hip ahsoka_amd/csrc/k_phase.hip
  make -s -C ahsoka_amd/csrc clean >/dev/null; make -s -C ahsoka_amd/csrc ../libahsoka_phase.so 2>&1 | grep -E "error" && exit 1
  bench new
done
cp /tmp/A.keep ahsoka_amd/Ahsoka
