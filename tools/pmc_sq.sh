#!/bin/bash
# SQ-level counters of k_phase: where do the wave cycles go?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CH=${1:-1024}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_sq$i -- python3 bench.py --chains $CH --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/pmc_sq$i.log 2>&1
  f=$(ls -t gpurun_out/pmc_sq$i/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_phase' in r['Kernel_Name']:
        print(r['Counter_Name'], r['Counter_Value'])
PY
done
