#!/bin/bash
# instruction counts of k_phase (one PMC pass): tools/pmc_inst.sh [chains]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CH=${1:-1280}
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_inst -- python3 bench.py --chains $CH --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/pmc_inst.log 2>&1
f=$(ls -t gpurun_out/pmc_inst/*/*counter_collection.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_phase' in r['Kernel_Name']:
        print(r['Counter_Name'], float(r['Counter_Value'])/1e9, 'G')
PY
