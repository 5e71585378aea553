#!/bin/bash
# SQ-level counters of k_phase: where do the wave cycles go?  (one rocprofv3 pass per set, the program
# directly after `--`; 1280 chains = one chain per slot of the persistent grid)
#   tools/pmc_sq.sh [chains] [out-prefix]   -> <out-prefix>.txt lines "COUNTER value"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CH=${1:-1280}
OUT=${2:-gpurun_out/pmc_sq}
i=0
: > $OUT.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQC_ICACHE_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d ${OUT}_$i -- python3 bench.py --chains $CH --steps 1 --warmup 0 --cpu-sample 0 > ${OUT}_$i.log 2>&1
  f=$(ls -t ${OUT}_$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "set $i failed: $(grep -m1 -o 'error code [0-9]*: [^"]*' ${OUT}_$i.log || tail -2 ${OUT}_$i.log)"; continue; fi
  python3 - "$f" <<'PY' | tee -a $OUT.txt
import csv,sys
acc={}
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_phase' in r['Kernel_Name']:
        acc[r['Counter_Name']] = acc.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
for k,v in acc.items(): print(k, v)
PY
done
