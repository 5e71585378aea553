#!/bin/bash
# vector-memory path counters of k_phase: is the per-CU address/L1 pipeline the limit?
# WARNING (rounds 1 and 2): the third counter set (TCP_*) makes rocprofv3 hang without output on this
# pool until the job is killed for silence -- run at most the first two sets, under `timeout -k 10 300`.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CH=${1:-1280}
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_ta$i -- python3 bench.py --chains $CH --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/pmc_ta$i.log 2>&1
  f=$(ls -t gpurun_out/pmc_ta$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "set $i failed: $(tail -2 gpurun_out/pmc_ta$i.log)"; continue; fi
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_phase' in r['Kernel_Name']:
        print(r['Counter_Name'], float(r['Counter_Value'])/1e9, 'G')
PY
done
