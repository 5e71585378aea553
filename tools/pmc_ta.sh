#!/bin/bash
# vector-memory path counters of k_phase: is the per-CU address/L1 pipeline the limit?
# One rocprofv3 pass per counter set, the program directly after `--`, each under `timeout -k 10 300`.
# A pass may hold at most TWO counters of one hardware block (TA_*, TCP_*): round 2 asked for four
# TA_* counters in one pass and rocprofiler refused -- rocprofiler_create_counter_config "error code
# 38: Request exceeds the capabilities of the hardware to collect", then SIGABRT of the profiled
# process (gpurun_out/pmc_ta3.log).  That was a profiler abort, not a kernel fault and not a hang.
# Sets 1 and 2 of round 2 completed: profiles/r02_v9_pmc_ta.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CH=${1:-1280}
OUT=${2:-gpurun_out/pmc_ta}
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
           "TA_BUSY_max GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d ${OUT}$i -- python3 bench.py --chains $CH --steps 1 --warmup 0 --cpu-sample 0 > ${OUT}$i.log 2>&1
  f=$(ls -t ${OUT}$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "set $i ($set) failed: $(grep -m1 -o 'error code [0-9]*: [^"]*' ${OUT}$i.log || tail -2 ${OUT}$i.log)"; continue; fi
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_phase' in r['Kernel_Name']:
        print(r['Counter_Name'], float(r['Counter_Value'])/1e9, 'G')
PY
done
