#!/bin/bash
# address translation of k_phase: how often does a vector-memory request miss the per-CU translation cache (UTCL1) and go
# to the shared one (UTCL2)?  ~8 GB of per-chain state touched as random 128-B lines.  One rocprofv3 pass per set (at most
# two TCP_* counters per pass, see pmc_ta.sh), the program directly after `--`.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CH=${1:-1280}
OUT=${2:-gpurun_out/pmc_tlb}
i=0
for set in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
           "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
           "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
           "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d ${OUT}$i -- python3 bench.py --chains $CH --steps 1 --warmup 0 --cpu-sample 0 > ${OUT}$i.log 2>&1
  f=$(ls -t ${OUT}$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -z "$f" ]; then echo "set $i ($set) failed: $(grep -m1 -o 'error code [0-9]*: [^"]*' ${OUT}$i.log || tail -2 ${OUT}$i.log)"; continue; fi
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_phase' in r['Kernel_Name'] and float(r['Counter_Value']) > 0:
        print(r['Kernel_Name'][:24], r['Counter_Name'], float(r['Counter_Value'])/1e9, 'G')
PY
done
