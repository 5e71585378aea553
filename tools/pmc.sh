#!/bin/bash
# HBM traffic of k_phase via rocprofv3 PMC counters, one counter per pass
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; MI355X_MICROARCH.md).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CH=${1:-512}
CFG=${2:-c3}
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_$ctr -- python3 bench.py --config $CFG --chains $CH --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/pmc_$ctr.log 2>&1
  f=$(ls -t gpurun_out/pmc_$ctr/*/*counter_collection.csv | head -1)
  python3 - "$f" $ctr <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(k in r['Kernel_Name'] for k in ('k_phase', 'k_readset', 'k_sortperm')):
        print(sys.argv[2], r['Kernel_Name'][:20], r['Counter_Name'], r['Counter_Value'])
PY
done
