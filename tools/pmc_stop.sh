#!/bin/bash
# FETCH_SIZE of k_phase when it stops after S4 (1), S5 init (2), the S5 loop (3) or not at all (0): diagnostic build
cp ahsoka_amd/Ahsoka /tmp/A.keep
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -s -C ahsoka_amd/csrc clean >/dev/null; make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="-DAHK_DIAG" 2>&1 | grep -E "error"
for sa in 1 2 3 0; do
  AHSOKA_STOP_AFTER=$sa rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_stop$sa -- python3 bench.py --chains 1280 --steps 1 --warmup 0 --cpu-sample 0 > gpurun_out/pmc_stop$sa.log 2>&1
  f=$(ls -t gpurun_out/pmc_stop$sa/*/*counter_collection.csv | head -1)
  python3 - "$f" $sa <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_phase' in r['Kernel_Name'] and float(r['Counter_Value']) > 1e6:
        print('stop_after', sys.argv[2], r['Counter_Name'], float(r['Counter_Value'])*2048/1e9, 'GB (x2 corrected)')
PY
done
cp /tmp/A.keep ahsoka_amd/Ahsoka
