#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o gpurun_out/calib_fetch || exit 1
./gpurun_out/calib_fetch
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/calib -- ./gpurun_out/calib_fetch > gpurun_out/calib.log 2>&1
f=$(ls -t gpurun_out/calib/*/*counter_collection.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'scattered' in r['Kernel_Name'] or 'stream16' in r['Kernel_Name']:
        print(r['Kernel_Name'][:40], r['Counter_Name'], float(r['Counter_Value'])*1024, 'bytes')
PY
