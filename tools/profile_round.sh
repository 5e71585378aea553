#!/bin/bash
# the judged artefacts of one kernel version, all from the same box:
#   gpurun_out/<tag>_bench.json          default bench.py line (with cpu_baseline)
#   gpurun_out/<tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>_pmc.txt             FETCH_SIZE / WRITE_SIZE passes (tools/pmc.sh, full workload)
# copy them into profiles/ afterwards (tools/mk_pmc_json.py builds the PMC summary).
set -e
TAG=${1:-r01_vX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python3 bench.py --steps 2 --warmup 0 --cpu-sample 0 > gpurun_out/${TAG}_stats.log 2>&1
cp $(ls gpurun_out/${TAG}_stats/*/*kernel_stats.csv | head -1) gpurun_out/${TAG}_kernel_stats.csv
bash tools/pmc.sh 10000 > gpurun_out/${TAG}_pmc.txt 2>&1
cat gpurun_out/${TAG}_bench.json; head -3 gpurun_out/${TAG}_kernel_stats.csv; cat gpurun_out/${TAG}_pmc.txt
