#!/bin/bash
# the judged artefacts of one kernel revision, all from the same box, same HEAD:
#   gpurun_out/<tag>_bench.json            default bench.py line (with cpu_baseline)
#   gpurun_out/<tag>_bench_c{2,4,5}.json   the other configurations on the same kernels (parity-test cases)
#   gpurun_out/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>_pmc.txt               FETCH_SIZE / WRITE_SIZE passes (tools/pmc.sh, full c3 workload)
#   gpurun_out/<tag>_c4_pmc.txt            the same for config 4 (with its own kernel time: <tag>_bench_c4.json)
#   gpurun_out/<tag>_end_to_end.json       tools/e2e_breakdown.py (host-buffer rate, multi-device prologue)
# copy them into profiles/ afterwards (tools/mk_pmc_json.py builds the PMC summaries).
set -e
TAG=${1:-r03_vX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
for c in c2 c4 c5; do python3 bench.py --config $c --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/${TAG}_bench_$c.json 2>/dev/null; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python3 bench.py --steps 2 --warmup 0 --cpu-sample 0 > gpurun_out/${TAG}_stats.log 2>&1
cp $(ls gpurun_out/${TAG}_stats/*/*kernel_stats.csv | head -1) gpurun_out/${TAG}_kernel_stats.csv
bash tools/pmc.sh 10000 c3 > gpurun_out/${TAG}_pmc.txt 2>&1
bash tools/pmc.sh 2000 c4 > gpurun_out/${TAG}_c4_pmc.txt 2>&1
python3 tools/e2e_breakdown.py c3 > gpurun_out/${TAG}_end_to_end.json 2>/dev/null
cat gpurun_out/${TAG}_bench.json; head -4 gpurun_out/${TAG}_kernel_stats.csv; cat gpurun_out/${TAG}_pmc.txt gpurun_out/${TAG}_c4_pmc.txt gpurun_out/${TAG}_end_to_end.json
for c in c2 c4 c5; do python3 -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_$c.json')); print('$c', round(d['value']), d['roofline']['kernel_ms'], d['roofline']['k_readset_ms'])"; done
