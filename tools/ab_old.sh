#!/bin/bash
# same-box A/B of the round-2 final tree against the working tree: bench.py of both, interleaved.  ${OLD:-old_r2}/ is not tracked:
# export it before the gpurun call with  git archive c2c743c ahsoka_amd bench.py include | tar -x -C ${OLD:-old_r2}  (mkdir -p ${OLD:-old_r2} first)
make -s -C ${OLD:-old_r2}/ahsoka_amd/csrc -j8 ../libahsoka_phase.so 2>&1 | grep -E "error"
run() { (cd $1 && python bench.py --cpu-sample 0 --steps ${VSTEPS:-4} --warmup 1 $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2 $3', round(d['value']), 'k_phase_ms', round(d['roofline']['kernel_ms'],1), 'k_readset_ms', round(d['roofline']['k_readset_ms'],2))"); }
for i in 1 2; do run ${OLD:-old_r2} OLD; run . NEW; done
run ${OLD:-old_r2} OLD "--config c2"; run . NEW "--config c2"
run ${OLD:-old_r2} OLD "--config c5"; run . NEW "--config c5"
