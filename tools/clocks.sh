#!/bin/bash
# shader / memory clock while k_phase runs on the whole chip and on every 8th CU (diagnostic build for the mask):
# is the full-chip slowdown a power-limited clock or the memory system?
make -s -C ahsoka_amd/csrc clean >/dev/null
make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="-DAHK_DIAG" 2>&1 | grep -E "error"
sample() { for i in $(seq 1 $1); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" | tr '\n' ' ' | sed 's/  */ /g' | cut -c1-300; echo; sleep 0.4; done; }
echo "== idle"; sample 2
echo "== full chip, 10000 chains"
python bench.py --steps 6 --warmup 1 --cpu-sample 0 > /tmp/b1.json 2>/dev/null &
sleep 6; sample 8; wait
python -c "import json; d=json.load(open('/tmp/b1.json')); print('full chip', round(d['value']), d['roofline']['kernel_ms'])"
echo "== every 8th CU, 1250 chains"
AHSOKA_CU_STRIDE=8 python bench.py --chains 1250 --steps 6 --warmup 1 --cpu-sample 0 > /tmp/b2.json 2>/dev/null &
sleep 4; sample 8; wait
python -c "import json; d=json.load(open('/tmp/b2.json')); print('1/8 chip', round(d['value']), d['roofline']['kernel_ms'])"
