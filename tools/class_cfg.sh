#!/bin/bash
# which LDS extras and how many workgroups per CU does each size class get?  (diagnostic build prints build_classes' decision)
make -s -C ahsoka_amd/csrc clean >/dev/null
make -s -C ahsoka_amd/csrc ../libahsoka_phase.so EXTRA="-DAHK_DIAG" 2>&1 | grep -E "error"
for c in c2 c3 c4 c5; do echo "== $c"; python bench.py --config $c --steps 1 --warmup 0 --cpu-sample 0 2>&1 | grep -E "^ahsoka: class" | sort | uniq -c; done
make -s -C ahsoka_amd/csrc clean >/dev/null; make -s -C ahsoka_amd/csrc ../libahsoka_phase.so 2>&1 | grep -E "error"
