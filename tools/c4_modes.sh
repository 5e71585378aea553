run() { python bench.py "$@" --cpu-sample 0 --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '->', round(d['value']), 'bubbles/s k_phase_ms', round(d['roofline']['kernel_ms'],2))"; }
run --config c4 --wave-mode 4
run --config c4 --wave-mode 8
run --config c4 --wave-mode 16
run --config c2 --wave-mode 4
run --config c2 --wave-mode 8
