# how the denoise chains share the machine: one chain alone vs 2 / 3 / 4 chains side by side
run() { echo -n "$1: "; shift; env "$@" 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value'],1), 'audio-s/s', round(r['ms_per_step'],2), 'ms/batch')"; }
run "B=16 one chain" LDC_NO_SPLIT=1 python bench.py --batch 16 --no-cpu-baseline --no-roofline
run "B=8 one chain" LDC_NO_SPLIT=1 python bench.py --batch 8 --no-cpu-baseline --no-roofline
run "B=32 one chain" LDC_NO_SPLIT=1 python bench.py --batch 32 --no-cpu-baseline --no-roofline
run "B=32 two chains" LDC_SPLIT=2 python bench.py --batch 32 --no-cpu-baseline --no-roofline
run "B=32 three chains" LDC_SPLIT=3 python bench.py --batch 32 --no-cpu-baseline --no-roofline
run "B=32 four chains" LDC_SPLIT=4 python bench.py --batch 32 --no-cpu-baseline --no-roofline
run "B=48 three chains" LDC_SPLIT=3 python bench.py --batch 48 --no-cpu-baseline --no-roofline
run "B=64 two chains" LDC_SPLIT=2 python bench.py --batch 64 --no-cpu-baseline --no-roofline
run "B=64 four chains" LDC_SPLIT=4 python bench.py --batch 64 --no-cpu-baseline --no-roofline
