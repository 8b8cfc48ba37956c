# A/B of environment knobs on one box: bash tools/ab_env.sh "NAME1 VAR=.. VAR=.." "NAME2 ..." ; three interleaved repetitions
cfgs=("$@")
for rep in 1 2 3; do
  for cfg in "${cfgs[@]}"; do
    name=${cfg%% *}; vars=${cfg#* }
    echo -n "$name: "; env $vars timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value'],1), round(r['ms_per_step'],2))"
  done
done
