# same-box A/B runs: bash tools/ab_r04.sh "<label> <ENV=.. ENV=..> [-- bench args]" ...
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-$PWD}
for spec in "$@"; do
  envs=${spec%%--*}; extra=""
  if [[ "$spec" == *--* ]]; then extra="--${spec#*--}"; fi
  set -- $envs
  label=$1; shift
  env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --no-roofline $extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['value'],1), round(d['ms_per_step'],2), 'host', round(d['host']['graph_launch_ms_per_step'],1), 'wait', round(d['host']['lookahead_wait_ms_per_step'],1))"
done
