#!/bin/bash
# One GPU-box session: record drift, run the GPU suite, bench, timed-mode timeline.  Everything lands in gpurun_out/.
export TMPDIR=/tmp
mkdir -p gpurun_out
STAGE=${1:-all}
if [[ $STAGE == all || $STAGE == drift ]]; then
  rm -f gpurun_out/drift.json
  LDC_RECORD_DRIFT=$PWD/gpurun_out/drift.json timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/drift_run.log 2>&1
  tail -5 gpurun_out/drift_run.log
  cat gpurun_out/drift.json
fi
if [[ $STAGE == all || $STAGE == test ]]; then
  timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gputest.log 2>&1; tail -15 gpurun_out/gputest.log
fi
if [[ $STAGE == all || $STAGE == bench ]]; then
  timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
fi
if [[ $STAGE == all || $STAGE == timeline ]]; then
  rm -rf /tmp/tl; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $OLDPWD/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/tl.json 2> /tmp/tl.err); tail -2 /tmp/tl.err; cat /tmp/tl.json
  python tools/timeline_summary.py /tmp/tl gpurun_out/timeline.md
fi
