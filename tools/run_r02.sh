#!/bin/bash
# One GPU-box session of round 2.  Everything lands in gpurun_out/; the summaries to be judged are copied into profiles/.
#   bash tools/run_r02.sh [test|bench|prof|pmc|shapes|all]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
STAGE=${1:-all}
cd $R
if [[ $STAGE == all || $STAGE == test ]]; then
  timeout 2400 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; tail -15 $O/gputest.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
fi
if [[ $STAGE == all || $STAGE == bench ]]; then
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cat $O/bench.json
fi
if [[ $STAGE == all || $STAGE == prof ]]; then
  rm -rf $O/prof_r02
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_r02 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-pipelined > $O/prof_r02.log 2>&1)
  python tools/prof_summary.py $(find $O/prof_r02 -name "*.db" | head -1) > $O/prof_r02_summary.md
  find $O/prof_r02 -name "*.db" -size +30M -delete
  head -40 $O/prof_r02_summary.md
fi
if [[ $STAGE == all || $STAGE == pmc ]]; then
  rm -rf $O/pmc2_FETCH $O/pmc2_WRITE
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc2_FETCH -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-pipelined > $O/pmc2_FETCH.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc2_WRITE -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-pipelined > $O/pmc2_WRITE.log 2>&1)
  python tools/pmc_traffic.py $O/pmc2_FETCH $O/pmc2_WRITE $O/conv_traffic_r02.json
  python tools/pmc_classes.py $O/pmc2_FETCH $O/pmc2_WRITE $O/conv_pmc_classes_r02.md > /dev/null
  find $O/pmc2_FETCH $O/pmc2_WRITE -name "*.csv" -size +20M -delete
  cat $O/conv_traffic_r02.json
fi
if [[ $STAGE == all || $STAGE == shapes ]]; then
  LDC_PROFILE_DUMP=/tmp/d.txt timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined > $O/bench_shapes.json 2> $O/bench_shapes.err
  python tools/prof_shapes.py /tmp/d.txt > $O/shapes_r02.txt 2>&1; head -60 $O/shapes_r02.txt
fi
