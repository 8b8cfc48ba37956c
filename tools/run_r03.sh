#!/bin/bash
# One GPU-box session of round 3.  Everything lands in gpurun_out/r03/; the summaries to be judged are copied into profiles/.
#   bash tools/run_r03.sh [test|bench|prof|pmc|shapes|train|all]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r03; mkdir -p $O
STAGE=${1:-all}
cd $R
if [[ $STAGE == all || $STAGE == test ]]; then
  ( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 > $O/gputest.log 2>&1 ) 2> $O/gputest.time; tail -14 $O/gputest.log; cat $O/gputest.time
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
fi
if [[ $STAGE == all || $STAGE == bench ]]; then
  timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -2 $O/bench_c2.err
  for c in c3 c8 c5 c4; do
    timeout 900 python bench.py --config $c --steps 3 --warmup 1 > $O/bench_$c.json 2> $O/bench_$c.err; tail -1 $O/bench_$c.err
  done
  LDC_TRAIN_FP32_MFMA=1 timeout 600 python bench.py --config c4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c4_fp32_mfma.json 2> /dev/null
  LDC_FP8_ACT=0 timeout 600 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined > $O/bench_c5_weights_only.json 2> /dev/null
  timeout 600 python bench.py --dtype fp8 --steps 4 --warmup 1 --no-cpu-baseline --no-pipelined > $O/bench_c2_fp8.json 2> /dev/null
  python - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms; roofline", round(d.get("roofline", {}).get("frac", 0), 4), "host", d.get("host"))
    except Exception as e:
        print(f, "unreadable", e)
PY
fi
if [[ $STAGE == all || $STAGE == prof ]]; then
  rm -rf $O/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-pipelined > $O/prof.log 2>&1)
  python tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) > $O/kernel_stats.md
  find $O/prof -name "*.db" -size +30M -delete
  head -30 $O/kernel_stats.md
fi
if [[ $STAGE == all || $STAGE == train ]]; then
  rm -rf $O/prof_train
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_train -- python $R/tools/train_step_time.py 32 > $O/prof_train.log 2>&1)
  python tools/prof_summary.py $(find $O/prof_train -name "*.db" | head -1) > $O/train_kernel_stats.md
  find $O/prof_train -name "*.db" -size +30M -delete
  head -24 $O/train_kernel_stats.md
fi
if [[ $STAGE == all || $STAGE == pmc ]]; then
  rm -rf $O/pmc_FETCH $O/pmc_WRITE $O/cal_FETCH $O/cal_WRITE
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_FETCH -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-pipelined > $O/pmc_FETCH.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_WRITE -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-pipelined > $O/pmc_WRITE.log 2>&1)
  python tools/pmc_traffic.py $O/pmc_FETCH $O/pmc_WRITE $O/conv_traffic.json
  python tools/pmc_classes.py $O/pmc_FETCH $O/pmc_WRITE $O/conv_pmc_classes.md > /dev/null
  # calibration of the counters in THIS kernel's access pattern: a layer with ONE N tile (every input byte is fetched once:
  # 32 x 1200 rows x 1024 channels = 78.6 MB of 64-byte row pieces, larger than the L2s) and known output bytes (4.9 MB)
  (cd /tmp && LDC_B=32 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cal_FETCH -- python $R/tools/conv_one.py 1200 1024 0 64 1 1 0 6 > $O/cal_FETCH.log 2>&1)
  (cd /tmp && LDC_B=32 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cal_WRITE -- python $R/tools/conv_one.py 1200 1024 0 64 1 1 0 6 > $O/cal_WRITE.log 2>&1)
  python tools/pmc_traffic.py $O/cal_FETCH $O/cal_WRITE $O/cal_traffic.json
  find $O/pmc_FETCH $O/pmc_WRITE $O/cal_FETCH $O/cal_WRITE -name "*.csv" -size +20M -delete
  cat $O/conv_traffic.json $O/cal_traffic.json
fi
if [[ $STAGE == all || $STAGE == shapes ]]; then
  LDC_PROFILE_DUMP=/tmp/d.txt timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined > $O/bench_shapes.json 2> $O/bench_shapes.err
  python tools/prof_shapes.py /tmp/d.txt > $O/layer_shapes.txt 2>&1; head -30 $O/layer_shapes.txt
fi
