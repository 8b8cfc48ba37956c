#!/bin/bash
# One GPU-box session of round 6.  Everything lands in gpurun_out/r06/; the summaries to be judged are copied into profiles/.
#   bash tools/run_r06.sh [probe|newtest|test|benchq|bench|issue|counters|prof|timed|pmc|configs|final] ...
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
SHAPES=("1200 256 0 256 3 1 0" "75 1024 0 1024 3 1 0" "300 512 0 512 3 1 0" "150 1024 512 1024 3 1 0" "1200 256 0 384 1 1 0")
for STAGE in "$@"; do
case $STAGE in
probe)   # LDS-DMA: immediate-offset semantics and issue cost per copy
  timeout 300 tools/bin/glds_probe > $O/glds_probe.txt 2>&1; cat $O/glds_probe.txt ;;
newtest)
  timeout 900 python -m pytest tests/test_gpu_bench_shape.py -m gpu -q -x -k "repeated_decode_with_per_part_ends" > $O/newtest.log 2>&1; tail -5 $O/newtest.log ;;
test)
  ( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $O/gputest.log 2>&1 ) 2> $O/gputest.time; tail -14 $O/gputest.log; cat $O/gputest.time
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log ;;
benchq)   # quick c2 line, no CPU baseline
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined > $O/benchq_c2.json 2> $O/benchq_c2.err; tail -2 $O/benchq_c2.err; cat $O/benchq_c2.json ;;
bench)
  timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -2 $O/bench_c2.err; cat $O/bench_c2.json ;;
leantest)
  timeout 900 python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_parity.py -m gpu -q -x -k "lean_conv_kernel or conv_fast_every_tile_shape" > $O/leantest.log 2>&1; tail -5 $O/leantest.log ;;
leanab)   # the lean kernel against conv_fast_kernel: hot microbenchmark with per-workgroup stamps, then the quick c2 line
  rm -f $O/leanab.txt
  for LEAN in 0 1; do
    for SH in "${SHAPES[@]}" "600 512 0 512 3 1 0" "75 1024 1024 1024 1 1 0" "150 512 0 512 3 1 0"; do
      echo "== lean=$LEAN $SH" >> $O/leanab.txt
      LDC_OPTIONS=conv_lean=$LEAN LDC_B=16 LDC_CONV_STAMPS=1 timeout 120 python tools/conv_one.py $SH 50 >> $O/leanab.txt 2>&1
    done
  done
  grep -v "XCC\|workgroup %" $O/leanab.txt
  for LEAN in 0 1 0 1; do
    LDC_OPTIONS=conv_lean=$LEAN timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --no-roofline > $O/benchq_lean$LEAN.json 2> $O/benchq_lean$LEAN.err
    python -c "import json; d=json.load(open('$O/benchq_lean$LEAN.json')); print('lean=$LEAN', round(d['value'],1), round(d['ms_per_step'],2))"
  done ;;
micro)   # hot microbenchmark of the five shape classes at B = 16 (one batch part), with the per-workgroup stamps
  for SH in "${SHAPES[@]}"; do
    echo "== $SH" >> $O/micro.txt
    LDC_B=16 LDC_CONV_STAMPS=1 timeout 120 python tools/conv_one.py $SH 50 >> $O/micro.txt 2>&1
  done
  cat $O/micro.txt ;;
issue)   # instruction-issue / instruction-cache counters of the pipelined conv kernel on the five top shape classes (B = 16: one batch part)
  rm -rf $O/iss
  for SH in "${SHAPES[@]}"; do
    set -- $SH
    NAME="k$5_c$2+$3-$4_L$1"
    p=0
    for SET in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
               "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
               "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_IFETCH SQ_IFETCH_LEVEL" \
               "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL"; do
      (cd /tmp && LDC_B=16 timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/iss/$NAME/p$p -- python $R/tools/conv_one.py $SH 12 > $O/iss_$NAME.p$p.log 2>&1)
      p=$((p+1))
    done
  done
  python tools/pmc_counters.py $O/iss $O/conv_issue_counters.md $O/conv_counters.json > /dev/null; head -70 $O/conv_issue_counters.md
  cp $O/conv_counters.json profiles/r06_conv_counters.json   # (bench.py attaches mfma_busy_frac from it while the kernel-source hash matches)
  find $O/iss -name "*.csv" -size +5M -delete ;;
prof)
  rm -rf $O/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-pipelined > $O/prof.log 2>&1)
  python tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) > $O/kernel_stats.md
  find $O/prof -name "*.db" -size +30M -delete
  head -30 $O/kernel_stats.md ;;
pmc)   # PMC_TAG=_nosk LDC_OPTIONS=conv_splitk=0 bash tools/run_r06.sh pmc : the same passes under other options, files with the tag
  G=${PMC_TAG:-}
  rm -rf $O/pmc_FETCH$G $O/pmc_WRITE$G
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_FETCH$G -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-pipelined > $O/pmc_FETCH$G.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_WRITE$G -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-pipelined > $O/pmc_WRITE$G.log 2>&1)
  python tools/pmc_traffic.py $O/pmc_FETCH$G $O/pmc_WRITE$G $O/conv_traffic$G.json
  python tools/pmc_classes.py $O/pmc_FETCH$G $O/pmc_WRITE$G $O/conv_pmc_classes$G.md > /dev/null
  find $O/pmc_FETCH$G $O/pmc_WRITE$G -name "*.csv" -size +20M -delete
  cat $O/conv_traffic$G.json ;;
timed)
  timeout 600 python tools/timed_mode_stats.py > $O/timed_mode_kernel_stats.md 2> $O/timed_mode.err; head -12 $O/timed_mode_kernel_stats.md ;;
configs)   # the other BASELINE configs (builder-run lines)
  timeout 300 python bench.py --config c1 --steps 20 --warmup 3 > $O/bench_c1.json 2> $O/bench_c1.err; tail -1 $O/bench_c1.err
  for c in c3 c8 c5 c4; do
    timeout 900 python bench.py --config $c --steps $([ $c = c4 ] && echo 6 || echo 3) --warmup $([ $c = c4 ] && echo 2 || echo 1) > $O/bench_$c.json 2> $O/bench_$c.err; tail -1 $O/bench_$c.err
  done
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
  ;;
final)   # PMC traffic first (bench.py reports it while the kernel-source hash matches), then the c2 line of record
  bash $0 pmc
  cp $O/conv_traffic.json profiles/r06_conv_traffic.json
  bash $0 bench ;;
*) echo "unknown stage $STAGE" ;;
esac
done
