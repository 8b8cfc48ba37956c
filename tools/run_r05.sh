#!/bin/bash
# One GPU-box session of round 5.  Everything lands in gpurun_out/r05/; the summaries to be judged are copied into profiles/.
#   bash tools/run_r05.sh [probe|counters|test|benchq|bench|prof|timed|pmc|configs|c1prof|final] ...
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
for STAGE in "$@"; do
case $STAGE in
probe)   # go / no-go numbers for the XCD-team persistent kernel
  timeout 300 tools/bin/xcd_team_probe 768 200 > $O/xcd_team_probe.txt 2>&1; cat $O/xcd_team_probe.txt ;;
counters)   # SQ / TCC counters of the pipelined conv kernel on the five top shape classes (B = 16: one batch part)
  (cd /tmp && rocprofv3 -L > $O/counters_list.txt 2>&1)
  rm -rf $O/ctr
  i=0
  for SH in "1200 256 0 256 3 1 0" "75 1024 0 1024 3 1 0" "300 512 0 512 3 1 0" "150 1024 512 1024 3 1 0" "1200 256 0 384 1 1 0"; do
    set -- $SH
    NAME="k$5_c$2+$3-$4_L$1"
    p=0
    for SET in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
               "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
               "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
      (cd /tmp && LDC_B=16 timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $O/ctr/$NAME/p$p -- python $R/tools/conv_one.py $SH 12 > $O/ctr_$NAME.p$p.log 2>&1)
      p=$((p+1))
    done
  done
  python tools/pmc_counters.py $O/ctr $O/conv_counters.md > /dev/null; head -60 $O/conv_counters.md
  find $O/ctr -name "*.csv" -size +5M -delete ;;
test)
  ( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $O/gputest.log 2>&1 ) 2> $O/gputest.time; tail -14 $O/gputest.log; cat $O/gputest.time
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log ;;
benchq)   # quick c2 line, no CPU baseline
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined > $O/benchq_c2.json 2> $O/benchq_c2.err; tail -2 $O/benchq_c2.err; cat $O/benchq_c2.json ;;
bench)
  timeout 900 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; tail -2 $O/bench_c2.err; cat $O/bench_c2.json ;;
prof)
  rm -rf $O/prof
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-pipelined > $O/prof.log 2>&1)
  python tools/prof_summary.py $(find $O/prof -name "*.db" | head -1) > $O/kernel_stats.md
  find $O/prof -name "*.db" -size +30M -delete
  head -30 $O/kernel_stats.md ;;
pmc)
  rm -rf $O/pmc_FETCH $O/pmc_WRITE
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_FETCH -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-pipelined > $O/pmc_FETCH.log 2>&1)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_WRITE -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-pipelined > $O/pmc_WRITE.log 2>&1)
  python tools/pmc_traffic.py $O/pmc_FETCH $O/pmc_WRITE $O/conv_traffic.json
  python tools/pmc_classes.py $O/pmc_FETCH $O/pmc_WRITE $O/conv_pmc_classes.md > /dev/null
  find $O/pmc_FETCH $O/pmc_WRITE -name "*.csv" -size +20M -delete
  cat $O/conv_traffic.json ;;
timed)
  timeout 600 python tools/timed_mode_stats.py > $O/timed_mode_kernel_stats.md 2> $O/timed_mode.err; head -12 $O/timed_mode_kernel_stats.md ;;
configs)   # the other BASELINE configs (builder-run lines)
  timeout 300 python bench.py --config c1 --steps 20 --warmup 3 > $O/bench_c1.json 2> $O/bench_c1.err; tail -1 $O/bench_c1.err
  for c in c3 c8 c5 c4; do
    timeout 900 python bench.py --config $c --steps 3 --warmup 1 > $O/bench_$c.json 2> $O/bench_$c.err; tail -1 $O/bench_$c.err
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
c1prof)   # kernel trace of the configs[0] round trip + dispatch listings of the codec ends
  rm -rf $O/prof_c1 $O/ends16 $O/ends_c1
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c1 -- python $R/bench.py --config c1 --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_c1.log 2>&1)
  python tools/prof_summary.py $(find $O/prof_c1 -name "*.db" | head -1) > $O/c1_kernel_stats.md
  find $O/prof_c1 -name "*.db" -size +30M -delete
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/ends16 -- python $R/tools/codec_ends.py 16 > $O/ends16.log 2>&1)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/ends_c1 -- python $R/tools/codec_ends.py 1 c1 > $O/ends_c1.log 2>&1)
  python tools/codec_ends.py --summarise $O/ends16 > $O/codec_ends_b16.md
  python tools/codec_ends.py --summarise $O/ends_c1 > $O/codec_ends_c1.md
  head -12 $O/c1_kernel_stats.md; tail -1 $O/codec_ends_b16.md; tail -1 $O/codec_ends_c1.md ;;
final)   # PMC traffic first (bench.py reports it while the kernel-source hash matches), then the c2 line of record
  bash $0 pmc
  cp $O/conv_traffic.json profiles/r05_conv_traffic.json
  bash $0 bench ;;
*) echo "unknown stage $STAGE" ;;
esac
done
