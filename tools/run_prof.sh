cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 600 python bench.py > $O/bench_r1f.log 2> $O/bench_r1f.err
rm -rf $O/prof_r1f $O/pmcF_FETCH $O/pmcF_WRITE
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_r1f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/prof_r1f.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcF_FETCH -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $O/pmcF_FETCH.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcF_WRITE -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $O/pmcF_WRITE.log 2>&1
python tools/pmc_traffic.py $O/pmcF_FETCH $O/pmcF_WRITE $O/conv_traffic_f.json
python tools/prof_summary.py $(find $O/prof_r1f -name "*.db" | head -1) > $O/prof_r1f_summary.md
# keep only small files
find $O/pmcF_FETCH $O/pmcF_WRITE -name "*.csv" -size +20M -delete
find $O/prof_r1f -name "*.db" -size +30M -delete
tail -1 $O/bench_r1f.log
