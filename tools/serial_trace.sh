cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
rm -rf $O/prof_serial
LDC_SERIAL=1 timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_serial -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $O/prof_serial.log 2>&1
python - <<'PY'
import sqlite3,glob,os
db=glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/prof_serial/**/*.db',recursive=True)[0]
con=sqlite3.connect(db)
rows=con.execute("select name, grid_x, grid_y, workgroup_x, count(*), avg(duration), min(duration) from kernels group by name, grid_x, grid_y order by sum(duration) desc limit 45").fetchall()
import re
for r in rows:
    nm=re.sub(r'\(.*','',r[0]).replace('ldc::','')[:70]
    print(f"{nm:70s} grid {r[1]//r[3]:5d}x{r[2]:3d} n {r[4]:5d} avg {r[5]/1e3:7.1f} min {r[6]/1e3:7.1f}")
PY
find $O/prof_serial -name "*.db" -size +30M -delete
