import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["ms_per_step"])
for k in r["roofline"]["other_kernels"]: print(k["kernel"], round(k["avg_launch_us"],2), k["launches"])
