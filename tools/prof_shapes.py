"""Aggregate an LDC_PROFILE_DUMP file (class flops bytes us per launch) by distinct (class, flops, bytes) shape.
Usage: LDC_PROFILE_DUMP=/tmp/d.txt python bench.py --no-cpu-baseline; python tools/prof_shapes.py /tmp/d.txt"""
import collections
import sys
names = ("other", "conv", "gn_apply", "layernorm", "linattn", "attn_full", "elementwise")
agg = collections.OrderedDict()
for line in open(sys.argv[1]):
    k, fl, by, us, info = line.split()
    e = agg.setdefault((int(k), float(fl), float(by), info), [0, 0.0, 1e9])
    e[0] += 1; e[1] += float(us); e[2] = min(e[2], float(us))
tot = sum(e[1] for e in agg.values())
print(f"{'class':12s} {'GFLOP':>8s} {'MB':>8s} {'n':>6s} {'avg us':>8s} {'min us':>8s} {'TFLOP/s':>8s} {'GB/s':>8s} {'% time':>7s}")
for (k, fl, by, info), e in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    avg = e[1] / e[0]
    print(f"{names[k]:12s} {fl / 1e9:8.3f} {by / 1e6:8.2f} {e[0]:6d} {avg:8.1f} {e[2]:8.1f} {fl / avg / 1e6:8.1f} {by / avg / 1e3:8.1f} {100 * e[1] / tot:7.2f}  {info}")
