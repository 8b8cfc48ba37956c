"""Per-layer-class HBM traffic of the conv launches from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of
`bench.py --steps 1 --warmup 0`): dispatches are grouped by (kernel template, grid size), which separates the UNet's layer
classes (taps / tile shape / grid).  gfx950: FETCH_SIZE of wide coalesced reads counts half the bytes -> doubled; both in KB.
usage: python tools/pmc_classes.py FETCH_DIR WRITE_DIR OUT.md"""
import collections
import csv
import glob
import re
import sys


def load(d, name):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name or not ("conv_fast_kernel" in r["Kernel_Name"] or "conv_lean_kernel" in r["Kernel_Name"]):
                continue
            k = r["Kernel_Name"]
            m = re.search(r"conv_(?:fast|lean)_kernel<(.*)>", k)
            args = [a.strip() for a in m.group(1).split(",")] if m else []
            ns = "f32" if "fast_f32" in k else ("bf16w8" if "w8" in k else "bf16")
            nums = [a for a in args if re.fullmatch(r"\d+", a)]
            # trailing template integers: ..., TG, KC, S, NA ; tile = 32*WM*TM x 32*WN*TN when all eight are printed
            key = (ns, ",".join(nums[-4:]), int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
            out[key].append(float(r["Counter_Value"]))
    return out


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for k in sorted(set(fe) | set(wr)):
    f = fe.get(k, [0.0]); w = wr.get(k, [0.0])
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    rows.append((len(f), k, 2.0 * fm * 1024, wm * 1024))
tot = sum(n * (a + b) for n, k, a, b in rows)
with open(sys.argv[3], "w") as o:
    o.write("# HBM traffic per conv launch by layer class (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over\n"
            "# `bench.py --steps 1 --warmup 0`; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md)\n\n"
            "class = (dtype, template tail `TG,KC,S,NA` = taps per unit, chunks per unit, ring stages, window sweeps; workgroups)\n\n"
            "| launches | dtype | TG,KC,S,NA | workgroups | fetch MB | write MB | share of conv HBM bytes |\n|---|---|---|---|---|---|---|\n")
    for n, k, a, b in sorted(rows, key=lambda r: -r[0] * (r[2] + r[3])):
        o.write(f"| {n} | {k[0]} | {k[1]} | {k[2]} | {a / 1e6:.2f} | {b / 1e6:.2f} | {100 * n * (a + b) / tot:.1f} % |\n")
    o.write(f"\ntotal {tot / 1e9:.2f} GB over {sum(r[0] for r in rows)} conv launches = {tot / sum(r[0] for r in rows) / 1e6:.2f} MB per launch\n")
print(open(sys.argv[3]).read()[:3000])
