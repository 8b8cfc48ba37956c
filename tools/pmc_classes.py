"""Per-layer-class HBM traffic of the conv launches from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of
`bench.py --steps 1 --warmup 0`): dispatches are grouped by (kernel template, grid size), which separates the UNet's layer
classes (taps / tile shape / grid).  gfx950: FETCH_SIZE of wide coalesced reads counts half the bytes -> doubled; both in KB.
usage: python tools/pmc_classes.py FETCH_DIR WRITE_DIR OUT.md"""
import collections
import csv
import glob
import re
import sys


KIND = {"0": "k3", "1": "k1", "2": "k1+LN"}
EPI = {"0": "plain", "1": "+res", "2": "GN", "3": "GN+res", "4": "qkv+ctx"}


def label(k):
    """conv_lean_kernel<T, TM, KIND, RF, EPI, SK> / conv_fast_kernel<T, WM, WN, TM, TN, TG, KC, S, NA, RF> as a short class label.  rocprofv3's
    demangler gives up on __bf16 (DF16b): such names arrive mangled, or with their first arguments out of step -- parsed from either form."""
    m = re.search(r"conv_lean_kernelI(?:DF16b|f)Li(\d)ELi(\d)ELb(\d)ELi(\d)ELb(\d)E", k)
    if m:
        tm, kind, rf, epi, sk = m.groups()
        return f"lean {64 * int(tm)}x64 {KIND[kind]}{'+fold' if rf == '1' else ''} {EPI[epi]}{' splitK' if sk == '1' else ''}"
    m = re.search(r"conv_lean_kernel<(.*)>", k)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        if len(a) == 6 and a[0] in ("float", "__bf16", "bf16"):      # a clean demangling
            tm, kind, rf, epi, sk = a[1], a[2], "1" if a[3] == "true" else "0", a[4], "1" if a[5] == "true" else "0"
            return f"lean {64 * int(tm)}x64 {KIND[kind]}{'+fold' if rf == '1' else ''} {EPI[epi]}{' splitK' if sk == '1' else ''}"
        # `<int, E, KIND, RF, EPI, SK>`: DF16b Li1E read as two arguments (seen for TM = 1 only; TM = 2 names stay mangled)
        kind, rf, epi, sk = a[-4], "1" if a[-3] == "true" else "0", a[-2], "1" if a[-1] == "true" else "0"
        if kind == "E":      # (`<int, EL, int, E, ...>`: Li1ELi1E, both eaten)
            kind = "1"
        return f"lean 64x64 {KIND.get(kind, kind)}{'+fold' if rf == '1' else ''} {EPI.get(epi, epi)}{' splitK' if sk == '1' else ''}"
    m = re.search(r"conv_fast_kernelI(?:DF16b|f|N\w+E)Li(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)E", k)
    if m:
        wm, wn, tm, tn, tg, kc, st, na, rf = m.groups()
        return f"fast {32 * int(wm) * int(tm)}x{32 * int(wn) * int(tn)} taps/unit {tg} chunks {kc} sweeps {na}{' +fold' if rf == '1' else ''}"
    m = re.search(r"conv_fast_kernel<(.*)>", k)
    nums = [x.strip() for x in m.group(1).split(",") if re.fullmatch(r"\s*\d+\s*", x)] if m else []
    return "fast taps/unit,chunks,stages,sweeps " + ",".join(nums[-4:])


def load(d, name):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name or not ("conv_fast_kernel" in r["Kernel_Name"] or "conv_lean_kernel" in r["Kernel_Name"]):
                continue
            k = r["Kernel_Name"]
            ns = "f32" if "fast_f32" in k else ("bf16w8" if "w8" in k else ("fp8" if "fast_fp8" in k else "bf16"))
            key = (ns, label(k), int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
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
            "class = (dtype, kernel and its template arguments -- tile, unit kind, folded second conv, epilogue --, workgroups)\n\n"
            "| launches | dtype | kernel | workgroups | fetch MB | write MB | share of conv HBM bytes |\n|---|---|---|---|---|---|---|\n")
    for n, k, a, b in sorted(rows, key=lambda r: -r[0] * (r[2] + r[3])):
        o.write(f"| {n} | {k[0]} | {k[1]} | {k[2]} | {a / 1e6:.2f} | {b / 1e6:.2f} | {100 * n * (a + b) / tot:.1f} % |\n")
    o.write(f"\ntotal {tot / 1e9:.2f} GB over {sum(r[0] for r in rows)} conv launches = {tot / sum(r[0] for r in rows) / 1e6:.2f} MB per launch\n")
print(open(sys.argv[3]).read()[:3000])
