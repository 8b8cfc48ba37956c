"""Summarise rocprofv3 --pmc passes over tools/conv_one.py into a markdown table: mean counter value per launch of the
pipelined conv kernel, per shape.  Usage: python tools/pmc_counters.py DIR OUT.md   (DIR/<shape>/<pass>/..._counter_collection.csv)
Derived rows follow MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES
and SQ_BUSY_CYCLES count cycles; FETCH_SIZE (KB) is doubled on gfx950."""
import csv
import glob
import os
import sys


def collect(shape_dir):
    vals, dur = {}, []
    for f in glob.glob(shape_dir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if not ("conv_fast_kernel" in r["Kernel_Name"] or "conv_lean_kernel" in r["Kernel_Name"]):
                continue
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob(shape_dir + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if ("conv_fast_kernel" in r["Kernel_Name"] or "conv_lean_kernel" in r["Kernel_Name"]):
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    # drop the first (cold) launch of every counter
    out = {k: sum(v[1:]) / max(1, len(v) - 1) if len(v) > 1 else v[0] for k, v in vals.items()}
    if dur:
        d = sorted(dur)
        out["_dur_us_median"] = d[len(d) // 2]
    return out


def main():
    root, out_md = sys.argv[1], sys.argv[2]
    shapes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
    rows = {s: collect(os.path.join(root, s)) for s in shapes}
    names = sorted({k for r in rows.values() for k in r})
    with open(out_md, "w") as f:
        f.write("| counter (mean per launch) | " + " | ".join(shapes) + " |\n|---|" + "---|" * len(shapes) + "\n")
        for n in names:
            f.write("| `%s` | " % n + " | ".join(("%.4g" % rows[s][n]) if n in rows[s] else "-" for s in shapes) + " |\n")
        f.write("\nDerived:\n\n| quantity | " + " | ".join(shapes) + " |\n|---|" + "---|" * len(shapes) + "\n")

        def ratio(a, b, scale=1.0):
            return [("%.3f" % (scale * rows[s][a] / rows[s][b])) if a in rows[s] and b in rows[s] and rows[s][b] else "-" for s in shapes]
        f.write("| MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES) | " + " | ".join(ratio("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", 0.25)) + " |\n")
        f.write("| MFMA busy / (GRBM_GUI_ACTIVE x 1024 SIMDs) | " + " | ".join(ratio("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", 1.0 / 1024)) + " |\n")
        f.write("| parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES | " + " | ".join(ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES")) + " |\n")
        f.write("| issue-stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES | " + " | ".join(ratio("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES")) + " |\n")
        f.write("| issuing = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES | " + " | ".join(ratio("SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES")) + " |\n")
        f.write("| LDS issue stall = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES | " + " | ".join(ratio("SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES")) + " |\n")
        f.write("| LDS bank conflicts / LDS active | " + " | ".join(ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")) + " |\n")
        f.write("| L2 hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) | " + " | ".join(
            ("%.3f" % (rows[s]["TCC_HIT_sum"] / (rows[s]["TCC_HIT_sum"] + rows[s]["TCC_MISS_sum"]))) if "TCC_HIT_sum" in rows[s] and "TCC_MISS_sum" in rows[s] else "-" for s in shapes) + " |\n")
        f.write("| VALU / SALU / LDS instructions per MFMA | " + " | ".join(
            ("%.1f / %.1f / %.1f" % (rows[s].get("SQ_INSTS_VALU", 0) / rows[s]["SQ_INSTS_MFMA"], rows[s].get("SQ_INSTS_SALU", 0) / rows[s]["SQ_INSTS_MFMA"],
                                       rows[s].get("SQ_INSTS_LDS", 0) / rows[s]["SQ_INSTS_MFMA"])) if rows[s].get("SQ_INSTS_MFMA") else "-" for s in shapes) + " |\n")
    print(open(out_md).read())


if __name__ == "__main__":
    main()
