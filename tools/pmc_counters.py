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
    if len(sys.argv) > 3:   # machine-readable summary for bench.py (roofline.mfma_busy_frac), stamped with the hash of the kernel sources
        import json
        from pmc_traffic import csrc_hash
        rec = {"kernel": "conv_lean_kernel / conv_fast_kernel", "csrc_sha256_16": csrc_hash(),
               "how": "rocprofv3 --pmc passes over tools/conv_one.py (LDC_B=16), tools/run_r06.sh issue; launch fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel-trace duration x 2.4 GHz)",
               "shapes": {}}
        tot_busy = tot_cap = 0.0
        for sname in shapes:
            r = rows[sname]
            if "SQ_VALU_MFMA_BUSY_CYCLES" not in r or "_dur_us_median" not in r:
                continue
            cap = 1024.0 * r["_dur_us_median"] * 2400.0
            e = {"mfma_busy_cycles": r["SQ_VALU_MFMA_BUSY_CYCLES"], "dur_us": r["_dur_us_median"], "mfma_busy_frac_of_launch": r["SQ_VALU_MFMA_BUSY_CYCLES"] / cap}
            if r.get("SQ_BUSY_CU_CYCLES"):
                e["mfma_busy_frac_of_busy_cu"] = 0.25 * r["SQ_VALU_MFMA_BUSY_CYCLES"] / r["SQ_BUSY_CU_CYCLES"]
            if r.get("SQ_INSTS_MFMA"):
                # (VERDICT r5's formula: VALU + SALU + LDS instructions per MFMA; SQ_INSTS_VALU counts the MFMAs themselves too)
                e["non_mfma_per_mfma"] = (r.get("SQ_INSTS_VALU", 0) + r.get("SQ_INSTS_SALU", 0) + r.get("SQ_INSTS_LDS", 0)) / r["SQ_INSTS_MFMA"]
            if r.get("SQC_ICACHE_REQ"):
                e["icache_hit"] = r.get("SQC_ICACHE_HITS", 0) / r["SQC_ICACHE_REQ"]
            rec["shapes"][sname] = e
            tot_busy += r["SQ_VALU_MFMA_BUSY_CYCLES"]
            tot_cap += cap
        rec["mfma_busy_frac"] = tot_busy / tot_cap if tot_cap else 0.0
        json.dump(rec, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
