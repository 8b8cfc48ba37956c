"""Summarise a rocprofv3 --kernel-trace CSV of bench.py (graph-replayed, two-stream decode) into per-queue busy time,
overlap factor and gaps for ONE steady-state decode (the span between the last two outnorm_apply kernels).

usage: python tools/timeline_summary.py <dir with *_kernel_trace.csv> <out.md>
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def short(n):
    n = n.split("(")[0]
    for p in ("void ", "ldc::", "(anonymous namespace)::"):
        n = n.replace(p, "")
    return n.split("<")[0][-48:]


def main():
    d, out = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no kernel trace csv under " + d
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if "outnorm_apply" in r[3]]
    assert len(ends) >= 2, "need two decodes in the trace"
    seg = rows[ends[-2] + 1: ends[-1] + 1]
    t0, t1 = seg[0][0], seg[-1][1]
    span = t1 - t0
    byq = defaultdict(list)
    byk = defaultdict(lambda: [0, 0])
    for s, e, q, n in seg:
        byq[q].append((s, e))
        k = short(n)
        byk[k][0] += e - s
        byk[k][1] += 1
    all_busy = union([(s, e) for s, e, _, _ in seg])
    sum_busy = sum(e - s for s, e, _, _ in seg)
    lines = ["# Timeline of one steady-state decode (graph-replayed, timed mode)", "",
             f"source: rocprofv3 --kernel-trace of `bench.py` (no --stats serialisation); segment = kernels between the last two "
             f"`outnorm_apply` dispatches: {len(seg)} dispatches", "",
             f"* span (first kernel start -> last kernel end): **{span / 1e6:.2f} ms**",
             f"* time with at least one kernel running: {all_busy / 1e6:.2f} ms ({100.0 * all_busy / span:.1f} % of the span); idle: {(span - all_busy) / 1e6:.2f} ms",
             f"* sum of kernel durations: {sum_busy / 1e6:.2f} ms -> overlap factor (sum / union) = **{sum_busy / max(1, all_busy):.2f}**", "",
             "| queue | dispatches | busy ms (union) | sum of durations ms | share of span | mean gap between consecutive kernels us |", "|---|---|---|---|---|---|"]
    for q, iv in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        iv.sort()
        gaps = [max(0, iv[i + 1][0] - iv[i][1]) for i in range(len(iv) - 1)]
        u = union(iv)
        lines.append(f"| {q} | {len(iv)} | {u / 1e6:.2f} | {sum(e - s for s, e in iv) / 1e6:.2f} | {100.0 * u / span:.1f} % | "
                     f"{(sum(gaps) / max(1, len(gaps))) / 1e3:.2f} |")
    crit = max(union(iv) for iv in byq.values())
    lines += ["", f"critical queue busy time {crit / 1e6:.2f} ms <= span {span / 1e6:.2f} ms (the span is what bench.py's ms_per_step measures, "
              f"plus host launch latency of the first graph)", "",
              "| kernel | launches | total ms | avg us |", "|---|---|---|---|"]
    for k, (t, n) in sorted(byk.items(), key=lambda kv: -kv[1][0])[:24]:
        lines.append(f"| {k} | {n} | {t / 1e6:.2f} | {t / n / 1e3:.2f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:16]))


if __name__ == "__main__":
    main()
