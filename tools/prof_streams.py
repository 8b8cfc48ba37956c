"""Per-stream totals of a rocprofv3 rocpd database: how much kernel time each HIP stream carried, and the top kernels of the busiest one.
Usage: python tools/prof_streams.py DB [N_TOP]   (round 6: what is left on the optimisation step's critical chain)"""
import sqlite3
import sys
from prof_summary import demangle, short


def main(path, ntop=25):
    con = sqlite3.connect(path)
    rows = con.execute("select S.kernel_name, K.stream_id, K.end - K.start, K.start, K.end from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S "
                       "on S.id = K.kernel_id and S.guid = K.guid").fetchall()
    uniq = sorted({r[0] for r in rows})
    dem = demangle(uniq) or uniq
    table = dict(zip(uniq, dem))
    per = {}
    for name, sid, dur, st, en in rows:
        e = per.setdefault(sid, {"n": 0, "t": 0, "k": {}, "first": st, "last": en})
        e["n"] += 1; e["t"] += dur; e["first"] = min(e["first"], st); e["last"] = max(e["last"], en)
        k = e["k"].setdefault(short(table[name]), [0, 0]); k[0] += 1; k[1] += dur
    print("| stream | dispatches | kernel ms | span ms |\n|---|---|---|---|")
    for sid, e in sorted(per.items(), key=lambda kv: -kv[1]["t"]):
        print(f"| {sid} | {e['n']} | {e['t'] / 1e6:.2f} | {(e['last'] - e['first']) / 1e6:.2f} |")
    for sid, e in sorted(per.items(), key=lambda kv: -kv[1]["t"])[:2]:
        print(f"\nstream {sid}: top kernels\n\n| kernel | calls | total ms | avg us |\n|---|---|---|---|")
        for k, (n, t) in sorted(e["k"].items(), key=lambda kv: -kv[1][1])[:ntop]:
            print(f"| `{k}` | {n} | {t / 1e6:.2f} | {t / n / 1e3:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
