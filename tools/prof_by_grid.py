"""Per (kernel, grid) durations from a rocprofv3 rocpd database: python tools/prof_by_grid.py DB [name-substring]"""
import re
import sqlite3
import sys


def main(path, sub=""):
    con = sqlite3.connect(path)
    rows = con.execute("select name, grid_x, grid_y, grid_z, workgroup_x, duration from kernels order by start").fetchall()
    agg, order = {}, []
    for name, gx, gy, gz, wx, dur in rows:
        if sub not in name:
            continue
        n = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", name)).replace("ldc::", "")
        k = (n[:60], gx // max(wx, 1), gy, gz)
        if k not in agg:
            agg[k] = [0, 0, 1 << 62]
            order.append(k)
        e = agg[k]
        e[0] += 1; e[1] += dur; e[2] = min(e[2], dur)
    for k in order:
        e = agg[k]
        print(f"{k[0]:60s} grid {k[1]:5d} x {k[2]:3d} x {k[3]:3d}  n {e[0]:3d}  avg {e[1] / e[0] / 1e3:8.1f} us  min {e[2] / 1e3:8.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
