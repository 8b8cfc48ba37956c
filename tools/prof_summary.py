"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats`) as a
per-kernel table: calls, total / average / min / max duration.  Usage: python tools/prof_summary.py DB [> profiles/x.md]"""
import re
import shutil
import sqlite3
import subprocess
import sys


def demangle(mangled):
    """rocprofv3's display names give up on __bf16 (`DF16b`): half of the conv kernels come out mangled, the other half with their template
    arguments out of step.  binutils' c++filt does not know DF16b either, but it knows `Dh` (half): substitute, demangle, rename."""
    names = [re.sub(r"\.kd$", "", m).replace("DF16b", "Dh") for m in mangled]
    exe = shutil.which("c++filt")
    if not exe:
        return None
    out = subprocess.run([exe], input="\n".join(names) + "\n", capture_output=True, text=True).stdout.splitlines()
    if len(out) != len(names):
        return None
    return [o.replace("half", "bf16") for o in out]


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("ldc::", "")
    name = re.sub(r"\(.*\)$", "", name)
    # template arguments in full (two conv_fast_kernel rows used to differ only past the 90th character: VERDICT r4); the spelled-out
    # names of the non-type parameters are dropped to keep the rows readable
    name = re.sub(r"\b(?:bool|int|unsigned) _?[A-Za-z]\w*, ?", "", name)
    return name[:200]


def main(path):
    con = sqlite3.connect(path)
    try:
        rows = con.execute("select S.kernel_name, K.grid_size_x*1.0/K.workgroup_size_x, K.end - K.start, K.group_segment_size, S.arch_vgpr_count, "
                           "S.accum_vgpr_count, S.sgpr_count, K.private_segment_size from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S "
                           "on S.id = K.kernel_id and S.guid = K.guid").fetchall()
        uniq = sorted({r[0] for r in rows})
        dem = demangle(uniq)
        if dem is None:
            raise RuntimeError("no c++filt")
        table = dict(zip(uniq, dem))
        rows = [(table[r[0]],) + tuple(r[1:]) for r in rows]
    except Exception:
        rows = con.execute("select name, grid_x*1.0/workgroup_x, duration, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size from kernels").fetchall()
    agg = {}
    for name, blocks, dur, lds, v, a, s, scr in rows:
        k = short(name)
        e = agg.setdefault(k, [0, 0, 1 << 62, 0, lds, v, a, s, scr])
        e[0] += 1; e[1] += dur; e[2] = min(e[2], dur); e[3] = max(e[3], dur)
    total = sum(e[1] for e in agg.values())
    print(f"| kernel | calls | total ms | % | avg us | min us | max us | VGPR | AGPR | SGPR | scratch |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for k, e in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {e[0]} | {e[1] / 1e6:.2f} | {100 * e[1] / total:.1f} | {e[1] / e[0] / 1e3:.1f} | {e[2] / 1e3:.1f} | {e[3] / 1e3:.1f} | {e[5]} | {e[6]} | {e[7]} | {e[8]} |")
    print(f"\ntotal kernel time: {total / 1e6:.2f} ms over {sum(e[0] for e in agg.values())} dispatches")


if __name__ == "__main__":
    main(sys.argv[1])
