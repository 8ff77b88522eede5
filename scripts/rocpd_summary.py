#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs into a small text file for profiles/.

    python scripts/rocpd_summary.py --stats gpurun_out/prof_stats --pmc gpurun_out/prof_fetch gpurun_out/prof_write \
        --cmd "<the profiled command>" -o profiles/r01_xxx.txt
"""
import argparse
import glob
import sqlite3


def db_of(d):
    f = sorted(glob.glob(d + "/**/*.db", recursive=True))
    assert f, f"no rocpd .db under {d}"
    return sqlite3.connect(f[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--cmd", default="")
    ap.add_argument("-o", "--out", required=True)
    a = ap.parse_args()
    lines = [f"# command: {a.cmd}", "# tool: rocprofv3 --kernel-trace --stats (kernel table) / rocprofv3 --pmc <C> --kernel-trace (counters, one pass each)", ""]
    if a.stats:
        db = db_of(a.stats)
        lines.append("## kernel stats (top_kernels): name | calls | total_us | avg_us | pct")
        for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            lines.append(f"{name} | {calls} | {total:.3f} | {avg:.3f} | {pct:.2f}")
        lines.append("")
        lines.append("## per-kernel resources: name | grid | workgroup | vgpr | accum_vgpr | sgpr | lds | scratch")
        for r in db.execute("select distinct name,grid_x,workgroup_x,vgpr_count,accum_vgpr_count,sgpr_count,lds_size,scratch_size from kernels"):
            lines.append(" | ".join(str(x) for x in r))
        lines.append("")
    for d in a.pmc:
        db = db_of(d)
        lines.append(f"## counters from {d}: kernel | counter | avg value per dispatch | dispatches")
        for k, c, v, n in db.execute("select kernel_name,counter_name,avg(value),count(*) from counters_collection group by kernel_name,counter_name"):
            lines.append(f"{k} | {c} | {v:.3f} | {n}")
        lines.append("")
    open(a.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
