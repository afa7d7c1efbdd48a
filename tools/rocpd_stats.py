#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output): name, grid, calls, total, average, min, max.

Usage: python tools/rocpd_stats.py gpurun_out/prof/x_results.db [> profiles/rNN_rocprof_kernel_stats.csv]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*\)$", "", name)
    name = name.replace("void ", "")
    return name[:110]


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, grid_x, grid_y, grid_z, workgroup_x, duration from kernels").fetchall()
    agg = {}
    for name, gx, gy, gz, wx, d in rows:
        k = (short(name), f"{gx // max(wx, 1)}x{gy}x{gz}")
        a = agg.setdefault(k, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print('"Name","Grid(workgroups)","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for (name, grid), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'"{name}","{grid}",{a[0]},{a[1]},{a[1] / a[0]:.0f},{100.0 * a[1] / total:.2f},{a[2]},{a[3]}')


if __name__ == "__main__":
    main(sys.argv[1])
