#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2) rocpd SQLite database into per-kernel statistics (the --stats view).

usage: summarize_rocpd.py trace_results.db [> summary.md]
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    m = re.search(r"(k_[a-z_]+)", name)
    return m.group(1) + ("<" + ",".join(re.findall(r"ILi(\d+)E|Li(\d+)E", name)[0:0]) + ">" if False else "") if m else name[:60]


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, duration, grid_x, workgroup_x, vgpr_count, sgpr_count, lds_size from kernels").fetchall()
    agg = {}
    for name, dur, gx, wx, vg, sg, lds in rows:
        a = agg.setdefault(name, {"n": 0, "tot": 0, "min": 1 << 62, "max": 0, "grid": gx, "wg": wx, "vgpr": vg, "sgpr": sg, "lds": lds})
        a["n"] += 1
        a["tot"] += dur
        a["min"] = min(a["min"], dur)
        a["max"] = max(a["max"], dur)
    total = sum(a["tot"] for a in agg.values()) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | grid | wg | vgpr | sgpr | lds B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
        m = re.search(r"(k_[a-z_]+)", name)
        nm = m.group(1) if m else name[:48]
        t = re.findall(r"Li(\d+)E", name)
        if t and m:
            nm += "<" + ",".join(t) + ">"
        print(f"| {nm} | {a['n']} | {a['tot']/1e6:.3f} | {a['tot']/a['n']/1e3:.1f} | {a['min']/1e3:.1f} | {a['max']/1e3:.1f} | "
              f"{100.0*a['tot']/total:.1f} | {a['grid']} | {a['wg']} | {a['vgpr']} | {a['sgpr']} | {a['lds']} |")


if __name__ == "__main__":
    main(sys.argv[1])
