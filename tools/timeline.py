#!/usr/bin/env python3
"""Per-kernel timeline statistics of a rocprofv3 --kernel-trace database (rocpd SQLite, ROCm 7.2).

usage: timeline.py loop.db [alone.db ...] [--skip-first N]  > table.md

For the first database: per kernel name -- launches, average duration, and the average QUEUE DELAY: start of the launch minus
max(end of the previous launch on the same stream, own enqueue is unknown to the trace) -- the time a launch whose stream
predecessor had finished still waited for the device, i.e. stood behind the kernels of OTHER streams.  For the further
databases (the same kernels run alone) the average duration is printed beside it.  Also: busy time of the device (union of all
kernel intervals) against the wall time of the traced window, per stream busy time.
"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    nm = m.group(1) if m else re.sub(r"\(.*", "", name)[:56]
    t = re.findall(r"Li(\d+)E|Lb([01])E", name)
    if t and m:
        nm += "<" + ",".join(a or b for a, b in t) + ">"
    return nm


def load(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end, stream_id, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
    return [(short(n), s, e, st, q, g, w) for n, s, e, st, q, g, w in rows]


def stats(rows, skip_frac=0.0):
    if not rows:
        return {}, 0, 0, {}
    t0 = rows[0][1]
    t1 = max(r[2] for r in rows)
    lo = t0 + (t1 - t0) * skip_frac
    agg = {}
    last_end = {}
    for nm, s, e, st, q, g, w in rows:
        prev = last_end.get(st)
        last_end[st] = e
        if s < lo:
            continue
        a = agg.setdefault(nm, {"n": 0, "dur": 0, "gap": 0, "ngap": 0, "grid": g, "wg": w, "streams": set()})
        a["n"] += 1
        a["dur"] += e - s
        a["streams"].add(st)
        if prev is not None:
            a["gap"] += max(0, s - prev)
            a["ngap"] += 1
    # device busy = union of intervals inside the window
    iv = sorted((max(s, lo), e) for _n, s, e, *_ in rows if e > lo)
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    per_stream = {}
    for _n, s, e, st, *_ in rows:
        if e > lo:
            per_stream[st] = per_stream.get(st, 0) + (e - max(s, lo))
    return agg, busy, t1 - lo, per_stream


def main(argv):
    skip = 0.25
    paths = []
    i = 0
    while i < len(argv):
        if argv[i] == "--skip-frac":
            skip = float(argv[i + 1])
            i += 2
        else:
            paths.append(argv[i])
            i += 1
    loop = load(paths[0])
    agg, busy, wall, per_stream = stats(loop, skip)
    alone = {}
    for p in paths[1:]:
        a2, _b, _w, _ps = stats(load(p), skip)
        for k, v in a2.items():
            alone.setdefault(k, v)
    print(f"window {wall/1e6:.2f} ms (the last {100*(1-skip):.0f} % of the trace), device busy {busy/1e6:.2f} ms = {100.0*busy/max(wall,1):.1f} %; "
          f"busy per stream: " + ", ".join(f"s{st}: {t/1e6:.2f}" for st, t in sorted(per_stream.items())))
    print()
    print("| kernel | streams | launches | alone avg us | in the loop avg us | stretch | avg delay behind its stream predecessor us | total in loop ms |")
    print("|---|---|---|---|---|---|---|---|")
    for nm, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
        al = alone.get(nm)
        al_us = al["dur"] / al["n"] / 1e3 if al else None
        in_us = a["dur"] / a["n"] / 1e3
        print(f"| {nm} | {','.join(str(s) for s in sorted(a['streams']))} | {a['n']} | {al_us:.1f} |" if al_us is not None else f"| {nm} | {','.join(str(s) for s in sorted(a['streams']))} | {a['n']} | - |", end="")
        print(f" {in_us:.1f} | {in_us / al_us:.2f} | " if al_us else f" {in_us:.1f} | - | ", end="")
        print(f"{a['gap'] / max(a['ngap'], 1) / 1e3:.1f} | {a['dur'] / 1e6:.3f} |")


if __name__ == "__main__":
    main(sys.argv[1:])
