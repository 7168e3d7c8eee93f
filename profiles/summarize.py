#!/usr/bin/env python3
"""Turn the rocprofv3 (rocpd sqlite) outputs of tools_profile.sh into a small committed summary.

  python profiles/summarize.py gpurun_out/prof_<tag> profiles/<tag>_summary.md

Reads <prefix>_stats/*.db (kernel trace), <prefix>_fetch/*.db (--pmc FETCH_SIZE) and
<prefix>_write/*.db (--pmc WRITE_SIZE).  FETCH_SIZE on gfx950 under-reports wide coalesced
reads by exactly 2x (MI355X_MICROARCH.md §HBM): both the raw and the doubled figure are listed;
WRITE_SIZE is uncalibrated and listed raw.
"""
import glob
import re
import sqlite3
import sys


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("unsigned short", "bf16")


def kernel_times(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    return [(short(n), c, t / 1e3, a, p) for n, c, t, a, p in rows]        # total ms, avg us


def counter(db, name):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for kn, val in cur.execute("select kernel_name, value from counters_collection where counter_name=?", (name,)):
        d = out.setdefault(short(kn), [0.0, 0])
        d[0] += val
        d[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in out.items()}


def main():
    prefix, outp = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    cmd = sys.argv[4] if len(sys.argv) > 4 else "python bench.py --steps 5 --warmup 2 --no-cpu-baseline` (7 train steps of SNUNet-ECAM bs=32 bf16)"
    stats = glob.glob(prefix + "_stats/*.db")[0]
    times = kernel_times(stats)
    fetch = counter(glob.glob(prefix + "_fetch/*.db")[0], "FETCH_SIZE") if glob.glob(prefix + "_fetch/*.db") else {}
    write = counter(glob.glob(prefix + "_write/*.db")[0], "WRITE_SIZE") if glob.glob(prefix + "_write/*.db") else {}
    tot = sum(t[2] for t in times)
    lines = ["# rocprofv3 summary: " + prefix, "", note, "",
             "`rocprofv3 --kernel-trace --stats` (+ separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes) of",
             "`" + cmd + ".  Plan construction (buffer zero fills) is included in the trace and excluded from the bench timing.", "",
             f"total kernel time {tot:.1f} ms over all dispatches", "",
             "| kernel | calls | total ms | avg us | % | FETCH_SIZE avg KB (raw) | x2 corrected MB | WRITE_SIZE avg KB (raw) |",
             "|---|---|---|---|---|---|---|---|"]
    for n, c, t, a, p in times[:40]:
        f = fetch.get(n, (None,))[0]
        w = write.get(n, (None,))[0]
        lines.append(f"| {n} | {c} | {t:.2f} | {a:.1f} | {p:.1f} | {'' if f is None else f'{f:.0f}'} | "
                     f"{'' if f is None else f'{2 * f / 1024:.1f}'} | {'' if w is None else f'{w:.0f}'} |")
    open(outp, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[8:24]))


if __name__ == "__main__":
    main()
