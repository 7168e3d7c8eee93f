#!/usr/bin/env python3
"""Turn the rocprofv3 (rocpd sqlite) outputs of tools_profile.sh into a small committed summary.

  python profiles/summarize.py gpurun_out/prof_<tag> profiles/<tag>_summary.md [note] [command] [traffic.json]

Reads <prefix>_stats/*.db (kernel trace), <prefix>_fetch/*.db (--pmc FETCH_SIZE), <prefix>_write/*.db (--pmc WRITE_SIZE) and
<prefix>_mfma/*.db (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE: MFMA utilisation = MFMA-busy cycles summed over
the SIMDs / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8): rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs -- calibrated on
wgrad3_kernel<4,1,4>, 885 TFLOP/s = 35 % of the dense bf16 peak by its launch time, 42 % MFMA-busy by this formula).  With a 5th argument also writes the per-kernel table as JSON
(profiles/<tag>_traffic.json), which bench.py reads to fill `roofline.traffic`.  FETCH_SIZE on gfx950 under-reports wide coalesced
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
    mdb = glob.glob(prefix + "_mfma/*.db")
    mbusy = counter(mdb[0], "SQ_VALU_MFMA_BUSY_CYCLES") if mdb else {}
    gui = counter(mdb[0], "GRBM_GUI_ACTIVE") if mdb else {}
    tot = sum(t[2] for t in times)
    passes = ["`rocprofv3 --kernel-trace --stats`"]
    if fetch or write:
        passes.append("separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes")
    if mbusy:
        passes.append("a `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE` pass")
    lines = ["# rocprofv3 summary: " + prefix, "", note, "",
             " + ".join(passes) + " of `" + cmd.rstrip("`") + "` (tools/profile.sh, tools/profile_all.sh).  Plan construction (buffer zero",
             "fills, weight packing tables) is part of the traced process and excluded from the bench timing; empty columns = counter pass not",
             "collected for this model.", "",
             f"total kernel time {tot:.1f} ms over all dispatches", "",
             "| kernel | calls | total ms | avg us | % | FETCH_SIZE avg KB (raw) | x2 corrected MB | WRITE_SIZE avg KB (raw) | MFMA busy % |",
             "|---|---|---|---|---|---|---|---|---|"]
    table = {}
    for n, c, t, a, p in times[:48]:
        f = fetch.get(n, (None,))[0]
        w = write.get(n, (None,))[0]
        mb, ga = mbusy.get(n, (None,))[0], gui.get(n, (None,))[0]
        util = None if mb is None or not ga else 100.0 * mb / (4.0 * 256.0 * ga / 8.0)   # busy cycles summed over all SIMDs; GUI_ACTIVE over 8 XCDs
        lines.append(f"| {n} | {c} | {t:.2f} | {a:.1f} | {p:.1f} | {'' if f is None else f'{f:.0f}'} | "
                     f"{'' if f is None else f'{2 * f / 1024:.1f}'} | {'' if w is None else f'{w:.0f}'} | {'' if util is None else f'{util:.1f}'} |")
        table[n] = {"calls": c, "avg_us": a, "fetch_kb_raw": f, "write_kb_raw": w, "mfma_busy_pct": util}
    open(outp, "w").write("\n".join(lines) + "\n")
    if len(sys.argv) > 5:
        import json
        json.dump({"note": "per-launch averages; HBM read bytes = 2 x FETCH_SIZE (gfx950 rocprofv3 counts 128-byte requests as 64, "
                           "MI355X_MICROARCH.md HBM section), write bytes = WRITE_SIZE (uncalibrated)", "kernels": table},
                  open(sys.argv[5], "w"), indent=1)
    print("\n".join(lines[8:24]))


if __name__ == "__main__":
    main()
