#!/usr/bin/env python3
"""Per-stage HBM traffic and MFMA-busy of ONE single-stream train step, from per-dispatch PMC rows.

  python profiles/stage_traffic.py gpurun_out/prof_<tag>_solo <launch_map.json> profiles/<tag>_<model>_stage_traffic.json

<prefix>_fetch / _write / _mfma are the rocpd databases of the single-stream `--pmc` passes of tools/profile.sh (one stream: dispatch
order = launch order).  <launch_map.json> is the per-launch table bench.py writes under BENCH_LAUNCH_MAP (kind, stage tag, the kernel
names the library reported for the launch).  The last step of each pass (from its last `pack_weights_batched_kernel` dispatch on) is
aligned with the launch table: a launch with reported kernel names anchors on its first name, dispatches in front of an anchor belong
to the launch before it (row folds, reducers, two-kernel launches), a launch without names (elementwise) takes the next dispatch.
Bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; x2 = the gfx950 correction of MI355X_MICROARCH.md), MFMA-busy = busy cycles over
the SIMDs / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), duration-free: weighted by GRBM_GUI_ACTIVE."""
import glob
import json
import re
import sqlite3
import sys


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("unsigned short", "bf16")


def dispatches(db, counters):
    """[(kernel name, {counter: value})] in dispatch order"""
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    order = next((c for c in ("dispatch_id", "id", "correlation_id") if c in cols), "rowid")
    rows = {}
    for did, kn, cn, val in con.execute(f"select {order}, kernel_name, counter_name, value from counters_collection order by {order}"):
        if cn in counters:
            d = rows.setdefault(did, [short(kn), {}])
            d[1][cn] = d[1].get(cn, 0.0) + val
    return [tuple(rows[k]) for k in sorted(rows)]


def last_step(seq):
    starts = [i for i, (n, _) in enumerate(seq) if n.startswith("pack_weights_batched_kernel")]
    if not starts:
        raise SystemExit("no pack_weights_batched_kernel dispatch: not a plan-based train step")
    return seq[starts[-1]:]


def align(step, table):
    """-> list of (stage, [dispatch values]) per launch-table row"""
    out, i = [], 0
    for row in table:
        names = row["kernels"]
        got = []
        if names:
            j = i
            while j < len(step) and j < i + 8 and step[j][0] != names[0]:
                j += 1
            if j < len(step) and step[j][0] == names[0]:
                if out:
                    out[-1][1].extend(step[i:j])           # dispatches in front of the anchor belong to the previous launch
                i = j
                for n in names:
                    if i < len(step) and step[i][0] == n:
                        got.append(step[i]); i += 1
            # (anchor not found within 8 dispatches: leave the launch empty rather than derail the walk)
        elif i < len(step):
            got.append(step[i]); i += 1
        out.append((row.get("stage") or "other", got))
    if out:
        out[-1][1].extend(step[i:])
    return out


def main():
    prefix, table_path, outp = sys.argv[1], sys.argv[2], sys.argv[3]
    table = json.load(open(table_path))
    res = {}

    def add(kind, counters, fold):
        dbs = glob.glob(prefix + f"_{kind}/*.db")
        if not dbs:
            return
        al = align(last_step(dispatches(dbs[0], counters)), table)
        for stage, ds in al:
            st = res.setdefault(stage, {"fetch_kb": 0.0, "write_kb": 0.0, "mfma_busy": 0.0, "gui": 0.0, "dispatches": 0, "launches": 0})
            fold(st, ds)
        res.setdefault("_matched", {})[kind] = sum(1 for _, ds in al if ds) / max(len(al), 1)

    add("fetch", {"FETCH_SIZE"}, lambda st, ds: st.__setitem__("fetch_kb", st["fetch_kb"] + sum(v.get("FETCH_SIZE", 0.0) for _, v in ds)))
    add("write", {"WRITE_SIZE"}, lambda st, ds: st.__setitem__("write_kb", st["write_kb"] + sum(v.get("WRITE_SIZE", 0.0) for _, v in ds)))

    def fold_mfma(st, ds):
        st["mfma_busy"] += sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for _, v in ds)
        st["gui"] += sum(v.get("GRBM_GUI_ACTIVE", 0.0) for _, v in ds)
        st["dispatches"] += len(ds)
        st["launches"] += 1
    add("mfma", {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"}, fold_mfma)
    matched = res.pop("_matched", {})
    out = {"note": "one single-stream train step; counted_GB = (2 x FETCH_SIZE + WRITE_SIZE) summed over the dispatches of the stage's launches; "
                   "mfma_busy_pct = MFMA-busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); profiles/stage_traffic.py",
           "launch_rows_with_dispatches": matched, "stages": {}}
    for stage, st in sorted(res.items()):
        out["stages"][stage] = {"counted_GB": round((2.0 * st["fetch_kb"] + st["write_kb"]) * 1024.0 / 1e9, 3),
                                "read_GB": round(2.0 * st["fetch_kb"] * 1024.0 / 1e9, 3), "write_GB": round(st["write_kb"] * 1024.0 / 1e9, 3),
                                "mfma_busy_pct": round(100.0 * st["mfma_busy"] / (1024.0 * st["gui"] / 8.0), 1) if st["gui"] else None,
                                "launches": st["launches"], "dispatches": st["dispatches"]}
    out["total_counted_GB"] = round(sum(v["counted_GB"] for v in out["stages"].values()), 3)
    json.dump(out, open(outp, "w"), indent=1)
    print(json.dumps(out)[:2000])


if __name__ == "__main__":
    main()
