"""Occupancy of the GPU timeline of a traced run (rocprofv3 --kernel-trace --output-format csv): busy fraction (>= 1 kernel running),
mean concurrency, the largest idle gaps with the kernels around them, per-stream busy time.
    python tools/timeline.py <kernel_trace.csv> [first_ms last_ms]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda e: e[0])
    t0 = ev[0][0]
    if len(sys.argv) == 3:             # "<name>:<k>": the window between the k-th last and the last launch of a kernel whose name contains <name>
        name, k = sys.argv[2].split(":")
        marks = [e[0] for e in ev if name in e[2]]
        lo, hi = marks[-int(k) - 1], marks[-1]
        ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
        print(f"window: {int(k)} intervals between launches of *{name}*: {(hi - lo) / 1e6 / int(k):.3f} ms each")
    if len(sys.argv) > 3:
        lo, hi = t0 + int(float(sys.argv[2]) * 1e6), t0 + int(float(sys.argv[3]) * 1e6)
        ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
    start, end = ev[0][0], max(e[1] for e in ev)
    pts = sorted([(s, 1) for s, _, _, _ in ev] + [(e, -1) for _, e, _, _ in ev])
    busy = conc = 0
    level, last = 0, start
    gaps = []
    for t, d in pts:
        if level > 0:
            busy += t - last
            conc += level * (t - last)
        elif t > last:
            gaps.append((t - last, last))
        level += d
        last = t
    wall = end - start
    print(f"{len(ev)} kernels over {wall / 1e6:.2f} ms: busy {busy / wall:.3f}, mean concurrency while busy {conc / max(busy, 1):.2f}, sum of durations {sum(e[1] - e[0] for e in ev) / 1e6:.2f} ms")
    per_q = {}
    for s, e, _, q in ev:
        per_q[q] = per_q.get(q, 0) + e - s
    print("per queue busy ms:", {q: round(v / 1e6, 2) for q, v in sorted(per_q.items(), key=lambda kv: -kv[1])})
    gaps.sort(reverse=True)
    print(f"idle: {sum(g for g, _ in gaps) / 1e6:.3f} ms in {len(gaps)} gaps; > 5 us: {sum(1 for g, _ in gaps if g > 5000)} gaps, {sum(g for g, _ in gaps if g > 5000) / 1e6:.3f} ms")
    for g, at in gaps[:12]:
        before = max((e for e in ev if e[1] <= at), key=lambda e: e[1], default=None)
        after = min((e for e in ev if e[0] >= at + g), key=lambda e: e[0], default=None)
        nm = lambda e: e[2].split("(")[0][-50:] if e else "-"
        print(f"  gap {g / 1e3:7.1f} us at +{(at - start) / 1e6:8.3f} ms   after [{nm(before)}]  before [{nm(after)}]")


if __name__ == "__main__":
    main()
