"""top launches of the last single-stream step per model (BENCH_LAUNCH_MAP json written by bench.py): python tools/launch_map_top.py m1 m2 ..."""
import json, sys
for m in sys.argv[1:]:
    rows = json.load(open(f"gpurun_out/lm_{m}.json"))
    tot = sum(r["ms"] for r in rows)
    print("==", m, len(rows), "launches", round(tot, 3), "ms solo")
    for r in sorted(rows, key=lambda r: -r["ms"])[:int(__import__("os").environ.get("TOP", "40"))]:
        print(r["i"], r["kind"], r["tag"], "|", " ".join(r["kernels"])[:80], "|", r["ms"], "ms", r["bytes"] // 1000000, "MB", round(r["bytes"] / r["ms"] / 1e9, 2), "TB/s")
