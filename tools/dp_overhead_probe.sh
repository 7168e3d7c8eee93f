#!/bin/bash
# where the one-rank RCCL overhead of the step comes from: per-kernel averages of the step with and without KSMI_DP_FORCE=1 (GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for tag in plain dp; do
  rm -rf /tmp/k_$tag
  if [ $tag = dp ]; then export KSMI_DP_FORCE=1; else unset KSMI_DP_FORCE; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k_$tag -o k -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-solo > /tmp/k_$tag.log 2>&1
  tail -1 /tmp/k_$tag.log | cut -c1-200
done
python - <<'PY'
import csv, glob
def load(tag):
    f = glob.glob(f"/tmp/k_{tag}/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
a, b = load("plain"), load("dp")
ta, tb = sum(v[2] for v in a.values()), sum(v[2] for v in b.values())
print(f"total kernel ms: plain {ta:.1f}  dp {tb:.1f}  ratio {tb / ta:.3f}")
for n, v in sorted(a.items(), key=lambda kv: -kv[1][2])[:14]:
    w = b.get(n)
    print(f"{n[:70]:70s} calls {v[0]:4d}/{w[0] if w else 0:4d} avg us {v[1]:8.1f} / {w[1] if w else 0:8.1f}")
print("only in dp:", [n[:60] for n in b if n not in a][:10])
PY
