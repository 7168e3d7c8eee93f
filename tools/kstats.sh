#!/bin/bash
# per-kernel durations of one short bench run: bash tools/kstats.sh <grep pattern> [bench args...]  (GPU box; prints matching rows of the rocprofv3 kernel stats)
set -u
PAT=${1:-.}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kstats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -o k -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > /tmp/kstats.log 2>&1 < /dev/null
f=$(find /tmp/kstats -name "*kernel_stats.csv" | head -1)
if [ -z "$f" ]; then echo "no kernel stats"; tail -5 /tmp/kstats.log; exit 1; fi
python - "$f" "$PAT" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
for r in rows:
    if pat.search(r["Name"]):
        name = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        print(f'{name[:70]:70s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"]) / 1e3:8.1f} total_ms {float(r["TotalDurationNs"]) / 1e6:8.2f} {float(r["Percentage"]):5.1f}%')
PY
