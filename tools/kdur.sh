#!/bin/bash
# GPU-side duration of the kernels of one short command (rocprofv3 kernel trace): bash tools/kdur.sh <grep pattern> <command...>
# (GPU box; prints calls / average us of every kernel whose name matches the pattern)
set -u
PAT=${1:-.}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kdur
( cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kdur -o k -- "$@" > /tmp/kdur.log 2>&1 < /dev/null )
f=$(find /tmp/kdur -name "*kernel_stats.csv" | head -1)
if [ -z "$f" ]; then echo "no kernel stats"; tail -5 /tmp/kdur.log; exit 1; fi
python - "$f" "$PAT" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
for r in rows:
    if pat.search(r["Name"]):
        name = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        print(f'{name[:90]:90s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"]) / 1e3:8.1f}')
PY
