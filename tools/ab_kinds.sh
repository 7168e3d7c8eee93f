#!/bin/bash
# same-box solo times (us) of launch kinds under environment settings: bash tools/ab_kinds.sh "<kind,kind>" "<VAR=val>" ...
kinds="$1"; shift
for cfg in "$@"; do
  env $cfg BENCH_LAUNCH_MAP=/tmp/map.json python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python - "$kinds" "$cfg" <<'PY'
import json, sys
m = json.load(open("/tmp/map.json"))
for k in sys.argv[1].split(","):
    r = [e["ms"] for e in m if e["kind"] == k]
    print(f"[{sys.argv[2]}] {k}: n={len(r)} total {sum(r)*1e3:.1f} us")
print(f"[{sys.argv[2]}] solo step {sum(e['ms'] for e in m):.3f} ms")
PY
done
