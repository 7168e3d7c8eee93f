#!/bin/bash
# same-box solo times (us) per (kind, tag) under environment settings, for kinds matching a prefix: bash tools/ab_tags.sh "<kind prefix>" "<VAR=val>" ...
pre="$1"; shift
for cfg in "$@"; do
  env $cfg BENCH_LAUNCH_MAP=/tmp/map.json python bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  python - "$pre" "$cfg" <<'PY'
import json, sys, collections
m = json.load(open("/tmp/map.json"))
agg = collections.OrderedDict()
for e in m:
    if e["kind"].startswith(sys.argv[1]):
        k = (e["kind"], e["tag"], tuple(e["kernels"][:1]))
        agg.setdefault(k, []).append(e["ms"])
for (kind, tag, kern), v in agg.items():
    print(f"[{sys.argv[2]}] {kind} {tag} n={len(v)} avg {sum(v)/len(v)*1e3:.1f} us {kern}")
PY
done
