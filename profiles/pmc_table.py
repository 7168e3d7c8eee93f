#!/usr/bin/env python3
"""Print per-kernel averages of the PMC counters collected by tools_pmc.sh (rocpd sqlite)."""
import glob, re, sqlite3, sys
from collections import defaultdict
def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*\)$", "", n).replace("unsigned short", "bf16")
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for pref in sys.argv[1:]:
    for db in glob.glob(pref + "/*.db"):
        cur = sqlite3.connect(db).cursor()
        for kn, cn, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
            a = acc[short(kn)][cn]; a[0] += val; a[1] += 1
for k, cs in acc.items():
    if "igemm" not in k: continue
    print(k)
    for cn, (s, n) in sorted(cs.items()):
        print(f"   {cn:28s} {s / n:16.0f}  (n={n})")
