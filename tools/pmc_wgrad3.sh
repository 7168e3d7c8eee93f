#!/bin/bash
# where do the waves of wgrad3_kernel wait?  PMC passes over profiles/wgrad3_probe.py (GPU box) -> gpurun_out/pmc_wgrad3.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw3_a /tmp/pw3_b /tmp/pw3_c
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace -d /tmp/pw3_a -o a -- python $R/profiles/wgrad3_probe.py > $R/gpurun_out/pmc_wgrad3_a.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pw3_b -o b -- python $R/profiles/wgrad3_probe.py > $R/gpurun_out/pmc_wgrad3_b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT --kernel-trace -d /tmp/pw3_c -o c -- python $R/profiles/wgrad3_probe.py > $R/gpurun_out/pmc_wgrad3_c.log 2>&1
cd $R
python - <<'PY' > gpurun_out/pmc_wgrad3.txt 2>&1
import glob, re, sqlite3
from collections import defaultdict
def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*\)$", "", n)
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for pref in ("/tmp/pw3_a", "/tmp/pw3_b", "/tmp/pw3_c"):
    for db in glob.glob(pref + "/**/*.db", recursive=True):
        cur = sqlite3.connect(db).cursor()
        try:
            rows = cur.execute("select kernel_name, counter_name, value from counters_collection")
        except Exception as e:
            print("no counters_collection in", db, e); continue
        for kn, cn, val in rows:
            a = acc[short(kn)][cn]; a[0] += val; a[1] += 1
for k, cs in sorted(acc.items()):
    if "wgrad3" not in k: continue
    print(k)
    for cn, (s, n) in sorted(cs.items()):
        print(f"   {cn:28s} {s / n:16.0f}  (n={n})")
PY
cat gpurun_out/pmc_wgrad3.txt | head -120; tail -22 gpurun_out/pmc_wgrad3_a.log | grep "H="
