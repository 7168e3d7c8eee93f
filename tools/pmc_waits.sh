#!/bin/bash
# what every kernel of the single-stream SNUNet step waits for: three PMC passes (GPU box) -> gpurun_out/pmc_waits.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pw_a /tmp/pw_b /tmp/pw_c
export KSMI_OVERLAP_WGRAD=0 KSMI_OVERLAP_LANES=0
CMD="python $R/bench.py ${PMC_ARGS:-} --steps 3 --warmup 2 --no-cpu-baseline --no-solo"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --kernel-trace -d /tmp/pw_a -o a -- $CMD > $R/gpurun_out/pmc_waits_a.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pw_b -o b -- $CMD > $R/gpurun_out/pmc_waits_b.log 2>&1
cd $R
python - <<'PY' > gpurun_out/pmc_waits.txt 2>&1
import glob, re, sqlite3
from collections import defaultdict
def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*\)$", "", n)[:64]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for pref in ("/tmp/pw_a", "/tmp/pw_b"):
    for db in glob.glob(pref + "/**/*.db", recursive=True):
        cur = sqlite3.connect(db).cursor()
        for kn, cn, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
            a = acc[short(kn)][cn]; a[0] += val; a[1] += 1
rows = []
for k, cs in acc.items():
    g = lambda c: cs[c][0] if c in cs else 0.0
    wc = g("SQ_WAVE_CYCLES")
    if wc <= 0: continue
    n = cs["SQ_WAVE_CYCLES"][1]
    gui = g("GRBM_GUI_ACTIVE")
    rows.append((gui, k, n, 100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_WAIT_INST_LDS") / wc, 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_ACTIVE_INST_VALU") / wc,
                 100 * g("SQ_ACTIVE_INST_SCA") / wc, 100 * g("SQ_ACTIVE_INST_LDS") / wc, 100 * g("SQ_ACTIVE_INST_VMEM") / wc,
                 100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * gui / 8) if gui else 0, 100 * g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1)))
tot = sum(r[0] for r in rows)
print("share of GUI-active cycles | kernel | dispatches | % of wave cycles: wait_inst_any, wait_inst_lds, wait_any, valu issue, scalar issue, lds issue, vmem issue | MFMA busy % | LDS conflict % of LDS cycles")
for r in sorted(rows, reverse=True)[:45]:
    print(f"{100 * r[0] / tot:5.1f}% {r[1]:64s} n={r[2]:4d}  wi_any {r[3]:5.1f} wi_lds {r[4]:5.1f} w_any {r[5]:5.1f} valu {r[6]:5.1f} sca {r[7]:5.1f} lds {r[8]:5.1f} vmem {r[9]:5.1f} | mfma {r[10]:5.1f} | conf {r[11]:5.1f}")
PY
cat gpurun_out/pmc_waits.txt
