#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for shape in "L0 conv0_4 224" "L1 conv1_3 384" "L2 conv2_2 640" "L3 conv3_1 1024"; do
  for var in "" "4,2"; do
    rm -rf /tmp/fp
    KSMI_IGEMM4_VAR=$var MB_ONLY="$shape" rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/fp -o f -- python $R/profiles/ig4_probe.py > /tmp/fp.log 2>&1
    python - "$shape" "$var" <<'PY'
import glob, sqlite3, sys
db = glob.glob("/tmp/fp/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
acc = {}
for kn, val in cur.execute("select kernel_name, value from counters_collection where counter_name='FETCH_SIZE'"):
    if "igemm" in kn:
        k = kn.split("(")[0][-60:]
        a = acc.setdefault(k, [0.0, 0]); a[0] += val; a[1] += 1
for k, (v, n) in acc.items():
    print(f"{sys.argv[1]:22s} var=[{sys.argv[2]:3s}] {k:55s} calls {n:3d} FETCH raw {v / n / 1024:9.1f} MB  x2 {2 * v / n / 1024:9.1f} MB")
PY
    grep -E "^(L[0-9]|X0)" /tmp/fp.log
  done
done
