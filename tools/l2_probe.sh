#!/bin/bash
# memory-side read requests by size for (a) the FETCH_SIZE calibration kernels (tools/fetch_calib.hip), (b) one long-K convolution:
# read bytes = 32 B x RDREQ_32B + 64 B x (RDREQ - RDREQ_32B - RDREQ_128B) + 128 B x RDREQ_128B
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value $R/tools/fetch_calib.hip -o /tmp/fetch_calib
report() {
python - "$1" <<'PY'
import glob, sqlite3, sys
dbs = glob.glob("/tmp/fp/**/*.db", recursive=True)
cur = sqlite3.connect(dbs[0]).cursor()
acc = {}
for kn, cn, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
    if sys.argv[1] in kn:
        a = acc.setdefault((kn.split("(")[0][-40:], cn), [0.0, 0]); a[0] += val; a[1] += 1
ks = sorted({k for k, _ in acc})
for k in ks:
    g = lambda c: acc.get((k, c), [0.0, 1])[0] / acc.get((k, c), [0.0, 1])[1]
    rd, r128, r32 = g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_RDREQ_128B_sum"), g("TCC_EA0_RDREQ_32B_sum")
    print(f"{k:42s} RDREQ {rd:12.0f}  128B {r128:12.0f}  32B {r32:10.0f}  -> read MB {(32 * r32 + 64 * (rd - r32 - r128) + 128 * r128) / 1e6:9.1f}   (FETCH_SIZE-style 64 B x RDREQ: {64 * rd / 1e6:9.1f})")
PY
}
rm -rf /tmp/fp; rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d /tmp/fp -o f -- /tmp/fetch_calib > /tmp/fp.log 2>&1; report calib
for shape in "L0 conv0_4 224" "L1 conv1_3 384" "L0 conv2 32->32 aff"; do
  rm -rf /tmp/fp; MB_ONLY="$shape" rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d /tmp/fp -o f -- python $R/profiles/ig4_probe.py > /tmp/fp.log 2>&1
  echo "$shape: $(grep -E "^[LX]" /tmp/fp.log)"; report igemm
done
