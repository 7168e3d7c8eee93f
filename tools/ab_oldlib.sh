#!/bin/bash
# Same-box A/B of the CURRENT library against one in which some kernel files are taken from an older commit.
#   here (no GPU):  bash tools/ab_oldlib.sh build <commit> igemm4.hip [wgrad3.hip ...]   -> kurosiwo_amd/libksmi_old.so (travels with gpurun)
#   GPU box:        bash tools/ab_oldlib.sh run [bench args]                              -> alternating bench lines, new / old
# (how the silent 2-3 % cost of run-time generalisations in shared kernels was found in round 5: LABNOTES)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "$1" = build ]; then
  c=$2; shift 2
  rm -rf /tmp/ksmi_old && mkdir -p /tmp/ksmi_old/kurosiwo_amd /tmp/ksmi_old/include
  cp -r $R/kurosiwo_amd/csrc /tmp/ksmi_old/kurosiwo_amd/ && cp $R/include/ksmi.h /tmp/ksmi_old/include/
  for f in "$@"; do git -C $R show $c:kurosiwo_amd/csrc/$f > /tmp/ksmi_old/kurosiwo_amd/csrc/$f; rm -f /tmp/ksmi_old/kurosiwo_amd/csrc/${f%.hip}.o; done
  make -C /tmp/ksmi_old/kurosiwo_amd/csrc -j8 > /tmp/ksmi_old/build.log 2>&1 || { tail -5 /tmp/ksmi_old/build.log; exit 1; }
  cp /tmp/ksmi_old/kurosiwo_amd/libksmi.so $R/kurosiwo_amd/libksmi_old.so; ls -la $R/kurosiwo_amd/libksmi_old.so
  exit 0
fi
shift || true
cd $R
cp kurosiwo_amd/libksmi.so kurosiwo_amd/libksmi_new.so
run() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-solo "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  echo -n "new: "; run "$@"
  cp kurosiwo_amd/libksmi_old.so kurosiwo_amd/libksmi.so; echo -n "old: "; run "$@"
  cp kurosiwo_amd/libksmi_new.so kurosiwo_amd/libksmi.so
done
