#!/bin/bash
# Per-kernel register / spill / scratch table of one translation unit (clean compile with -Rpass-analysis=kernel-resource-usage).
#   tools/spill_report.sh decoder_fast            -> every kernel of csrc/decoder_fast.hip that spills or uses scratch
#   ALL=1 tools/spill_report.sh kernels           -> every kernel
cd "$(dirname "$0")/../ubisoft-laforge-zeroeggs_amd/csrc"
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $ZEGGS_DEFS -Rpass-analysis=kernel-resource-usage \
      -c $f.hip -o /tmp/spill_$f.o 2>&1 | python3 -c '
import sys, re, os
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"remark: (.*)", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"): cur = t.split(":",1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":",1); rows[cur][k.strip()] = v.replace("[-Rpass-analysis=kernel-resource-usage]","").strip()
allk = os.environ.get("ALL")
for k, r in rows.items():
    sp = int(r.get("VGPRs Spill", 0)) + int(r.get("SGPRs Spill", 0)); sc = int(r.get("ScratchSize [bytes/lane]", 0))
    if allk or sp or sc:
        name = os.popen("echo %s | c++filt" % k).read().strip()[:110]
        print("%-112s vgpr %3s agpr %3s sgpr-spill %3s vgpr-spill %3s scratch %4s occ %s" % (name, r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), sc, r.get("Occupancy [waves/SIMD]")))
'
done
