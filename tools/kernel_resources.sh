#!/bin/bash
# usage: tools/kernel_resources.sh <tu.hip> [extra hipcc flags...]   (mujoco_humanoid4.hip: -mllvm -disable-machine-licm, as in the Makefile)
# Compiles the device side of one translation unit for gfx950 and prints one line
# per kernel: VGPRs / AGPRs / SGPR spills / scratch bytes per lane / LDS / occupancy.
set -e
cd "$(dirname "$0")/../envpool_amd/csrc"
TU=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage -c "$TU" -o ${KR_OUT:-/tmp/kr.co} 2>&1 |
python3 -c '
import re,sys,subprocess
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r"(Function Name|Name): (\S+)",l)
    if m:
        cur={"name":subprocess.run(["c++filt",m.group(2)],capture_output=True,text=True).stdout.strip()[:90]};rows.append(cur);continue
    m=re.search(r"remark: +([A-Za-z \[\]/]+): (\d+)",l)
    if m and cur is not None: cur[m.group(1).strip()]=int(m.group(2))
for r in rows:
    print("%-62s V=%3d A=%3d sgprspill=%3d vgprspill=%4d scratch=%5d lds=%6d occ=%d"%(r["name"][:60],r.get("VGPRs",0),r.get("AGPRs",0),r.get("SGPRs Spill",0),r.get("VGPRs Spill",0),r.get("ScratchSize [bytes/lane]",0),r.get("LDS Size [bytes/block]",0),r.get("Occupancy [waves/SIMD]",0)))
'
