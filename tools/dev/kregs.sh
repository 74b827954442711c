#!/bin/bash
# per-kernel register / spill / LDS figures: tools/dev/kregs.sh <source.hip> [extra hipcc flags]
# (compiles to /dev/null with -Rpass-analysis=kernel-resource-usage and condenses the remarks)
src=$1; shift
cd "$(dirname "$0")/../../bnp_vae_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -I../../include \
  -Rpass-analysis=kernel-resource-usage "$@" -c $src -o /dev/null 2>&1 | python3 -c '
import sys, re
rec = {}
for line in sys.stdin:
    m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|VGPR Spill|SGPRs): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        rec = {"name": v}
    rec[k] = v
    if k.startswith("LDS"):
        print("%-90s vgpr %-4s agpr %-3s scratch %-4s occ %-2s lds %s" % (rec["name"][:90], rec.get("VGPRs"), rec.get("AGPRs"), rec.get("ScratchSize [bytes/lane]"), rec.get("Occupancy [waves/SIMD]"), v))
'
