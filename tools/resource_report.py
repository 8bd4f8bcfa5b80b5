#!/usr/bin/env python3
"""Per-kernel VGPR/SGPR/spill/scratch/occupancy of sn_api.hip (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os, re, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from signerf_amd.build import CODEGEN_FLAGS  # noqa: E402  (the flags the library is built with)

cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *CODEGEN_FLAGS, "-fPIC", "-c",
       os.path.join(root, "signerf_amd/csrc/sn_api.hip"), "-o", "/tmp/sn_rr.o",
       "-Rpass-analysis=kernel-resource-usage", *sys.argv[1:]]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for k, v in rows.items():
    print("%-58s VGPR %4s AGPR %3s SGPR %3s spillV %4s scratch %5s occ %s LDS %s" % (
        k[:58], v.get("VGPRs"), v.get("AGPRs"), v.get("TotalSGPRs"), v.get("VGPRs Spill"),
        v.get("ScratchSize [bytes/lane]"), v.get("Occupancy [waves/SIMD]"), v.get("LDS Size [bytes/block]")))
