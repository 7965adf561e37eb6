#!/usr/bin/env python3
"""Print per-kernel register / scratch / occupancy figures for a .hip file (gfx950 cross-compile).
usage: python tools/kres.py chatglm_q_amd/csrc/w4_kernels.hip [filter-substring]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                     capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    txt = m.group(1).strip()
    if txt.startswith("Function Name:"):
        cur = {"name": txt.split(":", 1)[1].strip()}
        rows.append(cur)
    elif ":" in txt:
        k, v = txt.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    if flt and flt not in name:
        continue
    print(f"{name:60s} vgpr={r.get('VGPRs','?'):>4} agpr={r.get('AGPRs','?'):>3} sgpr={r.get('TotalSGPRs','?'):>3} "
          f"scratch={r.get('ScratchSize [bytes/lane]','?'):>4} occ={r.get('Occupancy [waves/SIMD]','?')} lds={r.get('LDS Size [bytes/block]','?')}")
