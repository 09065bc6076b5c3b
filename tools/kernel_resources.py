#!/usr/bin/env python3
"""Print VGPR / SGPR / scratch / LDS / occupancy of every kernel in pt_core.hip (hipcc -Rpass-analysis)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "rust-pathtracer_amd")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fPIC",
       "-c", f"-I{ROOT}/include", "csrc/device/pt_core.hip", "-o", "/tmp/pt_core_res.o", "-Rpass-analysis=kernel-resource-usage"]
cmd += sys.argv[1:]
r = subprocess.run(cmd, cwd=PKG, capture_output=True, text=True)
cur, rows = None, {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|TotalSGPRs|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur and "rocprim" not in cur:
        rows.setdefault(cur, {})[m.group(1).split(" ")[0]] = int(m.group(2))
for name, v in rows.items():
    m = re.search(r"\d+(k_[a-z_]+)(?:I((?:L[bij]\d+E)+)E)?", name)
    short = (m.group(1) + ("<" + ",".join(re.findall(r"L[bij](\d+)E", m.group(2))) + ">" if m.group(2) else "")) if m else name[:40]
    print(f"{short:42s} VGPR {v.get('VGPRs', 0):4d} AGPR {v.get('AGPRs', 0):3d} SGPR {v.get('TotalSGPRs', 0):4d} scratch {v.get('ScratchSize', 0):5d} "
          f"LDS {v.get('LDS', 0):6d} occ {v.get('Occupancy', 0)}")
if r.returncode != 0:
    sys.stderr.write(r.stderr[-3000:])
    sys.exit(1)
