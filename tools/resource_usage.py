#!/usr/bin/env python3
"""Per-kernel registers / spills / occupancy as hipcc reports them (-Rpass-analysis=kernel-resource-usage).
    python tools/resource_usage.py [name filter]"""
import re
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
r = subprocess.run([sys.executable, str(REPO / "amcpy_amd/csrc/build.py"), "--force", "--save-temps"],
                   capture_output=True, text=True)
cur, rows = None, {}
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark: \S+\s+(TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
if r.returncode != 0:
    sys.exit(r.stderr[-3000:])
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
    name = name.replace("void amcx::", "").replace("wave::", "")
    if flt in name:
        print(f"{name:48s} VGPR {v.get('VGPRs', -1):4d}  spilled {v.get('VGPRs Spill', -1):4d}  scratch {v.get('ScratchSize', -1):5d} B"
              f"  SGPR {v.get('TotalSGPRs', -1):4d}  waves/SIMD {v.get('Occupancy', -1)}")
