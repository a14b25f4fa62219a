#!/usr/bin/env python3
"""VGPR / SGPR / spill / occupancy table of the kernels in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_regs.py [source.hip] [filter-substring ...]"""
import re, subprocess, sys
src = sys.argv[1] if len(sys.argv) > 1 else "sc_gameengine_amd/csrc/sc_tick_kernels.hip"
flt = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
       "-fno-fast-math", "-fno-slp-vectorize", "-fPIC", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+([\w /\[\]]+?): (\d+)", line)
    if m and cur: rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem)
    if flt and not any(f in dem for f in flt): continue
    print(f"{dem:60s} vgpr {r.get('VGPRs', -1):3d} agpr {r.get('AGPRs', 0):3d} sgpr {r.get('SGPRs', -1):3d} spill v{r.get('VGPRs Spill', 0)} s{r.get('SGPRs Spill', 0)} "
          f"occ {r.get('Occupancy [waves/SIMD]', -1)} lds {r.get('LDS Size [bytes/block]', 0)}")
