"""Registers, spills and LDS of every kernel of a translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py rte-rrtmgp_amd/csrc/solvers.hip [name filter]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", "include",
       "-I", "rte-rrtmgp_amd/csrc", src, "-o", "/tmp/kr.so", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = {}
for ln in out.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur).split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|SGPRs|VGPR Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print(f"{k[:110]:110s} VGPR {v.get('VGPRs',0):3d} spill {v.get('VGPR Spill',0):3d} scratch {v.get('ScratchSize [bytes/lane]',0):4d} occ {v.get('Occupancy [waves/SIMD]',0)} LDS {v.get('LDS Size [bytes/block]',0)}")
