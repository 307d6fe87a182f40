"""Register / scratch / occupancy summary of every kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kres.py refil_amd/csrc/gemm_wres.hip [filter-substring]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
cur = None
rows = {}
names = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        names.append(cur)
        rows[cur] = {}
        continue
    m = re.search(r"(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
dem = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.splitlines()
for n, d in zip(names, dem):
    d = re.sub(r"^void refil::", "", d).split("(refil::")[0]
    v = rows[n]
    if flt in d:
        print(f"{d[:100]:100s} vgpr {v.get('VGPRs', 0):3d} agpr {v.get('AGPRs', 0):3d} scratch {v.get('ScratchSize', 0):3d} spill {v.get('VGPRs Spill', 0):3d} occ {v.get('Occupancy', 0)}")
