"""Per-kernel registers / scratch / occupancy of one source file, from hipcc's -Rpass-analysis=kernel-resource-usage.
    python tools/resource_usage.py refil_amd/csrc/attention_mfma.hip [substring]"""
import os
import re
import subprocess
import sys
import tempfile


def main():
    src = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    out = os.path.join(tempfile.mkdtemp(), "o.o")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Rpass-analysis=kernel-resource-usage"] + \
        os.environ.get("REFIL_EXTRA_FLAGS", "").split() + ["-c", src, "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    cur = None
    rows = {}
    for line in r.stdout.splitlines():
        m = re.search(r"remark: (?:\S+ )?\s*Function Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
            rows[cur] = {}
            continue
        m = re.search(r"remark: (?:\S+ )?\s*(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs|VGPR Spill): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).split(" ")[0]] = int(m.group(2))
    for k, v in rows.items():
        if pat in k:
            print(f"{k[:110]:110s} vgpr {v.get('VGPRs', -1):3d} agpr {v.get('AGPRs', -1):3d} scratch {v.get('ScratchSize', -1):4d} occ {v.get('Occupancy', -1)} lds {v.get('LDS', -1)}")


if __name__ == "__main__":
    main()
