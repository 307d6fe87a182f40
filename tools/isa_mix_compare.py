"""Instruction mix of the attn_qkv_fwd instantiations in two -save-temps assembly files (old source, new source):\n    hipcc --offload-arch=gfx950 -O3 -std=c++17 -save-temps -c attention_qkv.hip ; python tools/isa_mix_compare.py old.s new.s"""
import re, sys
def kernels(path):
    out, cur, name = {}, None, None
    for line in open(path):
        m = re.match(r"^(_ZN5refil12attn_qkv_fwd\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                out[name] = cur; cur = None
    return out
def mix(lines):
    c = dict(total=0, mfma=0, valu=0, salu=0, buffer=0, ds=0, scratch=0, waitcnt=0)
    for l in lines:
        t = l.strip().split(" ")[0].split("\t")[0]
        if not t or t.startswith((".", ";")) or t.endswith(":"): continue
        c["total"] += 1
        if t.startswith("v_mfma"): c["mfma"] += 1
        elif t.startswith("v_"): c["valu"] += 1
        elif t.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif t.startswith("s_"): c["salu"] += 1
        elif t.startswith("buffer_"): c["buffer"] += 1
        elif t.startswith("ds_"): c["ds"] += 1
        elif t.startswith("scratch_"): c["scratch"] += 1
    return c
a = kernels(sys.argv[1] if len(sys.argv) > 2 else "/tmp/isa5/attention_qkv-hip-amdgcn-amd-amdhsa-gfx950.s")
b = kernels(sys.argv[2] if len(sys.argv) > 2 else "/tmp/isa6/attention_qkv-hip-amdgcn-amd-amdhsa-gfx950.s")
for k5 in sorted(a):
    k6 = k5.replace("EEEvNS_4QkvME", "ELi1EEEvNS_4QkvME")
    if k6 in b and ("ILi2ELi2ELi4E" in k5 or "ILi1ELi2ELi4E" in k5 or "ILi2ELi1ELi2E" in k5):
        print(k5[20:44], "r05", mix(a[k5])); print(" " * 24, "r06", mix(b[k6]))
