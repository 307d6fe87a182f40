"""Builds librefil_hip.so (gfx950) in-tree with hipcc. No torch headers are involved: the library is a
plain C-ABI shared object (include/refil_hip.h) loaded through ctypes by refil_amd/_lib.py."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["gemm.hip", "gemm_wres.hip", "gemm_dw.hip", "gemm_dw4.hip", "attention.hip", "attention_mfma.hip", "attention_qkv.hip", "gru.hip", "mixer.hip", "replay.hip", "collective.hip", "learner.hip", "profile.hip"]
LIB = os.path.join(HERE, "librefil_hip.so")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "refil_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + os.environ.get("REFIL_EXTRA_FLAGS", "").split() + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
    # The dynamic symbol table holds the C ABI of include/refil_hip.h and nothing else: the launchers' C++ symbols and the kernel
    # handles stay local to the library (a linker version script: no object file is compiled differently for it).
    vs = os.path.join(HERE, "build", "exports.map")
    with open(vs, "w") as fh:
        fh.write("{ global: refil_*; local: *; };\n")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={vs}", "-o", LIB] + objs + ["-ldl"]
    subprocess.check_call(cmd)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
