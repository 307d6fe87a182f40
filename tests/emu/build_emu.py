"""TEST INFRASTRUCTURE: builds tests/emu/_build/librefil_emu.so -- the kernel SOURCES of refil_amd/csrc compiled as x86 C++ against the
CPU wavefront emulator (include/hip/hip_runtime.h, emu_rt.cpp). Used by the `-m "not gpu"` emulator tests only; the product never loads it.

The sources are compiled as they are except for four mechanical rewrites done on a scratch copy (the files under refil_amd/csrc are
not touched):
  * `extern __shared__ <attrs> T name[];`   -> `T* const name = static_cast<T*>(emu::dyn_smem());`   (dynamic LDS)
  * the gfx950 inline-asm statements, by template: empty / s_nop -> nothing (they are register-allocation and hazard hints),
    v_add_f32 / v_sub_f32 -> the C expression, `s_waitcnt lgkmcnt(0); s_barrier` -> the workgroup barrier. An unknown template is an error.
  * `#include "../../include/refil_hip.h"`  -> the absolute path (the scratch copy lives elsewhere).
  * `#pragma unroll` lines are dropped (host compile time; no semantic content).
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "refil_amd", "csrc")
# EMU_ASAN=1: an AddressSanitizer build of the same thing (its own directory): out-of-bounds global / LDS accesses of the kernels, which the
# GPU tolerates silently, abort with a report. Run python with LD_PRELOAD=$(clang++ -print-file-name=libclang_rt.asan-x86_64.so)
# ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 (tools/emu_asan.sh).
ASAN = os.environ.get("EMU_ASAN") == "1"
# EMU_UBSAN=1: the same with UndefinedBehaviorSanitizer (misaligned vector accesses, shifts past the width, signed overflow in index
# arithmetic, out-of-range float -> int conversions): LD_PRELOAD libclang_rt.ubsan_standalone-x86_64.so, UBSAN_OPTIONS=print_stacktrace=1.
UBSAN = os.environ.get("EMU_UBSAN") == "1"
OUT = os.path.join(HERE, "_build_asan" if ASAN else ("_build_ubsan" if UBSAN else "_build"))
LIB = os.path.join(OUT, "librefil_emu.so")
CLANG = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")


def _sources():
    sys.path.insert(0, ROOT)
    from refil_amd.build import SOURCES
    return list(SOURCES)


def _match_paren(s, i):
    """index just past the parenthesis that closes the one at s[i]"""
    depth, j, in_str = 0, i, False
    while j < len(s):
        c = s[j]
        if in_str:
            if c == "\\":
                j += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True
        elif c == "(":
            depth += 1
        elif c == ")":
            depth -= 1
            if depth == 0:
                return j + 1
        j += 1
    raise ValueError("unbalanced parenthesis")


def _split_top(s, sep):
    out, depth, in_str, cur, i = [], 0, False, "", 0
    while i < len(s):
        c = s[i]
        if in_str:
            cur += c
            if c == "\\":
                cur += s[i + 1]
                i += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True
            cur += c
        elif c in "([":
            depth += 1
            cur += c
        elif c in ")]":
            depth -= 1
            cur += c
        elif c == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += c
        i += 1
    out.append(cur)
    return out


def _operands(part):
    """'"+v"(a), "v"(b)' -> ['a', 'b']"""
    ops = []
    for o in _split_top(part, ","):
        o = o.strip()
        if not o:
            continue
        m = re.match(r'"[^"]*"\s*\((.*)\)\s*$', o, re.S)
        if not m:
            raise ValueError(f"asm operand not understood: {o!r}")
        ops.append(m.group(1).strip())
    return ops


def _asm_to_c(body, where):
    parts = _split_top(body, ":")
    template = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', parts[0]))
    outs = _operands(parts[1]) if len(parts) > 1 else []
    ins = _operands(parts[2]) if len(parts) > 2 else []
    t = template.replace("\\n", "\n").replace("\\t", " ").strip()
    if t == "" or re.fullmatch(r"s_nop \d+", t):
        return ";"          # (an empty statement: the asm may be the whole body of a for)
    if t == "v_add_f32 %0, %0, %1" and len(outs) == 1 and len(ins) == 1:
        return f"{outs[0]} += {ins[0]};"
    if t == "v_sub_f32_e32 %0, %1, %2" and len(outs) == 1 and len(ins) == 2:
        return f"{outs[0]} = {ins[0]} - {ins[1]};"
    if re.fullmatch(r"s_waitcnt lgkmcnt\(0\)\s*s_barrier", t):
        return "emu::block_sync();"
    raise ValueError(f"{where}: inline asm template not known to the emulator build: {template!r}")


def transform(text, name):
    # dynamic LDS
    text = re.sub(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][\w ]*?)\s+(\w+)\[\];",
                  lambda m: f"{m.group(1)}* const {m.group(2)} = static_cast<{m.group(1)}*>(emu::dyn_smem());", text)
    assert "extern __shared__" not in text, f"{name}: an extern __shared__ declaration was not rewritten"
    # inline asm
    out, i = "", 0
    for m in re.finditer(r"\basm\s*(?:volatile\s*)?\(", text):
        if m.start() < i:
            continue
        line_start = text.rfind("\n", 0, m.start()) + 1
        if "//" in text[line_start:m.start()]:
            continue        # inside a comment
        end = _match_paren(text, m.end() - 1)
        body = text[m.end():end - 1]
        semi = end
        while text[semi] in " \t":
            semi += 1
        assert text[semi] == ";", f"{name}: asm statement without ';'"
        out += text[i:m.start()] + _asm_to_c(body, name)
        i = semi + 1
    text = out + text[i:]
    # forced unrolling only costs host compile time (minutes for the heavily instantiated GEMM files): the loops stay loops here
    text = re.sub(r"^[ \t]*#pragma unroll.*$", "", text, flags=re.M)
    text = text.replace('#include "../../include/refil_hip.h"', f'#include "{os.path.join(ROOT, "include", "refil_hip.h")}"')
    return text


def _digest(*chunks):
    import hashlib
    h = hashlib.sha256()
    for c in chunks:
        h.update(c if isinstance(c, bytes) else c.encode())
    return h.hexdigest()


def build(force=False, verbose=False, opt="-O1"):
    """Incremental: an object is recompiled when its transformed source, any header, the shim or the flags changed (content hashes kept
    beside the objects), so a second call is a no-op and an edit of one kernel file costs one compile."""
    src_dir = os.path.join(OUT, "src")
    os.makedirs(src_dir, exist_ok=True)
    texts = {}
    for f in sorted(os.listdir(CSRC)):
        with open(os.path.join(CSRC, f)) as fh:
            texts[f] = transform(fh.read(), f)
    flags = [CLANG, "-x", "c++", "-std=c++17", opt, "-fPIC", "-pthread", "-fno-strict-aliasing", "-ffp-contract=off",
             "-I", os.path.join(HERE, "include"), "-I", src_dir, "-Wno-unknown-pragmas", "-Wno-pass-failed", "-Wno-unused-value",
             "-Wno-ignored-attributes", "-Wno-unknown-attributes"] + os.environ.get("EMU_EXTRA_FLAGS", "").split()
    if ASAN:
        flags += ["-fsanitize=address", "-shared-libasan", "-fno-omit-frame-pointer", "-g1"]
    if UBSAN:
        flags += ["-fsanitize=undefined", "-fno-sanitize=vptr,function", "-shared-libsan", "-fno-omit-frame-pointer", "-g1"]
    with open(os.path.join(HERE, "include", "hip", "hip_runtime.h")) as fh:
        shim = fh.read()
    with open(os.path.join(ROOT, "include", "refil_hip.h")) as fh:
        abi = fh.read()
    with open(os.path.join(HERE, "emu_rt.cpp")) as fh:
        rt = fh.read()
    headers = "".join(texts[f] for f in texts if f.endswith(".h")) + shim + abi + " ".join(flags)
    units = []      # (source path, object path, digest)
    for s in _sources():
        units.append((os.path.join(src_dir, s.replace(".hip", ".cpp")), os.path.join(OUT, s.replace(".hip", ".o")), _digest(texts[s], headers)))
    units.append((os.path.join(HERE, "emu_rt.cpp"), os.path.join(OUT, "emu_rt.o"), _digest(rt, shim, " ".join(flags))))
    for f, text in texts.items():
        dst = os.path.join(src_dir, f.replace(".hip", ".cpp"))
        old = None
        if os.path.exists(dst):
            with open(dst) as fh:
                old = fh.read()
        if old != text:
            with open(dst, "w") as fh:
                fh.write(text)
    todo = []
    for src, obj, dig in units:
        stamp = obj + ".sha"
        have = None
        if os.path.exists(stamp) and os.path.exists(obj):
            with open(stamp) as fh:
                have = fh.read()
        if force or have != dig:
            todo.append((src, obj, dig))
    if not todo and os.path.exists(LIB):
        return LIB
    procs = [(src, obj, dig, subprocess.Popen(flags + ["-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)) for src, obj, dig in todo]
    failed = False
    for src, obj, dig, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(f"--- {src}\n" + out.decode())
            failed = True
            continue
        if verbose and out:
            sys.stderr.write(out.decode())
        with open(obj + ".sha", "w") as fh:
            fh.write(dig)
    if failed:
        raise RuntimeError("emulator build failed")
    subprocess.check_call([CLANG, "-shared", "-fPIC", "-pthread"] + (["-fsanitize=address", "-shared-libasan"] if ASAN else []) + (["-fsanitize=undefined", "-shared-libsan"] if UBSAN else []) +
                          ["-o", LIB] + [obj for _, obj, _ in units] + ["-ldl", "-lm"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
