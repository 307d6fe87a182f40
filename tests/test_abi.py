"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports every symbol that
include/refil_hip.h declares; layout queries (host-only code) behave. No GPU compute is called."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from refil_amd import build
    build.build(verbose=False)
    from refil_amd import _lib
    return _lib.lib()


def test_header_symbols_are_exported(L):
    from refil_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "refil_hip.h")).read()
    declared = set(re.findall(r"\b(refil_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), f"{name} declared in refil_hip.h but not exported"


def test_library_exports_the_header_and_nothing_else(L):
    """The dynamic symbol table of librefil_hip.so is exactly the C ABI of include/refil_hip.h (no C++ launcher symbols, no kernel
    handles: refil_amd/build.py links with a version script)."""
    import subprocess
    from refil_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert exported == set(_lib.EXPORTS), sorted(exported ^ set(_lib.EXPORTS))[:20]


def test_struct_sizes_match_header():
    """ctypes mirrors must have the C sizes (computed from the header with the host compiler)."""
    import subprocess
    import tempfile
    from refil_amd import _lib
    src = '#include "refil_hip.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",' \
          'sizeof(refil_dims),sizeof(refil_param_layout),sizeof(refil_batch),sizeof(refil_debug_out),' \
          'sizeof(refil_gemm_desc),sizeof(refil_attn_desc),sizeof(refil_gru_desc),sizeof(refil_opt_hyper),' \
          'sizeof(refil_gather_field),sizeof(refil_profile_entry),sizeof(refil_attn_qkv_desc));return 0;}\n'
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    mine = [C.sizeof(t) for t in (_lib.Dims, _lib.ParamLayout, _lib.Batch, _lib.DebugOut, _lib.GemmDesc, _lib.AttnDesc,
                                  _lib.GruDesc, _lib.OptHyper, _lib.GatherField, _lib.ProfileEntry, _lib.AttnQkvDesc)]
    assert sizes == mine


def test_param_layout_counts(L):
    from refil_amd import _lib, flat
    # cfg-T shapes: P = 433 878 parameters
    d = _lib.make_dims(B=1, T1=2, ne=32, na=16, ed=62, A=22, d=128, heads=4, H=64, hyp=128, M=32,
                       entity_last_action=1, imagine=1, softmax_mixing_weights=1, mixer_tanh=0, double_q=1,
                       gamma=0.99, lmbda=0.5)
    import torch
    n = flat.total(d)
    a, m = flat.views(torch.zeros(n), d)
    assert sum(v.numel() for v in a.values()) + sum(v.numel() for v in m.values()) == 433878
    assert n >= 433878 and n % 4 == 0
    # views must not overlap
    buf = torch.zeros(n)
    a, m = flat.views(buf, d)
    for v in list(a.values()) + list(m.values()):
        v += 1
    assert buf.max().item() == 1.0


def test_bad_dims_are_rejected_with_message(L):
    from refil_amd import _lib
    d = _lib.make_dims(B=1, T1=2, ne=32, na=16, ed=62, A=22, d=128, heads=4, H=48, hyp=128, M=32)
    out = _lib.ParamLayout()
    rc = L.refil_get_param_layout(C.byref(d), C.byref(out))
    assert rc != 0 and b"rnn_hidden_dim" in L.refil_last_error()
    assert L.refil_learner_workspace_bytes(C.byref(d)) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from refil_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.lib()


@pytest.mark.parametrize("bad,needle", [
    (dict(ne=65), b"n_entities <= 64"),                      # the 64-bit mask words' limit
    (dict(na=33, ne=32), b"n_agents <= n_entities"),
    (dict(B=0), b"B and T1"),
    (dict(heads=3), b"divisible by attn_n_heads"),
    (dict(d=24, heads=4), b"head dim"),
    (dict(M=65), b"mixing_embed_dim"),
    (dict(pooling=3), b"pooling"),
    (dict(mixer_none=1, imagine=1), b"mixer=None"),
])
def test_every_dims_limit_is_reported(L, bad, needle):
    """check_dims (learner.hip): every limit of the path is rejected with a message naming it, through each sizing / layout entry."""
    from refil_amd import _lib
    kw = dict(B=2, T1=3, ne=32, na=16, ed=62, A=22, d=128, heads=4, H=64, hyp=128, M=32)
    kw.update(bad)
    d = _lib.make_dims(**kw)
    out = _lib.ParamLayout()
    assert L.refil_get_param_layout(C.byref(d), C.byref(out)) != 0
    assert needle in L.refil_last_error(), L.refil_last_error()
    assert L.refil_learner_workspace_bytes(C.byref(d)) == 0


def test_import_sets_hardware_queue_default(monkeypatch):
    """import refil_amd asks HIP for eight hardware queues unless the user chose otherwise: with HIP's four, the streams of an
    RCCL process group share queues with the step's four (every step 17-40 % slower)."""
    import importlib
    import os

    import refil_amd
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    importlib.reload(refil_amd)
    assert os.environ["GPU_MAX_HW_QUEUES"] == "8"
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "5")
    importlib.reload(refil_amd)
    assert os.environ["GPU_MAX_HW_QUEUES"] == "5"
