"""TEST INFRASTRUCTURE: runs the `-m gpu` parity tests' bodies against tests/emu (the kernel sources of refil_amd/csrc compiled for the host
and executed on the CPU wavefront emulator), so that the CPU tier exercises the product's kernel SOURCE where no GPU exists.

Nothing here touches the product: refil_amd/_lib.py keeps loading refil_amd/librefil_hip.so (and raising without it). While an emulator
test runs, the ctypes handle the test helpers go through is swapped for one of tests/emu/_build/librefil_emu.so and handed back afterwards."""
import contextlib
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_handle = None


def emu_handle():
    """ctypes handle of the emulator build, with the argtypes refil_amd/_lib.py declares (built on first use)."""
    global _handle
    if _handle is None:
        sys.path.insert(0, os.path.join(HERE, "emu"))
        import build_emu
        from refil_amd import _lib
        path = build_emu.build()
        saved = (_lib.LIB_PATH, _lib._lib)
        try:
            _lib.LIB_PATH, _lib._lib = path, None
            _handle = _lib.lib()            # (binds restype / argtypes exactly as for the product library)
        finally:
            _lib.LIB_PATH, _lib._lib = saved
    return _handle


def _host_batch(fields, group_bits=None, device=None, mask_words=None, mask_row_bits=None):
    """refil_batch over HOST tensors for the emulator (the product's refil_amd._lib.make_batch refuses host tensors: it has no CPU
    path). Same dtype conversions, same strides."""
    import torch
    from refil_amd import _lib
    b = _lib.Batch()
    b._converted = False
    keep = []
    names = {"entities": "ent", "obs_mask": "om", "entity_mask": "em", "actions": "ac", "avail_actions": "av",
             "reward": "rw", "terminated": "tm", "filled": "fl", "gt_mask": "gt"}
    want = _lib._field_dtypes()
    for name, short in names.items():
        t = fields.get(name)
        if t is None:
            continue
        assert not t.is_cuda
        if t.dtype != want[name]:
            t = t.view(torch.uint8) if (t.dtype == torch.bool and want[name] == torch.uint8) else t.to(want[name])
            keep.append(t)
            b._converted = True
        assert t[0, 0].is_contiguous()
        setattr(b, name, t.data_ptr())
        setattr(b, short + "_sB", t.stride(0))
        setattr(b, short + "_sT", t.stride(1))
    if group_bits is not None:
        group_bits = group_bits.to(torch.uint8).contiguous()
        keep.append(group_bits)
        b.group_bits = group_bits.data_ptr()
    if mask_words is not None:
        keep += [mask_words, mask_row_bits]
        b.mask_words, b.mask_row_bits = mask_words.data_ptr(), mask_row_bits.data_ptr()
    b._keep = keep
    return b


class _HostEvent:
    """torch.cuda.Event for code that runs on the emulator: its streams execute in host order, every event has happened when it is recorded.
    cuda_event is a non-null handle (the library passes it to hipStreamWaitEvent, a no-op there), so refil_batch.ready_event switches the
    early-prologue code path on exactly as a real event does."""
    cuda_event = 1

    def __init__(self, *a, **k):
        pass

    def record(self, *a, **k):
        pass

    def synchronize(self):
        pass

    def wait(self, *a, **k):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        return 0.0


@contextlib.contextmanager
def active():
    """route refil_amd._lib.lib() to the emulator build; null stream, host batches, no device synchronisation. For the plugin layer
    (QLearner, the only class of the package that asks for CUDA tensors, events and pinned memory): events that have always happened,
    pin_memory() as the identity, and the learner's GPU-only guard stood in for -- the guard itself (q_learner._require_gpu) is untouched."""
    import torch
    from refil_amd import _lib
    from refil_amd.learners import q_learner
    saved = (_lib._lib, _lib.current_stream_ptr, _lib.make_batch, torch.cuda.synchronize, torch.cuda.Event, torch.Tensor.pin_memory, q_learner._require_gpu)
    _lib._lib = emu_handle()
    _lib.current_stream_ptr = lambda: None
    _lib.make_batch = _host_batch
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.Event = _HostEvent
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    q_learner._require_gpu = lambda dev: None
    try:
        yield _lib._lib
    finally:
        (_lib._lib, _lib.current_stream_ptr, _lib.make_batch, torch.cuda.synchronize, torch.cuda.Event, torch.Tensor.pin_memory,
         q_learner._require_gpu) = saved


def load_copy(module, **patch):
    """a private copy of a tests/ module (its own globals: DEV = "cpu" there does not touch the module the gpu tier collects)"""
    spec = importlib.util.spec_from_file_location("emu__" + module, os.path.join(HERE, module + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for k, v in patch.items():
        setattr(m, k, v)
    return m
