"""CPU tier: the two replay-side C entry points (refil_replay_gather, refil_pack_mask_bits -- SURVEY.md section 8 f2) executed on the
CPU wavefront emulator (tests/emu; see test_emu_ops.py) against torch indexing, bit for bit: plain fields of several element sizes,
time-truncated copies, bit-packed byte masks expanded by the gather, repeated and out-of-order episode ids. The ReplayBuffer class
around them (streams, events, staging minibatches) is the gpu tier's tests/test_replay_buffer.py."""
import ctypes as C
import os
import shutil

import pytest
import torch

import emu_util

pytestmark = pytest.mark.skipif(not (shutil.which("clang++") or os.path.exists("/opt/rocm/lib/llvm/bin/clang++")),
                                reason="the emulator build needs a host clang++ (vector extensions, __bf16)")


@pytest.fixture(autouse=True)
def _emulated_library():
    with emu_util.active():
        yield


@pytest.mark.parametrize("rows,width", [(1, 1), (7, 5), (300, 16), (129, 33), (64, 64), (1000, 48)])
def test_pack_mask_bits(rows, width):
    from refil_amd import _lib
    torch.manual_seed(rows + width)
    m = (torch.rand(rows, width) < 0.4).to(torch.uint8) * torch.randint(1, 255, (rows, width), dtype=torch.uint8)     # any non-zero byte is a set bit
    out = torch.full((rows,), -1, dtype=torch.int64)
    _lib.check(_lib.lib().refil_pack_mask_bits(_lib.ptr(m), _lib.ptr(out), C.c_int64(rows), C.c_int32(width), None), "refil_pack_mask_bits")
    sh = torch.arange(width, dtype=torch.int64)
    want = ((m != 0).to(torch.int64) << sh).sum(dim=1)          # (bit 63 set = a negative int64: the same bits)
    assert torch.equal(out, want)


@pytest.mark.parametrize("cap,B,T1,tcopy,seed", [(16, 5, 7, 7, 0), (40, 32, 11, 4, 1), (9, 9, 3, 3, 2), (33, 1, 20, 13, 3)])
def test_replay_gather_equals_indexing(cap, B, T1, tcopy, seed):
    """dst[b] <- src[episode_ids[b]] for every field in ONE launch: float32 / int64 / uint8 / int32 fields, the first `tcopy` steps only,
    and a byte mask kept bit-packed in the buffer (one int64 word per row) that the gather expands."""
    from refil_amd import _lib
    torch.manual_seed(seed)
    ne, na, A = 11, 4, 6
    ids = torch.randint(0, cap, (B,), dtype=torch.int64)
    ids[0] = ids[-1]                                              # a repeated episode
    specs = [("entities", (ne, 9), torch.float32), ("actions", (na, 1), torch.int64), ("terminated", (1,), torch.uint8),
             ("avail_actions", (na, A), torch.int32), ("reward", (1,), torch.float32)]
    fields = (_lib.GatherField * (len(specs) + 1))()
    keep, checks = [], []
    for i, (name, shape, dt) in enumerate(specs):
        src = (torch.randn(cap, T1, *shape) * 5).to(dt) if dt.is_floating_point else torch.randint(0, 100, (cap, T1, *shape)).to(dt)
        dst = torch.full((B, T1, *shape), 77).to(dt)
        per_t = src[0, 0].numel() * src.element_size()
        f = fields[i]
        f.src, f.dst = src.data_ptr(), dst.data_ptr()
        f.src_episode_bytes = f.dst_episode_bytes = T1 * per_t
        f.copy_bytes = tcopy * per_t
        f.unpack_width = 0
        keep += [src, dst]
        checks.append((name, src, dst))
    # obs_mask [cap, T1, ne, ne] bytes, stored as [cap, T1, ne] words
    mask = (torch.rand(cap, T1, ne, ne) < 0.5).to(torch.uint8)
    words = torch.empty(cap, T1, ne, dtype=torch.int64)
    _lib.check(_lib.lib().refil_pack_mask_bits(_lib.ptr(mask), _lib.ptr(words), C.c_int64(words.numel()), C.c_int32(ne), None), "refil_pack_mask_bits")
    mdst = torch.full((B, T1, ne, ne), 77, dtype=torch.uint8)
    f = fields[len(specs)]
    f.src, f.dst = words.data_ptr(), mdst.data_ptr()
    f.src_episode_bytes, f.dst_episode_bytes = T1 * ne * 8, T1 * ne * ne
    f.copy_bytes = tcopy * ne * ne
    f.unpack_width = ne
    _lib.check(_lib.lib().refil_replay_gather(fields, len(specs) + 1, _lib.ptr(ids), B, C.c_int64(cap), None), "refil_replay_gather")
    for name, src, dst in checks:
        assert torch.equal(dst[:, :tcopy], src[ids][:, :tcopy]), name
        assert (dst[:, tcopy:] == 77).all(), f"{name}: steps past copy_bytes were written"
    assert torch.equal(mdst[:, :tcopy], mask[ids][:, :tcopy]), "expanded bit-packed mask"
    assert (mdst[:, tcopy:] == 77).all()
