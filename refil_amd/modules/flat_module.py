"""nn.Module base whose parameters are views into ONE flat fp32 buffer laid out as the HIP library
expects (include/refil_hip.h: refil_param_layout), while keeping the reference's state_dict names,
so checkpoints interchange with the reference."""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.nn as nn


def th_zero():
    return torch.zeros(0)


class FlatParamModule(nn.Module):
    """Subclasses implement _fields() -> [(param name, offset in floats, shape)] and _flat_size()."""

    def _fields(self) -> List[Tuple[str, int, Tuple[int, ...]]]:
        raise NotImplementedError

    def _flat_size(self) -> int:
        raise NotImplementedError

    def __deepcopy__(self, memo):
        # copy.deepcopy(mac) / deepcopy(mixer) (q_learner.py:35,41): parameters are copied, the flat
        # buffer binding and the engine (device workspace) are not -- the copy re-flattens lazily
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in ("_engine", "_flat", "_named_cache", "_flat_pairs") else copy.deepcopy(v, memo)
        return new

    def _named(self):
        # (name -> Parameter) is walked once per module instance: the Parameter objects are stable, only their .data moves
        named = self.__dict__.get("_named_cache")
        if named is None:
            named = dict(self.named_parameters())
            object.__setattr__(self, "_named_cache", named)
        return named

    def _is_flat(self) -> bool:
        flat = getattr(self, "_flat", None)
        if flat is None:
            return False
        # (per training step: the (Parameter, offset) pairs are walked once per module instance -- the layout is a function of the
        # constructor arguments, the Parameter objects are stable, only their .data can be moved by the user)
        pairs = self.__dict__.get("_flat_pairs")
        if pairs is None:
            named = self._named()
            pairs = [(named[name], 4 * off) for name, off, shape in self._fields()]
            object.__setattr__(self, "_flat_pairs", pairs)
        base, dev = flat.data_ptr(), flat.device
        for p, boff in pairs:
            if p.data_ptr() != base + boff or p.device != dev or not p.is_contiguous():
                return False
        return True

    def adopt(self, storage: torch.Tensor):
        """Move the parameter values into `storage` (a flat fp32 tensor of _flat_size() floats, e.g. a
        slice of the learner's combined buffer) and make every parameter a view of it."""
        assert storage.dtype == torch.float32 and storage.is_contiguous() and storage.numel() >= self._flat_size()
        named = self._named()
        with torch.no_grad():
            for name, off, shape in self._fields():
                n = 1
                for s in shape:
                    n *= s
                view = storage[off:off + n].view(*shape)
                view.copy_(named[name].data)
                named[name].data = view
        object.__setattr__(self, "_flat", storage)

    def flat(self) -> torch.Tensor:
        """The flat buffer holding all parameters (re-flattened after .cuda()/deepcopy broke the views)."""
        if not self._is_flat():
            dev = next(self.parameters(), th_zero()).device
            self.adopt(torch.zeros(self._flat_size(), dtype=torch.float32, device=dev))
        return self._flat
