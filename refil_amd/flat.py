"""Views of the library's flat parameter buffer under the reference's state_dict names.

The HIP library works on ONE flat fp32 buffer (layout: refil_get_param_layout in include/refil_hip.h).
This module maps it to/from the names a reference checkpoint uses (agent.th / mixer.th keys, SURVEY.md
section 5): agent = EntityAttentionRNNAgent (entity_rnn_agent.py:8-25), mixer = FlexQMixer
(flex_qmix.py:28-38,69-73).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import _lib

HYPERNETS = ("hyper_w_1", "hyper_w_final", "hyper_b_1", "V")   # storage order inside the flat buffer

_AGENT = [  # (state_dict key, layout field, shape lambda)
    ("fc1.weight", "ag_fc1_w", lambda c: (c["d"], c["E"])),
    ("fc1.bias", "ag_fc1_b", lambda c: (c["d"],)),
    ("attn.in_trans.weight", "ag_in_w", lambda c: (3 * c["d"], c["d"])),
    ("attn.out_trans.weight", "ag_out_w", lambda c: (c["d"], c["d"])),
    ("attn.out_trans.bias", "ag_out_b", lambda c: (c["d"],)),
    ("fc2.weight", "ag_fc2_w", lambda c: (c["H"], c["d"])),
    ("fc2.bias", "ag_fc2_b", lambda c: (c["H"],)),
    ("rnn.weight_ih", "ag_w_ih", lambda c: (3 * c["H"], c["H"])),
    ("rnn.weight_hh", "ag_w_hh", lambda c: (3 * c["H"], c["H"])),
    ("rnn.bias_ih", "ag_b_ih", lambda c: (3 * c["H"],)),
    ("rnn.bias_hh", "ag_b_hh", lambda c: (3 * c["H"],)),
    ("fc3.weight", "ag_fc3_w", lambda c: (c["A"], c["H"])),
    ("fc3.bias", "ag_fc3_b", lambda c: (c["A"],)),
]
_MIXER = [
    ("fc1.weight", "mix_fc1_w", lambda c: (c["h"], c["E"])),
    ("fc1.bias", "mix_fc1_b", lambda c: (c["h"],)),
    ("attn.in_trans.weight", "mix_in_w", lambda c: (3 * c["h"], c["h"])),
    ("attn.out_trans.weight", "mix_out_w", lambda c: (c["h"], c["h"])),
    ("attn.out_trans.bias", "mix_out_b", lambda c: (c["h"],)),
    ("fc2.weight", "mix_fc2_w", lambda c: (c["M"], c["h"])),
    ("fc2.bias", "mix_fc2_b", lambda c: (c["M"],)),
]


LIN_HYPERNETS = ("hyper_w_1", "V")                              # LinearFlexQMixer (flex_qmix.py:133-134)
_AGENT_FF = _AGENT[:5] + [("fc2.weight", "ag_fc2_w", lambda c: (c["A"], c["d"])), ("fc2.bias", "ag_fc2_b", lambda c: (c["A"],))]


def _consts(dims: _lib.Dims) -> Dict[str, int]:
    E = dims.ed + (dims.A if dims.entity_last_action else 0)
    return {"d": dims.d, "E": E, "H": dims.H, "A": dims.A, "h": dims.hyp, "M": dims.M}


def views(flat: torch.Tensor, dims: _lib.Dims) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """(agent, mixer) dicts of views into `flat` keyed by the reference's state_dict names."""
    L = _lib.param_layout(dims)
    c = _consts(dims)
    assert flat.numel() >= L.total and flat.is_contiguous()
    agent, mixer = {}, {}
    def put(dst, key, o, s):
        n = 1
        for x in s:
            n *= x
        dst[key] = flat[o:o + n].view(*s)

    for key, fld, shp in (_AGENT_FF if dims.agent_ff else _AGENT):
        if key == "attn.in_trans.weight" and dims.pooling:        # EntityPoolingLayer: W [d,d] then bias [d] in the same slot
            put(agent, key, L.ag_in_w, (c["d"], c["d"]))
            put(agent, "attn.in_trans.bias", L.ag_in_w + c["d"] * c["d"], (c["d"],))
            continue
        put(agent, key, getattr(L, fld), shp(c))
    for ni, net in enumerate(() if (dims.mixer_vdn or dims.mixer_none) else (LIN_HYPERNETS if dims.mixer_lin else HYPERNETS)):
        for key, fld, shp in _MIXER:
            o = getattr(L, fld) + ni * getattr(L, fld + "_stride")
            if key == "attn.in_trans.weight" and dims.pooling:
                put(mixer, f"{net}.{key}", o, (c["h"], c["h"]))
                put(mixer, f"{net}.attn.in_trans.bias", o + c["h"] * c["h"], (c["h"],))
                continue
            put(mixer, f"{net}.{key}", o, shp(c))
    return agent, mixer


def total(dims: _lib.Dims) -> int:
    return int(_lib.param_layout(dims).total)


def pack(dims: _lib.Dims, agent_sd: Dict[str, torch.Tensor], mixer_sd: Dict[str, torch.Tensor],
         device="cuda") -> torch.Tensor:
    """New flat buffer (+0 padding) filled from state_dicts (extra keys such as attn.scale_factor ignored)."""
    flat = torch.zeros(total(dims), dtype=torch.float32, device=device)
    a, m = views(flat, dims)
    for k, v in a.items():
        v.copy_(agent_sd[k])
    for k, v in m.items():
        v.copy_(mixer_sd[k])
    return flat
