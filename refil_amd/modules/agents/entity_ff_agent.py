"""Feed-forward entity-attention agents (reference: src/modules/agents/entity_ff_agent.py:7-135), the
agents of BASELINE.json configs[0] (group_matching / refil_group_matching.yaml). Parameter holders +
marshalling to the HIP library, like the recurrent agents."""
from __future__ import annotations

import torch as th
import torch.nn as nn

from ... import _lib
from .entity_rnn_agent import EntityAttentionRNNAgent, EntityInputs, ImagineGroups, _InTrans, in_trans_fields
from ..flat_module import FlatParamModule


class EntityAttentionFFAgent(EntityAttentionRNNAgent):
    """fc1 -> attention -> relu -> fc2 (entity_ff_agent.py:29-57). Re-uses the recurrent agent's marshalling;
    only the parameter set differs (no fc2->H, GRU, fc3)."""

    def __init__(self, input_shape, args):
        FlatParamModule.__init__(self)
        self.args = args
        assert args.agent.endswith("_ff")
        d = args.attn_embed_dim
        assert d % args.attn_n_heads == 0, "Embed dim must be divisible by n_heads"
        self.fc1 = nn.Linear(input_shape, d)                 # same construction order as the reference => same init
        self.attn = _InTrans(d, getattr(args, "pooling_type", None))
        self.fc2 = nn.Linear(d, args.n_actions)
        if getattr(args, "pooling_type", None) is None:
            self.attn.register_buffer("scale_factor", th.scalar_tensor(d // args.attn_n_heads).sqrt())
        self.input_shape = input_shape
        self._engine = None

    def _fields(self):
        L = _lib.param_layout(self._dims(ed=self.input_shape, last_action=0))
        a = self.args
        d, A, E = a.attn_embed_dim, a.n_actions, self.input_shape
        return [("fc1.weight", L.ag_fc1_w, (d, E)), ("fc1.bias", L.ag_fc1_b, (d,)),
                *in_trans_fields("", L.ag_in_w, d, getattr(a, "pooling_type", None)),
                ("attn.out_trans.weight", L.ag_out_w, (d, d)), ("attn.out_trans.bias", L.ag_out_b, (d,)),
                ("fc2.weight", L.ag_fc2_w, (A, d)), ("fc2.bias", L.ag_fc2_b, (A,))]

    def forward(self, inputs, hidden_state, ret_attn_logits=None):
        assert ret_attn_logits is None, "ret_attn_logits is not on the hot path"
        q, _, _ = self._run(inputs, None, imagine=False)
        return q[0], hidden_state          # no recurrent state (the reference returns attn_outs here; nobody reads it)


class ImagineEntityAttentionFFAgent(EntityAttentionFFAgent):
    def forward(self, inputs, hidden_state, imagine=False, use_gt_factors=False, use_rand_gt_factors=False,
                group_bits=None, **kwargs):
        if not imagine:
            return super().forward(inputs, hidden_state)
        q, _, gb = self._run(inputs, None, imagine=True, group_bits=group_bits, use_gt_factors=use_gt_factors,
                             use_rand_gt_factors=use_rand_gt_factors)
        G, bs, ts, na, A = q.shape
        ne = self.args.n_entities
        if isinstance(inputs, EntityInputs):
            entity_mask, gt_mask = inputs.entity_mask, inputs.gt_mask
        else:
            entity_mask, gt_mask = inputs[2], (inputs[3] if len(inputs) > 3 else None)
        inact = entity_mask[:, 0].bool()
        active = (inact[:, :na, None] | inact[:, None, :])[:, None]                          # :91
        if use_gt_factors:
            W = gt_mask.bool()                                                             # :93-95 (not repeated in time, :131-135)
            groups = ImagineGroups((W | active).to(th.uint8), (~W | active).to(th.uint8), gt_mask=gt_mask)
        else:
            g = gb.bool()
            same = (~inact)[:, :na, None] & (~inact)[:, None, :] & (g[:, :na, None] == g[:, None, :])
            W = (~same)[:, None]
            if use_rand_gt_factors:                                                        # :111-114 (time-dependent, :131-135)
                W = W | gt_mask.bool()
                groups = ImagineGroups((W | active).to(th.uint8), (~W | active).to(th.uint8), bits=gb, gt_mask=gt_mask)
            else:
                groups = ImagineGroups((W | active).to(th.uint8).repeat(1, ts, 1, 1), (~W | active).to(th.uint8).repeat(1, ts, 1, 1), bits=gb)
        return q.reshape(G * bs, ts, na, A), hidden_state, groups
