"""Entity-attention recurrent agents behind the reference's agent interface
(reference: src/modules/agents/entity_rnn_agent.py:7-126). The modules only own parameters (same
names/shapes/default init as the reference); forward() marshals to the HIP library
(refil_agent_forward) -- there is no PyTorch implementation of the math here."""
from __future__ import annotations

from collections import namedtuple

import torch as th
import torch.nn as nn

from ... import _lib
from ...engine import LearnerEngine, dims_from_args
from ..flat_module import FlatParamModule

# What EntityMAC._build_inputs hands to the agent. Raw batch fields instead of the materialised
# entities||last-action tensor: the library builds that tensor itself (once, for MAC and mixer).
EntityInputs = namedtuple("EntityInputs", ["entities", "actions", "obs_mask", "entity_mask", "first_step_zero", "gt_mask"],
                          defaults=(None,))


class ImagineGroups(tuple):
    """(Wmask_noobs, Imask_noobs) like the reference returns (entity_rnn_agent.py:126, entity_ff_agent.py:135),
    plus what the HIP mixers actually consume: the [bs,ne] partition bits or the ground-truth factor mask
    (the kernels re-derive the masks in registers)."""

    def __new__(cls, W, I, bits=None, gt_mask=None):
        obj = super().__new__(cls, (W, I))
        obj.bits, obj.gt_mask = bits, gt_mask
        return obj


class _InTrans(nn.Module):
    """Parameter holder with the reference's attribute names: EntityAttentionLayer (attention.py:21-22; in_trans
    [3w,w] without bias) or, when pooling_type is set, EntityPoolingLayer (attention.py:93-94; in_trans [w,w] + bias)."""

    def __init__(self, dim, pooling_type=None):
        super().__init__()
        if pooling_type is None:
            self.in_trans = nn.Linear(dim, dim * 3, bias=False)
        else:
            assert pooling_type in ("mean", "max"), "pooling_type must be None, 'mean' or 'max'"
            self.in_trans = nn.Linear(dim, dim)
        self.out_trans = nn.Linear(dim, dim)


def in_trans_fields(prefix, off, w, pooling_type):
    """_fields() entries of the in_trans slot of the flat layout (include/refil_hip.h: refil_dims.pooling)."""
    if pooling_type is None:
        return [(prefix + "attn.in_trans.weight", off, (3 * w, w))]
    return [(prefix + "attn.in_trans.weight", off, (w, w)), (prefix + "attn.in_trans.bias", off + w * w, (w,))]


class EntityAttentionRNNAgent(FlatParamModule):
    def __init__(self, input_shape, args):
        super().__init__()
        self.args = args
        d, H = args.attn_embed_dim, args.rnn_hidden_dim
        assert d % args.attn_n_heads == 0, "Embed dim must be divisible by n_heads"     # attention.py:16
        # same construction order as the reference => identical default init under the same seed
        self.fc1 = nn.Linear(input_shape, d)
        self.attn = _InTrans(d, getattr(args, "pooling_type", None))
        self.fc2 = nn.Linear(d, H)
        self.rnn = nn.GRUCell(H, H)
        self.fc3 = nn.Linear(H, args.n_actions)
        if getattr(args, "pooling_type", None) is None:
            self.attn.register_buffer("scale_factor", th.scalar_tensor(d // args.attn_n_heads).sqrt())   # checkpoint key parity
        self.input_shape = input_shape
        self._engine = None

    # ---- flat layout ---------------------------------------------------------------------
    def _dims(self, B=1, T1=1, ed=None, last_action=None):
        a = self.args
        d = dims_from_args(a, B, T1)
        if ed is not None:
            d.ed = ed
        if last_action is not None:
            d.entity_last_action = int(last_action)
        return d

    def _fields(self):
        # layout is determined by the fc1 input width only; express it with last_action=0, ed=input_shape
        L = _lib.param_layout(self._dims(ed=self.input_shape, last_action=0))
        a = self.args
        d, H, A, E = a.attn_embed_dim, a.rnn_hidden_dim, a.n_actions, self.input_shape
        return [("fc1.weight", L.ag_fc1_w, (d, E)), ("fc1.bias", L.ag_fc1_b, (d,)),
                *in_trans_fields("", L.ag_in_w, d, getattr(a, "pooling_type", None)),
                ("attn.out_trans.weight", L.ag_out_w, (d, d)), ("attn.out_trans.bias", L.ag_out_b, (d,)),
                ("fc2.weight", L.ag_fc2_w, (H, d)), ("fc2.bias", L.ag_fc2_b, (H,)),
                ("rnn.weight_ih", L.ag_w_ih, (3 * H, H)), ("rnn.weight_hh", L.ag_w_hh, (3 * H, H)),
                ("rnn.bias_ih", L.ag_b_ih, (3 * H,)), ("rnn.bias_hh", L.ag_b_hh, (3 * H,)),
                ("fc3.weight", L.ag_fc3_w, (A, H)), ("fc3.bias", L.ag_fc3_b, (A,))]

    def _flat_size(self):
        return int(_lib.param_layout(self._dims(ed=self.input_shape, last_action=0)).agent_total)

    # ---- reference interface -------------------------------------------------------------
    def init_hidden(self):
        return self.fc1.weight.new_zeros(1, self.args.rnn_hidden_dim)

    def engine(self):
        if self._engine is None:
            object.__setattr__(self, "_engine", LearnerEngine(self.fc1.weight.device))
        return self._engine

    def _run(self, inputs, hidden_state, imagine, group_bits=None, use_gt_factors=False, use_rand_gt_factors=False):
        na, H = self.args.n_agents, self.args.rnn_hidden_dim
        gt_mask = None
        if isinstance(inputs, EntityInputs):
            ents, obs_mask, entity_mask, gt_mask = inputs.entities, inputs.obs_mask, inputs.entity_mask, inputs.gt_mask
            fields = {"entities": ents, "obs_mask": obs_mask, "entity_mask": entity_mask}
            last_action = bool(self.args.entity_last_action)
            if last_action:
                fields["actions"] = inputs.actions
            dims = self._dims(ents.shape[0], ents.shape[1], ed=ents.shape[3], last_action=last_action)
            fsz = inputs.first_step_zero
        else:   # the reference's tuple form: entities already carry the last-action one-hots
            ents, obs_mask, entity_mask = inputs[:3]
            gt_mask = inputs[3] if len(inputs) > 3 else None
            fields = {"entities": ents, "obs_mask": obs_mask, "entity_mask": entity_mask}
            dims = self._dims(ents.shape[0], ents.shape[1], ed=ents.shape[3], last_action=False)
            fsz = True
        dims.imagine = int(imagine)
        assert not (use_gt_factors and use_rand_gt_factors), "Can only select one of use_rand_gt_factors and use_gt_factors"
        dims.gt_factors = 2 if use_rand_gt_factors else int(bool(use_gt_factors))
        if dims.gt_factors or dims.gt_obs_mask:
            assert gt_mask is not None, "gt_mask needed (env must provide it: gt_mask_avail)"
            fields["gt_mask"] = gt_mask
        bs, ts = ents.shape[0], ents.shape[1]
        G = 3 if imagine else 1
        fields = {k: (v if v[0, 0].is_contiguous() else v.contiguous()) for k, v in fields.items()}
        h0 = None
        if hidden_state is not None:
            h0 = hidden_state.reshape(-1, na, H)
            if imagine and h0.shape[0] == bs:
                h0 = h0.repeat(3, 1, 1)                                   # entity_rnn_agent.py:124
            h0 = h0.reshape(G, bs, na, H).contiguous().float()
        if imagine and group_bits is None and not use_gt_factors:
            # the reference's two RNG calls (entity_rnn_agent.py:94-96), on the CPU generator so that a
            # seed reproduces the same partition as the reference's CPU run
            p = th.rand(bs, 1, 1).repeat(1, 1, self.args.n_entities)
            group_bits = th.bernoulli(p).to(th.uint8).reshape(bs, -1)
        gb = group_bits.to(ents.device).contiguous() if group_bits is not None else None
        q, h = self.engine().agent_forward(dims, fields, gb, self.flat(), h0, first_step_zero=fsz)
        return q, h, gb

    def forward(self, inputs, hidden_state, ret_attn_logits=None):
        assert ret_attn_logits is None, "ret_attn_logits is not on the hot path (attention.py:68-78)"
        q, h, _ = self._run(inputs, hidden_state, imagine=False)
        bs = q.shape[1]
        return q[0], h[0].reshape(bs, 1, self.args.n_agents, -1)


class ImagineEntityAttentionRNNAgent(EntityAttentionRNNAgent):
    """REFIL agent: with imagine=True returns q for [real; within-group; between-group] stacked on
    the batch dim plus the mixer-side masks (entity_rnn_agent.py:87-126)."""

    def forward(self, inputs, hidden_state, imagine=False, group_bits=None, **kwargs):
        if not imagine:
            return super().forward(inputs, hidden_state)
        q, h, gb = self._run(inputs, hidden_state, imagine=True, group_bits=group_bits)
        G, bs, ts, na, A = q.shape
        entity_mask = inputs.entity_mask if isinstance(inputs, EntityInputs) else inputs[2]
        g = gb.bool()
        inact = entity_mask[:, 0].bool()
        act_pair = (~inact)[:, :, None] & (~inact)[:, None, :]
        same = act_pair & (g[:, :, None] == g[:, None, :])
        Wm = (~same).to(th.uint8)[:, None].repeat(1, ts, 1, 1)              # :111,126
        Im = (same | ~act_pair).to(th.uint8)[:, None].repeat(1, ts, 1, 1)   # :112,126
        return q.reshape(G * bs, ts, na, A), h.reshape(G * bs, 1, na, -1), ImagineGroups(Wm, Im, bits=gb)
