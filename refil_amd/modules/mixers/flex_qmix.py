"""FlexQMixer behind the reference's mixer interface (reference: src/modules/mixers/flex_qmix.py:7-121).
Parameter holder + marshalling to refil_mixer_forward; no PyTorch math."""
from __future__ import annotations

import torch as th
import torch.nn as nn

from ... import _lib
from ...engine import LearnerEngine, dims_from_args
from ..agents.entity_rnn_agent import _InTrans, in_trans_fields
from ..flat_module import FlatParamModule

HYPERNETS = ("hyper_w_1", "hyper_w_final", "hyper_b_1", "V")


class AttentionHyperNet(nn.Module):
    """fc1 -> entity attention -> fc2 (flex_qmix.py:15-38); mode only matters inside the kernels."""

    def __init__(self, args, mode="matrix"):
        super().__init__()
        self.mode = mode
        E = args.entity_shape + (args.n_actions if args.entity_last_action else 0)
        h = args.hypernet_embed
        self.fc1 = nn.Linear(E, h)
        self.attn = _InTrans(h, getattr(args, "pooling_type", None))
        if getattr(args, "pooling_type", None) is None:
            self.attn.register_buffer("scale_factor", th.scalar_tensor(h // args.attn_n_heads).sqrt())
        self.fc2 = nn.Linear(h, args.mixing_embed_dim)


def pack_mask_words(Wmask, Imask, entity_mask, n_agents):
    """(Wmask, Imask) as the imagine agents return them ([bs, T or 1, ne, ne], nonzero = masked) -> the 64-bit key
    words refil_batch.mask_words documents: [bs*T, 3, 16*ceil(na/16)] (variant 0: the hypernets' default entity mask
    1 - active_i active_j, attention.py / flex_qmix.py:43-46) and the per-row words [bs*T, 3]. Glue on small integer
    tensors, off the learner's hot path (QLearner.train hands the partition bits to the kernels instead)."""
    bs, T, ne = entity_mask.shape
    na_pad = 16 * ((n_agents + 15) // 16)
    dev = entity_mask.device
    em = entity_mask.reshape(bs, T, ne).bool()
    shifts = th.arange(ne, device=dev, dtype=th.int64)
    pad_keys = th.tensor(-1 << ne if ne < 64 else 0, dtype=th.int64, device=dev)        # bits >= ne set

    def words(m):                                   # m [bs,T,na,ne] bool -> [bs,T,na] int64 (bit j = key j masked)
        return (m.to(th.int64) << shifts).sum(-1) | pad_keys      # (distinct bits: the sum is a bitwise OR)

    ent = em[:, :, :n_agents, None] | em[:, :, None, :]
    var = [ent] + [g.bool().expand(bs, T, ne, ne)[:, :, :n_agents, :] for g in (Wmask.to(dev), Imask.to(dev))]
    mw = th.full((bs, T, 3, na_pad), -1, dtype=th.int64, device=dev)                     # padded agents: all ones
    for v, m in enumerate(var):
        mw[:, :, v, :n_agents] = words(m)
    rb = th.zeros(bs, T, 3, dtype=th.int64, device=dev)
    rb[:, :, 2] = (em.to(th.int64) << shifts).sum(-1)
    return mw.reshape(bs * T, 3, na_pad).contiguous(), rb.reshape(bs * T, 3).contiguous()


class FlexQMixer(FlatParamModule):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.n_agents = args.n_agents
        self.embed_dim = args.mixing_embed_dim
        self.hyper_w_1 = AttentionHyperNet(args, mode="matrix")
        self.hyper_w_final = AttentionHyperNet(args, mode="vector")
        self.hyper_b_1 = AttentionHyperNet(args, mode="vector")
        self.V = AttentionHyperNet(args, mode="scalar")
        self._engine = None

    def _dims(self, B=1, T1=2):
        return dims_from_args(self.args, B, T1)

    def _nets(self):
        return HYPERNETS

    def _layout(self):
        return _lib.param_layout(self._dims())

    def _fields(self):
        L = self._layout()
        a = self.args
        h, M = a.hypernet_embed, a.mixing_embed_dim
        E = a.entity_shape + (a.n_actions if a.entity_last_action else 0)
        base = L.agent_total
        out = []
        for n, net in enumerate(self._nets()):
            out += [(f"{net}.fc1.weight", L.mix_fc1_w + n * L.mix_fc1_w_stride - base, (h, E)),
                    (f"{net}.fc1.bias", L.mix_fc1_b + n * L.mix_fc1_b_stride - base, (h,)),
                    *in_trans_fields(net + ".", L.mix_in_w + n * L.mix_in_w_stride - base, h, getattr(a, "pooling_type", None)),
                    (f"{net}.attn.out_trans.weight", L.mix_out_w + n * L.mix_out_w_stride - base, (h, h)),
                    (f"{net}.attn.out_trans.bias", L.mix_out_b + n * L.mix_out_b_stride - base, (h,)),
                    (f"{net}.fc2.weight", L.mix_fc2_w + n * L.mix_fc2_w_stride - base, (M, h)),
                    (f"{net}.fc2.bias", L.mix_fc2_b + n * L.mix_fc2_b_stride - base, (M,))]
        return out

    def _flat_size(self):
        L = self._layout()
        return int(L.total - L.agent_total)

    def engine(self):
        if self._engine is None:
            object.__setattr__(self, "_engine", LearnerEngine(self.V.fc1.weight.device))
        return self._engine

    def _mix(self, agent_qs, inputs, imagine_groups, want_ingroup):
        """agent_qs [bs,T,na] (or [bs,T,2na] with imagine_groups); inputs = (entities||last-action
        [bs,T,ne,E], entity_mask [bs,T,ne]) as built by QLearner._get_mixer_ins (q_learner.py:45-64).
        imagine_groups: what the imagine agents return (ImagineGroups carrying the partition bits or the
        gt mask), or the [bs,ne] bits tensor itself; the kernels re-derive the masks in registers."""
        from ..agents.entity_rnn_agent import ImagineGroups
        entities, entity_mask = inputs
        bs, T, ne, E = entities.shape
        dims = self._dims(bs, T)
        dims.ed, dims.entity_last_action = E, 0          # entities already carry the one-hots
        fields = {"entities": entities.contiguous(), "entity_mask": entity_mask.contiguous()}
        gb = qs_im = mw = rb = None
        dims.imagine = dims.gt_factors = 0
        if imagine_groups is not None:
            if isinstance(imagine_groups, ImagineGroups):
                if imagine_groups.gt_mask is not None:
                    gt = imagine_groups.gt_mask
                    fields["gt_mask"] = (gt[:, :T] if gt.shape[1] != T else gt).contiguous()
                    gb = imagine_groups.bits                 # bits AND gt mask: randomised ground-truth factors
                    dims.gt_factors = 2 if gb is not None else 1
                else:
                    gb = imagine_groups.bits
            elif isinstance(imagine_groups, (tuple, list)):
                # the reference's own calling convention (flex_qmix.py:85-94): (Wmask, Imask) [bs, T or 1, ne, ne]
                mw, rb = pack_mask_words(imagine_groups[0], imagine_groups[1], entity_mask, self.n_agents)
            else:
                gb = imagine_groups
            if gb is not None:
                gb = gb.to(entities.device).to(th.uint8).contiguous()
            qs_im = agent_qs.reshape(bs, T, 2 * self.n_agents).contiguous().float()
            real = th.zeros(bs, T, self.n_agents, dtype=th.float32, device=entities.device)
            dims.imagine = 1
        else:
            real = agent_qs.reshape(bs, T, self.n_agents).contiguous().float()
        L = self._layout()
        # the library indexes the mixer tensors at absolute offsets of the combined [agent|mixer] buffer
        base_ptr = self.flat().data_ptr() - 4 * int(L.agent_total)
        out = self.engine().mixer_forward(dims, fields, gb, None, real, qs_im, 0, T, params_ptr=base_ptr,
                                          want_ingroup=want_ingroup, mask_words=mw, mask_row_bits=rb)
        q = (out[1] if qs_im is not None else out[0]).reshape(bs, T, 1)
        if want_ingroup:
            return q, out[2][0] / (bs * T)
        return q

    def forward(self, agent_qs, inputs, imagine_groups=None):
        return self._mix(agent_qs, inputs, imagine_groups, False)


class LinearFlexQMixer(FlexQMixer):
    """flex_qmix.py:124-172: q_tot = sum_i q_i * w1_i + V with w1 from an 'alt_vector' attention hypernet."""

    def __init__(self, args):
        FlatParamModule.__init__(self)
        self.args = args
        self.n_agents = args.n_agents
        self.embed_dim = args.mixing_embed_dim
        self.hyper_w_1 = AttentionHyperNet(args, mode="alt_vector")
        self.V = AttentionHyperNet(args, mode="scalar")
        self._engine = None

    def _nets(self):
        return ("hyper_w_1", "V")

    def forward(self, agent_qs, inputs, imagine_groups=None, ret_ingroup_prop=False):
        return self._mix(agent_qs, inputs, imagine_groups, bool(ret_ingroup_prop))
