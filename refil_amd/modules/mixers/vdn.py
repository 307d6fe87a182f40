"""VDN mixer (reference: src/modules/mixers/vdn.py:5-10): q_tot = sum of the agents' Qs. Parameter-free;
inside QLearner.train the sum (and its backward) is a mode of the HIP mixing kernels. This module only exists
for the reference's surface (learner.mixer(...) / target_mixer)."""
import torch as th

from ..flat_module import FlatParamModule


class VDNMixer(FlatParamModule):
    def _fields(self):
        return []

    def _flat_size(self):
        return 0

    def forward(self, agent_qs, batch, imagine_groups=None):
        return th.sum(agent_qs, dim=2, keepdim=True)      # glue-level op on a [bs,T,na] tensor, off the hot path
