"""Shared-parameter multi-agent controller with the reference's surface
(reference: src/controllers/basic_controller.py:7-121): used by runners for acting
(select_actions / init_hidden) and by the learner for its parameters and (de)serialisation."""
import torch as th

from ..components.action_selectors import REGISTRY as action_REGISTRY
from ..modules.agents import REGISTRY as agent_REGISTRY


class BasicMAC:
    def __init__(self, scheme, groups, args):
        self.n_agents = args.n_agents
        self.args = args
        self._build_agents(self._get_input_shape(scheme))
        self.agent_output_type = args.agent_output_type
        assert self.agent_output_type == "q", "only Q-value agents are on the REFIL hot path (basic_controller.py:43-62 is dead for it)"
        self.action_selector = action_REGISTRY[args.action_selector](args)
        self.hidden_states = None

    def select_actions(self, ep_batch, t_ep, t_env, bs=slice(None), test_mode=False, ret_agent_outs=False):
        avail_actions = ep_batch["avail_actions"][:, t_ep]
        agent_outputs = self.forward(ep_batch, t_ep, test_mode=test_mode)
        chosen = self.action_selector.select_action(agent_outputs[bs], avail_actions[bs], t_env, test_mode=test_mode)
        if ret_agent_outs:
            return chosen, agent_outputs[bs]
        return chosen

    def forward(self, ep_batch, t, test_mode=False, **kwargs):
        int_t = False
        if t is None:
            t = slice(0, ep_batch["avail_actions"].shape[1])
        elif type(t) is int:
            t, int_t = slice(t, t + 1), True
        agent_inputs = self._build_inputs(ep_batch, t)
        groups = None
        if kwargs.get("imagine", False):
            agent_outs, self.hidden_states, groups = self.agent(agent_inputs, self.hidden_states, **kwargs)
        else:
            agent_outs, self.hidden_states = self.agent(agent_inputs, self.hidden_states)
        if int_t:                                           # (a single step returns the outputs alone, imagined or not)
            return agent_outs.squeeze(1)
        return (agent_outs, groups) if kwargs.get("imagine", False) else agent_outs

    def init_hidden(self, batch_size):
        self.hidden_states = self.agent.init_hidden().unsqueeze(0).expand(batch_size, self.n_agents, -1)

    def parameters(self):
        return self.agent.parameters()

    def load_state(self, other_mac):
        self.agent.load_state_dict(other_mac.agent.state_dict())

    def cuda(self):
        self.agent.cuda()

    def eval(self):
        self.agent.eval()

    def train(self):
        self.agent.train()

    def save_models(self, path):
        th.save(self.agent.state_dict(), "{}/agent.th".format(path))

    def load_models(self, path):
        self.agent.load_state_dict(th.load("{}/agent.th".format(path), map_location=lambda storage, loc: storage))

    def _build_agents(self, input_shape):
        self.agent = agent_REGISTRY[self.args.agent](input_shape, self.args)

    def _build_inputs(self, batch, t):
        raise NotImplementedError("flat-observation agents are out of scope; use entity_mac (SURVEY.md section 2, row 7)")

    def _get_input_shape(self, scheme):
        raise NotImplementedError("flat-observation agents are out of scope; use entity_mac")
