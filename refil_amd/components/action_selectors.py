"""Action selectors used by the MAC while acting (reference: src/components/action_selectors.py:10-63).
Rollout-side glue: plain torch ops on [bs, n_agents, n_actions] tensors."""
import torch as th
from torch.distributions import Categorical

from .epsilon_schedules import DecayThenFlatSchedule

REGISTRY = {}


class MultinomialActionSelector:
    def __init__(self, args):
        self.args = args
        self.schedule = DecayThenFlatSchedule(args.epsilon_start, args.epsilon_finish, args.epsilon_anneal_time, decay="linear")
        self.epsilon = self.schedule.eval(0)
        self.test_greedy = getattr(args, "test_greedy", True)

    def select_action(self, agent_inputs, avail_actions, t_env, test_mode=False):
        policies = agent_inputs.masked_fill(avail_actions == 0, 0.0)
        self.epsilon = self.schedule.eval(t_env)
        if test_mode and self.test_greedy:
            return policies.max(dim=2)[1]
        return Categorical(policies, validate_args=False).sample().long()     # (same draw; the argument checks cost ~0.1 ms per env step)


class EpsilonGreedyActionSelector:
    def __init__(self, args):
        self.args = args
        self.schedule = DecayThenFlatSchedule(args.epsilon_start, args.epsilon_finish, args.epsilon_anneal_time, decay="linear")
        self.epsilon = self.schedule.eval(0)

    def select_action(self, agent_inputs, avail_actions, t_env, test_mode=False):
        self.epsilon = 0.0 if test_mode else self.schedule.eval(t_env)
        q = agent_inputs.masked_fill(avail_actions == 0, -float("inf"))      # never pick an unavailable action
        explore = (th.rand_like(agent_inputs[:, :, 0]) < self.epsilon).long()
        random_actions = Categorical(avail_actions.float(), validate_args=False).sample().long()     # (same draw, no argument checks)
        return explore * random_actions + (1 - explore) * q.max(dim=2)[1]


REGISTRY["multinomial"] = MultinomialActionSelector
REGISTRY["epsilon_greedy"] = EpsilonGreedyActionSelector
