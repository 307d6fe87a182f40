"""Entity-scheme MAC (reference: src/controllers/entity_controller.py:6-36)."""
from ..modules.agents.entity_rnn_agent import EntityInputs
from .basic_controller import BasicMAC


class EntityMAC(BasicMAC):
    def _build_inputs(self, batch, t):
        """The reference materialises entities || one-hot(previous action) here with zeros/cat
        (entity_controller.py:13-27). The HIP library builds that tensor itself, so only the raw
        slices travel: entities[:, t], the actions whose one-hots are appended, the masks."""
        ents = batch["entities"][:, t]
        actions = None
        first_step_zero = True
        if self.args.entity_last_action:
            if t.start == 0:
                actions = batch["actions"][:, t]          # one-hot of actions[t-1] is appended at step t, zeros at t=0
            else:
                actions = batch["actions"][:, slice(t.start - 1, t.stop - 1)]
                first_step_zero = False
        gt = batch["gt_mask"][:, t] if getattr(self.args, "gt_mask_avail", False) else None     # entity_controller.py:28-29
        return EntityInputs(ents, actions, batch["obs_mask"][:, t], batch["entity_mask"][:, t], first_step_zero, gt)

    def _get_input_shape(self, scheme):
        shape = scheme["entities"]["vshape"]
        if self.args.entity_last_action:
            shape += scheme["actions_onehot"]["vshape"][0]
        return shape
