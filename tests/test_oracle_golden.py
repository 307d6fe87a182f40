"""Pins oracle/refil_oracle.py against golden vectors produced by the REAL reference learner
(tools/make_golden.py). CPU only."""
import numpy as np
import pytest
import torch

from oracle import refil_oracle as orc
from golden_util import CASES, load, rel_err

TOL = 2e-5   # fp32, different op order than the reference (shared fc1/K/V, fused masks)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    g = load(name)
    z, cfg, case = g["z"], g["cfg"], g["case"]
    torch.manual_seed(0)
    agent, mixer = dict(g["agent"]), dict(g["mixer"])
    out, grads, gnorm = orc.train_step(cfg, agent, mixer, g["tagent"], g["tmixer"], g["batch"], g["bits"])
    B, T = case["B"], case["T"]
    assert rel_err(out.q.detach(), z["q"]) < TOL
    assert rel_err(out.chosen_q[0].detach(), z["chosen_q_real"]) < TOL
    assert rel_err(out.target_max_q, z["target_max_q"]) < TOL
    assert rel_err(out.q_tot.detach(), z["q_tot"]) < TOL
    assert rel_err(out.target_q_tot, z["target_q_tot"]) < TOL
    assert abs(out.loss.item() - float(z["stat.loss"])) < TOL * abs(float(z["stat.loss"]))
    if cfg.imagine:
        caq_im = torch.cat([out.chosen_q[1], out.chosen_q[2]], dim=2).detach()
        assert rel_err(caq_im, z["chosen_q_imagine"]) < TOL
        assert rel_err(out.q_tot_imagine.detach(), z["q_tot_imagine"]) < TOL
        assert abs(out.im_loss.item() - float(z["stat.im_loss"])) < TOL * abs(float(z["stat.im_loss"]))
        Wm, Im = orc.imagine_masks(g["bits"], g["batch"]["entity_mask"][:, 0])
        assert np.array_equal(Wm.numpy().astype(np.uint8), z["Wmask_noobs"])
        assert np.array_equal(Im.numpy().astype(np.uint8), z["Imask_noobs"])
    assert abs(gnorm - float(z["stat.grad_norm"])) < 1e-4 * float(z["stat.grad_norm"])
    for k in ("td_error_abs", "q_taken_mean", "target_mean"):
        assert abs(out.stats[k] - float(z["stat." + k])) < 1e-4 * max(abs(float(z["stat." + k])), 1e-3)
    gmax = max(v.abs().max().item() for v in grads.values())
    for k, gv in grads.items():
        if ("grad." + k) in z.files:
            ref = torch.from_numpy(z["grad." + k])
            assert (gv - ref).abs().max().item() < 1e-4 * gmax + 1e-9, k
            which, nm = k.split(".", 1)
            post = (agent if which == "agent" else mixer)[nm]
            assert (post - torch.from_numpy(z["post." + k])).abs().max().item() < 2e-6, k
        else:
            ref = float(z["gradnorm." + k])
            assert abs(gv.double().norm().item() - ref) < 1e-4 * max(ref, 1e-6), k


def test_partition_draw_matches_reference_rng_calls():
    g = load("refil_tiny")
    torch.manual_seed(g["case"]["seed"] + 7)
    bits = orc.draw_partition_bits(g["case"]["B"], g["case"]["ne"])
    assert torch.equal(bits, g["bits"])
