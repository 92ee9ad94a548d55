"""Pins oracle/oracle_recurrent.py (the CPU restatement of RecurrentActorProb / RecurrentCritic) to the reference's own
outputs and autograd gradients (tests/golden/recurrent_nets.npz, oracle/gen_golden.py::gen_recurrent)."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle_recurrent as OR
from tests.conftest import GOLDEN


@pytest.mark.parametrize("tag", ["utils", "small", "free"])
def test_oracle_matches_reference(tag):
    g = np.load(os.path.join(GOLDEN, "recurrent_nets.npz"))
    pre = tag + "_"
    obs_dim, act_dim, hidden, layers, B, T = (int(x) for x in g[pre + "dims"])
    kw = dict(max_action=float(g[pre + "max_action"]), unbounded=bool(g[pre + "unbounded"]))
    pa = {k: torch.from_numpy(g[pre + "actor." + k]).requires_grad_(True) for k in OR.actor_keys(layers)}
    pc = {k: torch.from_numpy(g[pre + "critic." + k]).requires_grad_(True) for k in OR.critic_keys(layers)}
    mu, sig, st = OR.actor_forward(pa, g[pre + "obs"], **kw)
    np.testing.assert_allclose(mu.detach().numpy(), g[pre + "mu"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(sig.detach().numpy(), g[pre + "sigma"], rtol=1e-7)
    np.testing.assert_allclose(st[0].detach().numpy(), g[pre + "hidden"], rtol=1e-6, atol=2e-7)
    np.testing.assert_allclose(st[1].detach().numpy(), g[pre + "cell"], rtol=1e-6, atol=2e-7)
    mu_s, _, st_s = OR.actor_forward(pa, g[pre + "obs"][:, -1], state=(g[pre + "state_hidden"], g[pre + "state_cell"]), **kw)
    np.testing.assert_allclose(mu_s.detach().numpy(), g[pre + "mu_s"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(st_s[0].detach().numpy(), g[pre + "hidden_s"], rtol=1e-6, atol=2e-7)
    (mu * torch.from_numpy(g[pre + "w_mu"])).sum().backward()
    for k in OR.actor_keys(layers):
        want = g[pre + "actor_grad." + k]
        got = pa[k].grad.numpy() if pa[k].grad is not None else np.zeros_like(want)
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-6 * max(float(np.abs(want).max()), 1e-3), err_msg=k)
    v = OR.critic_forward(pc, g[pre + "obs"], g[pre + "act"])
    np.testing.assert_allclose(v.detach().numpy(), g[pre + "value"], rtol=1e-6, atol=1e-7)
    (v * torch.from_numpy(g[pre + "w_v"])).sum().backward()
    for k in OR.critic_keys(layers):
        want = g[pre + "critic_grad." + k]
        np.testing.assert_allclose(pc[k].grad.numpy(), want, rtol=2e-5, atol=1e-6 * max(float(np.abs(want).max()), 1e-3), err_msg=k)
