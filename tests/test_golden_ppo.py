"""PPO-side arithmetic against golden vectors generated from the reference's own modules
(tests/golden/make_golden_ppo.py)."""
import os

import numpy as np

import pytest
import torch

from solorl_amd.ppo import Policy, RolloutStorage, PPO
from solorl_amd.vec_env import Box
from tests.util import GOLDEN

G = torch.load(os.path.join(GOLDEN, "ppo_golden.pt"), weights_only=False)
T, N, O, A = 12, 5, 84, 12


def storage_from_golden():
    s = RolloutStorage(T, N, (O,), A, torch.device("cpu"))
    for k, v in G["buf"].items():
        getattr(s, k).copy_(v)
    return s


def test_gae_and_discounted_returns():
    s = storage_from_golden()
    s.compute_returns(G["next_value"], True, 0.99, 0.95)
    assert torch.allclose(s.returns[:-1], G["returns_gae"][:-1], atol=1e-5)
    assert torch.allclose(s.value_preds, G["value_preds_after_gae"])
    s.compute_returns(G["next_value"], False, 0.99, 0.95)
    assert torch.allclose(s.returns, G["returns_disc"], atol=1e-5)


def test_policy_matches_reference():
    import numpy as np
    pol = Policy((O,), Box(-np.ones(A), np.ones(A)), None, {"hidden_size": 64})
    assert sum(p.numel() for p in pol.parameters()) == G["n_params"] == 20057
    assert set(pol.state_dict().keys()) == set(G["state_dict"].keys())            # checkpoint compatibility
    pol.load_state_dict(G["state_dict"])
    x = G["buf"]["obs"][0]
    with torch.no_grad():
        v, a, lp = pol.act(x, deterministic=True)
        v2, lp2, ent = pol.evaluate_actions(x, G["buf"]["actions"][0])
    assert torch.allclose(v, G["act_det"]["value"], atol=1e-6) and torch.allclose(a, G["act_det"]["action"], atol=1e-6)
    assert torch.allclose(lp, G["act_det"]["logp"], atol=1e-5)
    assert torch.allclose(v2, G["eval"]["value"], atol=1e-6) and torch.allclose(lp2, G["eval"]["logp"], atol=1e-4)
    assert torch.allclose(ent, G["eval"]["entropy"], atol=1e-6)
    # stochastic action: log-prob consistent with the sample
    torch.manual_seed(0)
    with torch.no_grad():
        v, a, lp = pol.act(x)
        _, lp_chk, _ = pol.evaluate_actions(x, a)
    assert torch.allclose(lp, lp_chk, atol=1e-5)


def test_init_is_orthogonal_gain_sqrt2():
    import numpy as np
    torch.manual_seed(3)
    pol = Policy((O,), Box(-np.ones(A), np.ones(A)), None, {"hidden_size": 64})
    w = pol.base.features[2].weight
    assert torch.allclose(w @ w.t(), 2.0 * torch.eye(64), atol=1e-4)
    assert float(pol.base.features[0].bias.abs().max()) == 0.0 and float(pol.pi_dist.logstd.abs().max()) == 0.0


def test_ppo_update_matches_reference():
    import numpy as np
    pol = Policy((O,), Box(-np.ones(A), np.ones(A)), None, {"hidden_size": 64})
    pol.load_state_dict(G["state_dict"])
    s = storage_from_golden()
    s.compute_returns(G["next_value"], True, 0.99, 0.95)
    agent = PPO(pol, 0.1, 1, T * N, 0.5, 0.01, lr=2.5e-4, l2_coef=0.0, max_grad_norm=0.5)
    losses = agent.update(s)
    assert losses == pytest.approx(G["update_losses"], rel=1e-4, abs=1e-5)
    after = pol.state_dict()
    for k, v in G["state_dict_after"].items():
        assert torch.allclose(after[k], v, atol=2e-6), k


def test_linear_lr_schedule():
    from solorl_amd.ppo.train import update_linear_schedule
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    got = []
    for ep in (0, 1, 50, 99):
        update_linear_schedule(opt, ep, 100, 2.5e-4)
        got.append(opt.param_groups[0]["lr"])
    assert got == pytest.approx(G["linear_schedule"], rel=1e-12)


def test_split_reduction_linear_has_nn_linear_gradients():
    """ppo/policy.py Linear: training batches take the sliced weight-gradient path; values and gradients are nn.Linear's
    (float64: equal up to summation order), parameter names unchanged (checkpoints, agents/ppo/train.py:121-131)."""
    from solorl_amd.ppo.policy import Linear
    torch.manual_seed(3)
    for m in (4096, 5000, 300):                         # whole slices / slices + remainder / below the threshold
        a, b = Linear(7, 5).double(), torch.nn.Linear(7, 5).double()
        b.load_state_dict(a.state_dict())
        x = torch.randn(m, 7, dtype=torch.float64)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        wgt = torch.randn(m, 5, dtype=torch.float64)
        ya, yb = a(xa), b(xb)
        assert torch.allclose(ya, yb, rtol=0, atol=1e-14)
        (ya * wgt).sum().backward(); (yb * wgt).sum().backward()
        for ga, gb in ((a.weight.grad, b.weight.grad), (a.bias.grad, b.bias.grad), (xa.grad, xb.grad)):
            assert torch.allclose(ga, gb, rtol=1e-12, atol=1e-12)
    assert list(Linear(3, 2).state_dict()) == ["weight", "bias"]


def test_act_into_is_act():
    """Policy.act_into (rollout loops writing into their storage rows) computes act()'s value / action / log-prob."""
    torch.manual_seed(0)
    pol = Policy((76,), Box(-np.ones(12), np.ones(12)), None, {"hidden_size": 64})
    with torch.no_grad():
        pol.pi_dist.logstd.normal_(0, 0.3)
    x = torch.randn(50, 76)
    torch.manual_seed(4); v, a, lp = pol.act(x)
    vo, ao, lo = torch.empty(50, 1), torch.empty(50, 12), torch.empty(50, 1)
    torch.manual_seed(4); pol.act_into(x, vo, ao, lo)
    assert torch.allclose(v, vo, atol=1e-6) and torch.allclose(a, ao, atol=1e-6) and torch.allclose(lp, lo, atol=1e-5)
