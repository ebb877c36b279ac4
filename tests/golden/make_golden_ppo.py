#!/usr/bin/env python3
"""Golden vectors for the PPO-side arithmetic, produced by IMPORTING the reference's pybullet-free
modules by file path: agents/ppo/storage.py (OPBuffer.compute_returns),
agents/ppo/policy.py (Policy), agents/ppo/ppo.py (PPO.update), agents/utils.py
(update_linear_schedule).  Only inputs/outputs are written (tests/golden/ppo_golden.pt); no reference
source or bytecode is copied.  Run in the build container only (/root/reference is not on the GPU box).
"""
import importlib.util
import os
import sys
import tempfile
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
REF = "/root/reference"


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def main():
    # `from soloRL.agents.utils import init_layer` (policy.py:8): expose the reference under the package
    # name it expects through a throw-away symlink OUTSIDE the repo
    tmp = tempfile.mkdtemp()
    os.symlink(REF, os.path.join(tmp, "soloRL"))
    sys.path.insert(0, tmp)
    for pkg in ("soloRL", "soloRL.agents"):
        sys.modules[pkg] = types.ModuleType(pkg)
        sys.modules[pkg].__path__ = [os.path.join(tmp, *pkg.split("."))]
    utils = load("soloRL.agents.utils", "agents/utils.py")
    storage = load("ref_storage", "agents/ppo/storage.py")
    policy = load("ref_policy", "agents/ppo/policy.py")
    ppo = load("ref_ppo", "agents/ppo/ppo.py")

    class Box:                      # policy.py:22-31 dispatches on the class name
        def __init__(self, n):
            self.shape = (n,)

    out = {}
    torch.manual_seed(0)
    T, N, O, A = 12, 5, 84, 12
    buf = storage.OPBuffer(T, N, (O,), A, torch.device("cpu"))
    buf.obs.copy_(torch.randn_like(buf.obs))
    buf.rewards.copy_(torch.randn_like(buf.rewards))
    buf.value_preds.copy_(torch.randn_like(buf.value_preds))
    buf.actions.copy_(torch.randn_like(buf.actions))
    buf.action_log_probs.copy_(torch.randn_like(buf.action_log_probs) - 10)
    buf.masks.copy_((torch.rand_like(buf.masks) > 0.2).float())
    next_value = torch.randn(N, 1)
    out["buf"] = {k: getattr(buf, k).clone() for k in ("obs", "rewards", "value_preds", "actions", "action_log_probs", "masks")}
    out["next_value"] = next_value.clone()
    buf.compute_returns(next_value, True, 0.99, 0.95)
    out["returns_gae"] = buf.returns.clone()
    vp_after = buf.value_preds.clone()
    buf.compute_returns(next_value, False, 0.99, 0.95)
    out["returns_disc"] = buf.returns.clone()
    buf.compute_returns(next_value, True, 0.99, 0.95)

    torch.manual_seed(1)
    pol = policy.Policy((O,), Box(A), None, {"hidden_size": 64})
    out["n_params"] = sum(p.numel() for p in pol.parameters())
    with torch.no_grad():
        pol.pi_dist.logstd.copy_(torch.linspace(-0.5, 0.3, A))
    out["state_dict"] = {k: v.clone() for k, v in pol.state_dict().items()}
    x = out["buf"]["obs"][0]
    with torch.no_grad():
        v, a, lp = pol.act(x, deterministic=True)
        v2, lp2, ent = pol.evaluate_actions(x, out["buf"]["actions"][0])
    out["act_det"] = dict(value=v, action=a, logp=lp)
    out["eval"] = dict(value=v2, logp=lp2, entropy=ent)

    # one PPO update with ONE mini-batch holding every sample (order-independent means)
    agent = ppo.PPO(pol, 0.1, 1, T * N, 0.5, 0.01, lr=2.5e-4, l2_coef=0.0, max_grad_norm=0.5)
    losses = agent.update(buf)
    out["update_losses"] = [float(z) for z in losses]
    out["state_dict_after"] = {k: v.clone() for k, v in pol.state_dict().items()}
    out["value_preds_after_gae"] = vp_after

    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    sched = []
    for ep in (0, 1, 50, 99):
        utils.update_linear_schedule(opt, ep, 100, 2.5e-4)
        sched.append(opt.param_groups[0]["lr"])
    out["linear_schedule"] = sched
    torch.save(out, os.path.join(HERE, "ppo_golden.pt"))
    print("n_params", out["n_params"], "losses", out["update_losses"], "sched", sched)


if __name__ == "__main__":
    main()
