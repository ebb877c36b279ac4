#!/usr/bin/env python3
"""Reference-derived vectors at a batch size the HIP policy kernels accept (Next #2b): T = 16, N = 32 -> 512 rows
(whole wavefronts), observation widths 76 (Solo12 walk, h = 1) and 84 (Solo12 pointgoal, h = 1).

Produced by IMPORTING the reference's pybullet-free modules by file path (as make_golden_ppo.py does): agents/ppo/policy.py
(Policy.act / evaluate_actions), agents/ppo/ppo.py (PPO.update), agents/ppo/storage.py (OPBuffer, compute_returns).  Only
inputs and outputs are written (tests/golden/ppo_golden_kernel.pt); no reference source or bytecode is stored.  Per width:

  buf                 the rollout buffer the reference's own OPBuffer holds (seeded inputs, returns from its compute_returns)
  state_dict          initial weights of the reference's Policy (its own orthogonal init), logstd set to a ramp
  act_det             Policy.act(obs, deterministic=True) on all 512 rows: value, action (= mean), log-prob
  grads               p.grad of every parameter after PPO.update with ONE epoch, ONE mini-batch of all 512 rows and no effective
                      norm clip (max_grad_norm 1e9): the mean is order-independent, so the sampler's permutation does not matter
  update_losses       what that update returned (value loss, action loss, entropy)
  state_dict_after    weights after the same update run with the README's max_grad_norm 0.5 (clip + Adam step)
Run in the build container only (/root/reference is not on the GPU box).
"""
import copy
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True


def main():
    import importlib.util
    import tempfile
    import types
    REF = "/root/reference"

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m

    tmp = tempfile.mkdtemp()
    os.symlink(REF, os.path.join(tmp, "soloRL"))
    sys.path.insert(0, tmp)
    for pkg in ("soloRL", "soloRL.agents"):
        sys.modules[pkg] = types.ModuleType(pkg)
        sys.modules[pkg].__path__ = [os.path.join(tmp, *pkg.split("."))]
    load("soloRL.agents.utils", "agents/utils.py")
    storage = load("ref_storage", "agents/ppo/storage.py")
    policy = load("ref_policy", "agents/ppo/policy.py")
    ppo = load("ref_ppo", "agents/ppo/ppo.py")

    class Box:                      # policy.py:22-31 dispatches on the class name
        def __init__(self, n):
            self.shape = (n,)

    out = {}
    T, N, A = 16, 32, 12
    for O in (76, 84):
        torch.manual_seed(100 + O)
        buf = storage.OPBuffer(T, N, (O,), A, torch.device("cpu"))
        buf.obs.copy_(torch.randn_like(buf.obs) * 0.5)
        buf.rewards.copy_(torch.randn_like(buf.rewards))
        buf.actions.copy_(torch.randn_like(buf.actions) * 0.7)
        buf.masks.copy_((torch.rand_like(buf.masks) > 0.1).float())
        pol = policy.Policy((O,), Box(A), None, {"hidden_size": 64})
        with torch.no_grad():
            pol.pi_dist.logstd.copy_(torch.linspace(-0.5, 0.3, A))
            flat_obs = buf.obs[:-1].reshape(T * N, O)
            # the rollout's own values / log-probs come from the same policy plus a perturbation, so that the ratio and both clip
            # branches of ppo.py:54-68 are exercised
            v, lp, _ = pol.evaluate_actions(flat_obs, buf.actions.reshape(T * N, A))
            buf.value_preds[:-1].copy_((v + 0.3 * torch.randn_like(v)).view(T, N, 1))
            buf.action_log_probs.copy_((lp + 0.15 * torch.randn_like(lp)).view(T, N, 1))
            next_value = pol.get_value(buf.obs[-1])
        buf.compute_returns(next_value, True, 0.99, 0.95)
        e = {"T": T, "N": N, "O": O, "A": A}
        e["buf"] = {k: getattr(buf, k).clone() for k in ("obs", "rewards", "value_preds", "returns", "actions", "action_log_probs", "masks")}
        e["state_dict"] = {k: v.clone() for k, v in pol.state_dict().items()}
        with torch.no_grad():
            v, a, lp = pol.act(flat_obs, deterministic=True)
        e["act_det"] = dict(value=v.clone(), action=a.clone(), logp=lp.clone())
        pol_g = copy.deepcopy(pol)
        agent = ppo.PPO(pol_g, 0.1, 1, T * N, 0.5, 0.01, lr=2.5e-4, l2_coef=0.0, max_grad_norm=1e9)
        losses = agent.update(buf)
        e["update_losses"] = [float(z) for z in losses]
        e["grads"] = {k: p.grad.clone() for k, p in pol_g.named_parameters()}
        pol_s = copy.deepcopy(pol)
        agent = ppo.PPO(pol_s, 0.1, 1, T * N, 0.5, 0.01, lr=2.5e-4, l2_coef=0.0, max_grad_norm=0.5)
        agent.update(buf)
        e["state_dict_after"] = {k: v.clone() for k, v in pol_s.state_dict().items()}
        out[O] = e
        print("O", O, "losses", e["update_losses"], "grad norm", float(torch.sqrt(sum((g ** 2).sum() for g in e["grads"].values()))))
    torch.save(out, os.path.join(HERE, "ppo_golden_kernel.pt"))


if __name__ == "__main__":
    main()
