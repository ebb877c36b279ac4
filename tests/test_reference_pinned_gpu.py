"""GPU tests that compare the HIP kernels DIRECTLY with outputs of the reference's own importable arithmetic (Next #2).

The reference holds no physics fixtures (its arithmetic is PyBullet's), but four of its modules run without PyBullet and pin what
they can: controllers/PD.py:3-10 (the PD law of the hot path's `apply_action`), agents/ppo/policy.py:33-58, agents/ppo/ppo.py:34-89,
agents/ppo/storage.py:35-55.  The fixtures under tests/golden/ were generated in the build container by importing those modules by
file path (make_golden.py, make_golden_ppo_kernel.py); here they are pushed through the C ABI on the MI355X:

  * every pd_golden.json case through the step kernel (team mode, lane mode, the fp64 instantiation): the torque the kernel
    applied (solorl_info_soa.applied_torque) against the reference's PD() output;
  * the reference's Policy.act on 512 rows against solorl_policy_act;
  * the reference's PPO.update gradients (one mini-batch of all 512 rows) against solorl_ppo_grad_stage1/2;
  * the reference's weights after one clipped Adam step, and its losses, against GraphedPPO.update (stage 1/2 + solorl_ppo_clip_adam
    replayed from a HIP graph).
"""
import json
import os

import numpy as np
import pytest
import torch

from solorl_amd.config import (default_config, ROBOT_SOLO12, TASK_STAND, CONTROL_PD, PRECISION_F32, PRECISION_F64)
from tests.util import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("team,precision", [("1", PRECISION_F32), ("0", PRECISION_F32), ("1", PRECISION_F64)])
def test_step_kernel_pd_torque_matches_reference_pd_vectors(gpu_device, monkeypatch, team, precision):
    """controllers/PD.py::PD golden vectors through solorl_step: set q / q_dot (robot airborne so that nothing else interferes),
    action = q_ref / 10 (solo.py:234: q_ref = clip(a, -1, 1) * joint limit 10), PD control with the case's gains; the torque the
    kernel applied equals the reference's tau to float32 rounding of the inputs (|q| <= 10, |q_dot| <= 100 are stored as float32
    by the fp32 engine; the action always crosses the boundary as float32)."""
    from solorl_amd.vec_env import SoloVecEnv
    monkeypatch.setenv("SOLORL_TEAM", team)
    cases = json.load(open(os.path.join(GOLDEN, "pd_golden.json")))
    assert cases[0]["tau"] == [3.0, -3.0, -3.0]                                 # the reference's own example
    worst = 0.0
    for cse in cases:
        c = default_config(ROBOT_SOLO12, TASK_STAND)
        c.control, c.kp, c.kd, c.max_torque = CONTROL_PD, cse["Kp"], cse["Kd"], cse["limit"]
        c.disable_termination, c.settle_min, c.settle_max, c.precision = 1, 5, 5, precision
        N = 5                                                                   # ragged: one full team-mode wavefront + one env
        env = SoloVecEnv(c, N, device="cuda:0", seed=1)
        env.reset()
        tau_out = env.record_applied_torque()
        n = len(cse["q"])                                                       # 3 (the survey's example) or 12
        a = np.zeros((N, 12), np.float32)
        for i in range(N):
            s = env.get_state(i); s.pos[2] = 5.0
            for j in range(12):
                s.q[j] = cse["q"][j] if j < n else 0.0
                s.qd[j] = cse["q_dot"][j] if j < n else 0.0
            env.set_state(i, s)
            a[i, :n] = np.asarray(cse["q_ref"]) / 10.0
        env.step(torch.from_numpy(a).cuda())
        got = tau_out.cpu().numpy().astype(np.float64)
        want = np.zeros(12); want[:n] = cse["tau"]
        # float32 inputs: |d tau| <= Kp (|d q_ref| + |d q|) + Kd |d q_dot| with relative input errors 6e-8
        tol = (cse["Kp"] * 20 + cse["Kd"] * 100) * 1.2e-7 + 1e-6
        err = np.abs(got - want[None]).max()
        worst = max(worst, err / tol)
        assert err <= tol, (cse["Kp"], cse["Kd"], err, tol)
        env.close()
    print("PD torque vs reference: worst error / float32 input-rounding bound = %.2f" % worst)


def _load_policy(e, dev):
    from solorl_amd.ppo import Policy
    from solorl_amd.vec_env import Box
    pol = Policy((e["O"],), Box(-np.ones(e["A"]), np.ones(e["A"])), None, {"hidden_size": 64})
    pol.load_state_dict(e["state_dict"])                    # the reference's parameter names and layouts, as is
    return pol.to(dev)


def _load_storage(e, dev):
    from solorl_amd.ppo import RolloutStorage
    st = RolloutStorage(e["T"], e["N"], (e["O"],), e["A"], dev)
    for k in ("obs", "rewards", "value_preds", "returns", "actions", "action_log_probs", "masks"):
        getattr(st, k).copy_(e["buf"][k])
    return st


@pytest.fixture(scope="module")
def kernel_golden():
    return torch.load(os.path.join(GOLDEN, "ppo_golden_kernel.pt"), weights_only=False)


@pytest.mark.parametrize("O", [76, 84])
def test_policy_act_kernel_matches_reference_policy(gpu_device, kernel_golden, O):
    """solorl_policy_act on the fixture's 512 rows against the reference's Policy.act(deterministic=True) outputs
    (agents/ppo/policy.py:33-49): value, action (= mean), log-prob.  Tolerance: float32 sums in another order + the kernel's
    tanh (absolute error <= 2e-7 per activation)."""
    from solorl_amd.ppo.fused import policy_act, policy_params, policy_kernels_supported
    e = kernel_golden[O]
    dev = torch.device("cuda:0")
    pol = _load_policy(e, dev)
    assert policy_kernels_supported(pol)
    n = e["T"] * e["N"]
    obs = e["buf"]["obs"][:-1].reshape(n, O).to(dev).contiguous()
    v, a, lp = torch.empty(n, 1, device=dev), torch.empty(n, e["A"], device=dev), torch.empty(n, 1, device=dev)
    policy_act(policy_params(pol), obs, None, v, a, lp)
    ref = e["act_det"]
    dv, da, dl = (v.cpu() - ref["value"]).abs().max().item(), (a.cpu() - ref["action"]).abs().max().item(), (lp.cpu() - ref["logp"]).abs().max().item()
    print("policy_act vs reference Policy.act (O=%d): max |dv| %.2e |da| %.2e |dlogp| %.2e" % (O, dv, da, dl))
    assert dv < 2e-5 and da < 2e-5 and dl < 1e-4


@pytest.mark.parametrize("O", [76, 84])
def test_minibatch_grad_kernels_match_reference_ppo_gradients(gpu_device, kernel_golden, O):
    """solorl_ppo_grad_stage1/2 on one mini-batch holding all 512 rows against p.grad of the reference's PPO.update
    (agents/ppo/ppo.py:34-77; max_grad_norm 1e9, i.e. unclipped) -- every parameter -- and against the losses it returned."""
    from solorl_amd.ppo import dist as D
    from solorl_amd.ppo.fused import MiniBatchGrad
    e = kernel_golden[O]
    dev = torch.device("cuda:0")
    pol = _load_policy(e, dev)
    st = _load_storage(e, dev)
    n = e["T"] * e["N"]
    adv = st.returns[:-1] - st.value_preds[:-1]
    adv = ((adv - adv.mean()) / (adv.std() + 1e-5)).reshape(n, 1).contiguous()          # ppo.py:35-37
    perm = torch.arange(n, device=dev)
    off = torch.zeros((), dtype=torch.long, device=dev)
    bucket = D.FlatGradBucket(pol.parameters())
    mb = MiniBatchGrad(pol, st, n, 0.1, 0.5, 0.01, True, perm, off, adv)
    bucket.flat.fill_(123.0)
    mb()
    worst = 0.0
    scale = max(g.abs().max().item() for g in e["grads"].values())
    for k, p in pol.named_parameters():
        d = (p.grad.cpu() - e["grads"][k]).abs().max().item()
        worst = max(worst, d)
        assert torch.allclose(p.grad.cpu(), e["grads"][k], rtol=2e-3, atol=2e-5 * scale), (k, d, scale)
    vl, al, ent = mb.losses(1)
    rv, ra, re = e["update_losses"]
    print("gradients vs reference PPO.update (O=%d): max |dg| %.2e at scale %.2e; losses %.6f/%.6f %.6f/%.6f %.6f/%.6f" % (O, worst, scale, vl, rv, al, ra, ent, re))
    assert abs(vl - rv) < 1e-4 * max(1.0, abs(rv)) and abs(al - ra) < 1e-5 + 1e-4 * abs(ra) and abs(ent - re) < 1e-5


@pytest.mark.parametrize("O", [76, 84])
def test_graphed_update_matches_reference_ppo_update(gpu_device, kernel_golden, O):
    """GraphedPPO.update (stage 1 + stage 2 + solorl_ppo_clip_adam replayed from a HIP graph) against the reference's PPO.update
    with the README's max_grad_norm 0.5: returned losses and every weight after the step.  One Adam step from zero moments moves
    each weight by lr * g / (|g| + eps) ~ +-2.5e-4, so the comparison is on the UPDATE (after - before), to 2 % of lr."""
    from solorl_amd.ppo.graphs import GraphedPPO
    e = kernel_golden[O]
    dev = torch.device("cuda:0")
    pol = _load_policy(e, dev)
    st = _load_storage(e, dev)
    n = e["T"] * e["N"]
    agent = GraphedPPO(pol, 0.1, 1, n, 0.5, 0.01, lr=2.5e-4, max_grad_norm=0.5)
    assert agent.fused_mlp
    vl, al, ent = agent.update(st)
    rv, ra, re = e["update_losses"]
    assert abs(vl - rv) < 1e-4 * max(1.0, abs(rv)) and abs(al - ra) < 1e-5 + 1e-4 * abs(ra) and abs(ent - re) < 1e-5
    worst, lr = 0.0, 2.5e-4
    moved = 0
    for k, v in pol.state_dict().items():
        du = (v.cpu() - e["state_dict"][k]) - (e["state_dict_after"][k] - e["state_dict"][k])
        ref_step = (e["state_dict_after"][k] - e["state_dict"][k]).abs()
        moved += int((ref_step > 0.5 * lr).sum())
        # where |g| is comparable to Adam's eps (1e-8) or to the kernels' summation error, the sign-like step g/(|g|+eps) is
        # ill-conditioned: compare where the reference moved by most of a full step, bound the rest by one step
        big = ref_step > 0.9 * lr
        if big.any():
            worst = max(worst, du[big].abs().max().item())
        assert du.abs().max().item() <= 2.0 * lr * 1.01, k
    print("update vs reference PPO.update (O=%d): max |d step| %.2e (lr %.1e) over %d moved weights" % (O, worst, lr, moved))
    assert moved > 1000 and worst < 0.02 * lr
