"""Where the rollout's per-step time goes (not a pytest file): the step kernel alone under U(-1,1) and under clipped N(0,1) actions (what a
freshly initialised Gaussian policy emits), and solorl_step_act (policy tail in the same launch) against solorl_step + solorl_policy_act
(two launches), each as one HIP graph of K steps at 4096 Solo12-walk envs after the usual burn-in."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv, Box
from solorl_amd.ppo import Policy
from solorl_amd.ppo.fused import policy_act, policy_params
dev = torch.device("cuda:0")
N, K = 4096, 200
c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
torch.manual_seed(1)
pol = Policy((76,), Box(-np.ones(12), np.ones(12)), None, {"hidden_size": 64}).to(dev)
P = policy_params(pol)
def timed(body):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for t in range(K): body(t)
    ts = []
    for r in range(5):
        torch.cuda.synchronize(); t0 = time.time(); g.replay(); torch.cuda.synchronize(); ts.append(time.time() - t0)
    return sorted(ts)[2] / K * 1e3
for name, acts in (("U(-1,1) actions", torch.rand(64, N, 12, device=dev) * 2 - 1), ("N(0,1) actions (clipped by the env)", torch.randn(64, N, 12, device=dev))):
    env = SoloVecEnv(c, N, device=dev, seed=1); env.reset()
    for t in range(450): env.step_inplace(acts[t % 64])
    print("%-40s step kernel alone: %.4f ms/step" % (name, timed(lambda t: env.step_inplace(acts[t % 64]))), flush=True)
    env.close()
# closed loop with the policy: two launches vs one
noise = torch.randn(64, N, 12, device=dev)
for mode in ("two launches (step + policy_act)", "one launch (step_act)"):
    env = SoloVecEnv(c, N, device=dev, seed=1)
    obs = env.reset().clone()
    v, a, l = torch.empty(N, 1, device=dev), torch.empty(N, 12, device=dev), torch.empty(N, 1, device=dev)
    policy_act(P, obs, noise[0], v, a, l)
    o2 = torch.empty_like(obs)
    def two(t):
        env.step_inplace(a, obs_out=o2); policy_act(P, o2, noise[t % 64], v, a, l)
    def one(t):
        env.step_act_inplace(a, P, noise[t % 64], v, a, l, obs_out=o2)
    body = two if mode.startswith("two") else one
    for t in range(450): body(t)
    print("%-40s closed loop: %.4f ms/step" % (mode, timed(body)), flush=True)
    env.close()
# the tail's own cost: the SAME recorded closed-loop action sequence replayed with and without the policy tail (identical physics)
env = SoloVecEnv(c, N, device=dev, seed=1)
obs = env.reset().clone()
v, a, l = torch.empty(N, 1, device=dev), torch.empty(N, 12, device=dev), torch.empty(N, 1, device=dev)
policy_act(P, obs, noise[0], v, a, l)
rec = torch.empty(450 + K, N, 12, device=dev)
for t in range(450 + K):
    rec[t].copy_(a)
    env.step_act_inplace(rec[t], P, noise[t % 64], v, a, l)
env.close()
sv, sa, sl = torch.empty(N, 1, device=dev), torch.empty(N, 12, device=dev), torch.empty(N, 1, device=dev)
for mode in ("plain step", "step_act (outputs to scratch)", "step + policy_act"):
    env = SoloVecEnv(c, N, device=dev, seed=1); env.reset()
    for t in range(450): env.step_inplace(rec[t])
    if mode == "plain step": body = lambda t: env.step_inplace(rec[450 + t])
    elif mode.startswith("step_act"): body = lambda t: env.step_act_inplace(rec[450 + t], P, noise[t % 64], sv, sa, sl)
    else:
        def body(t):
            o, _, _, _ = env.step_inplace(rec[450 + t]); policy_act(P, o, noise[t % 64], sv, sa, sl)
    print("recorded closed-loop actions, %-32s %.4f ms/step" % (mode + ":", timed(body)), flush=True)
    env.close()
