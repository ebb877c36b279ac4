"""Reward / velocity tail of the engine under a Gaussian random policy (dev tool): how often does a step's reward leave the
[-12, 2] band the oracle stays in, and what state produces it?  usage: reward_tail.py [config.yaml] [precision f32|f64]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from solorl_amd.config import load_yaml, config_from_dict, PRECISION_F64
from solorl_amd.vec_env import SoloVecEnv
d = load_yaml(os.path.join(ROOT, "configs", sys.argv[1] if len(sys.argv) > 1 else "basic12.yaml")); d["task"] = os.environ.get("TASK", "walk")
cfg = config_from_dict(d)
if len(sys.argv) > 2 and sys.argv[2] == "f64": cfg.precision = PRECISION_F64
N = 4096
env = SoloVecEnv(cfg, N, device="cuda:0", seed=1); env.reset()
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
big, worst, tot = 0, 0.0, 0
vmax = zmax = 0.0
for t in range(600):
    a = torch.randn(N, env.act_dim, device="cuda:0", generator=g)
    o, r, dn, info = env.step_inplace(a)
    x = r[(r != -10.0)]
    m = x.abs().max().item()
    live = dn == 0
    vmax = max(vmax, o[live, 4:7].norm(dim=1).max().item()); zmax = max(zmax, o[live, 0].max().item())
    big += int((x.abs() > 20).sum()); tot += x.numel()
    if m > worst:
        worst = m; i = int(r.abs().argmax()); wo = o[i, :14].tolist(); wt = t
print("max base speed %.1f m/s, max z %.2f m;" % (vmax, zmax), end=" ")
print("steps*envs %d; |reward| > 20: %d (%.2e); largest |reward| %.1f at step %d; that env's obs[:14] (z, rpy/2pi, v, w, ...) = %s" % (
    tot, big, big / tot, worst, wt, ["%.2f" % v for v in wo]))
