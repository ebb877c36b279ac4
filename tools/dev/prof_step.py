"""Minimal workload for rocprofv3 PMC passes (not a pytest file): the headline configuration (Solo12 walk, 4096 envs,
random policy), burn-in of episode_length + 50 steps to the stationary regime, then 40 more steps; tools/dev/rocpd_export.py
averages the counters over those last 40 full-size dispatches."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
N = int(os.environ.get("PROF_ENVS", "4096"))
c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
g = torch.Generator(device="cuda:0"); g.manual_seed(1234)
a = torch.rand(64, N, 12, device="cuda:0", generator=g) * 2 - 1
for t in range(int(os.environ.get("PROF_STEPS", "490"))): env.step_inplace(a[t % 64])
torch.cuda.synchronize()
