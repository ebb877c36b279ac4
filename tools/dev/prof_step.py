"""Minimal workload for rocprofv3 PMC passes (not a pytest file)."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
N = 4096
c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
a = torch.rand(16, N, 12, device="cuda:0") * 2 - 1
for t in range(40): env.step_inplace(a[t % 16])
torch.cuda.synchronize()
