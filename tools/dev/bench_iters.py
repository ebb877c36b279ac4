"""Time per step vs PGS iterations / frame_skip (not a pytest file)."""
import os, sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for iters, fs in [(50, 4), (1, 4), (25, 4), (50, 1), (1, 1)]:
    c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1; c.solver_iterations = iters; c.frame_skip = fs
    env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
    a = torch.rand(16, N, 12, device="cuda:0") * 2 - 1
    for t in range(30): env.step_inplace(a[t % 16])
    torch.cuda.synchronize(); t0 = time.time(); K = 200
    for t in range(K): env.step_inplace(a[t % 16])
    torch.cuda.synchronize(); dt = time.time() - t0
    print("N %d epw %s iters %2d frame_skip %d: %.3f ms/step" % (N, os.environ.get("SOLORL_ENVS_PER_WAVE", "auto"), iters, fs, dt / K * 1e3), flush=True)
