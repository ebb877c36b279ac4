"""ms/step at 4096 / 8192 envs (Solo12 walk, random policy, steady state) for the solver settings of DESIGN.md section 3:
residual exit on/off x warm-start factor (not a pytest file)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
for N in (4096, 8192):
    for thr, warm in ((1e-7, 0.0), (0.0, 0.85), (1e-7, 0.85), (0.0, 0.0)):
        c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
        c.solver_residual_threshold, c.warmstart = thr, warm
        env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
        a = torch.rand(64, N, env.act_dim, device="cuda:0") * 2 - 1
        for t in range(450): env.step_inplace(a[t % 64])
        K = 200
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in range(K): env.step_inplace(a[t % 64])
        ts = []
        for r in range(3):
            torch.cuda.synchronize(); t0 = time.time(); g.replay(); torch.cuda.synchronize(); ts.append(time.time() - t0)
        dt = sorted(ts)[1]
        st = env.episode_stats()
        print("N %5d thr %g warm %.2f: %.4f ms/step  %.2f M env-steps/s | mean episode length %.1f" % (N, thr, warm, dt / K * 1e3, N * K / dt / 1e6, st["episode_length"]), flush=True)
        env.close()
