"""ms/step at 4096 / 8192 envs (Solo12 walk, random policy, steady state) for the friction model x contact ERP of the [K] ledger
(DESIGN.md section 3): round 3's pair (pyramid, 0.2), each change alone, and round 4's defaults (cone, 0.08).  Not a pytest file."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from solorl_amd.config import *
from solorl_amd.vec_env import SoloVecEnv
for N in (4096, 8192, 65536):
    for cone, cerp in ((0, 0.2), (1, 0.2), (0, 0.08), (1, 0.08)):
        c = default_config(ROBOT_SOLO12, TASK_WALK); c.num_history_stack = 1
        c.friction_model, c.contact_erp = cone, cerp
        env = SoloVecEnv(c, N, device="cuda:0", seed=1); env.reset()
        a = torch.rand(64, N, env.act_dim, device="cuda:0") * 2 - 1
        for t in range(450): env.step_inplace(a[t % 64])
        K = 200
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in range(K): env.step_inplace(a[t % 64])
        ts = []
        for r in range(5):
            torch.cuda.synchronize(); t0 = time.time(); g.replay(); torch.cuda.synchronize(); ts.append(time.time() - t0)
        dt = sorted(ts)[2]
        st = env.episode_stats()
        print("N %6d %-7s contact_erp %.2f: %.4f ms/step  %.2f M env-steps/s | mean episode length %.1f" % (
            N, "cone" if cone else "pyramid", cerp, dt / K * 1e3, N * K / dt / 1e6, st["episode_length"]), flush=True)
        env.close()
